// handoff_scan_check.cpp — the predecessor / successor scans of the hit_stack hand-off (csrc/handoff.hpp) on the CPU,
// against a brute-force walk over random touched-bit patterns: one chain through the whole tile, and tiles whose rows
// start chains of their own (halo slots in front of every row).  Prints "ok <cases>" or the first mismatch.
// Built and run by tests/test_host_logic.py (hipcc, host code only: no GPU needed).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "handoff.hpp"

using namespace p3d;

static uint32_t rnd_state = 2463534242u;
static uint32_t rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 17; rnd_state ^= rnd_state << 5; return rnd_state; }

int main() {
  long cases = 0;
  for (int trial = 0; trial < 400; ++trial) {
    Handoff H{};
    const uint32_t w = 1 + rnd() % 97, rows = 1 + rnd() % 23;
    const bool chains = trial % 2 == 1;
    H.halo = chains ? kHaloChain : 0;
    H.row_units = w + H.halo;
    H.rows = rows;
    H.n_units = rows * H.row_units;
    std::vector<uint32_t> bits(H.n_units / 32 + 2, 0);
    std::vector<uint8_t> chain(rows, 0), touched(H.n_units, 0);
    const uint32_t density = rnd() % 5;  // 0: almost empty ... 4: almost full
    for (uint32_t u = 0; u < H.n_units; ++u) {
      const bool t = density == 0 ? rnd() % 61 == 0 : (density == 4 ? rnd() % 61 != 0 : rnd() % (density + 1) != 0);
      touched[u] = t;
      if (t) bits[u >> 5] |= 1u << (u & 31u);
    }
    if (chains)
      for (uint32_t r = 0; r < rows; ++r) chain[r] = r == 0 || rnd() % 3 == 0;
    H.touched = bits.data();
    H.row_chain = chains ? chain.data() : nullptr;
    for (uint32_t u = 0; u < H.n_units; ++u) {
      // brute force: walk back / forward unit by unit; a row that starts a chain cuts the link to the rows before it
      int want_pred = -1;
      for (long v = (long)u - 1; v >= 0; --v) {
        if (touched[v]) { want_pred = (int)v; break; }
        if (chains && v % H.row_units == 0 && chain[v / H.row_units]) break;  // first unit of a chain row: nothing before it
      }
      if (chains && u % H.row_units == 0 && chain[u / H.row_units]) want_pred = -1;
      int want_succ = -1;
      for (uint32_t v = u + 1; v < H.n_units; ++v) {
        if (chains && v % H.row_units == 0 && chain[v / H.row_units]) break;  // the next row starts on its own halo
        if (touched[v]) { want_succ = (int)v; break; }
      }
      const int got_pred = handoff_pred(H, u), got_succ = handoff_succ(H, u);
      if (got_pred != want_pred || got_succ != want_succ) {
        std::printf("MISMATCH trial %d unit %u (w %u rows %u chains %d): pred %d want %d, succ %d want %d\n", trial, u, w, rows, (int)chains,
                    got_pred, want_pred, got_succ, want_succ);
        return 1;
      }
      ++cases;
    }
  }
  std::printf("ok %ld\n", cases);
  return 0;
}
