// handoff.hpp — the reference's ONE BVH::hit_stack for the whole frame (bvh.cpp:86), reproduced on a GPU.
//
// The reference renders pixels one after the other (main.cpp:747-751) and its any-hit query leaves entries on
// the member stack whenever it returns `true` (bvh.cpp:322).  The next closest-hit query that gets past the
// root test pops them AFTER its own entries (bvh.cpp:256-265: LIFO) and visits those that are nearer than its
// hit — which can re-normalise its ray copy once more (ray.h:16-18) and move the hit point by an ulp.  So the
// colour of pixel p depends on what the last pixel before it that touched the stack left behind.
//
// What the kernels do (P3D_STACK_LITERAL, DESIGN.md "hit_stack hand-off"):
//   pass 1   every pixel ("unit") is rendered on an EMPTY stack; it records its leftover, whether it touched the
//            stack at all (a primary ray that fails the root test, bvh.cpp:203-205, passes its predecessor's
//            leftover on unchanged) and the result of its first closest hit (hit point + object).
//   check    every unit whose predecessor left something re-traces that first closest hit on the predecessor's
//            leftover.  Everything a pixel computes after that query depends on the stack only through the query's
//            result, so an unchanged result means an unchanged pixel.
//   redo     the units whose result changed are rendered again, seeded; if a unit's own leftover changes, its
//            successor is looked at in the next round (check + redo again), until no leftover changes: a fixed
//            point of "every unit was rendered on its predecessor's current leftover" is the serial result,
//            because the first unit's stack is empty in both and each next one is determined by the one before.
// Units are numbered in the reference's order: tile row by tile row, x ascending; every row has `halo` extra slots
// in front for the chain of pixels that precedes the row in the FRAME when the row above it in the tile is not
// its predecessor there (stripes of a multi-GPU frame, sub-rectangles): see halo_find_kernel.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace p3d {

constexpr uint32_t kHaloChain = 16;       // slots in front of a row that starts a chain of its own: at most that many frame pixels are rendered for their leftovers
constexpr uint32_t kMetaTouched = 1u << 17;
constexpr uint32_t kNoUnit = 0xffffffffu;

enum HandoffCounter { kHoChecked = 0, kHoRedone, kHoRounds, kHoPoolTop, kHoListA, kHoListB, kHoListC, kHoListD, kHoCheckN, kHoRound0, kHoRound1, kHoNumCounters };
constexpr uint32_t kHoErrLeftoverCap = 1u;   // a leftover had more entries than a slot holds, or the leftover pool is full
constexpr uint32_t kHoErrNoFixedPoint = 2u;  // round bound reached
constexpr uint32_t kHoErrTrips = 4u;         // sample hand-out loop hit its trip bound (SUB = 4 kernels)
constexpr uint32_t kHoErrList = 8u;          // a work list or a ray queue segment overflowed
constexpr uint32_t kHoErrHalo = 16u;         // no pixel in front of a row could be shown to leave a stack that does not depend on what it found (halo_find_kernel)

struct Handoff {
  uint32_t n_units;     // rows * row_units
  uint32_t row_units;   // halo + tile width
  uint32_t halo;        // 0 or kHaloChain
  uint32_t rows;        // tile rows
  uint32_t cap;         // entries per leftover slot
  uint32_t persistent;  // list kernel: one workgroup, loop over rounds until the list stays empty
  uint32_t list_cap;
  uint32_t max_rounds;  // rounds (work-list launches + trips of the persistent workgroup) after which a non-empty list is an error
  uint32_t round_base;  // rounds that came before this launch
  uint32_t count;       // keep the checked / redone statistics (one contended atomic per wave: only on request)
  uint32_t lanes;       // list kernel: list entries per wave (fewer = less divergence between unrelated pixels)
  // Leftover stacks, two slots per unit (a unit rendered again writes its other slot: nothing is read while it is written).
  // Compact (default): `entries` is a pool, a leftover of n entries takes n consecutive ones from a bump pointer (one atomic
  // per wave) and where[slot][unit] says where - most units leave nothing, the worst case is cap entries (lights x tree depth).
  // Dense (per-level launches, p3d_config.handoff_records = P3D_HANDOFF_DENSE): slot s of unit u at (s * n_units + u) * cap.
  uint2* entries;
  uint32_t* where;         // compact: [2][n_units] first pool entry of the slot
  uint32_t* pool_top;      // compact: next free pool entry
  // compact: the units pass 1 left something behind for, in the order their waves finished (one atomic per wave, next to the
  // pool's).  The check launch runs over THIS list - 64 entries per wave, each checking the unit that starts on the listed
  // unit's leftover - instead of over every tile with a few lanes to do (100k triangles 2048x2048: 398 k of 4.2 M units).
  uint32_t* check_list;    // [n_units]; null: the check launch walks the tiles (dense records, per-level launches, LDS-staged scenes)
  uint32_t* check_n;
  uint32_t pool_cap;       // entries in the pool
  uint32_t dense;
  uint32_t* meta;          // [n_units] bits 0..15 entries of the current slot, bit 16 which slot, bit 17 touched
  uint32_t* meta0;         // [n_units] the word as pass 1 wrote it (slot 0), read-only afterwards
  float4* first;           // [n_units] {hit point, object id} of the first closest hit of the first touching sample
  uint32_t* first_sample;  // [n_units] index of that sample (anti-aliased launches)
  uint32_t* touched;       // bit per unit
  const uint8_t* row_chain;  // [rows] row starts a chain of its own; null: one chain through the whole tile
  const uint32_t* halo_pix;  // [rows * halo] frame pixel (y * res_x + x) of a halo slot, kNoUnit: none
  uint4* list_in;            // work of this launch {unit, predecessor, predecessor's slot << 16 | entries, 1 = check first}
  uint4* list_out;           // units to look at in the next round
  uint32_t* n_in;
  uint32_t* n_out;
  uint32_t* counters;        // kHoNumCounters words
  // collect_stats under P3D_STACK_LITERAL: the counters of every unit's CURRENT render (so that a unit rendered again
  // replaces its first pass) and, within them, those of its first closest hit (which a new predecessor leftover changes
  // even when the hit itself stays: the stale entries are visited).  Summed over the tile's pixels at the end.
  uint32_t* ucount;          // [kNumStats][n_units]
  uint32_t* uch0;            // [kCh0Counters][n_units]
};
constexpr int kCh0Counters = 5;  // node, sphere, triangle, box, plane tests of the first closest hit

// highest set bit in [lo, i), -1: none  (host + device: tests/handoff_scan_check.cpp runs these on the CPU)
__host__ __device__ inline int find_prev_bit(const uint32_t* bits, uint32_t lo, uint32_t i) {
  if (i <= lo) return -1;
  uint32_t w = (i - 1) >> 5;
  const uint32_t wlo = lo >> 5;
  uint32_t word = bits[w] & (0xffffffffu >> (31u - ((i - 1) & 31u)));
  while (true) {
    if (w == wlo) word &= 0xffffffffu << (lo & 31u);
    if (word) return (int)(w * 32u + 31u - (uint32_t)__builtin_clz(word));
    if (w == wlo) return -1;
    --w;
    word = bits[w];
  }
}
// lowest set bit in [i, hi), -1: none
__host__ __device__ inline int find_next_bit(const uint32_t* bits, uint32_t i, uint32_t hi) {
  if (i >= hi) return -1;
  uint32_t w = i >> 5;
  const uint32_t whi = (hi - 1) >> 5;
  uint32_t word = bits[w] & (0xffffffffu << (i & 31u));
  while (true) {
    if (w == whi) word &= 0xffffffffu >> (31u - ((hi - 1) & 31u));
    if (word) return (int)(w * 32u + (uint32_t)__builtin_ffs((int)word) - 1u);
    if (w == whi) return -1;
    ++w;
    word = bits[w];
  }
}

// the unit whose leftover `u` starts on: the last unit before it that touched the stack (-1: u starts empty)
__host__ __device__ inline int handoff_pred(const Handoff& H, uint32_t u) {
  if (!H.row_chain) return find_prev_bit(H.touched, 0, u);
  uint32_t row = u / H.row_units, i = u;
  while (true) {
    const uint32_t lo = row * H.row_units;
    const int b = find_prev_bit(H.touched, lo, i);
    if (b >= 0) return b;
    if (H.row_chain[row] || row == 0) return -1;
    i = lo;
    --row;
  }
}
// the unit that starts on u's leftover (-1: nobody in this tile does)
__host__ __device__ inline int handoff_succ(const Handoff& H, uint32_t u) {
  if (!H.row_chain) return find_next_bit(H.touched, u + 1, H.n_units);
  uint32_t row = u / H.row_units, i = u + 1;
  while (true) {
    const uint32_t hi = (row + 1) * H.row_units;
    const int b = find_next_bit(H.touched, i, hi);
    if (b >= 0) return b;
    ++row;
    if (row >= H.rows || H.row_chain[row]) return -1;
    i = hi;
  }
}

__device__ __forceinline__ uint32_t leftover_at(const Handoff& H, uint32_t slot, uint32_t unit) {
  return H.dense ? (slot * H.n_units + unit) * H.cap : H.where[slot * H.n_units + unit];
}
// Room for the n entries a lane wants to leave in `slot` of `unit` (n = 0: none).  EVERY lane of the wave has to call this
// together (wave prefix sum, one atomic per wave).  Returns the first entry, kNoUnit if the pool is full (status raised).
// announce (pass 1): the units with n != 0 also go on the check list.
__device__ __forceinline__ uint32_t leftover_alloc(const Handoff& H, uint32_t n, uint32_t slot, uint32_t unit, uint32_t* status, bool announce) {
  if (H.dense) return (slot * H.n_units + unit) * H.cap;
  const uint32_t lane = threadIdx.x & 63u;
  if (announce && H.check_list) {
    const unsigned long long m = __ballot(n != 0);
    uint32_t first = 0;
    if (lane == 63 && m) first = atomicAdd(H.check_n, (uint32_t)__popcll(m));
    first = __shfl(first, 63, 64);
    if (n != 0) H.check_list[first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = unit;  // (every unit at most once: room for all)
  }
  uint32_t incl = n;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o, 64);
    if ((int)lane >= o) incl += v;
  }
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t base = 0;
  if (lane == 63 && total) base = atomicAdd(H.pool_top, total);
  base = __shfl(base, 63, 64);
  const uint32_t at = base + incl - n;
  if (n == 0) return kNoUnit;
  if ((unsigned long long)at + n > H.pool_cap) {
    atomicOr(status, kHoErrLeftoverCap);
    return kNoUnit;
  }
  H.where[slot * H.n_units + unit] = at;
  return at;
}

__device__ __forceinline__ bool handoff_touched(const Handoff& H, uint32_t u) { return (H.touched[u >> 5] >> (u & 31u)) & 1u; }

// `status`: the scene's error word (never cleared by a kernel; p3d_scene_status and every call with `stats` read it), so
// that an overflow is not lost on a caller that renders into device buffers without asking for statistics.
__device__ __forceinline__ void handoff_append(uint4* list, uint32_t* n, uint32_t cap, uint32_t* status, uint4 e) {
  const uint32_t i = atomicAdd(n, 1u);
  if (i < cap) list[i] = e;
  else atomicOr(status, kHoErrList);
}

}  // namespace p3d
