// jpeg_decode.cpp — baseline JPEG -> 8-bit RGB, for Scene::LoadSkybox (scene.cpp:329-377 loads `<dir>/*.jpg` through
// DevIL's ilLoadImage + ilConvertImage(IL_RGB, IL_UNSIGNED_BYTE); the image has neither DevIL nor libjpeg headers).
//
// What is decoded: sequential DCT, Huffman coding, 8-bit samples (SOF0 / SOF1), one component (grey) or three (YCbCr),
// luma sampling 1x1, 2x1 or 2x2 over 1x1 chroma, restart intervals.  Not decoded: progressive / arithmetic / lossless /
// 12-bit frames, CMYK - the call fails and says which.
//
// Which decoder, bit for bit: DevIL's version is unpinned in the reference, and it hands the file to
// the IJG library.  This file follows the IJG library's documented default decompression path so that the texels are
// the ones that library family produces (PIL's libjpeg-turbo is bit-compatible with it and is what the Python binding
// and the committed fixtures use; tests/test_host_logic.py compares the six shipped faces byte for byte):
//   * the accurate integer inverse DCT (13-bit constants, two passes with 2 extra bits after the first),
//   * "fancy" chroma upsampling: triangle filter, 3/4 of the nearer and 1/4 of the further sample in each direction,
//     rounding alternating between +8 and +7 (2x2) or +1 and +2 (2x1), edge samples replicated,
//   * YCbCr -> RGB with 16-bit fixed-point factors 1.40200, 0.34414, 0.71414, 1.77200.
#include "jpeg_decode.hpp"

#include <cstring>

namespace p3d {
namespace {

struct Huff {
  // canonical code lengths 1..16: first code, first value index and count per length; values in code order
  int32_t mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  uint8_t look_len[256], look_val[256];  // the codes of up to 8 bits, by the next 8 bits of the stream
  bool present = false;
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int dc_pred = 0;
  uint32_t w = 0, hgt = 0;          // plane size in samples, padded to whole blocks of whole MCUs
  std::vector<uint8_t> plane;
};

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t bits = 0;
  int n_bits = 0;
  bool hit_marker = false;
  // entropy-coded bytes: 0xFF 0x00 is a data byte 0xFF; any other marker ends the segment (zeros are fed from then on)
  void fill() {
    while (n_bits <= 24) {
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;
          else { hit_marker = true; b = 0; }
        } else {
          ++p;
        }
      }
      bits |= b << (24 - n_bits);
      n_bits += 8;
    }
  }
  int get(int n) {  // n <= 16
    if (n == 0) return 0;
    if (n_bits < n) fill();
    const int v = (int)(bits >> (32 - n));
    bits <<= n;
    n_bits -= n;
    return v;
  }
  int peek8() {
    if (n_bits < 8) fill();
    return (int)(bits >> 24);
  }
  void skip(int n) { bits <<= n; n_bits -= n; }
  void reset() { bits = 0; n_bits = 0; hit_marker = false; }
};

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool build_huff(Huff& H, const uint8_t counts[16], const uint8_t* vals, int n_vals) {
  int code = 0, k = 0;
  std::memset(H.look_len, 0, sizeof(H.look_len));
  for (int len = 1; len <= 16; ++len) {
    H.valptr[len] = k;
    H.mincode[len] = code;
    for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
      if (k >= n_vals || k >= 256) return false;
      H.vals[k] = vals[k];
      if (len <= 8) {
        const int lo = code << (8 - len), hi = lo + (1 << (8 - len));
        if (hi > 256) return false;
        for (int j = lo; j < hi; ++j) { H.look_len[j] = (uint8_t)len; H.look_val[j] = vals[k]; }
      }
    }
    H.maxcode[len] = counts[len - 1] ? code - 1 : -1;
    if (code > (1 << len)) return false;
    code <<= 1;
  }
  H.maxcode[17] = 0x7fffffff;
  H.present = true;
  return true;
}

inline int decode_symbol(Reader& R, const Huff& H) {
  const int look = R.peek8();
  if (H.look_len[look]) {
    R.skip(H.look_len[look]);
    return H.look_val[look];
  }
  int code = R.get(8);
  for (int len = 9; len <= 16; ++len) {
    code = (code << 1) | R.get(1);
    if (H.maxcode[len] >= 0 && code <= H.maxcode[len] && code >= H.mincode[len]) return H.vals[H.valptr[len] + code - H.mincode[len]];
  }
  return -1;
}

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// The accurate integer inverse DCT: factored 1-D transform (12 multiplies) on columns, then on rows.
constexpr int kConstBits = 13, kPass1Bits = 2;
constexpr int32_t F0_298631336 = 2446, F0_390180644 = 3196, F0_541196100 = 4433, F0_765366865 = 6270, F0_899976223 = 7373,
                  F1_175875602 = 9633, F1_501321110 = 12299, F1_847759065 = 15137, F1_961570560 = 16069, F2_053119869 = 16819,
                  F2_562915447 = 20995, F3_072711026 = 25172;
inline int32_t descale(int32_t x, int n) { return (x + ((int32_t)1 << (n - 1))) >> n; }

void idct_block(const int32_t* in /* dequantised, natural order */, uint8_t* out, size_t stride) {
  int32_t ws[64];
  for (int c = 0; c < 8; ++c) {
    const int32_t* s = in + c;
    int32_t* w = ws + c;
    if ((s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56]) == 0) {
      const int32_t dc = s[0] * ((int32_t)1 << kPass1Bits);
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    int32_t z2 = s[16], z3 = s[48];
    int32_t z1 = (z2 + z3) * F0_541196100;
    int32_t tmp2 = z1 + z3 * (-F1_847759065);
    int32_t tmp3 = z1 + z2 * F0_765366865;
    z2 = s[0]; z3 = s[32];
    int32_t tmp0 = (z2 + z3) * ((int32_t)1 << kConstBits);
    int32_t tmp1 = (z2 - z3) * ((int32_t)1 << kConstBits);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = s[56]; tmp1 = s[40]; tmp2 = s[24]; tmp3 = s[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * F1_175875602;
    tmp0 *= F0_298631336; tmp1 *= F2_053119869; tmp2 *= F3_072711026; tmp3 *= F1_501321110;
    z1 *= -F0_899976223; z2 *= -F2_562915447; z3 *= -F1_961570560; z4 *= -F0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    w[0] = descale(tmp10 + tmp3, kConstBits - kPass1Bits);  w[56] = descale(tmp10 - tmp3, kConstBits - kPass1Bits);
    w[8] = descale(tmp11 + tmp2, kConstBits - kPass1Bits);  w[48] = descale(tmp11 - tmp2, kConstBits - kPass1Bits);
    w[16] = descale(tmp12 + tmp1, kConstBits - kPass1Bits); w[40] = descale(tmp12 - tmp1, kConstBits - kPass1Bits);
    w[24] = descale(tmp13 + tmp0, kConstBits - kPass1Bits); w[32] = descale(tmp13 - tmp0, kConstBits - kPass1Bits);
  }
  constexpr int kOut = kConstBits + kPass1Bits + 3;
  for (int r = 0; r < 8; ++r) {
    const int32_t* w = ws + 8 * r;
    uint8_t* o = out + (size_t)r * stride;
    int32_t z2 = w[2], z3 = w[6];
    int32_t z1 = (z2 + z3) * F0_541196100;
    int32_t tmp2 = z1 + z3 * (-F1_847759065);
    int32_t tmp3 = z1 + z2 * F0_765366865;
    int32_t tmp0 = (w[0] + w[4]) * ((int32_t)1 << kConstBits);
    int32_t tmp1 = (w[0] - w[4]) * ((int32_t)1 << kConstBits);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * F1_175875602;
    tmp0 *= F0_298631336; tmp1 *= F2_053119869; tmp2 *= F3_072711026; tmp3 *= F1_501321110;
    z1 *= -F0_899976223; z2 *= -F2_562915447; z3 *= -F1_961570560; z4 *= -F0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = clamp8(descale(tmp10 + tmp3, kOut) + 128); o[7] = clamp8(descale(tmp10 - tmp3, kOut) + 128);
    o[1] = clamp8(descale(tmp11 + tmp2, kOut) + 128); o[6] = clamp8(descale(tmp11 - tmp2, kOut) + 128);
    o[2] = clamp8(descale(tmp12 + tmp1, kOut) + 128); o[5] = clamp8(descale(tmp12 - tmp1, kOut) + 128);
    o[3] = clamp8(descale(tmp13 + tmp0, kOut) + 128); o[4] = clamp8(descale(tmp13 - tmp0, kOut) + 128);
  }
}

inline uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

}  // namespace

bool jpeg_decode_rgb(const uint8_t* data, size_t size, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height, std::string& err) {
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { err = "not a JPEG file (no SOI marker)"; return false; }
  uint16_t quant[4][64];
  bool have_q[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Component comp[3];
  int n_comp = 0, hmax = 1, vmax = 1, restart_interval = 0;
  bool have_frame = false;
  size_t pos = 2;
  while (true) {
    if (pos + 4 > size) { err = "truncated before the scan"; return false; }
    if (data[pos] != 0xFF) { err = "marker expected"; return false; }
    const uint8_t m = data[pos + 1];
    if (m == 0xFF) { ++pos; continue; }  // fill byte
    pos += 2;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (m == 0xD9) { err = "end of image before any scan"; return false; }
    const size_t len = be16(data + pos);
    if (len < 2 || pos + len > size) { err = "bad segment length"; return false; }
    const uint8_t* seg = data + pos + 2;
    const size_t n = len - 2;
    if (m == 0xDB) {  // DQT
      for (size_t i = 0; i < n;) {
        const int pq = seg[i] >> 4, tq = seg[i] & 15;
        ++i;
        if (tq > 3 || pq > 1 || i + (pq ? 128u : 64u) > n) { err = "bad quantisation table"; return false; }
        for (int k = 0; k < 64; ++k, i += pq ? 2 : 1) quant[tq][kZigzag[k]] = pq ? be16(seg + i) : seg[i];
        have_q[tq] = true;
      }
    } else if (m == 0xC4) {  // DHT
      for (size_t i = 0; i < n;) {
        if (i + 17 > n) { err = "bad Huffman table"; return false; }
        const int tc = seg[i] >> 4, th = seg[i] & 15;
        int total = 0;
        for (int k = 0; k < 16; ++k) total += seg[i + 1 + k];
        if (tc > 1 || th > 3 || total > 256 || i + 17 + total > n) { err = "bad Huffman table"; return false; }
        if (!build_huff(tc ? ac[th] : dc[th], seg + i + 1, seg + i + 17, total)) { err = "inconsistent Huffman table"; return false; }
        i += 17 + total;
      }
    } else if (m == 0xC0 || m == 0xC1) {  // baseline / extended sequential, Huffman
      if (n < 6 || seg[0] != 8) { err = "only 8-bit samples are decoded"; return false; }
      height = be16(seg + 1); width = be16(seg + 3); n_comp = seg[5];
      if (width == 0 || height == 0) { err = "empty image"; return false; }
      if ((n_comp != 1 && n_comp != 3) || n < 6 + 3u * n_comp) { err = "only grey and YCbCr images are decoded (1 or 3 components)"; return false; }
      for (int c = 0; c < n_comp; ++c) {
        comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c];
        if (comp[c].tq > 3) { err = "bad frame header"; return false; }
      }
      hmax = comp[0].h; vmax = comp[0].v;
      if (hmax < 1 || hmax > 2 || vmax < 1 || vmax > 2 || (vmax == 2 && hmax == 1)) { err = "unsupported luma sampling factors (1x1, 2x1, 2x2 are decoded)"; return false; }
      for (int c = 1; c < n_comp; ++c)
        if (comp[c].h != 1 || comp[c].v != 1) { err = "unsupported chroma sampling factors (only 1x1 chroma is decoded)"; return false; }
      if (n_comp == 1) { comp[0].h = comp[0].v = hmax = vmax = 1; }  // a single-component scan is never interleaved
      have_frame = true;
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      err = m == 0xC2 ? "progressive JPEG is not decoded (baseline only)" : "this JPEG coding process is not decoded (baseline only)";
      return false;
    } else if (m == 0xDD) {
      if (n < 2) { err = "bad DRI"; return false; }
      restart_interval = be16(seg);
    } else if (m == 0xDA) {  // SOS: the one scan of a baseline file with all components
      if (!have_frame) { err = "scan before frame header"; return false; }
      if (n < 1 || seg[0] != n_comp || n < 1 + 2u * n_comp + 3) { err = "only single-scan (fully interleaved) baseline files are decoded"; return false; }
      for (int c = 0; c < n_comp; ++c) {
        int which = -1;
        for (int k = 0; k < n_comp; ++k) if (comp[k].id == seg[1 + 2 * c]) which = k;
        if (which != c) { err = "scan components out of order"; return false; }
        comp[c].td = seg[2 + 2 * c] >> 4; comp[c].ta = seg[2 + 2 * c] & 15;
        if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].present || !ac[comp[c].ta].present || !have_q[comp[c].tq]) { err = "scan refers to a missing table"; return false; }
      }
      pos += len;
      break;
    }
    pos += len;
  }

  // ---- entropy-coded segment: MCU by MCU into the component planes ----
  const uint32_t mcu_w = 8u * hmax, mcu_h = 8u * vmax;
  const uint32_t mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
  for (int c = 0; c < n_comp; ++c) {
    comp[c].w = mcus_x * 8u * comp[c].h;
    comp[c].hgt = mcus_y * 8u * comp[c].v;
    comp[c].plane.assign((size_t)comp[c].w * comp[c].hgt, 0);
    comp[c].dc_pred = 0;
  }
  Reader R{data + pos, data + size};
  int until_restart = restart_interval;
  int32_t block[64];
  for (uint32_t my = 0; my < mcus_y; ++my) {
    for (uint32_t mx = 0; mx < mcus_x; ++mx) {
      if (restart_interval && until_restart == 0) {  // RSTn: byte-align, skip the marker, reset the predictors
        R.reset();
        while (R.p + 1 < R.end && !(R.p[0] == 0xFF && R.p[1] >= 0xD0 && R.p[1] <= 0xD7)) ++R.p;
        if (R.p + 1 < R.end) R.p += 2;
        for (int c = 0; c < n_comp; ++c) comp[c].dc_pred = 0;
        until_restart = restart_interval;
      }
      for (int c = 0; c < n_comp; ++c) {
        Component& C = comp[c];
        const uint16_t* q = quant[C.tq];
        for (int by = 0; by < C.v; ++by)
          for (int bx = 0; bx < C.h; ++bx) {
            std::memset(block, 0, sizeof(block));
            int s = decode_symbol(R, dc[C.td]);
            if (s < 0 || s > 11) { err = "corrupt entropy-coded data (DC)"; return false; }
            if (s) C.dc_pred += extend(R.get(s), s);
            block[0] = C.dc_pred * q[0];
            for (int k = 1; k < 64;) {
              const int rs = decode_symbol(R, ac[C.ta]);
              if (rs < 0) { err = "corrupt entropy-coded data (AC)"; return false; }
              const int run = rs >> 4, sz = rs & 15;
              if (sz == 0) {
                if (run == 15) { k += 16; continue; }
                break;  // end of block
              }
              k += run;
              if (k > 63) { err = "corrupt entropy-coded data (run past the block)"; return false; }
              block[kZigzag[k]] = extend(R.get(sz), sz) * q[kZigzag[k]];
              ++k;
            }
            const size_t x0 = ((size_t)mx * C.h + bx) * 8, y0 = ((size_t)my * C.v + by) * 8;
            idct_block(block, &C.plane[y0 * C.w + x0], C.w);
          }
      }
      if (restart_interval) --until_restart;
    }
  }

  // ---- upsample the chroma planes ("fancy": triangle filter) and convert ----
  rgb.assign((size_t)width * height * 3, 0);
  if (n_comp == 1) {
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x) {
        const uint8_t v = comp[0].plane[(size_t)y * comp[0].w + x];
        uint8_t* o = &rgb[((size_t)y * width + x) * 3];
        o[0] = o[1] = o[2] = v;
      }
    return true;
  }
  // the chroma samples that exist for this image (the planes are padded to whole MCUs; the filter replicates the image's own
  // last sample, not the padding)
  const uint32_t cw = (width + hmax - 1) / hmax, ch = (height + vmax - 1) / vmax;
  std::vector<uint8_t> up[2];
  for (int c = 1; c <= 2; ++c) {
    const Component& C = comp[c];
    std::vector<uint8_t>& U = up[c - 1];
    U.assign((size_t)width * height, 0);
    auto in = [&](uint32_t x, uint32_t y) -> int { return C.plane[(size_t)y * C.w + x]; };
    if (hmax == 1 && vmax == 1) {
      for (uint32_t y = 0; y < height; ++y) std::memcpy(&U[(size_t)y * width], &C.plane[(size_t)y * C.w], width);
    } else if (hmax == 2 && vmax == 1) {
      for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < cw; ++x) {
          const int cur = in(x, y), left = in(x ? x - 1 : 0, y), right = in(x + 1 < cw ? x + 1 : cw - 1, y);
          const int a = cw == 1 || x == 0 ? cur : (cur * 3 + left + 1) >> 2;
          const int b = cw == 1 || x + 1 == cw ? cur : (cur * 3 + right + 2) >> 2;
          if (2 * x < width) U[(size_t)y * width + 2 * x] = (uint8_t)a;
          if (2 * x + 1 < width) U[(size_t)y * width + 2 * x + 1] = (uint8_t)b;
        }
      }
    } else {  // 2x2
      for (uint32_t y = 0; y < height; ++y) {
        const uint32_t cy = y >> 1;
        const uint32_t other = (y & 1) ? (cy + 1 < ch ? cy + 1 : ch - 1) : (cy ? cy - 1 : 0);  // the further input row: above for even rows, below for odd
        for (uint32_t x = 0; x < cw; ++x) {
          const int cur = 3 * in(x, cy) + in(x, other);
          const int left = x ? 3 * in(x - 1, cy) + in(x - 1, other) : cur;
          const int right = x + 1 < cw ? 3 * in(x + 1, cy) + in(x + 1, other) : cur;
          const int a = x == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + left + 8) >> 4;
          const int b = x + 1 == cw ? (cur * 4 + 7) >> 4 : (cur * 3 + right + 7) >> 4;
          if (2 * x < width) U[(size_t)y * width + 2 * x] = (uint8_t)a;
          if (2 * x + 1 < width) U[(size_t)y * width + 2 * x + 1] = (uint8_t)b;
        }
      }
    }
  }
  auto fix = [](double v) { return (int32_t)(v * 65536.0 + 0.5); };
  const int32_t f140 = fix(1.40200), f177 = fix(1.77200), f071 = fix(0.71414), f034 = fix(0.34414), half = 1 << 15;
  for (uint32_t y = 0; y < height; ++y)
    for (uint32_t x = 0; x < width; ++x) {
      const int Y = comp[0].plane[(size_t)y * comp[0].w + x];
      const int cb = (int)up[0][(size_t)y * width + x] - 128, cr = (int)up[1][(size_t)y * width + x] - 128;
      uint8_t* o = &rgb[((size_t)y * width + x) * 3];
      o[0] = clamp8(Y + ((f140 * cr + half) >> 16));
      o[1] = clamp8(Y + ((-f034 * cb + half - f071 * cr) >> 16));
      o[2] = clamp8(Y + ((f177 * cb + half) >> 16));
    }
  return true;
}

}  // namespace p3d
