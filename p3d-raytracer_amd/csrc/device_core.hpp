// device_core.hpp — CDNA4 device code shared by the Whitted and path-tracing kernels:
// float3 arithmetic with the reference's exact operation order, the shape tests, and the
// three traversal back ends.  gfx950 only; compiled with -ffp-contract=off so that no
// multiply-add is fused: every + - * / sqrt below rounds exactly where
// the reference's scalar code rounds.
//
// Conventions
//   * A wavefront (64 lanes) is one packet of rays = an 8x8 pixel block (path tracer from 16 spp:
//     a 4x4 block, four lanes per pixel).  Traversal is
//     PER LANE (own current node, own stack): the reference's any-hit traversal is
//     order-dependent, so lanes cannot share a stack.  The wave as a
//     whole iterates `while any lane is still walking`.
//   * The per-lane node stack lives in LDS, entry e of lane l at  stack[e * kBlock + l]
//     (8 bytes: node index, entry distance).  A wave's 64 lanes then cover all 64 banks
//     exactly twice per ds_read_b64/ds_write_b64 whatever their individual depths are,
//     i.e. the access is bank-conflict free by construction.
//   * Scenes of up to 26 KB (most packaged .p3f) are staged into LDS once per workgroup; large
//     ones (100k triangles = 8.8 MB) are read through L1/L2 from the linearised arrays.
//   * Mixed-precision literals of the reference are folded into float comparisons where
//     that is exact (see the cmp_* helpers); real double arithmetic stays double.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "p3d.h"

namespace p3d {

constexpr int kBlock = 64;  // one wavefront per workgroup

// ---------------------------------------------------------------------------
// float3 with the operation order of vector.cpp
// ---------------------------------------------------------------------------
struct F3 {
  float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float f) { return f3(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ F3 operator/(F3 a, float f) { return f3(a.x / f, a.y / f, a.z / f); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }  // Color*Color
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vector.cpp:55
__device__ __forceinline__ F3 cross(F3 u, F3 v) {                                               // vector.cpp:84-99
  return f3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ __forceinline__ float length(F3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
// vector.cpp:65-70: l = 1.0 / length() is a double division narrowed to float.  For a
// float divisor that equals the correctly rounded float quotient 1.0f / len (53 >= 2*24+2
// bits: double rounding is innocuous), which is what IEEE-exact v_div gives.
__device__ __forceinline__ F3 normalized(F3 a) {
  const float l = 1.0f / length(a);
  return f3(a.x * l, a.y * l, a.z * l);
}
__device__ __forceinline__ bool same_bits(F3 a, F3 b) {
  return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
         __float_as_uint(a.z) == __float_as_uint(b.z);
}
__device__ __forceinline__ F3 xyz(float4 v) { return f3(v.x, v.y, v.z); }

// color.h:10,39-44 CLAMP(0.0, v, 1.0): NaN passes through
__device__ __forceinline__ float clamp01(float v) { return (v < 0.0f) ? 0.0f : ((v > 1.0f) ? 1.0f : v); }
__device__ __forceinline__ F3 clamp01(F3 c) { return f3(clamp01(c.x), clamp01(c.y), clamp01(c.z)); }

// Comparisons of a float against a DOUBLE literal, folded to float exactly:
//   (double)t <  0.0001  <=>  t <= 0.0001f   (0.0001f is the float just below 1e-4)
//   (double)t >  0.0001  <=>  t >  0.0001f
//   (double)a >  0.1     <=>  a >= 0.1f      (0.1f is the float just above 0.1)
__device__ __forceinline__ bool lt_1em4(float t) { return t <= 0.0001f; }
__device__ __forceinline__ bool gt_1em4(float t) { return t > 0.0001f; }
__device__ __forceinline__ bool gt_0p1(float a) { return a >= 0.1f; }

// scene.h:17-22 MIN3 / MAX3 as written (select chains; NaN-order sensitive, unlike v_max3)
__device__ __forceinline__ float max3_ref(float a, float b, float c) {
  return (a > b) ? ((a > c) ? a : c) : ((b > c) ? b : c);
}
__device__ __forceinline__ float min3_ref(float a, float b, float c) {
  return (a < b) ? ((a < c) ? a : c) : ((b < c) ? b : c);
}


// ---------------------------------------------------------------------------
// pow(max(0, H.N), shine) of main.cpp:224.  The reference evaluates libm pow in double and
// narrows the product to float, so only ~1e-8 relative accuracy is observable.  ocml's
// fully accurate double pow costs ~36 VGPRs of peak pressure and a few hundred instructions
// in the light loop.  Round 4: table-driven, about 45 double operations instead of the 120 of the series version it replaces
// (an ablation build without the call had shown it to be a tenth of the bench kernel):
//   log x  = e ln2 + log(c_j) + log1p(r),  x = 2^e m, m in [1, 2), j = the top six fraction bits of m, c_j the middle of that
//            interval (halved, with e + 1, from m = 1.5 on, so that x just below 1 has e = 0 and nothing cancels),
//            r = m / c_j - 1 as ONE fused multiply-add with the tabulated 1 / c_j (|r| < 1/128; the tabulated log is that of the
//            reciprocal as rounded, so the identity is exact), log1p(r) to the seventh power;
//   x^y    = 2^(k / 32) exp(r2),  t = y log x = k ln2 / 32 + r2 (Cody-Waite), 2^(j / 32) tabulated, exp(r2) to the sixth power.
// Same accuracy class as the version it replaces (both are bounded by the rounding of t: 1.4e-13 relative at the edge of the float
// range; 5e-15 where the result is visible): on 159 M shading-like inputs with visible results (cosines in [0, 1], the packaged
// shine values) both differ from libm's float-rounded value three times; tests/pow_spec_check.hip is the device-side check.
// Domain: x >= 0 (callers clamp), any finite y.
// ---------------------------------------------------------------------------
__device__ const double kPowLogTable[128] = {  // {1 / c_j as rounded, log(c_j) or log(c_j / 2)} for j = 0 .. 63
    0x1.fc07f01fc07f0p-1, 0x1.fe02a6b106799p-8, 0x1.f44659e4a4271p-1, 0x1.7b91b07d5b126p-6,
    0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd68p-5, 0x1.e573ac901e574p-1, 0x1.b42dd711971b9p-5,
    0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae3p-4, 0x1.d77b654b82c34p-1, 0x1.51b073f06183cp-4,
    0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b23p-4, 0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4,
    0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabcp-4, 0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a2p-3,
    0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f75p-3, 0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3,
    0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21bp-3, 0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3,
    0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b561p-3, 0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd4p-3,
    0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb94p-3, 0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5935p-3,
    0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2, 0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad4p-2,
    0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2, 0x1.7f405fd017f40p-1, 0x1.2895a13de86a4p-2,
    0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d56p-2, 0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2,
    0x1.724287f46debcp-1, 0x1.4be5f957778a1p-2, 0x1.6e1f76b4337c7p-1, 0x1.5767717455a6cp-2,
    0x1.6a13cd1537290p-1, 0x1.62c82f2b9c796p-2, 0x1.661ec6a5122f9p-1, 0x1.6e08eaa2ba1e4p-2,
    0x1.623fa77016240p-1, 0x1.792a55fdd47a1p-2, 0x1.5e75bb8d015e7p-1, 0x1.842d1da1e8b18p-2,
    0x1.5ac056b015ac0p-1, 0x1.8f11e873662c8p-2, 0x1.571ed3c506b3ap-1, 0x1.99d958117e08ap-2,
    0x1.5390948f40febp-1, -0x1.214456d0eb8d5p-2, 0x1.5015015015015p-1, -0x1.16b5ccbacfb73p-2,
    0x1.4cab88725af6ep-1, -0x1.0c42d676162e2p-2, 0x1.49539e3b2d067p-1, -0x1.01eae5626c691p-2,
    0x1.460cbc7f5cf9ap-1, -0x1.ef5ade4dcffe5p-3, 0x1.42d6625d51f87p-1, -0x1.db13db0d48941p-3,
    0x1.3fb013fb013fbp-1, -0x1.c6ffbc6f00f71p-3, 0x1.3c995a47babe7p-1, -0x1.b31d8575bce3bp-3,
    0x1.3991c2c187f63p-1, -0x1.9f6c407089663p-3, 0x1.3698df3de0748p-1, -0x1.8beafeb38fe8fp-3,
    0x1.33ae45b57bcb2p-1, -0x1.7898d85444c74p-3, 0x1.30d190130d190p-1, -0x1.6574ebe8c1339p-3,
    0x1.2e025c04b8097p-1, -0x1.527e5e4a1b58dp-3, 0x1.2b404ad012b40p-1, -0x1.3fb45a59928cap-3,
    0x1.288b01288b013p-1, -0x1.2d1610c86813dp-3, 0x1.25e22708092f1p-1, -0x1.1aa2b7e23f729p-3,
    0x1.23456789abcdfp-1, -0x1.08598b59e3a07p-3, 0x1.20b470c67c0d9p-1, -0x1.ec739830a1126p-4,
    0x1.1e2ef3b3fb874p-1, -0x1.c885801bc4b20p-4, 0x1.1bb4a4046ed29p-1, -0x1.a4e7640b1bc38p-4,
    0x1.19453808ca29cp-1, -0x1.8197e2f40e3f0p-4, 0x1.16e0689427379p-1, -0x1.5e95a4d9791cdp-4,
    0x1.1485f0e0acd3bp-1, -0x1.3bdf5a7d1ee5ep-4, 0x1.12358e75d3033p-1, -0x1.1973bd1465561p-4,
    0x1.0fef010fef011p-1, -0x1.eea31c006b87cp-5, 0x1.0db20a88f4696p-1, -0x1.aaef2d0fb1108p-5,
    0x1.0b7e6ec259dc8p-1, -0x1.67c94f2d4bb65p-5, 0x1.0953f39010954p-1, -0x1.252f32f8d1840p-5,
    0x1.073260a47f7c6p-1, -0x1.c63d2ec14aad7p-6, 0x1.05197f7d73404p-1, -0x1.432a925980cbcp-6,
    0x1.03091b51f5e1ap-1, -0x1.82448a388a283p-7, 0x1.0101010101010p-1, -0x1.0080559588b25p-8,
};
__device__ const double kPowExp2Table[32] = {  // 2^(j / 32)
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0,
};
__device__ __forceinline__ double pow_spec(double x, double y) {
  if (y == 0.0) return 1.0;
  if (!(x > 0.0)) return (x == 0.0) ? (y > 0.0 ? 0.0 : __longlong_as_double(0x7ff0000000000000LL)) : x;
  long long bits = __double_as_longlong(x);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  if (((bits >> 52) & 0x7ff) == 0) {  // subnormal: scaled into the normal range first
    bits = __double_as_longlong(x * 0x1p64);
    e = (int)((bits >> 52) & 0x7ff) - 1023 - 64;
  }
  const int j = (int)((bits >> 46) & 63);
  const double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);  // [1, 2)
  e += j >> 5;  // from m = 1.5 on the table holds log(c_j / 2)
  const double r = __builtin_fma(m, kPowLogTable[2 * j], -1.0);
  double p = 1.0 / 7;
  p = p * r - 1.0 / 6; p = p * r + 1.0 / 5; p = p * r - 1.0 / 4; p = p * r + 1.0 / 3; p = p * r - 0.5; p = p * r + 1.0;
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double t = y * ((double)e * ln2_hi + ((double)e * ln2_lo + (kPowLogTable[2 * j + 1] + r * p)));
  const double kd = floor(t * 0x1.71547652b82fep+5 + 0.5);  // 32 / ln2
  if (kd < -32000.0) return 0.0;
  if (kd > 32000.0) return __longlong_as_double(0x7ff0000000000000LL);
  const double r2 = (t - kd * 0x1.62e42fee00000p-6) - kd * 0x1.a39ef35793c76p-38;  // ln2 / 32, high part (21 trailing zero bits) and low part
  double q = 1.0 / 720.0;
  q = q * r2 + 1.0 / 120.0; q = q * r2 + 1.0 / 24.0; q = q * r2 + 1.0 / 6.0; q = q * r2 + 0.5; q = q * r2 + 1.0; q = q * r2 + 1.0;
  const long long k = (long long)kd;
  return (kPowExp2Table[k & 31] * q) * __longlong_as_double(((k >> 5) + 1023LL) << 52);
}

// ---------------------------------------------------------------------------
// Ray with the mutable direction of ray.h:16-18 (Q8).  `inv` caches 1.0f/d for the slab
// test (boundingBox.cpp:57,67,77 recompute it per test; the value is the same until the
// direction is re-normalised).  `settled` is set once a re-normalisation left the bits
// unchanged: normalize() is a pure function of d, so from then on it is the identity.
// ---------------------------------------------------------------------------
struct RayS {
  F3 o, d, inv;
  bool settled;
  bool odd_inv;  // some 1/d component is +-inf or NaN: the slab test can then produce NaNs (0 * inf)
};
// "odd": some 1/d component is +-inf or NaN (the slab test can then produce NaNs, 0 * inf), or smaller than 0.75 in
// magnitude (|d| > 1.33: never a ray of the renderer, whose directions are normalised; possible through p3d_trace_*).  For a
// wave without an odd lane the slab products keep the sign of their factors exactly (|x * inv| >= 0.75 |x| cannot round
// to zero for x != 0) and no NaN can arise: the fast paths of aabb_intercepts and bvh_closest rely on both.
__device__ __forceinline__ bool inv_is_odd(F3 inv) {
  const float ax = fabsf(inv.x), ay = fabsf(inv.y), az = fabsf(inv.z);
  return !(ax < INFINITY && ax >= 0.75f) || !(ay < INFINITY && ay >= 0.75f) || !(az < INFINITY && az >= 0.75f);
}
__device__ __forceinline__ void ray_set(RayS& r, F3 o, F3 d) {
  r.o = o;
  r.d = d;
  r.inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.settled = false;
  r.odd_inv = inv_is_odd(r.inv);
}
__device__ __forceinline__ F3 get_direction(RayS& r) {
  if (!r.settled) {
    const F3 n = normalized(r.d);
    if (same_bits(n, r.d)) {
      r.settled = true;
    } else {
      r.d = n;
      r.inv = f3(1.0f / n.x, 1.0f / n.y, 1.0f / n.z);
      r.odd_inv = inv_is_odd(r.inv);
    }
  }
  return r.d;
}

// ---------------------------------------------------------------------------
// Counters (only in the STATS instantiation)
// ---------------------------------------------------------------------------
enum StatSlot {
  kRaysPrimary = 0, kRaysShadow, kRaysReflect, kRaysRefract, kRaysBounce, kRaysLight,
  kNodeTests, kSphereTests, kTriTests, kBoxTests, kPlaneTests, kShadedHits, kPixels, kMaxStack,
  kNumStats
};
template <bool ON>
struct Counters {
  __device__ __forceinline__ void add(int, uint32_t = 1) {}
  __device__ __forceinline__ void stack_depth(int) {}
  __device__ __forceinline__ void clear() {}
  __device__ __forceinline__ uint32_t get(int) const { return 0; }
};
template <>
struct Counters<true> {
  uint32_t c[kNumStats];
  __device__ __forceinline__ void clear() {
    for (int i = 0; i < kNumStats; ++i) c[i] = 0;
  }
  __device__ __forceinline__ uint32_t get(int slot) const { return c[slot]; }
  __device__ __forceinline__ void add(int slot, uint32_t n = 1) { c[slot] += n; }
  __device__ __forceinline__ void stack_depth(int d) {
    if ((uint32_t)d > c[kMaxStack]) c[kMaxStack] = d;
  }
};

// ---------------------------------------------------------------------------
// Device scene.  All arrays are float4-granular so every fetch is a 16-byte load
// (global_load_dwordx4 / ds_read_b128).
//   nodes    : 2 x float4 per BVH node   {bmin.xyz, descriptor} {bmax.xyz, -}
//   bgeom    : 3 x float4 per BVH leaf slot (geometry gathered into leaf order)
//   ogeom    : 3 x float4 per object (object order: brute force + grid)
//              {v0..v3} {v4..v7} {v8, type | material << 8, object id, -}
//   normals  : 1 x float4 per object (triangle unit normal)
//   mats     : 4 x float4  {cd, Kd} {cs, Ks} {shine, T, ior, refl} {emission, -}
//   lights   : 2 x float4  {pos, -} {col, -}
// ---------------------------------------------------------------------------
struct DevCamera {
  F3 eye, u, v, n;
  float w, h, plane_dist, focal_ratio, aperture;
  int res_x, res_y;
};
struct DevGrid {
  F3 bmin, bmax;
  int nx, ny, nz;
  const uint32_t* cell_start;
  const uint32_t* cell_items;
};
struct DevScene {
  const float4* nodes;
  const float4* bgeom;
  const float4* ogeom;
  const float4* normals;
  const float4* mats;
  const float4* lights;
  const uint32_t* emitters;  // object ids of emissive spheres, object order (main.cpp:407-415)
  uint32_t n_nodes, n_slots, n_objs, n_mats, n_lights, n_emitters;
  uint32_t odd_boxes;  // some node box is not finite or has min > max: the slab fast paths are off for this scene
  DevCamera cam;
  F3 bg;
  DevGrid grid;
  // cubemap faces as RGBA8 words, row 0 = bottom row; null when no skybox was supplied
  const uint32_t* sky[6];
  uint32_t sky_w[6], sky_h[6];
};

// Per-lane traversal stack (bvh.cpp:86 hit_stack, one per pixel instead of one per process:
// see DESIGN.md "Sequential state").
//   SPILL == false: the host guarantees that the worst-case height fits the LDS part, and that node and leaf-slot indices
//                   are below 4096 (an LDS-staged scene: at most 26 KB).  An entry is SIX bytes: the distance in a dword
//                   array (tbase[e * kBlock]) and the descriptor packed into 16 bits (leaf bit, count, 12 index bits) in a
//                   halfword array behind it (dbase[e * kBlock]).  LDS bytes per workgroup decide how many waves of these
//                   kernels a CU holds (cfg2: 2 KB of scene + 16 entries; 8-byte entries = 10 KB = 16 workgroups per CU,
//                   6-byte entries = 20, and the frames-in-flight loop gains a quarter: experiments r03 §13).
//   SPILL == true : LDS holds a WINDOW of the `cap` (a power of two) most recent entries, entry e in slot e & (cap - 1);
//                   entries [lo, sp) are in the window, entries [0, lo) in a per-thread column of a global backing
//                   array.  A push into a full window moves the window's oldest entry out (it sits in the very slot the
//                   new entry takes); a pop below the window reads the backing array.  The traversals work at the top of the
//                   stack, so the backing array is only touched by what sinks out of the window and comes back: the stale
//                   leftovers of earlier any-hit queries (Q2) under a deep closest-hit traversal.  (Until round 3 the LDS
//                   part held the FIRST cap entries: under a leftover of a dozen entries every push and pop of the next
//                   query went to global memory; 100k triangles 2048x2048: 1.7 GB written per frame, profiles/r03.)
// The SPILL parameter of everything below: bool-compatible (false = kStackLds6, true = kStackWindow), plus a third form
constexpr int kStackLds6 = 0;    // whole stack in LDS, six-byte entries
constexpr int kStackWindow = 1;  // LDS window + global backing array, eight-byte entries
constexpr int kStackLds8 = 2;    // whole stack in LDS, eight-byte entries: kernels whose registers, not their LDS, set how many
                                 // waves a CU holds (the path tracer: 128 VGPRs) - one ds_*_b64 per access instead of two
                                 // narrower ones and the packing (Cornell box 1024x1024 256 spp: 144.2 -> 139.8 ms)
typedef __attribute__((address_space(3))) unsigned long long lds_uint2;  // explicit LDS pointer to one 8-byte entry: ds_read/write_b64, never flat_*
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
struct Stack {
  lds_uint2* base;  // SPILL: &window[lane]; slot s at base[s * kBlock]
  lds_u16* dbase;   // !SPILL: &desc[lane]; packed descriptor of entry e at dbase[e * kBlock] (LDS address D0 + 2 lane + 128 e) ...
  uint32_t tfix;    // ... and its distance at LDS address 2 * (that address) + tfix = T0 + 4 lane + 256 e (tfix = T0 - 2 D0 is
                    // wave-uniform: one per-lane address register for both arrays)
  uint2* spill;     // global backing array (wave-uniform base: scalar registers); entry e of this thread at spill[e * spill_stride + tid]
  uint32_t spill_stride, tid;  // (a 32-bit element offset from a uniform base costs one address register, a per-lane pointer two)
  int sp;
  int lo;           // SPILL only: first entry held in the LDS window
  int cap;          // LDS entries per lane (SPILL: a power of two)
};
__device__ __forceinline__ lds_uint2* lds_stack_ptr(float4* smem_base, size_t float4_offset, uint32_t lane) {
  return (lds_uint2*)(reinterpret_cast<unsigned long long*>(smem_base + float4_offset)) + lane;
}
// LDS of one workgroup's node stacks, in float4 units
__host__ __device__ constexpr uint32_t stack_lds_f4(int mode, uint32_t cap) { return cap * kBlock * (mode == kStackLds6 ? 6u : 8u) / 16u; }
// the 16-bit form of a descriptor (indices < 4096) and back
__device__ __forceinline__ uint32_t pack_desc16(uint32_t desc) { return desc | (desc >> 16); }  // low half (bits 27..12 are zero: the flags land in bits 15..12)
__device__ __forceinline__ uint32_t unpack_desc16(uint32_t pk) { return ((pk & 0xf000u) << 16) | (pk & 0x0fffu); }
__device__ __forceinline__ void stack_clear(Stack& s) { s.sp = 0; s.lo = 0; }
__device__ __forceinline__ lds_u32* stack_t_of(const Stack& s, const lds_u16* d) {  // (!SPILL) where the distance of the entry with descriptor *d lives
  return (lds_u32*)(uintptr_t)(2u * (uint32_t)(uintptr_t)d + s.tfix);
}
// the stacks of a workgroup start `float4_offset` float4s into its LDS; `cap` entries per lane
__device__ __forceinline__ void stack_bind(Stack& s, float4* smem_base, size_t float4_offset, uint32_t lane, int cap, uint2* backing,
                                           uint32_t backing_stride, uint32_t tid) {
  s.base = lds_stack_ptr(smem_base, float4_offset, lane);
  lds_u32* t0 = (lds_u32*)(reinterpret_cast<uint32_t*>(smem_base + float4_offset));  // distances first, descriptors behind them
  lds_u16* d0 = (lds_u16*)(t0 + (size_t)cap * kBlock);
  s.dbase = d0 + lane;
  s.tfix = (uint32_t)(uintptr_t)t0 - 2u * (uint32_t)(uintptr_t)d0;
  s.spill = backing;
  s.spill_stride = backing_stride;
  s.tid = tid;
  s.cap = cap;
  stack_clear(s);
}
__device__ __forceinline__ uint2 unpack_entry(unsigned long long v) { return make_uint2((uint32_t)v, (uint32_t)(v >> 32)); }
template <int SPILL, class CT>
__device__ __forceinline__ void push(Stack& s, uint32_t node, float t, CT& ct) {
  if (SPILL == kStackLds6) {
    lds_u16* d = s.dbase + s.sp * kBlock;
    *d = (uint16_t)pack_desc16(node);
    *stack_t_of(s, d) = __float_as_uint(t);
  } else if (SPILL == kStackLds8) {
    *(s.base + s.sp * kBlock) = (unsigned long long)node | ((unsigned long long)__float_as_uint(t) << 32);
  } else {
    const unsigned long long e = (unsigned long long)node | ((unsigned long long)__float_as_uint(t) << 32);
    lds_uint2* p = s.base + (s.sp & (s.cap - 1)) * kBlock;
    if (s.sp - s.lo == s.cap) {  // window full: its oldest entry lives in this slot and moves to the backing array
      s.spill[(uint32_t)s.lo * s.spill_stride + s.tid] = unpack_entry(*p);
      ++s.lo;
#ifdef P3D_ABL_COUNT_SPILLS  // (ablation builds only: entries that left the window, counted in the box-test slot)
      ct.add(kBoxTests);
#endif
    }
    *p = e;
  }
  ++s.sp;
  ct.stack_depth(s.sp);
}
// entry i < sp, wherever it lives; the stack is not changed
template <int SPILL>
__device__ __forceinline__ uint2 stack_read(const Stack& s, int i) {
  uint2 e;
  if (SPILL == kStackLds6) {
    const lds_u16* d = s.dbase + i * kBlock;
    e = make_uint2(unpack_desc16(*d), *stack_t_of(s, d));
  } else if (SPILL == kStackLds8) {
    e = unpack_entry(*(s.base + i * kBlock));
  } else if (i >= s.lo) {
    e = unpack_entry(*(s.base + (i & (s.cap - 1)) * kBlock));
    asm volatile("" : "+v"(e.x), "+v"(e.y));  // keep the two address spaces apart (no flat_load)
  } else {
    e = s.spill[(uint32_t)i * s.spill_stride + s.tid];
    asm volatile("" : "+v"(e.x), "+v"(e.y));
  }
  return e;
}
// --sp and the entry that was on top
template <int SPILL>
__device__ __forceinline__ uint2 pop(Stack& s) {
  --s.sp;
  const uint2 e = stack_read<SPILL>(s, s.sp);
  if (SPILL == kStackWindow && s.sp < s.lo) s.lo = s.sp;  // popped below the window: the window is empty now
  return e;
}

// Shading state of a Whitted level that is written once and read once or twice per level but would otherwise sit in
// registers through every traversal of the level: the hit point (only a refraction needs it again) and the diffuse / specular
// sums of the light loop.  ON: nine dwords per lane in LDS behind the node stack (ds_read / ds_write around the light loop's
// bookkeeping) for the kernels that trade registers for waves (scenes traversed from L2: the register allocator spilled
// these very values to scratch memory, 1.4 GB of HBM writes per 2048x2048 frame of 100k triangles, profiles/r03).
// OFF: plain registers.
typedef __attribute__((address_space(3))) float lds_float;
template <bool ON>
struct ColdState {
  F3 pn_, diff_, spec_;
  __device__ __forceinline__ void bind(float4*, size_t, uint32_t) {}
  __device__ __forceinline__ void set_pn(F3 v) { pn_ = v; }
  __device__ __forceinline__ F3 pn() const { return pn_; }
  __device__ __forceinline__ void clear_sums() { diff_ = f3(0, 0, 0); spec_ = f3(0, 0, 0); }
  __device__ __forceinline__ void add(F3 d, F3 s) { diff_ = diff_ + d; spec_ = spec_ + s; }
  __device__ __forceinline__ F3 diff() const { return diff_; }
  __device__ __forceinline__ F3 spec() const { return spec_; }
};
template <>
struct ColdState<true> {
  lds_float* p;  // &cold[0][lane]; dword k at p[k * kBlock]
  __device__ __forceinline__ void bind(float4* smem_base, size_t float4_offset, uint32_t lane) {
    p = (lds_float*)(reinterpret_cast<float*>(smem_base + float4_offset)) + lane;
  }
  __device__ __forceinline__ void put(int k, F3 v) { p[k * kBlock] = v.x; p[(k + 1) * kBlock] = v.y; p[(k + 2) * kBlock] = v.z; }
  __device__ __forceinline__ F3 get(int k) const { return f3(p[k * kBlock], p[(k + 1) * kBlock], p[(k + 2) * kBlock]); }
  __device__ __forceinline__ void set_pn(F3 v) { put(0, v); }
  __device__ __forceinline__ F3 pn() const { return get(0); }
  __device__ __forceinline__ void clear_sums() { put(3, f3(0, 0, 0)); put(6, f3(0, 0, 0)); }
  __device__ __forceinline__ void add(F3 d, F3 s) { put(3, get(3) + d); put(6, get(6) + s); }
  __device__ __forceinline__ F3 diff() const { return get(3); }
  __device__ __forceinline__ F3 spec() const { return get(6); }
};
constexpr uint32_t kColdDwords = 9;

struct Geom {
  float4 a, b, c;
};
__device__ __forceinline__ Geom load_geom(const float4* g, uint32_t slot) {
  Geom r;
  r.a = g[3 * slot];
  r.b = g[3 * slot + 1];
  r.c = g[3 * slot + 2];
  return r;
}
__device__ __forceinline__ uint32_t geom_type(const Geom& g) { return __float_as_uint(g.c.y) & 0xffu; }
__device__ __forceinline__ uint32_t geom_material(const Geom& g) { return __float_as_uint(g.c.y) >> 8; }
__device__ __forceinline__ uint32_t geom_object(const Geom& g) { return __float_as_uint(g.c.z); }

// boundingBox.cpp:44-98.  Raw direction; t0 < t1 strict; t1 > 0.0001 (double literal).
// MAX3/MIN3 (scene.h:17-22) are select chains whose result depends on operand order when a
// NaN is present (0 * inf at an axis-parallel ray touching a slab plane).  `all_finite` is a
// WAVE-UNIFORM promise that no lane's 1/d has an inf/NaN component: then no NaN can arise,
// the chains equal plain max/min and one v_max3_f32 / v_min3_f32 each replaces 12 compares
// and selects (the sign of a zero result is never observable: only orderings of t matter).
// t0 / t1 (optional): the slab interval itself, for bvh_closest's fast isInside.
__device__ __forceinline__ bool aabb_intercepts(F3 mn, F3 mx, const RayS& r, float& t, bool all_finite, float* t0_out = nullptr,
                                                float* t1_out = nullptr) {
  const float a = r.inv.x, b = r.inv.y, c = r.inv.z;
  const float tx_min = ((a >= 0 ? mn.x : mx.x) - r.o.x) * a, tx_max = ((a >= 0 ? mx.x : mn.x) - r.o.x) * a;
  const float ty_min = ((b >= 0 ? mn.y : mx.y) - r.o.y) * b, ty_max = ((b >= 0 ? mx.y : mn.y) - r.o.y) * b;
  const float tz_min = ((c >= 0 ? mn.z : mx.z) - r.o.z) * c, tz_max = ((c >= 0 ? mx.z : mn.z) - r.o.z) * c;
  // the common case falls through (one v_max3 / v_min3); the select chains sit out of line
  float t0 = __builtin_fmaxf(__builtin_fmaxf(tx_min, ty_min), tz_min);
  float t1 = __builtin_fminf(__builtin_fminf(tx_max, ty_max), tz_max);
  if (__builtin_expect(!all_finite, 0)) {
    t0 = max3_ref(tx_min, ty_min, tz_min);
    t1 = min3_ref(tx_max, ty_max, tz_max);
  }
  t = (t0 < 0) ? t1 : t0;
  if (t0_out) *t0_out = t0;
  if (t1_out) *t1_out = t1;
  return (t0 < t1) && gt_1em4(t1);
}
__device__ __forceinline__ bool is_inside(F3 mn, F3 mx, F3 p) {  // boundingBox.cpp:39-42 (strict)
  return (p.x > mn.x && p.x < mx.x) && (p.y > mn.y && p.y < mx.y) && (p.z > mn.z && p.z < mx.z);
}

// Object::intercepts for the four kinds.  The ray is taken by reference: the sphere test
// re-normalises its direction in place (scene.cpp:156, Q8).
template <class CT>
__device__ __forceinline__ bool intercepts(const Geom& g, RayS& ray, float& time, CT& ct) {
  const uint32_t type = geom_type(g);
  if (type == P3D_PRIM_TRIANGLE) {  // scene.cpp:47-94
    ct.add(kTriTests);
    const F3 P0 = f3(g.a.x, g.a.y, g.a.z), P1 = f3(g.a.w, g.b.x, g.b.y), P2 = f3(g.b.z, g.b.w, g.c.x);
    const float a = P0.x - P1.x, b = P0.x - P2.x, c = ray.d.x, d = P0.x - ray.o.x;
    const float e = P0.y - P1.y, f = P0.y - P2.y, gg = ray.d.y, h = P0.y - ray.o.y;
    const float i = P0.z - P1.z, j = P0.z - P2.z, k = ray.d.z, l = P0.z - ray.o.z;
    const float m = f * k - gg * j, n = h * k - gg * l, p = f * l - h * j;
    const float q = gg * i - e * k, s = e * j - f * i;
    const float inv_denom = 1.0f / (a * m + b * q + c * s);  // 1.0/x narrowed == 1.0f/x
    const float e1 = d * m - b * n - c * p;
    const float beta = e1 * inv_denom;
    if (beta < 0.0f) return false;
    const float r = e * l - h * i;
    const float e2 = a * n + d * q + c * r;
    const float gamma = e2 * inv_denom;
    if (gamma < 0.0f) return false;
    if (beta + gamma > 1.0f) return false;
    const float e3 = a * p - b * r + d * s;
    const float t = e3 * inv_denom;
    if (lt_1em4(t)) return false;
    time = t;  // a NaN t passes every rejection and is reported as a hit (A10)
    return true;
  } else if (type == P3D_PRIM_SPHERE) {  // scene.cpp:149-186
    ct.add(kSphereTests);
    const F3 Rd = get_direction(ray);
    const F3 co = f3(g.a.x, g.a.y, g.a.z) - ray.o;
    const float doc2 = co.x * co.x + co.y * co.y + co.z * co.z;
    const float b = dot(co, Rd);
    const float c = doc2 - g.a.w * g.a.w;
    if (c > 0 && b < 0) return false;
    const float discriminant = b * b - c;
    if (discriminant < 0) return false;
    time = (c > 0) ? b - sqrtf(discriminant) : b + sqrtf(discriminant);
    return true;
  } else if (type == P3D_PRIM_BOX) {  // scene.cpp:215-227
    ct.add(kBoxTests);
    return aabb_intercepts(f3(g.a.x, g.a.y, g.a.z), f3(g.a.w, g.b.x, g.b.y), ray, time, false);
  } else {  // plane, scene.cpp:116-137
    ct.add(kPlaneTests);
    const F3 PN = f3(g.a.x, g.a.y, g.a.z), A = f3(g.a.w, g.b.x, g.b.y);
    const float numer = dot(ray.o - A, PN);
    const float divid = dot(PN, ray.d);
    if (fabsf(divid) <= 0.0001f) return false;  // fabs(divid) < 0.0001 (double literal)
    time = -(numer / divid);
    return !(time <= 0);
  }
}

// Object::getNormal (scene.cpp:41-44,139-142,188-192,229-267)
__device__ __forceinline__ F3 get_normal(const Geom& g, const float4* normals, F3 point) {
  const uint32_t type = geom_type(g);
  if (type == P3D_PRIM_SPHERE) return normalized(point - f3(g.a.x, g.a.y, g.a.z));
  if (type == P3D_PRIM_TRIANGLE) return xyz(normals[geom_object(g)]);
  if (type == P3D_PRIM_PLANE) return f3(g.a.x, g.a.y, g.a.z);
  const F3 mn = f3(g.a.x, g.a.y, g.a.z), mx = f3(g.a.w, g.b.x, g.b.y);
  const F3 co = point - (mx + mn) / 2;
  int dir = (fabsf(co.x) > fabsf(co.y)) ? 0 : 1;
  if (dir == 0 && fabsf(co.z) > fabsf(co.x)) dir = 2;
  else if (dir == 1 && fabsf(co.z) > fabsf(co.y)) dir = 2;
  if (dir == 0) return f3(co.x >= 0 ? 1.f : -1.f, 0, 0);
  if (dir == 1) return f3(0, co.y >= 0 ? 1.f : -1.f, 0);
  return f3(0, 0, co.z >= 0 ? 1.f : -1.f);
}

// ---------------------------------------------------------------------------
// BVH traversal — bvh.cpp:198-340.
// A node record is two float4s; both children of an inner node are adjacent, so one
// inner-node visit fetches 64 contiguous bytes.
// ---------------------------------------------------------------------------
// Node descriptor, packed at upload into one word (carried in lo.w and on the stack):
//   bit 31 = leaf, bits 30..28 = object count of a leaf (<= 7), bits 27..0 = left child
//   (inner) or first leaf slot (leaf).
#ifndef P3D_VOTE_NUM  // a step down is taken when lanes_on_inner * DEN >= lanes_on_leaf * NUM (bvh_closest, VOTE)
#define P3D_VOTE_NUM 1
#define P3D_VOTE_DEN 1
#endif
constexpr uint32_t kDescLeaf = 0x80000000u;
constexpr uint32_t kDescDone = 0xffffffffu;  // traversal state "no node left"; never a descriptor (it would be a leaf of 7 objects at slot 2^28 - 1)
constexpr uint32_t kDescHit = 0xfffffffeu;   // any-hit traversal state "a primitive was hit" (nor this: slot 2^28 - 2); both have the leaf bit
__device__ __host__ __forceinline__ uint32_t desc_index(uint32_t d) { return d & 0x0fffffffu; }
__device__ __host__ __forceinline__ uint32_t desc_count(uint32_t d) { return (d >> 28) & 7u; }

struct NodeRec {
  float4 lo, hi;  // {bmin, descriptor} {bmax, -}
};
__device__ __forceinline__ NodeRec load_node(const float4* nodes, uint32_t i) {
  NodeRec n;
  n.lo = nodes[2 * i];
  n.hi = nodes[2 * i + 1];
  return n;
}

// Both traversals are written "while-while": every lane first DESCENDS through inner nodes
// until it stands on a leaf (or has finished), then the wave processes leaves together.  The
// per-lane sequence of node visits, tests, pushes and pops is exactly the reference's; only
// the interleaving between lanes differs from a one-node-per-iteration loop, which would make
// the wave execute the leaf code AND the inner-node code in almost every iteration (some lane
// is always on a leaf).  The stack holds packed child descriptors, so a pop needs no node
// fetch.

// bvh.cpp:256-265: pop until an entry is nearer than the best hit; kDescDone = stack exhausted.  One loop condition
// (`more`) and no exit from the middle: the compiler's loop then needs one lane mask instead of three.
template <int SPILL>
__device__ __forceinline__ uint32_t pop_closer(Stack& st, float tmin) {
  uint32_t desc = kDescDone;
  bool more = st.sp > 0;
  while (more) {
    const uint2 e = pop<SPILL>(st);
    const bool take = __uint_as_float(e.y) < tmin;
    if (take) desc = e.x;
    more = !take && st.sp > 0;
  }
  return desc;
}

// Closest hit.  `ray` is the traversal's private copy (bvh.cpp:198 takes Ray by value).
// Returns the leaf slot of the hit (-1 = miss) and the hit point d*tmin + o (bvh.cpp:271).
// *root_passed (optional): the ray got past the root test of bvh.cpp:203-205, i.e. the query touched hit_stack
// and left it empty; a ray that fails it returns with the stack exactly as it found it.
// *final_ray (optional): the private copy as the traversal leaves it (its direction re-normalised by the sphere tests it ran).
// The two kinds of step of the closest-hit traversal, as text: both loop skeletons below (nested loops / vote) get exactly
// this code inline (as lambdas they cost the global-memory kernels 70 bytes of scratch per lane and 15 % of their speed).
// One step down: bvh.cpp:208-239 (ties go right, Q10).  A leaf: bvh.cpp:241-252, then the pop loop.
// (in the step down) bvh.cpp:216-217: an origin strictly inside a child's box makes that child's distance 0.  Without an odd
// lane in the wave every slab product has exactly the sign of (plane - origin) * (1/d) and none is a NaN, so the origin lies
// strictly between the two planes of an axis iff that axis' near product is < 0 and its far product > 0, and inside the box
// iff t0 = max(near products) < 0 and t1 = min(far products) > 0: two compares instead of the twelve of AABB::isInside
// (boundingBox.cpp:39-42), same truth value (boxes with min <= max: what a BVH build produces; a descriptor with an inverted
// box makes the scene take the slow path, p3d_scene_create).  Otherwise the test as the reference writes it.
// May the slab tests of this wave use v_max3 / v_min3 (no NaN can arise)?  Wave-uniform.  FASTIN: ... and may bvh_closest take
// isInside from the slab interval (needs boxes with min <= max as well: DevScene::odd_boxes)?  Only the loops that wait
// on memory per visit use that form (VOTE, scenes traversed from global memory); for LDS-staged scenes the extra scalar
// state costs more than the ten compares it saves (cfg2 frames-in-flight loop: 3 %, experiments r03 §13).
template <bool FASTIN>
__device__ __forceinline__ bool slab_fast_path(const DevScene& sc, const RayS& ray) {
  return FASTIN ? (!sc.odd_boxes && !__any(ray.odd_inv)) : !__any(ray.odd_inv);
}
#define P3D_CLOSEST_DESCEND_STEP \
  {\
      const uint32_t index = desc_index(desc);\
      const NodeRec l = load_node(sc.nodes, index), r = load_node(sc.nodes, index + 1);\
      float l_t, r_t;\
      ct.add(kNodeTests, 2);\
      const bool fin = slab_fast_path<FASTIN>(sc, ray);\
      float l_t0, l_t1, r_t0, r_t1;\
      const bool l_hit = aabb_intercepts(xyz(l.lo), xyz(l.hi), ray, l_t, fin, &l_t0, &l_t1);\
      const bool r_hit = aabb_intercepts(xyz(r.lo), xyz(r.hi), ray, r_t, fin, &r_t0, &r_t1);\
      if (FASTIN && fin) {\
        if (l_t0 < 0 && l_t1 > 0) l_t = 0;\
        if (r_t0 < 0 && r_t1 > 0) r_t = 0;\
      } else {\
        if (is_inside(xyz(l.lo), xyz(l.hi), ray.o)) l_t = 0;\
        if (is_inside(xyz(r.lo), xyz(r.hi), ray.o)) r_t = 0;\
      }\
      const uint32_t ld = __float_as_uint(l.lo.w), rd = __float_as_uint(r.lo.w);\
      if (l_hit && r_hit) {\
        if (l_t < r_t) { desc = ld; push<SPILL>(st, rd, r_t, ct); }\
        else           { desc = rd; push<SPILL>(st, ld, l_t, ct); }\
      } else if (l_hit) { desc = ld; }\
      else if (r_hit)   { desc = rd; }\
      else desc = pop_closer<SPILL>(st, tmin);\
  }
#define P3D_CLOSEST_LEAF_STEP \
  {\
      const uint32_t index = desc_index(desc), n = desc_count(desc);\
      for (uint32_t s = index; s < index + n; ++s) {\
        const Geom g = load_geom(sc.bgeom, s);\
        float curr_t;\
        if (intercepts(g, ray, curr_t, ct) && curr_t < tmin) {\
          tmin = curr_t;\
          hit = (int)s;\
          hit_geom = g;\
        }\
      }\
      desc = pop_closer<SPILL>(st, tmin);\
  }
template <int SPILL, class CT, bool VOTE = (SPILL == kStackWindow), bool FASTIN_ = VOTE>
__device__ int bvh_closest(const DevScene& sc, Stack& st, RayS ray, F3& hit_point, Geom& hit_geom, CT& ct,
                           bool* root_passed = nullptr, float* t_out = nullptr, RayS* final_ray = nullptr) {
  constexpr bool FASTIN = FASTIN_;
  float tmp, tmin = FLT_MAX;
  int hit = -1;
  const NodeRec root = load_node(sc.nodes, 0);
  ct.add(kNodeTests);
  if (!aabb_intercepts(xyz(root.lo), xyz(root.hi), ray, tmp, slab_fast_path<FASTIN>(sc, ray))) return -1;  // stale entries stay (Q2)
  if (root_passed) *root_passed = true;
  // The traversal state is ONE word: the descriptor of the node the lane stands on, or kDescDone (leaf bit set, so
  // that a finished lane also falls out of the descend loop): fewer lane masks for the compiler to carry round the loops.
  uint32_t desc = __float_as_uint(root.lo.w);
  if (VOTE) {
    // Every lane walks its own sequence of steps down and leaves; the wave can only take one kind of step at a time and
    // the lanes that need the other kind sit it out.  "Leaves only when nobody is on an inner node" (the nested loops
    // below) lets a few long descents hold up everyone who already stands on a leaf; here the kind that has the
    // majority goes next (a leaf step costs about as much as a step down).  The order of a LANE's visits, tests, pushes
    // and pops is untouched, so are the bits.  100k triangles 2048x2048: 16.9 -> 15.0 ms (literal 18.1 -> 16.2); not for
    // LDS-staged scenes, whose phases are a handful of steps long (the ballots cost more than the waiting: cfg2 -7 %).
    while (true) {
      const bool on_inner = !(desc & kDescLeaf), on_leaf = !on_inner && desc != kDescDone;
      const unsigned long long m_inner = __ballot(on_inner), m_leaf = __ballot(on_leaf);
      if ((m_inner | m_leaf) == 0) break;
      const bool descend = m_leaf == 0 || __popcll(m_inner) * P3D_VOTE_DEN >= __popcll(m_leaf) * P3D_VOTE_NUM;  // wave-uniform
      if (descend && on_inner) P3D_CLOSEST_DESCEND_STEP
      if (!descend && on_leaf) P3D_CLOSEST_LEAF_STEP
    }
  } else {
    while (desc != kDescDone) {
      while (!(desc & kDescLeaf)) P3D_CLOSEST_DESCEND_STEP
      if (desc != kDescDone) P3D_CLOSEST_LEAF_STEP
    }
  }
  if (hit >= 0) hit_point = ray.d * tmin + ray.o;
  if (t_out) *t_out = tmin;
  if (final_ray) *final_ray = ray;
  return hit;
}

// Any hit.  Q1: after a dead end the reference pops EVERYTHING and resumes at the
// bottom-most entry; Q2: an early `return true` leaves its entries on the stack for the
// next query of the same pixel.
// ... and of the any-hit traversal (bvh.cpp:278-340): a step down; a leaf (a hit ends the query, its entries stay behind: Q2).
#define P3D_ANY_DESCEND_STEP \
  {\
      const uint32_t index = desc_index(desc);\
      const NodeRec l = load_node(sc.nodes, index), r = load_node(sc.nodes, index + 1);\
      float l_t, r_t;\
      ct.add(kNodeTests, 2);\
      const bool fin = slab_fast_path<FASTIN>(sc, ray);\
      const bool l_hit = aabb_intercepts(xyz(l.lo), xyz(l.hi), ray, l_t, fin);\
      const bool r_hit = aabb_intercepts(xyz(r.lo), xyz(r.hi), ray, r_t, fin);\
      const uint32_t ld = __float_as_uint(l.lo.w), rd = __float_as_uint(r.lo.w);\
      if (l_hit && r_hit) {\
        if (l_t < r_t) { desc = ld; push<SPILL>(st, rd, r_t, ct); }\
        else           { desc = rd; push<SPILL>(st, ld, l_t, ct); }\
      } else if (l_hit) { desc = ld; }\
      else if (r_hit)   { desc = rd; }\
      else restart_from_bottom();\
  }
#define P3D_ANY_LEAF_STEP \
  {\
      uint32_t s = desc_index(desc);\
      const uint32_t end = s + desc_count(desc);\
      bool occluded = false, more = s < end;\
      while (more) {\
        const Geom g = load_geom(sc.bgeom, s);\
        float curr_t;\
        occluded = intercepts(g, ray, curr_t, ct);\
        ++s;\
        more = !occluded && s < end;\
      }\
      if (occluded) desc = kDescHit;\
      else restart_from_bottom();\
  }
template <int SPILL, class CT, bool VOTE = (SPILL == kStackWindow), bool FASTIN_ = VOTE>
__device__ bool bvh_any(const DevScene& sc, Stack& st, RayS ray, CT& ct) {
  constexpr bool FASTIN = FASTIN_;
  float tmp;
  const NodeRec root = load_node(sc.nodes, 0);
  ct.add(kNodeTests);
  if (!aabb_intercepts(xyz(root.lo), xyz(root.hi), ray, tmp, slab_fast_path<FASTIN>(sc, ray))) return false;
  // state word as in bvh_closest, with a second end state: kDescHit = a primitive was hit (no flag to carry round the loops)
  uint32_t desc = __float_as_uint(root.lo.w);
  // bvh.cpp:329-338: pop all, continue from the first-pushed entry; nothing left = done
  auto restart_from_bottom = [&]() {
    if (st.sp > 0) {
      desc = stack_read<SPILL>(st, 0).x;
      stack_clear(st);
    } else {
      desc = kDescDone;
    }
  };
  if (VOTE) {  // (see bvh_closest)
    while (true) {
      const bool on_inner = !(desc & kDescLeaf), on_leaf = !on_inner && desc < kDescHit;
      const unsigned long long m_inner = __ballot(on_inner), m_leaf = __ballot(on_leaf);
      if ((m_inner | m_leaf) == 0) break;
      const bool descend = m_leaf == 0 || __popcll(m_inner) * P3D_VOTE_DEN >= __popcll(m_leaf) * P3D_VOTE_NUM;
      if (descend && on_inner) P3D_ANY_DESCEND_STEP
      if (!descend && on_leaf) P3D_ANY_LEAF_STEP
    }
  } else {
    while (desc < kDescHit) {
      while (!(desc & kDescLeaf)) P3D_ANY_DESCEND_STEP
      if (desc < kDescHit) P3D_ANY_LEAF_STEP
    }
  }
  return desc == kDescHit;
}

// ---------------------------------------------------------------------------
// Brute force — main.cpp:116-124 (closest) and main.cpp:208-216 (any).  The closest-hit
// loop works on the CALLER's ray (Object::intercepts takes Ray&), so the caller sees Q8.
// ---------------------------------------------------------------------------
template <class CT>
__device__ int brute_closest(const DevScene& sc, RayS& ray, float& min_t, Geom& hit_geom, CT& ct) {
  int min_obj = -1;
  min_t = FLT_MAX;
  for (uint32_t i = 0; i < sc.n_objs; ++i) {
    const Geom g = load_geom(sc.ogeom, i);
    float t;
    if (intercepts(g, ray, t, ct) && t < min_t) {
      min_obj = (int)i;
      min_t = t;
      hit_geom = g;
    }
  }
  return min_obj;
}
template <class CT>
__device__ bool brute_any(const DevScene& sc, RayS& ray, CT& ct) {
  for (uint32_t i = 0; i < sc.n_objs; ++i) {
    const Geom g = load_geom(sc.ogeom, i);
    float t;
    if (intercepts(g, ray, t, ct)) return true;
  }
  return false;
}

// ---------------------------------------------------------------------------
// Uniform grid — grid.cpp:71-370 (3D-DDA).  t_next / dt are doubles as in the reference.
// ---------------------------------------------------------------------------
struct GridWalk {
  int ix, iy, iz, ix_step, iy_step, iz_step, ix_stop, iy_stop, iz_stop;
  double dtx, dty, dtz, tx_next, ty_next, tz_next;
};
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

__device__ inline bool grid_init(const DevGrid& G, const RayS& ray, GridWalk& k) {  // grid.cpp:261-370
  const F3 o = ray.o, dir = ray.d, mn = G.bmin, mx = G.bmax;
  const int nx = G.nx, ny = G.ny, nz = G.nz;
  float tx_min = (mn.x - o.x) / dir.x, ty_min = (mn.y - o.y) / dir.y, tz_min = (mn.z - o.z) / dir.z;
  float tx_max = (mx.x - o.x) / dir.x, ty_max = (mx.y - o.y) / dir.y, tz_max = (mx.z - o.z) / dir.z;
  if (tx_min > tx_max) { const float s = tx_max; tx_max = tx_min; tx_min = s; }
  if (ty_min > ty_max) { const float s = ty_max; ty_max = ty_min; ty_min = s; }
  if (tz_min > tz_max) { const float s = tz_max; tz_max = tz_min; tz_min = s; }
  const float t0 = max3_ref(tx_min, ty_min, tz_min);
  const float t1 = min3_ref(tx_max, ty_max, tz_max);
  if (t0 > t1 || t1 < 0) return false;
  k.dtx = (double)((tx_max - tx_min) / nx);
  k.dty = (double)((ty_max - ty_min) / ny);
  k.dtz = (double)((tz_max - tz_min) / nz);
  F3 p = o;
  if (!is_inside(mn, mx, o)) p = o + dir * t0;
  // clamp() passes a NaN through (0 * inf on a slab plane); the reference then dies in cells.at().
  // Map it to cell 0 explicitly so that no lane can ever index outside the cell arrays.
  auto coord = [](double v, int n) {
    const double c = clampd(v, 0, n - 1);
    return (c >= 0) ? (int)c : 0;
  };
  k.ix = coord((double)((p.x - mn.x) * nx / (mx.x - mn.x)), nx);
  k.iy = coord((double)((p.y - mn.y) * ny / (mx.y - mn.y)), ny);
  k.iz = coord((double)((p.z - mn.z) * nz / (mx.z - mn.z)), nz);
  if (dir.x > 0) { k.tx_next = tx_min + (k.ix + 1) * k.dtx; k.ix_step = 1; k.ix_stop = nx; }
  else           { k.tx_next = tx_min + (nx - k.ix) * k.dtx; k.ix_step = -1; k.ix_stop = -1; }
  if (dir.x == 0.0f) k.tx_next = FLT_MAX;
  if (dir.y > 0) { k.ty_next = ty_min + (k.iy + 1) * k.dty; k.iy_step = 1; k.iy_stop = ny; }
  else           { k.ty_next = ty_min + (ny - k.iy) * k.dty; k.iy_step = -1; k.iy_stop = -1; }
  if (dir.y == 0.0f) k.ty_next = FLT_MAX;
  if (dir.z > 0) { k.tz_next = tz_min + (k.iz + 1) * k.dtz; k.iz_step = 1; k.iz_stop = nz; }
  else           { k.tz_next = tz_min + (nz - k.iz) * k.dtz; k.iz_step = -1; k.iz_stop = -1; }
  if (dir.z == 0.0f) k.tz_next = FLT_MAX;
  return true;
}

// advance one cell; returns false when the walk leaves the grid.  `limit` receives the
// t_next of the axis about to be crossed (the acceptance bound of grid.cpp:108,123,137).
__device__ __forceinline__ bool grid_step(GridWalk& k, double& limit) {
  if (k.tx_next < k.ty_next && k.tx_next < k.tz_next) {
    limit = k.tx_next; k.tx_next += k.dtx; k.ix += k.ix_step;
    return k.ix != k.ix_stop;
  } else if (k.ty_next < k.tz_next) {
    limit = k.ty_next; k.ty_next += k.dty; k.iy += k.iy_step;
    return k.iy != k.iy_stop;
  } else {
    limit = k.tz_next; k.tz_next += k.dtz; k.iz += k.iz_step;
    return k.iz != k.iz_stop;
  }
}

// grid.cpp:71-151: the caller's ray (Ray&) is tested and mutated; hit point = o + d*min_t
template <class CT>
__device__ int grid_closest(const DevScene& sc, RayS& ray, F3& hit_point, Geom& hit_geom, CT& ct, float* t_out = nullptr) {
  const DevGrid& G = sc.grid;
  GridWalk k;
  if (!grid_init(G, ray, k)) return -1;
  int min_obj = -1;
  float min_t = FLT_MAX;
  while (true) {
    const uint32_t c = (uint32_t)(k.ix + G.nx * k.iy + G.nx * G.ny * k.iz);
    const uint32_t b = G.cell_start[c], e = G.cell_start[c + 1];
    for (uint32_t i = b; i < e; ++i) {
      const uint32_t obj = G.cell_items[i];
      const Geom g = load_geom(sc.ogeom, obj);
      float t;
      if (intercepts(g, ray, t, ct) && t < min_t) {
        min_t = t;
        min_obj = (int)obj;
        hit_geom = g;
      }
    }
    double limit;
    const bool more = grid_step(k, limit);
    if (min_obj >= 0 && (double)min_t < limit) {
      hit_point = ray.o + ray.d * min_t;
      if (t_out) *t_out = min_t;
      return min_obj;
    }
    if (!more) return -1;
  }
}
// grid.cpp:154-208
template <class CT>
__device__ bool grid_any(const DevScene& sc, RayS& ray, CT& ct) {
  const DevGrid& G = sc.grid;
  GridWalk k;
  if (!grid_init(G, ray, k)) return false;
  while (true) {
    const uint32_t c = (uint32_t)(k.ix + G.nx * k.iy + G.nx * G.ny * k.iz);
    const uint32_t b = G.cell_start[c], e = G.cell_start[c + 1];
    for (uint32_t i = b; i < e; ++i) {
      const Geom g = load_geom(sc.ogeom, G.cell_items[i]);
      float t;
      if (intercepts(g, ray, t, ct)) return true;
    }
    double limit;
    if (!grid_step(k, limit)) return false;
  }
}

// ---------------------------------------------------------------------------
// Closest hit as rayTracing / Radiance select it (main.cpp:103-125, 324-343, 449-469).
// Returns the object id (-1 = miss); `P` is the reference's interceptNotPrecise
// (main.cpp:164): o + d*min_t with the caller's (mutated) ray for accel None, the
// traversal's hit point otherwise.
// ---------------------------------------------------------------------------
// VOTE: the BVH loops take the kind of step the majority of the wave's lanes needs (bvh_closest); for scenes traversed from
// global memory, not for LDS-staged ones.
template <int ACCEL, int SPILL, bool VOTE, bool FASTIN = VOTE, class CT>
__device__ __forceinline__ int closest_hit(const DevScene& sc, Stack& st, RayS& ray, F3& P, Geom& g, CT& ct,
                                           bool* root_passed = nullptr, float* t_out = nullptr) {
  if (ACCEL == P3D_ACCEL_BVH) {
    const int slot = bvh_closest<SPILL, CT, VOTE, FASTIN>(sc, st, ray, P, g, ct, root_passed, t_out);
    return slot < 0 ? -1 : (int)geom_object(g);
  } else if (ACCEL == P3D_ACCEL_GRID) {
    return grid_closest(sc, ray, P, g, ct, t_out);
  } else {
    float min_t;
    const int obj = brute_closest(sc, ray, min_t, g, ct);
    if (obj >= 0) P = ray.o + ray.d * min_t;
    if (t_out) *t_out = min_t;
    return obj;
  }
}
// Shadow feeler (main.cpp:196-217).  Q6: with the grid, brute force runs as well.
template <int ACCEL, int SPILL, bool VOTE, bool FASTIN = VOTE, class CT>
__device__ __forceinline__ bool any_hit(const DevScene& sc, Stack& st, RayS& feeler, CT& ct) {
  if (ACCEL == P3D_ACCEL_BVH) return bvh_any<SPILL, CT, VOTE, FASTIN>(sc, st, feeler, ct);
  bool occluded = false;
  if (ACCEL == P3D_ACCEL_GRID) occluded = grid_any(sc, feeler, ct);
  const bool b = brute_any(sc, feeler, ct);
  return occluded || b;
}

// Scene::GetSkyboxColor — scene.cpp:379-457, indexed by the RAW ray direction.  The reference's
// two clamp lines (scene.cpp:448,450) are expression statements without effect; here the index
// IS clamped so that a NaN / zero direction cannot read out of bounds.
__device__ inline F3 skybox_color(const DevScene& sc, F3 d) {
  float ma;
  int side;  // RIGHT 0, LEFT 1, TOP 2, BOTTOM 3, FRONT 4, BACK 5
  if (fabsf(d.x) > fabsf(d.y)) { ma = fabsf(d.x); side = d.x >= 0 ? 1 : 0; }
  else                         { ma = fabsf(d.y); side = d.y >= 0 ? 2 : 3; }
  if (fabsf(d.z) > ma) { ma = fabsf(d.z); side = d.z >= 0 ? 4 : 5; }
  float s_c, t_c;
  switch (side) {
    case 0: s_c = -d.z; t_c = d.y; break;
    case 1: s_c = d.z; t_c = d.y; break;
    case 2: s_c = -d.x; t_c = -d.z; break;
    case 3: s_c = -d.x; t_c = d.z; break;
    case 4: s_c = -d.x; t_c = d.y; break;
    default: s_c = d.x; t_c = d.y; break;
  }
  const double invMa = (double)(1 / ma);  // `1 / ma` is a float division widened afterwards
  const float s = (float)((s_c * invMa + 1) / 2);
  const float t = (float)((t_c * invMa + 1) / 2);
  const uint32_t width = sc.sky_w[side], height = sc.sky_h[side];
  uint32_t xp = (uint32_t)(int)((float)(width - 1) * s);
  uint32_t yp = (uint32_t)(int)((float)(height - 1) * t);
  xp = xp >= width ? width - 1 : xp;
  yp = yp >= height ? height - 1 : yp;
  const uint32_t px = sc.sky[side][(size_t)yp * width + xp];
  // u8tofloat, maths.h:89-92
  return f3((float)(px & 0xffu) / 255.99f, (float)((px >> 8) & 0xffu) / 255.99f, (float)((px >> 16) & 0xffu) / 255.99f);
}
// miss colour of main.cpp:144-147 / 350-355.  Pass the KERNEL ARGUMENT copy of the scene (P.sc): the faces are
// indexed dynamically, and indexing a local copy would force the whole struct into scratch memory.
__device__ __forceinline__ F3 miss_color(const DevScene& sc, bool skybox, F3 raw_direction) {
  return skybox ? skybox_color(sc, raw_direction) : sc.bg;
}

__device__ __forceinline__ F3 offset_intersection(F3 inter, F3 normal) { return inter + normal * .0001f; }  // main.cpp:82-84

// ---------------------------------------------------------------------------
// RNG: one PCG-XSH-RR 64/32 stream per (pixel, sample); same definition as the oracle's.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Rng {
  uint64_t state, inc;
  __device__ __forceinline__ void seed_stream(uint64_t seed, uint32_t pixel, uint32_t sample) {
    const uint64_t k = mix64(seed ^ mix64(((uint64_t)pixel << 32) | (uint64_t)sample));
    inc = (mix64(k) << 1) | 1ull;
    state = k * 6364136223846793005ull + inc;
  }
  __device__ __forceinline__ uint32_t next31() {
    const uint64_t old = state;
    state = old * 6364136223846793005ull + inc;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    const uint32_t rot = (uint32_t)(old >> 59);
    return ((xs >> rot) | (xs << ((0u - rot) & 31u))) >> 1;
  }
  // maths.h:67-70 with RAND_MAX = 2^31-1 ((float)RAND_MAX == 2^31); can return exactly 1.0f
  __device__ __forceinline__ float rand_float() { return (float)next31() / 2147483648.0f; }
  // main.cpp:75-77
  __device__ __forceinline__ double erand48() { return (double)next31() / 2147483647.0; }
};

// camera.h:65-82
__device__ __forceinline__ void primary_ray(const DevCamera& c, float px, float py, F3& o, F3& d) {
  const float psx = c.w * (px / c.res_x - 0.5f);
  const float psy = c.h * (py / c.res_y - 0.5f);
  const float psz = -c.plane_dist;
  d = normalized((c.u * psx + c.v * psy) + c.n * psz);
  o = c.eye;
}
// camera.h:84-115
__device__ __forceinline__ void primary_ray_lens(const DevCamera& c, float lx, float ly, float px, float py, F3& o,
                                                 F3& d) {
  const float psx = c.w * (px / c.res_x - 0.5f);
  const float psy = c.h * (py / c.res_y - 0.5f);
  const float lsx = lx * c.aperture, lsy = ly * c.aperture;
  const float qx = psx * c.focal_ratio, qy = psy * c.focal_ratio;
  d = normalized((c.u * (qx - lsx) + c.v * (qy - lsy)) + c.n * -(c.focal_ratio * c.plane_dist));
  o = (c.eye + c.u * lsx) + c.v * lsy;
}

__device__ __forceinline__ uint8_t u8fromfloat(float x) {  // maths.h:81-86
  const float s = x * 255.99f;
  return s >= 255.0f ? (uint8_t)255 : (uint8_t)s;
}

}  // namespace p3d
