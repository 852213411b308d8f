// kernels.hpp — the HIP kernels of the hot path (gfx950).
//
//   whitted_kernel : primary ray -> closest hit -> Blinn-Phong with shadow feelers ->
//                    reflect / refract chain with per-level clamp      (main.cpp:92-309 + 753-820)
//   pt_kernel      : the smallpt-style Radiance loop as a persistent per-lane bounce loop
//                    (main.cpp:313-516 + 758-800)
//   trace_kernel   : batched closest / any-hit queries (bvh.cpp:198-340, grid.cpp:71-208,
//                    main.cpp:116-124,208-216)
//
// One workgroup = one wavefront = an 8x8 pixel block.  See device_core.hpp for the
// traversal conventions and DESIGN.md for the kernel-by-kernel roofline discussion.
#pragma once

#include "device_core.hpp"
#include "handoff.hpp"
#include "p3d.h"

// minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ arg)
#ifndef P3D_WHITTED_WAVES
#define P3D_WHITTED_WAVES 2
#endif
// Same for the no-AA instantiations that traverse the scene from L2: they wait on memory, not on the VALU, and
// trade registers for waves.  100k triangles 1024x1024 / 2048x2048 with 16 / 12 / 12 / 10 / 8 LDS stack entries:
// 4 waves (111 VGPRs) 8.44 / 22.9 ms, 5 (96) 7.69 / 20.7, 6 (80 VGPRs, 46 spilled dwords) 7.26 / 19.4,
// 7 (72) 7.40 / 18.7, 8 (64) 7.37 / 18.7.
// The work-list launches of the hit_stack hand-off (LIT == 2) are short lists of unrelated deep pixels: every wave waits
// on its own dependent chain, so what counts is how many waves are resident at once, not registers per wave.
#ifndef P3D_LIST_WAVES
#define P3D_LIST_WAVES 4
#endif
#ifndef P3D_WHITTED_GLOBAL_WAVES
#define P3D_WHITTED_GLOBAL_WAVES 6
#endif
// One-sample-per-pixel instantiations over an LDS-staged scene: 96 VGPRs = five waves per SIMD.  The per-pixel kernel needs
// exactly that; the literal pass 1 needs 99 without the zero-weight-reflection machinery (GHOSTS = false) and is held to 96
// without a spilled dword (with it: 107, and holding it to 96 spills 12 dwords - profiles/r03/experiments §13).
#ifndef P3D_WHITTED_LDS_WAVES
#define P3D_WHITTED_LDS_WAVES 5
#endif

namespace p3d {

#ifdef P3D_TIMELINE  // debug builds only (build/variants): per-workgroup start/end clock + HW id
__device__ unsigned long long* g_timeline = nullptr;
#define P3D_TL_BEGIN() const unsigned long long tl_t0 = wall_clock64();
#define P3D_TL_END()                                                                                  \
  if (g_timeline && threadIdx.x == 0) {                                                               \
    g_timeline[3 * (size_t)blockIdx.x] = tl_t0;                                                       \
    g_timeline[3 * (size_t)blockIdx.x + 1] = wall_clock64();                                          \
    g_timeline[3 * (size_t)blockIdx.x + 2] =                                                          \
        ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
  }
#else
#define P3D_TL_BEGIN()
#define P3D_TL_END()
#endif

#ifdef P3D_PT_PROFILE  // debug builds only: wave-time, entries and active lanes per region of a kernel's loop
__device__ unsigned long long* g_pt_prof = nullptr;
constexpr int kProfRegions = 12;
// The clock belongs to the wave, not to a lane: whichever lanes reach a marker, the time since the previous marker
// (reached by any lanes) was spent executing the region that marker opened — divergent paths run one after the other.
struct RegionProfShared { unsigned long long acc[kProfRegions], lanes[kProfRegions], iters[kProfRegions], last; int cur; };
struct RegionProf {
  volatile RegionProfShared* s;
  __device__ void init() {
    __shared__ RegionProfShared sh;  // one wave per workgroup
    s = &sh;
    if (__ffsll((unsigned long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) {
      for (int i = 0; i < kProfRegions; ++i) s->acc[i] = s->lanes[i] = s->iters[i] = 0;
      s->cur = 0; s->last = __builtin_readcyclecounter();
    }
  }
  __device__ void enter(int r) {
    __builtin_amdgcn_s_waitcnt(0);  // outstanding loads belong to the region that issued them
    const unsigned long long now = __builtin_readcyclecounter(), m = __ballot(1);
    if (__ffsll(m) - 1 == (int)(threadIdx.x & 63)) {
      s->acc[s->cur] += now - s->last; s->last = now; s->cur = r;
      s->lanes[r] += __popcll(m); s->iters[r] += 1;
    }
    __builtin_amdgcn_s_waitcnt(0);
  }
  __device__ void flush() {
    if (g_pt_prof && __ffsll((unsigned long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63))
      for (int i = 0; i < kProfRegions; ++i) {
        atomicAdd(&g_pt_prof[3 * i], s->acc[i]); atomicAdd(&g_pt_prof[3 * i + 1], s->lanes[i]); atomicAdd(&g_pt_prof[3 * i + 2], s->iters[i]);
      }
  }
};
#define PT_REGION(r) prof.enter(r);
#else
#define PT_REGION(r)
#endif

struct RenderParams {
  DevScene sc;
  const float4* blob;   // all float4 scene arrays, contiguous (for the LDS staging copy)
  uint32_t blob_f4;     // float4s of blob staged in LDS (= lds_scene_f4 when the scene is staged): the arrays a kernel of this
                        // launch reads in its loops, without the alignment pad in front and - Whitted over the BVH, which only
                        // reads the BVH-ordered copy - without the object-order geometry at the end
  uint32_t stage_lo;    // first staged float4 of blob
  // byte offsets (in float4) of the arrays inside blob, same order as DevScene
  uint32_t off_nodes, off_bgeom, off_ogeom, off_normals, off_mats, off_lights;
  // options (p3d_config)
  int32_t max_depth;
  uint32_t spp_sqrt, antialiasing, depth_of_field, sample_disk, soft_shadows, sample_mode;
  float light_side, gamma;
  uint32_t skybox;  // miss = cubemap texel (only set when the scene has a cubemap)
  uint32_t debug_view;  // P3D_DEBUG_* (constants.h:18,33)
  uint64_t seed;
  // tile
  int32_t x0, y0, w, h, stripe_h, stripe_stride;
  uint32_t tiles_x, tiles_y, xcd_chunk;
  uint32_t tile_w_shift, tile_h_shift;  // one-lane-per-pixel kernels: a wave renders a (1 << w) x (1 << h) pixel tile: 8x8, 8x4 or 4x4 (host: tile_shape)
  // outputs (device memory, any may be null)
  float* rgb;
  int32_t* hit_id;
  uint8_t* rgb8;
  // scratch
  float4* levels;          // Whitted per-level records {local colour, child weight}, [level][thread]
  uint32_t level_stride;   // threads in this launch
  unsigned long long* stats;  // kNumStats counters
  uint2* spill;               // backing array of the node stack (entries that sank out of the LDS window), [entry][thread]
  int32_t stack_cap;          // node-stack entries per lane held in LDS (spilling stack: the window, a power of two)
  uint32_t lds_scene_f4;      // float4s reserved for the staged scene (0 when not staged)
  // Cost-ordered tile schedule (DESIGN.md "Tile schedule"); both null: frame order, nothing recorded.
  //   sched: see sched_build_kernel.   tile_cost[t]: wall-clock ticks the workgroup of tile t was resident.
  const uint32_t* sched;
  uint32_t* tile_cost;
  uint32_t stack_spills;  // LDS-staged scene with the spilling stack (host-side kernel choice)
  // P3D_STACK_LITERAL (handoff.hpp)
  Handoff hand;
  int32_t row0;            // first tile row of this launch (units are numbered over the whole tile)
  uint32_t tile_blocks;    // workgroups that render tiles; the ones behind them render halo chains (pass 1, check)
  float4* deferred;        // zero-weight reflection rays put aside, 2 x float4 each, [2 * entry][thread]
  uint32_t* status;        // device-detected errors of this scene (kHoErr* bits), never cleared by a kernel
  uint32_t debug_trip_bound;  // tests only: trip bound of the four-lanes-per-pixel sample loops (0 = the real bound)
  // Per-level ("wavefront") launches of the Whitted chain, scenes traversed from L2 (wf_level_kernel)
  uint32_t wf_level;       // chain level this launch renders
  uint32_t wf_seg_cap;     // entries per queue segment
  float4* wf_ray_in;       // [kWfSegments][wf_seg_cap][2]: rays this level traces, written by the level above
  float4* wf_ray_out;
  const uint32_t* wf_n_in; // [kWfSegments] counters, one per 128-byte line
  uint32_t* wf_n_out;
  float4* wf_final;        // [units] {colour the chain ended with, number of level records below it}
  uint32_t level_stride2;  // units: stride of the level records of the per-level launches
  // binning of the child rays by origin cell and direction octant between two levels (wf_scan_kernel, wf_scatter_kernel)
  uint32_t* wf_key_out;    // [kWfSegments][wf_seg_cap] bin of every staged ray
  const uint32_t* wf_key_in;
  uint32_t* wf_hist;       // [kWfBins + 1] rays per bin of the level being written; after the scan: first slot of each bin
  const uint32_t* wf_total;  // rays in the sorted queue this launch reads
  float4* wf_sorted;       // [units][2] the queue in bin order
  F3 wf_cell_origin, wf_cell_scale;  // cell = (origin - wf_cell_origin) * wf_cell_scale, clamped to [0, kWfCellsPerAxis)
};

// LDS map of one workgroup:  [ staged scene (lds_scene_f4 float4) | node stack (cap * 64 * 8 B) | per-pixel sample ring (PT, 4 lanes per pixel) ]
// OBJECTS: the launch stages the object-order geometry too (every kernel but Whitted over the BVH; the host's stage range)
template <bool LDS, bool OBJECTS>
__device__ __forceinline__ void stage_scene(DevScene& sc, const RenderParams& P, float4* smem) {
  if (LDS) {
    for (uint32_t i = threadIdx.x; i < P.blob_f4; i += kBlock) smem[i] = P.blob[P.stage_lo + i];
    __syncthreads();
    sc.nodes = smem + (P.off_nodes - P.stage_lo);
    sc.bgeom = smem + (P.off_bgeom - P.stage_lo);
    sc.normals = smem + (P.off_normals - P.stage_lo);
    sc.mats = smem + (P.off_mats - P.stage_lo);
    sc.lights = smem + (P.off_lights - P.stage_lo);
    if (OBJECTS) sc.ogeom = smem + (P.off_ogeom - P.stage_lo);  // (otherwise: stays in global memory)
  }
}

// XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs, so blocks
// b, b+8, b+16, ... share an L2.  Tiles are grouped into chunks of `xcd_chunk` consecutive
// 8x8 tiles (one tile row of the frame for big scenes) and chunks are dealt to XCDs round-
// robin: each XCD's L2 sees spatially coherent rays (matters for the 8.8 MB scene, which
// does not fit one 4 MiB L2) while expensive and cheap image regions are still spread over
// all XCDs.  xcd_chunk = 1 is the identity map (LDS-staged scenes have no L2 working set).
//
// With a schedule (DESIGN.md "Tile schedule") blockIdx order is "most expensive class first":
// workgroups are dispatched in blockIdx order, so the few long-running tiles start at once and
// the many short ones fill in behind them instead of the other way round.
__device__ __forceinline__ bool tile_of_block(const RenderParams& P, uint32_t& tx, uint32_t& ty) {
  const uint32_t b = blockIdx.x;
  const uint32_t n = P.tiles_x * P.tiles_y;
  uint32_t tile;
  if (P.sched) {
    if (b >= n) return false;
    tile = P.sched[b];
  } else {
    const uint32_t j = b >> 3;
    tile = ((j / P.xcd_chunk) * 8 + (b & 7u)) * P.xcd_chunk + (j % P.xcd_chunk);
    if (tile >= n) return false;
  }
  tx = tile % P.tiles_x;
  ty = tile / P.tiles_x;
  return true;
}

// Cost of a finished tile, for the schedule of the following launches.
__device__ __forceinline__ void record_tile_cost(const RenderParams& P, uint32_t tx, uint32_t ty, unsigned long long t_begin) {
  if (P.tile_cost && threadIdx.x == 0) P.tile_cost[ty * P.tiles_x + tx] = (uint32_t)(wall_clock64() - t_begin);
}

// One workgroup: counting sort of the tiles by cost class, most expensive class first.  Classes
// are quarter octaves of cost / mean cost, from below 1/4 to above 3.4 (order inside a class:
// arrival).  sched[i] = the tile workgroup i renders.
constexpr int kSchedBuildThreads = 1024;
constexpr int kSchedClasses = 16;
__device__ __forceinline__ int sched_class(uint32_t cost, float mean) {
  if (cost == 0) return 0;
  const int c = (int)floorf(log2f((float)cost / mean) * 4.0f) + 8;
  return c < 0 ? 0 : (c >= kSchedClasses ? kSchedClasses - 1 : c);
}

__global__ void __launch_bounds__(kSchedBuildThreads) sched_build_kernel(const uint32_t* cost, uint32_t n, uint32_t* sched) {
  __shared__ unsigned long long total;
  __shared__ uint32_t cursor[kSchedClasses];
  if (threadIdx.x == 0) total = 0;
  if (threadIdx.x < kSchedClasses) cursor[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long mine = 0;
  for (uint32_t t = threadIdx.x; t < n; t += kSchedBuildThreads) mine += cost[t];
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&total, mine);
  __syncthreads();
  const float mean = fmaxf((float)total / (float)n, 1.0f);
  for (uint32_t t = threadIdx.x; t < n; t += kSchedBuildThreads) atomicAdd(&cursor[sched_class(cost[t], mean)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {  // class sizes -> first slot of each class, most expensive class first
    uint32_t at = 0;
    for (int c = kSchedClasses - 1; c >= 0; --c) {
      const uint32_t size = cursor[c];
      cursor[c] = at;
      at += size;
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < n; t += kSchedBuildThreads) sched[atomicAdd(&cursor[sched_class(cost[t], mean)], 1u)] = t;
}

// One launch instead of a hipMemsetAsync per buffer (each of those is a fill kernel of its own, ~5 us): zeroes up to
// three word ranges at the head of a frame (counters, touched bits, statistics).
struct ClearParams {
  uint32_t* p[3];
  uint32_t n[3];
  const uint32_t* halo_verdict;  // kHoErrHalo or 0 as the last halo_find_kernel launch for this tile left it (the search is
  uint32_t* status;              // memoised per tile; its verdict holds for every frame that uses the memoised chain)
};
__global__ void __launch_bounds__(256) clear_kernel(const ClearParams C) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  if (i == 0 && C.halo_verdict && *C.halo_verdict) atomicOr(C.status, *C.halo_verdict);
  for (int k = 0; k < 3; ++k)
    for (uint32_t j = i; j < C.n[k]; j += stride) C.p[k][j] = 0;
}

template <bool STATS>
__device__ __forceinline__ void flush_stats(Counters<STATS>&, unsigned long long*) {}
template <>
__device__ __forceinline__ void flush_stats<true>(Counters<true>& ct, unsigned long long* out) {
  for (int s = 0; s < kNumStats; ++s) {
    uint32_t v = ct.c[s];
    if (s == kMaxStack) {
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(v, o, 64);
        v = other > v ? other : v;
      }
      if ((threadIdx.x & 63) == 0) atomicMax(&out[s], (unsigned long long)v);
    } else {
      unsigned long long w = v;
      for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
      if ((threadIdx.x & 63) == 0 && w) atomicAdd(&out[s], w);
    }
  }
}

// Pixel sample of main.cpp:758-787: sub-pixel position (jitter / tent), optional lens sample.
__device__ __forceinline__ void make_primary(const RenderParams& P, const DevCamera& cam, int x, int y, int i, int j,
                                             Rng& rng, F3& o, F3& d) {
  const int SPP = (int)P.spp_sqrt;
  float px, py;
  if (!P.antialiasing) {  // main.cpp:805-808
    px = (float)(x + 0.5);
    py = (float)(y + 0.5);
    primary_ray(cam, px, py, o, d);
    return;
  }
  if (P.sample_mode == P3D_SAMPLE_JITTER) {  // main.cpp:763-766
    px = x + (i + rng.rand_float()) / SPP;
    py = y + (j + rng.rand_float()) / SPP;
  } else {  // tent, main.cpp:767-773
    const double r1 = 2 * rng.erand48(), dx = r1 < 1 ? sqrt(r1) - 1 : 1 - sqrt(2 - r1);
    const double r2 = 2 * rng.erand48(), dy = r2 < 1 ? sqrt(r2) - 1 : 1 - sqrt(2 - r2);
    px = (float)(x + (0.5 + dx) / SPP);
    py = (float)(y + (0.5 + dy) / SPP);
  }
  if (P.depth_of_field) {  // main.cpp:776-784
    float lx, ly;
    if (P.sample_disk) {  // sampler.cpp:5-11; argument order as g++ evaluates it: y first
      do {
        const float b = rng.rand_float();
        const float a = rng.rand_float();
        lx = a * 2 - 1.0f;
        ly = b * 2 - 1.0f;
      } while (lx * lx + ly * ly + 0.0f * 0.0f >= 1.0f);
    } else {
      lx = (i + rng.rand_float()) / SPP;
      ly = (j + rng.rand_float()) / SPP;
    }
    primary_ray_lens(cam, lx, ly, px, py, o, d);
  } else {
    primary_ray(cam, px, py, o, d);
  }
}

// Four lanes per pixel (SUB = 4 of pt_kernel and of the anti-aliased whitted_kernel; a wave renders a 4x4-pixel tile).  The samples of a pixel are
// independent (own RNG stream each) but main.cpp:792-800 adds them up in sample order, and float
// addition does not commute: the four lanes take the pixel's samples in order from a shared
// counter, post each finished sample's radiance to a small ring in LDS, and lane 0 of the pixel
// adds the ring entries to the pixel colour strictly in sample order.  Same samples, same sum,
// four times more and four times shorter workgroups (DESIGN.md "Tile schedule": the path
// tracer's frame ends with a tail as long as its last tiles).
constexpr int kPtRing = 16;  // finished samples a pixel can hold before the oldest one is added
struct PtPixelShared {       // [..][pixel]: the 16 pixels of the tile are the fastest index (LDS banks)
  uint32_t next_start[16];   // next sample index to hand out (runs past the last sample: one ticket per finished lane)
  uint32_t next_add[16];     // next sample index to add to the pixel colour
  int32_t first_hit[16];
  float colour[3][16];             // the pixel's running sum (kept here, not in lane 0's registers)
  uint32_t tag[kPtRing][16];       // sample index + 1 of the radiance in that ring slot
  float radiance[kPtRing][3][16];
};

typedef __attribute__((address_space(3))) volatile PtPixelShared LdsPtPixelShared;

// Where a unit of the hand-off (handoff.hpp) lies in the image: tile pixels and, in front of every row, the halo slots.
struct UnitPlace {
  int c, r, x, y;  // tile column / tile row (output index), image pixel
  bool halo;       // a frame pixel rendered only for what it leaves on the stack: no output
  bool valid;
};
__device__ __forceinline__ int image_row(const RenderParams& P, int r) {
  const int sh = P.stripe_h > 0 ? P.stripe_h : 1, ss = P.stripe_h > 0 ? P.stripe_stride : 1;
  return P.y0 + (r / sh) * sh * ss + (r % sh);
}
__device__ __forceinline__ UnitPlace place_of_unit(const RenderParams& P, uint32_t unit) {
  const Handoff& H = P.hand;
  UnitPlace u;
  const uint32_t row = unit / H.row_units, j = unit - row * H.row_units;
  u.r = (int)row - P.row0;
  u.halo = j < H.halo;
  u.valid = true;
  if (u.halo) {
    const uint32_t fp = H.halo_pix[row * H.halo + j];
    u.valid = fp != kNoUnit;
    u.c = 0;
    u.x = (int)(fp % (uint32_t)P.sc.cam.res_x);
    u.y = (int)(fp / (uint32_t)P.sc.cam.res_x);
  } else {
    u.c = (int)(j - H.halo);
    u.x = P.x0 + u.c;
    u.y = image_row(P, u.r);
  }
  return u;
}
// pass 1 / check: the unit of a lane of a halo workgroup (8 chains of kHaloChain slots per wave)
__device__ __forceinline__ bool halo_unit_of_lane(const RenderParams& P, uint32_t lane, uint32_t& unit) {
  const Handoff& H = P.hand;
  const uint32_t slot = (blockIdx.x - P.tile_blocks) * kBlock + lane;
  const uint32_t row = slot / kHaloChain, j = slot % kHaloChain;
  if (H.halo == 0 || row >= H.rows || !H.row_chain[row]) return false;
  unit = row * H.row_units + j;
  return H.halo_pix[row * H.halo + j] != kNoUnit;
}

template <bool SPILL, class CT>
__device__ __forceinline__ void seed_stack(Stack& st, const Handoff& H, uint32_t pred, uint32_t slot_count, CT& ct) {
  stack_clear(st);
  const uint32_t n = slot_count & 0xffffu, slot = slot_count >> 16;
  const uint32_t at = n ? leftover_at(H, slot, pred) : 0u;
  for (uint32_t e = 0; e < n; ++e) {
    const uint2 v = H.entries[at + e];
    push<SPILL>(st, v.x, __uint_as_float(v.y), ct);
  }
}
// collect_stats under P3D_STACK_LITERAL (handoff.hpp: ucount / uch0).  The deepest stack goes straight to the global
// maximum: it is taken over everything that was traced, speculative passes included.
template <bool STATS>
__device__ __forceinline__ void store_unit_counters(const Handoff& H, uint32_t unit, const Counters<STATS>& ct, const uint32_t* ch0,
                                                    unsigned long long* stats) {
  if (!STATS) return;
  for (int s = 0; s < kNumStats; ++s)
    if (s != kMaxStack) H.ucount[(size_t)s * H.n_units + unit] = ct.get(s);
  for (int k = 0; k < kCh0Counters; ++k) H.uch0[(size_t)k * H.n_units + unit] = ch0[k];
  atomicMax(&stats[kMaxStack], (unsigned long long)ct.get(kMaxStack));
}
// a unit whose first closest hit was re-traced on a new leftover and came out the same: only that query's tests change
template <bool STATS>
__device__ __forceinline__ void replace_ch0_counters(const Handoff& H, uint32_t unit, const Counters<STATS>& cc, unsigned long long* stats) {
  if (!STATS) return;
  for (int k = 0; k < kCh0Counters; ++k) {
    const size_t at = (size_t)k * H.n_units + unit;
    const uint32_t now = cc.get(kNodeTests + k);
    H.ucount[(size_t)(kNodeTests + k) * H.n_units + unit] += now - H.uch0[at];
    H.uch0[at] = now;
  }
  atomicMax(&stats[kMaxStack], (unsigned long long)cc.get(kMaxStack));
}
// sum of the unit counters over the pixels of the tile (halo units are not pixels of the tile)
__global__ void __launch_bounds__(256) ucount_reduce_kernel(const Handoff H, uint32_t w, unsigned long long* stats) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n_pix = w * H.rows;
  for (int s = 0; s < kNumStats; ++s) {
    if (s == kMaxStack) continue;
    unsigned long long v = 0;
    for (uint32_t p = i; p < n_pix; p += gridDim.x * 256) {
      const uint32_t unit = (p / w) * H.row_units + H.halo + (p % w);
      v += H.ucount[(size_t)s * H.n_units + unit];
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[s], v);
  }
}

__device__ __forceinline__ bool same_first(float4 a, float4 b) {
  return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
         __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}
// The first closest hit of a unit's first touching sample, traced on whatever the stack holds (bvh.cpp:198-276).
template <bool SPILL, bool VOTE, class CT>
__device__ __forceinline__ float4 first_closest_hit(const RenderParams& P, const DevScene& sc, Stack& st, int x, int y, uint32_t sample, CT& ct) {
  Rng rng;
  rng.state = 0; rng.inc = 1;
  const int SPP = P.antialiasing ? (int)P.spp_sqrt : 1;
  const int si = (int)sample / SPP, sj = (int)sample % SPP;
  if (P.antialiasing) rng.seed_stream(P.seed, (uint32_t)(y * sc.cam.res_x + x), sample);
  F3 o, d, Pn;
  make_primary(P, sc.cam, x, y, si, sj, rng, o, d);
  RayS ray;
  ray_set(ray, o, d);
  Geom g;
  const int obj = closest_hit<P3D_ACCEL_BVH, SPILL, VOTE>(sc, st, ray, Pn, g, ct);
  return obj < 0 ? make_float4(0.f, 0.f, 0.f, __int_as_float(-1)) : make_float4(Pn.x, Pn.y, Pn.z, __int_as_float(obj));
}

// ---------------------------------------------------------------------------
// Whitted megakernel
// ---------------------------------------------------------------------------
//
// SUB = 4 (anti-aliased launches with at least four samples per pixel over a scene traversed from L2): four lanes per pixel, as in the path
// tracer — the samples of a pixel are handed out by ticket, finished samples wait in a ring in LDS and lane 0 of
// the pixel adds them in sample order (main.cpp:792-800 sums in that order).  An 8x8 tile of 4 x 4 = 16 chained
// samples per lane was the longest thing in such a frame; now a wave has a 4x4 tile and a quarter of the samples.
//
// LIT (P3D_STACK_LITERAL, handoff.hpp): 0 = the stack is emptied at every primary sample (one launch, nothing kept);
// 1 = pass 1: the pixel starts on an empty stack, the samples of the pixel hand the stack on, and the pixel's leftover,
// its touched flag and its first closest hit are recorded; 2 = the lanes take units from a work list, seed the stack
// with the predecessor's leftover, re-trace the first closest hit if the list entry asks for it, and render the unit again
// if that hit changed; 3 = round 0 of the hand-off over the TILES, check and repair in one launch (LDS-staged scenes): every lane
// whose unit starts on a non-empty leftover re-traces its first closest hit on it (what handoff_check_kernel does) and, if that
// hit changed, renders the unit again in the same wave (what the first work-list launch did with 64 unrelated pixels per
// wave: the units to repair of one tile are neighbours and walk the scene together - round 3, one box: the list launch
// cost the frames-in-flight loop of the bench workload 0.023 of its 0.126 ms per frame, profiles/r04/experiments).
// GHOSTS: the scene has a material that is transmissive AND reflective, i.e. LITERAL frames trace zero-weight reflection rays
// (whitted_sample.inc); without one the machinery is compiled out (pass 1 over the bench scene: 107 -> 96 VGPRs, five waves
// per SIMD like the per-pixel kernel).
template <int ACCEL, bool LDS, bool STATS, bool AA, bool SPILL = !LDS, int SUB = 1, int LIT = 0, bool GHOSTS = true>
__global__ void __launch_bounds__(kBlock, LIT >= 2 ? P3D_LIST_WAVES : (AA ? P3D_WHITTED_WAVES : (LDS ? ((LIT == 1 && GHOSTS) ? P3D_WHITTED_WAVES : P3D_WHITTED_LDS_WAVES) : P3D_WHITTED_GLOBAL_WAVES))) whitted_kernel(const RenderParams P) {
  static_assert(SUB == 1 || AA, "four lanes per pixel need more than one sample per pixel");
  static_assert(LIT == 0 || (ACCEL == P3D_ACCEL_BVH && SUB == 1), "only the BVH has a stack to hand on; one lane per pixel");
  constexpr bool REDO = LIT >= 2;               // launches that render units again, seeded with their predecessor's leftover
  constexpr bool GHOST = LIT != 0 && GHOSTS;    // zero-weight reflection rays are put aside and traced (whitted_sample.inc)
  extern __shared__ float4 smem[];
  uint32_t tx = 0, ty = 0;
  const bool halo_block = (LIT == 1 || LIT == 3) && blockIdx.x >= P.tile_blocks;
  if (LIT != 2 && !halo_block && !tile_of_block(P, tx, ty)) return;
  if (LIT == 2) {  // nothing on the list for this workgroup: leave before the scene is staged
    uint32_t n0 = __hip_atomic_load(P.hand.n_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    n0 = n0 > P.hand.list_cap ? P.hand.list_cap : n0;
    if ((size_t)blockIdx.x * P.hand.lanes >= n0) return;
  }
  const uint32_t lane = threadIdx.x;
  const uint32_t tws = SUB == 4 ? 2u : P.tile_w_shift, ths = SUB == 4 ? 2u : P.tile_h_shift;  // tile = (1 << tws) x (1 << ths) pixels
  // LIT == 3: which lanes have anything to re-trace is known before the scene is staged, and most waves leave here.  The
  // predecessor's record is read from meta0, pass 1's copy: a predecessor repaired by another wave of THIS launch changes
  // its meta word while this wave may still be reading it (Jacobi: everybody starts from pass 1's leftovers).
  uint32_t t_unit = 0, t_pred = 0, t_slot_count = 0;
  bool t_need = false;
  if (LIT == 3) {
    const Handoff& H0 = P.hand;
    bool act;
    if (halo_block) {
      act = halo_unit_of_lane(P, lane, t_unit);
    } else {
      const int c = (int)((tx << tws) + (lane & ((1u << tws) - 1u))), r = (int)((ty << ths) + (lane >> tws));
      act = lane < (1u << (tws + ths)) && c < P.w && r < P.h;
      t_unit = (uint32_t)(P.row0 + r) * H0.row_units + H0.halo + (uint32_t)c;
    }
    if (act && handoff_touched(H0, t_unit)) {
      const int pred = handoff_pred(H0, t_unit);
      if (pred >= 0) {
        const uint32_t pm = H0.meta0[pred];
        t_need = (pm & 0xffffu) != 0;  // otherwise the predecessor left nothing: pass 1's empty stack was right
        t_pred = (uint32_t)pred;
        t_slot_count = pm & 0x1ffffu;
      }
    }
    if (__ballot(t_need) == 0) return;
  }
  P3D_TL_BEGIN()
  const unsigned long long t_begin = (LIT < 2 && P.tile_cost) ? wall_clock64() : 0;
  DevScene sc = P.sc;
  stage_scene<LDS, ACCEL != P3D_ACCEL_BVH>(sc, P, smem);

  const uint32_t px = SUB == 4 ? lane >> 2 : lane;  // pixel of the tile this lane works for
  const uint32_t sub = SUB == 4 ? lane & 3u : 0u;
  Counters<STATS> ct;
  if (STATS) reinterpret_cast<Counters<true>&>(ct).clear();
  Stack st;
  stack_bind(st, smem, P.lds_scene_f4, lane, P.stack_cap, P.spill, P.level_stride, blockIdx.x * kBlock + lane);
  const uint32_t gid = blockIdx.x * kBlock + lane;
  // cold shading state behind the node stack (device_core.hpp ColdState): only the kernels that trade registers for waves
  constexpr bool COLD = !LDS && !AA;
  constexpr bool VOTE = !LDS;  // BVH loops by majority vote (device_core.hpp): scenes traversed from global memory only
  ColdState<COLD> cold;
  cold.bind(smem, P.lds_scene_f4 + stack_lds_f4(SPILL, P.stack_cap), lane);
  // per-pixel sample hand-out state behind the node stack (only allocated for SUB == 4); explicit LDS address space
  LdsPtPixelShared& shared = *(LdsPtPixelShared*)(smem + P.lds_scene_f4 + stack_lds_f4(SPILL, P.stack_cap));
  if (SUB == 4 && sub == 0) {
    shared.next_start[px] = 0;
    shared.next_add[px] = 0;
    shared.first_hit[px] = -1;
    shared.colour[0][px] = 0.0f; shared.colour[1][px] = 0.0f; shared.colour[2][px] = 0.0f;
    for (int k = 0; k < kPtRing; ++k) shared.tag[k][px] = 0;
  }

  const Handoff& H = P.hand;
  uint4* list_in = H.list_in;
  uint4* list_out = H.list_out;
  uint32_t* n_in_p = H.n_in;
  uint32_t* n_out_p = H.n_out;
  for (uint32_t round = 0;; ++round) {  // LIT == 2 in one persistent workgroup: a trip per round; otherwise one trip
    uint32_t n_in = 0;
    if (LIT == 2) {
      n_in = __hip_atomic_load(n_in_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      n_in = n_in > H.list_cap ? H.list_cap : n_in;
      if (n_in == 0) break;
      if (H.round_base + round >= H.max_rounds) {  // work left after the last round allowed: the frame is not the serial one
        if (lane == 0) atomicOr(P.status, kHoErrNoFixedPoint);
        break;
      }
      if (H.count && blockIdx.x == 0 && lane == 0) {  // rounds that found work: round 1 as a flag (the tile launch's own re-checks raise it too)
        if (H.round_base + round == 1) atomicOr(&H.counters[kHoRound1], 1u);
        else atomicAdd(&H.counters[kHoRounds], 1u);
      }
    }
    for (uint32_t chunk = blockIdx.x;; chunk += gridDim.x) {  // LIT == 2: 64 list entries per trip
      bool active;
      UnitPlace up;
      uint32_t unit = 0, pred = 0, pred_slot_count = 0, flags = 0;
      if (LIT == 2) {
        if ((size_t)chunk * H.lanes >= n_in) break;
        const uint32_t i = chunk * H.lanes + lane;
        active = lane < H.lanes && i < n_in;
        up.c = up.r = up.x = up.y = 0; up.halo = false; up.valid = false;
        if (active) {
          const uint32_t* e = reinterpret_cast<const uint32_t*>(list_in + i);
          unit = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pred = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pred_slot_count = __hip_atomic_load(e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          flags = __hip_atomic_load(e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          up = place_of_unit(P, unit);
          active = up.valid;
        }
      } else if (LIT == 3) {  // the tile's lanes that start on a non-empty leftover: check first, like a list entry with flag 1
        active = t_need;
        unit = t_unit; pred = t_pred; pred_slot_count = t_slot_count; flags = 1u;
        up.c = up.r = up.x = up.y = 0; up.halo = false; up.valid = false;
        if (active) {
          up = place_of_unit(P, unit);
          active = up.valid;
        }
      } else if (halo_block) {
        active = halo_unit_of_lane(P, lane, unit);
        up.c = up.r = up.x = up.y = 0; up.halo = true; up.valid = active;
        if (active) up = place_of_unit(P, unit);
      } else {
        up.c = (int)((tx << tws) + (px & ((1u << tws) - 1u)));
        up.r = (int)((ty << ths) + (px >> tws));
        up.halo = false;
        active = px < (1u << (tws + ths)) && up.c < P.w && up.r < P.h;
        up.valid = active;
        up.x = P.x0 + up.c;
        up.y = image_row(P, up.r);
        if (LIT == 1) unit = (uint32_t)(P.row0 + up.r) * H.row_units + H.halo + (uint32_t)up.c;
      }

      bool unit_touched = false;
      uint32_t unit_ch0[kCh0Counters] = {0, 0, 0, 0, 0};
      if (STATS && LIT != 0) ct.clear();  // LITERAL: counters per unit (store_unit_counters), not per lane
      if (REDO && active) {  // seed with the predecessor's leftover; re-trace the first closest hit if asked to
        seed_stack<SPILL>(st, H, pred, pred_slot_count, ct);
        if (flags & 1u) {
          if (H.count) atomicAdd(&H.counters[kHoChecked], 1u);
          const float4 now = first_closest_hit<SPILL, !LDS>(P, sc, st, up.x, up.y, AA ? H.first_sample[unit] : 0u, ct);
          if (same_first(now, H.first[unit])) {  // nothing this unit computes can differ
            active = false;
            replace_ch0_counters<STATS>(H, unit, ct, P.stats);
          } else {
            seed_stack<SPILL>(st, H, pred, pred_slot_count, ct);
          }
          if (STATS) ct.clear();
        }
        if (active && H.count) {
          atomicAdd(&H.counters[kHoRedone], 1u);
          if (LIT == 3) atomicOr(&H.counters[kHoRound0], 1u);  // (the tile launch is a round of its own if it repaired anything)
        }
      }
      if (LIT == 1) stack_clear(st);
      uint32_t leave_n = 0, leave_slot = 0;  // LIT != 0: entries this lane's unit leaves, stored after the lanes have come together again
      bool leave_succ = false;

      if (active) {
        const int c = up.c, r = up.r, x = up.x, y = up.y;
        const int SPP = AA ? (int)P.spp_sqrt : 1;
        if (sub == 0 && !(LIT != 0 && up.halo)) ct.add(kPixels);

        F3 color = f3(0, 0, 0);
        int first_hit = -1;
        Rng rng;
        rng.state = 0; rng.inc = 1;
#ifdef P3D_PT_PROFILE
        RegionProf prof; prof.init();
#endif
        // one sample per pixel: the hit ID goes to memory when it is known instead of riding through the whole chain in a register
        constexpr bool EARLY_HIT = !AA && !LDS;  // (kernels that trade registers for waves; over an LDS-staged scene the extra store only costs)
        if (EARLY_HIT && P.hit_id && !(LIT != 0 && up.halo)) P.hit_id[(size_t)r * P.w + c] = -1;  // a primary ray that is never traced (none: every sample is) would leave -1
#define P3D_FIRST_HIT(obj)                                                                                   \
  do {                                                                                                       \
    if (EARLY_HIT) { if (P.hit_id && !(LIT != 0 && up.halo)) P.hit_id[(size_t)r * P.w + c] = (obj); }        \
    else first_hit = (obj);                                                                                  \
  } while (0)
        if (SUB == 1) {
          for (int si = 0; si < SPP; ++si) {
            for (int sj = 0; sj < SPP; ++sj) {
#include "whitted_sample.inc"
              color = color + result;
            }
          }
        } else {
          // Lanes of one wave wait for each other here (full ring, lane 0 waiting for the last samples of its pixel),
          // so the loop is wave-uniform — a ballot every lane takes part in decides its end, a waiting lane sits out
          // the rest of the trip — exactly as in pt_kernel (where a per-lane `continue` got split off as an inner loop).
          const int n_samples = SPP * SPP;
          const unsigned long long trips_max = P.debug_trip_bound ? (unsigned long long)P.debug_trip_bound : (unsigned long long)n_samples * 2ull + 1024ull;
          uint32_t trips_left = trips_max > 0xffffffffull ? 0xffffffffu : (uint32_t)trips_max;
          const uint32_t spp_magic = (uint32_t)((0x100000000ull + (unsigned)SPP - 1) / (unsigned)SPP);
          bool done = false, holding = false;
          int s = 0;
          while (true) {
            if (trips_left-- == 0) {  // backstop: the pixel would be written with samples missing -> the call fails
              if (!done) atomicOr(P.status, kHoErrTrips);
              done = true;
            }
            if (__ballot(!done) == 0) break;
            if (done) continue;
            if (sub == 0) {  // add finished samples to the pixel colour, strictly in sample order
              uint32_t na = shared.next_add[px];
              if (na < (uint32_t)n_samples && shared.tag[na % kPtRing][px] == na + 1) {
                F3 sum = f3(shared.colour[0][px], shared.colour[1][px], shared.colour[2][px]);
                do {
                  const int k = (int)(na % kPtRing);
                  sum = sum + f3(shared.radiance[k][0][px], shared.radiance[k][1][px], shared.radiance[k][2][px]);
                  ++na;
                } while (na < (uint32_t)n_samples && shared.tag[na % kPtRing][px] == na + 1);
                shared.colour[0][px] = sum.x; shared.colour[1][px] = sum.y; shared.colour[2][px] = sum.z;
                shared.next_add[px] = na;
              }
            }
            if (!holding) {
              s = (int)__hip_atomic_fetch_add(&shared.next_start[px], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              holding = true;
            }
            if (s >= n_samples) {  // nothing left to start: finished, except lane 0 while samples remain to be added
              done = sub != 0 || shared.next_add[px] >= (uint32_t)n_samples;
              continue;
            }
            if ((uint32_t)s >= shared.next_add[px] + kPtRing) continue;  // wait for room in the ring
            holding = false;
            const int si = (int)__umulhi((uint32_t)s, spp_magic);  // s / SPP (exact: s * SPP < 2^32, SPP >= 2)
            const int sj = s - si * SPP;
#include "whitted_sample.inc"
            if (s == 0) shared.first_hit[px] = first_hit;
            const int k = s % kPtRing;
            shared.radiance[k][0][px] = result.x; shared.radiance[k][1][px] = result.y; shared.radiance[k][2][px] = result.z;
            shared.tag[k][px] = (uint32_t)s + 1;
          }
          if (sub == 0) {
            color = f3(shared.colour[0][px], shared.colour[1][px], shared.colour[2][px]);
            first_hit = shared.first_hit[px];
          }
        }
#undef P3D_FIRST_HIT
        if (AA) color = color / (float)(SPP * SPP);  // main.cpp:800
#ifdef P3D_PT_PROFILE
        PT_REGION(8)
        prof.flush();
#endif

        if ((SUB == 1 || sub == 0) && !(LIT != 0 && up.halo)) {  // SUB == 4: lane 0 of the pixel holds its colour
          const size_t k = (size_t)r * P.w + c;
          if (P.rgb) {
            P.rgb[3 * k] = color.x; P.rgb[3 * k + 1] = color.y; P.rgb[3 * k + 2] = color.z;
          }
          if (!EARLY_HIT && P.hit_id) P.hit_id[k] = first_hit;
          if (P.rgb8) {  // main.cpp:814-820
            F3 gc = color;
            if (P.gamma != 1.0f) {
              const double ig = (double)(1 / P.gamma);
              gc = f3((float)pow_spec((double)color.x, ig), (float)pow_spec((double)color.y, ig), (float)pow_spec((double)color.z, ig));
            }
            P.rgb8[3 * k] = u8fromfloat(gc.x); P.rgb8[3 * k + 1] = u8fromfloat(gc.y); P.rgb8[3 * k + 2] = u8fromfloat(gc.z);
          }
        }

#ifndef P3D_ABL_NO_RECORDS  // (timing ablation only: pass 1 keeps no records, the frame is the per-pixel one)
        if (LIT == 1) {  // what this unit leaves behind (slot 0), whether it touched the stack, its first closest hit
          uint32_t meta = 0;
          if (unit_touched) {
            uint32_t n = (uint32_t)st.sp;
            if (n > H.cap) { atomicOr(P.status, kHoErrLeftoverCap); n = H.cap; }
            leave_n = n;
            meta = n | kMetaTouched;
            atomicOr(&H.touched[unit >> 5], 1u << (unit & 31u));
          }
          H.meta[unit] = meta;
          H.meta0[unit] = meta;  // pass 1's record, never changed by a repair (what the tile launch of round 0 reads of its predecessors)
        }
#endif
        if (LIT != 0 && STATS) {
          if (up.halo) ct.clear();  // a halo pixel is rendered for its leftover, it is not a pixel of this tile
          store_unit_counters<STATS>(H, unit, ct, unit_ch0, P.stats);
        }
        if (REDO) {  // a changed leftover goes to the unit's other slot and sends the successor to the next round
          const uint32_t meta = H.meta[unit];
          const uint32_t cur = (meta >> 16) & 1u, cnt = meta & 0xffffu;
          uint32_t n = (uint32_t)st.sp;
          if (n > H.cap) { atomicOr(P.status, kHoErrLeftoverCap); n = H.cap; }
          bool same = n == cnt;
          const uint32_t old_at = (same && n) ? leftover_at(H, cur, unit) : 0u;
          for (uint32_t e = 0; same && e < n; ++e) {
            const uint2 a = H.entries[old_at + e], b = stack_read<SPILL>(st, (int)e);
            same = a.x == b.x && a.y == b.y;
          }
          if (!same) {
            leave_n = n;
            leave_slot = cur ^ 1u;
            leave_succ = true;
            H.meta[unit] = n | (leave_slot << 16) | kMetaTouched;
          }
        }
      }
#ifdef P3D_ABL_NO_RECORDS
      if (REDO) {
#else
      if (LIT != 0) {  // every lane: room in the leftover pool (one atomic per wave), then the entries
#endif
        const uint32_t at = leftover_alloc(H, leave_n, leave_slot, unit, P.status, LIT == 1 && !LDS);
        if (at != kNoUnit) {
          for (uint32_t e = 0; e < leave_n; ++e) H.entries[at + e] = stack_read<SPILL>(st, (int)e);
        } else if (leave_n) {  // pool full: the call fails (status); the records must still not point anywhere
          leave_n = 0;
          H.meta[unit] = (leave_slot << 16) | kMetaTouched;
        }
        if (REDO && leave_succ) {
          const int succ = handoff_succ(H, unit);
          if (succ >= 0) handoff_append(list_out, n_out_p, H.list_cap, P.status, make_uint4((uint32_t)succ, unit, (leave_slot << 16) | leave_n, 1u));
        }
      }
      if (LIT != 2) break;
    }
    if (LIT != 2 || !H.persistent) break;
    // next round of the persistent workgroup: what was written becomes the work list; the one just done is emptied
    __threadfence();
    if (lane == 0) __hip_atomic_store(n_in_p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    uint4* tl = list_in; list_in = list_out; list_out = tl;
    uint32_t* tn = n_in_p; n_in_p = n_out_p; n_out_p = tn;
  }
  if (STATS && LIT == 0) flush_stats<STATS>(ct, P.stats);  // LITERAL: ucount_reduce_kernel
  if (LIT < 2 && !halo_block) record_tile_cost(P, tx, ty, t_begin);
  P3D_TL_END()
}

// Object::intercepts / Object::getNormal / Scene::GetSkyboxColor for batches (host-class forwarding, unit parity)
struct ObjectQueryParams {
  DevScene sc;
  uint32_t object, n;
  const float* a;   // origins | points | directions
  float* b;         // directions (in/out) | normals | rgb
  uint8_t* hit;
  float* t;
};
template <int WHAT>  // 0 intercepts, 1 normal, 2 skybox colour
__global__ void __launch_bounds__(kBlock) object_query_kernel(const ObjectQueryParams P) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= P.n) return;
  Counters<false> ct;
  if (WHAT == 0) {
    const Geom g = load_geom(P.sc.ogeom, P.object);
    RayS ray;
    ray_set(ray, f3(P.a[3 * i], P.a[3 * i + 1], P.a[3 * i + 2]), f3(P.b[3 * i], P.b[3 * i + 1], P.b[3 * i + 2]));
    float t = 0.0f;
    const bool h = intercepts(g, ray, t, ct);
    P.hit[i] = h ? 1 : 0;
    if (h) P.t[i] = t;
    P.b[3 * i] = ray.d.x; P.b[3 * i + 1] = ray.d.y; P.b[3 * i + 2] = ray.d.z;
  } else if (WHAT == 1) {
    const Geom g = load_geom(P.sc.ogeom, P.object);
    const F3 nrm = get_normal(g, P.sc.normals, f3(P.a[3 * i], P.a[3 * i + 1], P.a[3 * i + 2]));
    P.b[3 * i] = nrm.x; P.b[3 * i + 1] = nrm.y; P.b[3 * i + 2] = nrm.z;
  } else {
    const F3 c = skybox_color(P.sc, f3(P.a[3 * i], P.a[3 * i + 1], P.a[3 * i + 2]));
    P.b[3 * i] = c.x; P.b[3 * i + 1] = c.y; P.b[3 * i + 2] = c.z;
  }
}

// For every tile row that starts a chain of its own (its predecessor in the FRAME is not the end of the tile row above:
// first row of a stripe, any row of a sub-rectangle, a tile that does not start at the frame's first pixel): the frame
// pixels in front of the row whose leftovers have to be known for the row's first pixel to start on the right stack.
//
// A pixel "touches" the stack if the primary ray of one of its samples gets past the root test (bvh.cpp:203-205).  What
// the row starts on is the leftover of the last touching pixel before it, which depends on ITS predecessor's leftover,
// and so on back to the frame's first pixel - but only through the result of each pixel's first closest hit
// (handoff.hpp).  The chain can therefore be cut at a pixel whose first closest hit provably does not depend on the
// stack it finds.  The stale entries of a found stack are popped AFTER the query's own traversal and the popped
// subtrees are walked with the query's ray (bvh.cpp:256-265); that changes the result only if (a) one of the primitive
// tests it runs returns a hit nearer than the own traversal's, or (b) a sphere test re-normalises the ray's direction
// (scene.cpp:156, ray.h:16-18).  So a pixel is CERTIFIED if, after its own traversal on an empty stack,
//   (a) NO primitive of the scene - all of them are tested, whatever boxes they sit in - is hit nearer than tmin by the
//       ray as the traversal left it, and
//   (b) the scene has no sphere, or normalising that ray's direction once more leaves its bits unchanged (then every
//       further normalisation is the identity).
// Under (a) and (b) no sequence of stale subtree walks can change tmin, the hit or the direction, whatever the stack
// held: the pixel rendered on an empty stack is the pixel of the serial frame, its leftover included.  The kernel walks
// back from the row, takes the touching pixels most recent first, and stops at the first one it can certify (or at the
// frame's first touching pixel, whose stack IS empty); the pixels it collected are rendered in front of the row for
// their leftovers, the oldest on an empty stack, each next one checked against its predecessor's leftover like any
// other unit.  If `max_chain` (<= kHaloChain) pixels are collected without a certificate and an older touching pixel
// exists, the row cannot be started exactly: kHoErrHalo is raised and the call fails (P3D_ERR_CAPACITY) instead of
// returning a frame that is only probably right.
//
// One workgroup per row.  Every thread looks at one pixel of the kHaloFindThreads before the row; candidates are taken
// one at a time: wave 0 runs the own traversal (all lanes the same ray), all threads share the all-primitives test.
constexpr int kHaloFindThreads = 1024;
struct HaloFindShared {
  unsigned long long touched[kHaloFindThreads / kBlock];
  float tmin, dx, dy, dz;
  uint32_t settled, closer;
};
__device__ __forceinline__ bool pixel_first_touching_ray(const RenderParams& P, const NodeRec& root, long long f, RayS& ray) {
  const DevScene& sc = P.sc;
  const int res_x = sc.cam.res_x, SPP = P.antialiasing ? (int)P.spp_sqrt : 1;
  const int x = (int)(f % res_x), y = (int)(f / res_x);
  for (int s = 0; s < SPP * SPP; ++s) {
    Rng rng;
    rng.state = 0; rng.inc = 1;
    if (P.antialiasing) rng.seed_stream(P.seed, (uint32_t)(y * res_x + x), (uint32_t)s);
    F3 o, d;
    make_primary(P, sc.cam, x, y, s / SPP, s % SPP, rng, o, d);
    ray_set(ray, o, d);
    float t;
    if (aabb_intercepts(xyz(root.lo), xyz(root.hi), ray, t, false)) return true;
  }
  return false;
}
// Which rows of a tile start a chain of their own: a function of the tile alone, worked out on the launch stream so
// that two tiles queued on one stream cannot see each other's flags.  Also resets the verdict of the search that follows.
struct RowChainParams {
  uint8_t* chain;
  uint32_t* verdict;
  int32_t rows, x0, y0, w, res_x, sh, ss;
};
__global__ void __launch_bounds__(256) row_chain_kernel(const RowChainParams C) {
  const int r = (int)(blockIdx.x * 256 + threadIdx.x);
  if (r == 0) *C.verdict = 0;
  if (r >= C.rows) return;
  auto image_y = [&](int row) { return (long long)C.y0 + (long long)(row / C.sh) * C.sh * C.ss + (row % C.sh); };
  const long long y = image_y(r);
  const bool full_width = C.x0 == 0 && C.w == C.res_x;
  C.chain[r] = r == 0 ? !(C.x0 == 0 && y == 0) : !(full_width && y == image_y(r - 1) + 1);
}

__global__ void __launch_bounds__(kHaloFindThreads) halo_find_kernel(const RenderParams P, uint32_t* halo_pix, uint32_t* verdict, uint32_t max_chain,
                                                                     uint32_t has_spheres, uint32_t window, uint32_t backing_stride) {
  extern __shared__ float4 smem[];  // wave 0's node stack: window * 64 entries
  __shared__ HaloFindShared sh;
  const Handoff& H = P.hand;
  const uint32_t row = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (row >= H.rows || !H.row_chain[row]) return;
  const DevScene& sc = P.sc;
  const long long f0 = (long long)image_row(P, (int)row) * sc.cam.res_x + P.x0;  // the row's first pixel: search below it
  const NodeRec root = load_node(sc.nodes, 0);
  Counters<false> ct;
  uint32_t found = 0;
  bool certified = false, older_exists = false;
  for (long long base = f0; base > 0 && !certified && !older_exists; base -= kHaloFindThreads) {
    const long long f = base - 1 - (long long)tid;  // thread 0 looks at the most recent pixel
    RayS ray;
    const bool touched = f >= 0 && pixel_first_touching_ray(P, root, f, ray);
    const unsigned long long m = __ballot(touched);
    if (lane == 0) sh.touched[wave] = m;
    __syncthreads();
    for (uint32_t w = 0; w < kHaloFindThreads / kBlock && !certified && !older_exists; ++w) {
      unsigned long long mask = sh.touched[w];
      while (mask && !certified && !older_exists) {  // workgroup-uniform
        const int l = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        if (found == max_chain) {  // the chain is full and here is a touching pixel it would still need
          older_exists = true;
          break;
        }
        const long long pix = base - 1 - (long long)(w * kBlock + (uint32_t)l);
        if (tid == 0) halo_pix[row * kHaloChain + (kHaloChain - 1 - found)] = (uint32_t)pix;
        ++found;
        // ---- certificate ----
        pixel_first_touching_ray(P, root, pix, ray);  // (every thread: the same ray)
        if (wave == 0) {
          Stack st;
          stack_bind(st, smem, 0, lane, (int)window, P.spill, backing_stride, blockIdx.x * kBlock + lane);
          F3 hp;
          Geom g;
          float tmin = FLT_MAX;
          RayS left;
          bvh_closest<true>(sc, st, ray, hp, g, ct, nullptr, &tmin, &left);
          if (lane == 0) {
            sh.tmin = tmin;
            sh.dx = left.d.x; sh.dy = left.d.y; sh.dz = left.d.z;
            sh.settled = same_bits(normalized(left.d), left.d) ? 1u : 0u;
            sh.closer = 0;
          }
        }
        __syncthreads();
        const float tmin = sh.tmin;
        const bool settled = sh.settled != 0;
        bool closer = false;
        if (settled || !has_spheres) {
          RayS r = ray;
          r.d = f3(sh.dx, sh.dy, sh.dz);
          r.inv = f3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
          r.odd_inv = inv_is_odd(r.inv);
          r.settled = settled;  // a settled direction is returned as it is by the sphere test (ray.h:16-18 would give the same bits)
          for (uint32_t i = tid; i < sc.n_objs && !closer; i += kHaloFindThreads) {
            const Geom g = load_geom(sc.ogeom, i);
            float t;
            closer = intercepts(g, r, t, ct) && t < tmin;
          }
        }
        if (closer) sh.closer = 1;  // (benign race: every writer stores the same value)
        __syncthreads();
        certified = (settled || !has_spheres) && sh.closer == 0;
        __syncthreads();
      }
    }
    __syncthreads();
  }
  if (older_exists && !certified && tid == 0) atomicOr(verdict, kHoErrHalo);
  if (tid < kHaloChain - found) halo_pix[row * kHaloChain + tid] = kNoUnit;
}

// The check of round 0 of the hand-off for a whole launch: every unit that touched the stack and whose predecessor left something
// re-traces its first closest hit on that leftover; the units whose hit changed go on the work list of the redo launch.
template <bool LDS, bool SPILL, bool STATS = false>
__global__ void __launch_bounds__(kBlock) handoff_check_kernel(const RenderParams P) {
  extern __shared__ float4 smem[];
  uint32_t tx = 0, ty = 0;
  const bool halo_block = blockIdx.x >= P.tile_blocks;
  if (!halo_block && !tile_of_block(P, tx, ty)) return;
  const uint32_t lane = threadIdx.x;
  const Handoff& H = P.hand;
  uint32_t unit = 0;
  bool active;
  if (halo_block) {
    active = halo_unit_of_lane(P, lane, unit);
  } else {
    const uint32_t tws = P.tile_w_shift, ths = P.tile_h_shift;
    const int c = (int)((tx << tws) + (lane & ((1u << tws) - 1u))), r = (int)((ty << ths) + (lane >> tws));
    active = lane < (1u << (tws + ths)) && c < P.w && r < P.h;
    unit = (uint32_t)(P.row0 + r) * H.row_units + H.halo + (uint32_t)c;
  }
  // which lanes have anything to re-trace is known before the scene is staged: most waves leave here
  int pred = -1;
  uint32_t pm = 0;
  if (active && handoff_touched(H, unit)) {
    pred = handoff_pred(H, unit);
    if (pred >= 0) pm = H.meta[pred];
  }
  const bool need = (pm & 0xffffu) != 0;  // otherwise the predecessor left nothing: pass 1's empty stack was right
  if (__ballot(need) == 0) return;
  DevScene sc = P.sc;
  stage_scene<LDS, false>(sc, P, smem);  // (the hand-off exists for the BVH only)
  if (!need) return;
  Counters<STATS> ct;
  ct.clear();
  Stack st;
  stack_bind(st, smem, P.lds_scene_f4, lane, P.stack_cap, P.spill, P.level_stride, blockIdx.x * kBlock + lane);
  const UnitPlace up = place_of_unit(P, unit);
  const uint32_t slot_count = pm & 0x1ffffu;
  seed_stack<SPILL>(st, H, (uint32_t)pred, slot_count, ct);
  if (H.count) atomicAdd(&H.counters[kHoChecked], 1u);
  const float4 now = first_closest_hit<SPILL, !LDS>(P, sc, st, up.x, up.y, P.antialiasing ? H.first_sample[unit] : 0u, ct);
  if (!same_first(now, H.first[unit]))
    handoff_append(H.list_out, H.n_out, H.list_cap, P.status, make_uint4(unit, (uint32_t)pred, slot_count, 0u));
  else
    replace_ch0_counters<STATS>(H, unit, ct, P.stats);
}

// The same round over the list pass 1 wrote (Handoff::check_list: the units that left something): entry -> the unit that starts on
// that leftover (the next one that touched the stack) -> re-trace its first closest hit on it.  Every lane has work; one launch
// for the whole tile however many launches pass 1 took.
template <bool LDS, bool SPILL, bool STATS = false>
__global__ void __launch_bounds__(kBlock) handoff_check_list_kernel(const RenderParams P) {
  extern __shared__ float4 smem[];
  const Handoff& H = P.hand;
  uint32_t n = __hip_atomic_load(H.check_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  n = n > H.n_units ? H.n_units : n;
  if ((size_t)blockIdx.x * kBlock >= n) return;  // nothing on the list for this workgroup: leave before the scene is staged
  const uint32_t lane = threadIdx.x;
  DevScene sc = P.sc;
  stage_scene<LDS, false>(sc, P, smem);
  Stack st;
  stack_bind(st, smem, P.lds_scene_f4, lane, P.stack_cap, P.spill, P.level_stride, blockIdx.x * kBlock + lane);
  for (uint32_t chunk = blockIdx.x; (size_t)chunk * kBlock < n; chunk += gridDim.x) {
    const uint32_t i = chunk * kBlock + lane;
    if (i >= n) continue;
    const uint32_t pred = H.check_list[i];
    const uint32_t pm = H.meta[pred];
    const int succ = handoff_succ(H, pred);
    if (succ < 0 || (pm & 0xffffu) == 0) continue;  // nobody starts on it (end of a chain) / the pool was full (the call fails)
    const uint32_t unit = (uint32_t)succ;
    Counters<STATS> ct;
    ct.clear();
    const UnitPlace up = place_of_unit(P, unit);
    const uint32_t slot_count = pm & 0x1ffffu;
    seed_stack<SPILL>(st, H, pred, slot_count, ct);
    if (H.count) atomicAdd(&H.counters[kHoChecked], 1u);
    const float4 now = first_closest_hit<SPILL, !LDS>(P, sc, st, up.x, up.y, P.antialiasing ? H.first_sample[unit] : 0u, ct);
    if (!same_first(now, H.first[unit]))
      handoff_append(H.list_out, H.n_out, H.list_cap, P.status, make_uint4(unit, pred, slot_count, 0u));
    else
      replace_ch0_counters<STATS>(H, unit, ct, P.stats);
  }
}

// Round 1 of the hand-off as a LIGHT launch.  List B holds the successors of the units whose leftover changed in round 0; nearly
// all of them only need their first closest hit re-traced on the new leftover to find that nothing changes.  Until round 4 that was
// done by the work-list instantiation of whitted_kernel (LIT = 2: the whole Whitted chain, 128 VGPRs + scratch), whose few waves had
// to wait for a double-width slot among the other frames' pass-1 waves; this kernel only checks, and passes the rare entry whose hit
// does change (or that asks for no check) on to list C, which the persistent workgroup behind it renders again.
template <bool LDS, bool SPILL, bool STATS = false>
__global__ void __launch_bounds__(kBlock) handoff_check_entries_kernel(const RenderParams P) {
  extern __shared__ float4 smem[];
  const Handoff& H = P.hand;
  uint32_t n = __hip_atomic_load(H.n_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  n = n > H.list_cap ? H.list_cap : n;
  if ((size_t)blockIdx.x * H.lanes >= n) return;  // nothing on the list for this workgroup: leave before the scene is staged
  const uint32_t lane = threadIdx.x;
  if (H.round_base >= H.max_rounds) {  // work left after the last round allowed: the frame is not the serial one
    if (lane == 0) atomicOr(P.status, kHoErrNoFixedPoint);
    return;
  }
  if (H.count && blockIdx.x == 0 && lane == 0) atomicOr(&H.counters[kHoRound1], 1u);
  DevScene sc = P.sc;
  stage_scene<LDS, false>(sc, P, smem);
  Stack st;
  stack_bind(st, smem, P.lds_scene_f4, lane, P.stack_cap, P.spill, P.level_stride, blockIdx.x * kBlock + lane);
  for (uint32_t chunk = blockIdx.x; (size_t)chunk * H.lanes < n; chunk += gridDim.x) {
    const uint32_t i = chunk * H.lanes + lane;
    if (lane >= H.lanes || i >= n) continue;
    const uint32_t* e = reinterpret_cast<const uint32_t*>(H.list_in + i);
    const uint32_t unit = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t pred = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t slot_count = __hip_atomic_load(e + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t flags = __hip_atomic_load(e + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const UnitPlace up = place_of_unit(P, unit);
    if (!up.valid) continue;
    if (flags & 1u) {
      Counters<STATS> ct;
      ct.clear();
      seed_stack<SPILL>(st, H, pred, slot_count, ct);
      if (H.count) atomicAdd(&H.counters[kHoChecked], 1u);
      const float4 now = first_closest_hit<SPILL, !LDS>(P, sc, st, up.x, up.y, P.antialiasing ? H.first_sample[unit] : 0u, ct);
      if (same_first(now, H.first[unit])) {  // nothing this unit computes can differ
        replace_ch0_counters<STATS>(H, unit, ct, P.stats);
        continue;
      }
    }
    handoff_append(H.list_out, H.n_out, H.list_cap, P.status, make_uint4(unit, pred, slot_count, 0u));  // rendered again by the launch behind this one
  }
}

// ---------------------------------------------------------------------------
// Batched queries (unit-level parity of the traversal back ends)
// ---------------------------------------------------------------------------
struct TraceParams {
  DevScene sc;
  uint32_t n;
  const float* origin;
  const float* direction;
  int32_t* hit_id;
  float* t;
  float* hit_point;
  uint8_t* occluded;
  uint2* spill;
  uint32_t spill_stride;
  int32_t stack_cap;
};

template <int ACCEL, bool ANY>
__global__ void __launch_bounds__(kBlock) trace_kernel(const TraceParams P) {
  extern __shared__ float4 smem[];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  Stack st;
  stack_bind(st, smem, 0, threadIdx.x, P.stack_cap, P.spill, P.spill_stride, i);
  if (i >= P.n) return;
  Counters<false> ct;
  RayS ray;
  ray_set(ray, f3(P.origin[3 * i], P.origin[3 * i + 1], P.origin[3 * i + 2]),
          f3(P.direction[3 * i], P.direction[3 * i + 1], P.direction[3 * i + 2]));
  if (ANY) {
    P.occluded[i] = any_hit<ACCEL, true, true>(P.sc, st, ray, ct) ? 1 : 0;
  } else {
    F3 hp = f3(0, 0, 0);
    Geom g;
    float t = FLT_MAX;
    const int obj = closest_hit<ACCEL, true, true>(P.sc, st, ray, hp, g, ct, nullptr, &t);
    P.hit_id[i] = obj;
    if (P.t) P.t[i] = obj < 0 ? FLT_MAX : t;
    if (obj < 0) hp = f3(0, 0, 0);
    if (P.hit_point) {
      P.hit_point[3 * i] = hp.x; P.hit_point[3 * i + 1] = hp.y; P.hit_point[3 * i + 2] = hp.z;
    }
  }
}

}  // namespace p3d

#include "wavefront.hpp"
