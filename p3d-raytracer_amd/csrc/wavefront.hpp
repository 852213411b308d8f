// wavefront.hpp — the Whitted chain as one launch per level with ray compaction and binning between levels
// (p3d_config.chain_launch = P3D_CHAIN_PER_LEVEL).  Included at the end of kernels.hpp: uses its RenderParams, tile map,
// unit placement and the per-level body whitted_level.inc.
#pragma once

namespace p3d {

// ---------------------------------------------------------------------------
// Whitted chain as one launch per level ("wavefront"), for scenes traversed from L2
// ---------------------------------------------------------------------------
// In the megakernel a lane keeps its pixel from the primary ray to the end of the reflection chain.  Over a big scene
// most of the work is in the reflection levels (100k triangles, depth 6: 82 % of the shaded hits), where the lanes of
// an 8x8 tile die one by one and the survivors walk unrelated parts of the tree: 22 % of the lanes of an issued VALU
// instruction are active.  The idea tested here: fill the waves again.  Every level is a launch of its own: level 0 takes the pixels tile by tile as
// before; each lane that spawns a child ray appends it to a queue; the next launch takes 64 queue entries per wave, so
// its lanes are all alive.  What a lane's pixel carries from level to level is (a) the ray, (b) the hit_stack as the
// feelers of this level left it (Q2: the next level's closest-hit query drains it) — kept in the unit's leftover
// record of handoff.hpp, which after the last level IS the pixel's leftover for the cross-pixel hand-off — and (c) the
// {local colour, child weight} record per level; wf_fold_kernel folds those bottom-up with the per-level clamp
// (main.cpp:305-307).  Same queries in the same per-pixel order as the megakernel: same bits.
//
// MEASURED (profiles/r02/experiments/README.md): bit-identical, 13 % fewer instructions, but slower than the megakernel
// (100k triangles 2048x2048: 25.3 against 19.6 ms) — with every lane alive at the start of a level the reflection levels
// still run at 20-29 % lane utilisation: the idle lanes wait for the slowest traversal of their wave inside one query.
// Hence opt-in only.
//
// The child rays are staged in kWfSegments segments with a counter each (a single counter would serialise ~40k returning
// atomics per level); a producer appends to the segment of its blockIdx.x & 7.
//
// Between two levels the rays are put in order: a ray's bin is the Morton code of the cell of its origin (16 cells
// per axis of the root box) followed by its direction octant; rays of one wave then start in the same few cells and
// walk the same top of the tree, and every XCD takes one eighth of the sorted queue, i.e. one region of the scene,
// which is what its L2 then holds.  A counting sort: the level kernel counts the bins as it stages its child rays,
// wf_scan_kernel turns the counts into first slots, wf_scatter_kernel moves every staged ray to its slot.  (Without
// the sort the queue is in arrival order, a wave mixes rays from unrelated tiles, and the per-level launches are
// slower than the megakernel: 100k triangles 2048x2048 22.5 ms against 19.7 ms.)
constexpr uint32_t kWfCellsPerAxis = 16;
constexpr uint32_t kWfBins = kWfCellsPerAxis * kWfCellsPerAxis * kWfCellsPerAxis * 8;
__device__ __forceinline__ uint32_t spread4(uint32_t v) {  // 4 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ uint32_t ray_bin(const RenderParams& P, F3 o, F3 d) {
  const F3 c = (o - P.wf_cell_origin) * P.wf_cell_scale;
  auto cell = [](float v) { return (uint32_t)(v > 0.0f ? (v < (float)(kWfCellsPerAxis - 1) ? (int)v : (int)kWfCellsPerAxis - 1) : 0); };  // NaN -> 0
  const uint32_t morton = spread4(cell(c.x)) | (spread4(cell(c.y)) << 1) | (spread4(cell(c.z)) << 2);
  const uint32_t octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  return (morton << 3) | octant;
}
constexpr uint32_t kWfSegments = 8;
constexpr uint32_t kWfCounterStride = 32;  // words: every counter on its own 128-byte line
#ifndef P3D_WF_WAVES
#define P3D_WF_WAVES 6
#endif

template <bool STATS, int LIT>
__global__ void __launch_bounds__(kBlock, P3D_WF_WAVES) wf_level_kernel(const RenderParams P) {
  constexpr int ACCEL = P3D_ACCEL_BVH;
  constexpr bool SPILL = true, AA = false;
  extern __shared__ float4 smem[];
  const DevScene& sc = P.sc;
  const Handoff& H = P.hand;
  const uint32_t lane = threadIdx.x, level = P.wf_level;
  const uint32_t gid = blockIdx.x * kBlock + lane;
  Counters<STATS> ct;
  if (STATS) reinterpret_cast<Counters<true>&>(ct).clear();
  Stack st;
  stack_bind(st, smem, 0, lane, P.stack_cap, P.spill, P.level_stride, gid);
  const uint32_t seg = blockIdx.x & (kWfSegments - 1);
  uint32_t tx = 0, ty = 0;
  const bool halo_block = level == 0 && LIT == 1 && blockIdx.x >= P.tile_blocks;
  if (level == 0 && !halo_block && !tile_of_block(P, tx, ty)) return;
  // the sorted queue of this level: XCD label `seg` takes the seg-th eighth of its chunks
  const uint32_t n_in = level == 0 ? 0u : *P.wf_total;
  const float4* ray_in = P.wf_sorted;
  const uint32_t chunks = (n_in + kBlock - 1) / kBlock, per_xcd = (chunks + kWfSegments - 1) / kWfSegments;
  const uint32_t chunk_end = (seg + 1) * per_xcd < chunks ? (seg + 1) * per_xcd : chunks;

  for (uint32_t chunk = seg * per_xcd + blockIdx.x / kWfSegments;; chunk += gridDim.x / kWfSegments) {
    bool active;
    UnitPlace up;
    uint32_t unit = 0;
    RayS ray;
    float ior_1 = 1.0f;
    bool inside = false;
    if (level == 0) {
      if (halo_block) {
        active = halo_unit_of_lane(P, lane, unit);
        up.c = up.r = up.x = up.y = 0; up.halo = true; up.valid = active;
        if (active) up = place_of_unit(P, unit);
        active = up.valid;
      } else {
        up.c = (int)(tx * 8 + (lane % 8));
        up.r = (int)(ty * 8 + (lane / 8));
        up.halo = false;
        active = up.c < P.w && up.r < P.h;
        up.valid = active;
        up.x = P.x0 + up.c;
        up.y = image_row(P, up.r);
        unit = (uint32_t)(P.row0 + up.r) * H.row_units + H.halo + (uint32_t)up.c;
      }
      if (active) {
        F3 o, d;
        primary_ray(sc.cam, (float)(up.x + 0.5), (float)(up.y + 0.5), o, d);  // main.cpp:805-808
        ray_set(ray, o, d);
        ct.add(kRaysPrimary);
        if (!up.halo) ct.add(kPixels);
      }
      stack_clear(st);
    } else {
      if (chunk >= chunk_end) break;
      const uint32_t i = chunk * kBlock + lane;
      active = i < n_in;
      up.c = up.r = up.x = up.y = 0; up.halo = false; up.valid = active;
      if (active) {
        const float4 q0 = ray_in[2 * (size_t)i], q1 = ray_in[2 * (size_t)i + 1];
        unit = __float_as_uint(q0.w);
        ray_set(ray, f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z));
        ior_1 = fabsf(q1.w);
        inside = q1.w < 0.0f;
        seed_stack<SPILL>(st, H, unit, H.meta[unit] & 0xffffu, ct);  // slot 0: what the level above left (Q2)
      }
    }
    if (active) {
      const int si = 0, sj = 0, SPP = 1;
      int depth = P.max_depth - (int)level;
      const bool ghost = false;
      constexpr bool GHOST = false;  // (the per-level launches are not used for scenes with zero-weight reflection rays)
      int n_deferred = 0, first_hit = -1;
      Rng rng;
      rng.state = 0; rng.inc = 1;
      bool unit_touched = false;
      uint32_t unit_ch0[kCh0Counters] = {0, 0, 0, 0, 0};
      F3 chain_result = f3(0, 0, 0);
      constexpr bool COLD = false, VOTE = true;
      ColdState<COLD> cold;
#ifdef P3D_PT_PROFILE
      RegionProf prof; prof.init();  // (instrumented build: the level body marks its regions; only the megakernel reports them)
#endif
#define P3D_FIRST_HIT(obj) first_hit = (obj)
#include "whitted_level.inc"
#undef P3D_FIRST_HIT
      (void)n_deferred;
      if (STATS && LIT == 1) {  // counters per unit, accumulated level by level (handoff.hpp: ucount)
        if (up.halo) ct.clear();
        if (level == 0) {
          store_unit_counters<STATS>(H, unit, ct, unit_ch0, P.stats);
        } else {
          for (int s = 0; s < kNumStats; ++s)
            if (s != kMaxStack) H.ucount[(size_t)s * H.n_units + unit] += ct.get(s);
          atomicMax(&P.stats[kMaxStack], (unsigned long long)ct.get(kMaxStack));
        }
        ct.clear();
      }
      if (level == 0 && !up.halo && P.hit_id) P.hit_id[(size_t)up.r * P.w + up.c] = first_hit;
      if (chain_ended) {
        P.wf_final[unit] = make_float4(chain_result.x, chain_result.y, chain_result.z, __uint_as_float(level));
      } else {
        P.levels[(size_t)level * P.level_stride2 + unit] = make_float4(col.x, col.y, col.z, weight);
        const uint32_t at = atomicAdd(&P.wf_n_out[seg * kWfCounterStride], 1u);
        if (at < P.wf_seg_cap) {
          float4* q = P.wf_ray_out + ((size_t)seg * P.wf_seg_cap + at) * 2;
          q[0] = make_float4(child_o.x, child_o.y, child_o.z, __uint_as_float(unit));
          q[1] = make_float4(child_d.x, child_d.y, child_d.z, child_inside ? -child_ior : child_ior);
          const uint32_t bin = ray_bin(P, child_o, child_d);
          P.wf_key_out[(size_t)seg * P.wf_seg_cap + at] = bin;
          atomicAdd(&P.wf_hist[bin], 1u);
        } else {
          atomicOr(P.status, kHoErrList);
        }
      }
      // the stack as this level leaves it: read by the next level (Q2) and, after the last one, by the next pixel (LIT)
      if (!chain_ended || LIT == 1) {
        uint32_t n = (uint32_t)st.sp;
        if (n > H.cap) { atomicOr(P.status, kHoErrLeftoverCap); n = H.cap; }
        for (uint32_t e = 0; e < n; ++e) H.entries[(size_t)unit * H.cap + e] = stack_read<SPILL>(st, (int)e);  // (dense records: slot 0 of the unit)
        if (level == 0) {
          uint32_t meta = n;
          if (LIT == 1 && unit_touched) {
            meta |= kMetaTouched;
            atomicOr(&H.touched[unit >> 5], 1u << (unit & 31u));
          }
          H.meta[unit] = meta;
        } else {
          H.meta[unit] = (H.meta[unit] & ~0xffffu) | n;
        }
      } else if (level == 0) {
        H.meta[unit] = 0;
      }
    }
    if (level == 0) break;
  }
  if (STATS && LIT == 0) flush_stats<STATS>(ct, P.stats);
}

// counts per bin -> first slot of each bin (exclusive prefix sum, in place); hist[kWfBins] = number of rays
__global__ void __launch_bounds__(1024) wf_scan_kernel(uint32_t* hist) {
  __shared__ uint32_t part[1024];
  constexpr uint32_t kPer = kWfBins / 1024;
  const uint32_t t = threadIdx.x;
  uint32_t local[kPer];
  uint32_t sum = 0;
  for (uint32_t k = 0; k < kPer; ++k) { local[k] = hist[t * kPer + k]; sum += local[k]; }
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {  // Hillis-Steele over the 1024 partial sums
    const uint32_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t at = part[t] - sum;
  for (uint32_t k = 0; k < kPer; ++k) { hist[t * kPer + k] = at; at += local[k]; }
  if (t == 1023) hist[kWfBins] = part[1023];
}

// every staged ray to the next free slot of its bin
__global__ void __launch_bounds__(256) wf_scatter_kernel(const RenderParams P, const uint32_t* total) {
  const uint32_t seg = blockIdx.x & (kWfSegments - 1);
  const uint32_t n_raw = P.wf_n_in[seg * kWfCounterStride];
  const uint32_t n = n_raw > P.wf_seg_cap ? P.wf_seg_cap : n_raw;
  const uint32_t limit = *total;  // (= the sum of the segment counts unless a segment overflowed)
  const float4* in = P.wf_ray_in + (size_t)seg * P.wf_seg_cap * 2;
  const uint32_t* key = P.wf_key_in + (size_t)seg * P.wf_seg_cap;
  for (uint32_t i = (blockIdx.x / kWfSegments) * 256 + threadIdx.x; i < n; i += (gridDim.x / kWfSegments) * 256) {
    const uint32_t at = atomicAdd(&P.wf_hist[key[i]], 1u);
    if (at < limit) {
      P.wf_sorted[2 * (size_t)at] = in[2 * (size_t)i];
      P.wf_sorted[2 * (size_t)at + 1] = in[2 * (size_t)i + 1];
    }
  }
}

// Folds the level records of every pixel bottom-up with the per-level clamp (main.cpp:305-307, Q4) and writes the frame.
__global__ void __launch_bounds__(256) wf_fold_kernel(const RenderParams P) {
  const Handoff& H = P.hand;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint32_t)P.w * (uint32_t)P.h) return;
  const uint32_t r = i / (uint32_t)P.w, c = i - r * (uint32_t)P.w;
  const uint32_t unit = r * H.row_units + H.halo + c;
  const float4 fin = P.wf_final[unit];
  F3 result = f3(fin.x, fin.y, fin.z);
  for (int level = (int)__float_as_uint(fin.w); level > 0;) {
    --level;
    const float4 rec = P.levels[(size_t)level * P.level_stride2 + unit];
    result = clamp01(f3(rec.x, rec.y, rec.z) + result * rec.w);
  }
  if (P.rgb) {
    P.rgb[3 * (size_t)i] = result.x; P.rgb[3 * (size_t)i + 1] = result.y; P.rgb[3 * (size_t)i + 2] = result.z;
  }
  if (P.rgb8) {  // main.cpp:814-820
    F3 gc = result;
    if (P.gamma != 1.0f) {
      const double ig = (double)(1 / P.gamma);
      gc = f3((float)pow_spec((double)result.x, ig), (float)pow_spec((double)result.y, ig), (float)pow_spec((double)result.z, ig));
    }
    P.rgb8[3 * (size_t)i] = u8fromfloat(gc.x); P.rgb8[3 * (size_t)i + 1] = u8fromfloat(gc.y); P.rgb8[3 * (size_t)i + 2] = u8fromfloat(gc.z);
  }
}

}  // namespace p3d
