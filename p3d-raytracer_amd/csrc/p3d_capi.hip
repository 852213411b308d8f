// p3d_capi.hip — device half of include/p3d.h: scene upload, kernel dispatch, batched
// queries.  There is no CPU fallback anywhere in this file: every entry point needs a
// HIP device and fails with P3D_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../host/p3d_error.hpp"
#include "kernels.hpp"
#include "lbvh.hpp"
#include "p3d.h"
#include "p3d_debug.h"
#include "pt_kernel.hpp"

using namespace p3d;

namespace {

#define P3D_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(P3D_ERR_NO_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));     \
  } while (0)

#ifndef P3D_PT_SUB4_MIN_SPP_SQRT
#define P3D_PT_SUB4_MIN_SPP_SQRT 4
#endif
#ifndef P3D_WHITTED_SUB4_MIN_SPP_SQRT
#define P3D_WHITTED_SUB4_MIN_SPP_SQRT 2
#endif
constexpr uint32_t kWhittedSub4MinSppSqrt = P3D_WHITTED_SUB4_MIN_SPP_SQRT;
constexpr uint32_t kPtSub4MinSppSqrt = P3D_PT_SUB4_MIN_SPP_SQRT;  // from 16 samples per pixel: 4 lanes per pixel
constexpr uint32_t kLdsSceneLimitBytesPt = 16 * 1024;  // same for the path tracer (not re-tuned: its packaged scenes are 1-2 KB)
constexpr uint32_t kLdsSceneLimitBytes = 26 * 1024;  // stage the scene in LDS up to this size
static_assert(kLdsSceneLimitBytes / 32 < 4096 && kLdsSceneLimitBytesPt / 32 < 4096, "the 6-byte stack entries of LDS-staged scenes keep 12 index bits (device_core.hpp Stack)");
// A frame is rendered by as few launches as the per-thread scratch (level records + stack spill)
// allows: every launch ends with a tail of partly idle CUs (2048x2048, 100k triangles: 30.3 ms in
// two launches, 28.1 ms in one).
constexpr size_t kLaunchScratchBudget = (size_t)4 << 30;
constexpr uint32_t kMaxLaunchThreads = 1u << 24;

inline F3 to_f3(const float v[3]) { return F3{v[0], v[1], v[2]}; }

struct Scratch {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return P3D_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    P3D_HIP(hipMalloc(&p, need));
    bytes = need;
    return P3D_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

// One memoised tile schedule.  The cost of a tile is a function of the scene (camera included),
// the chain depth, the back end, the sampling and the pixel rectangle: the first launch with a
// given key runs in frame order and records what every tile cost, the launches after it take
// the tiles most-expensive-class first.
struct SchedEntry {
  uint32_t accel = 0, aa = 0, spp = 0, pt = 0, tiles_x = 0, tiles_y = 0;  // (tiles: 8x8 or 4x4 pixels, by kernel variant)
  int32_t max_depth = 0, x0 = 0, y0 = 0, w = 0, h = 0, stripe_h = 0, stripe_stride = 0;
  Scratch cost, sched;
  hipEvent_t ready = nullptr;
  hipStream_t built_on = nullptr;
  uint64_t last_use = 0;
  bool built = false;  // the recording launch and sched_build_kernel were enqueued: `sched` may be used
  bool same_key(const SchedEntry& o) const {
    return accel == o.accel && aa == o.aa && spp == o.spp && pt == o.pt && tiles_x == o.tiles_x && tiles_y == o.tiles_y && max_depth == o.max_depth && x0 == o.x0 && y0 == o.y0 &&
           w == o.w && h == o.h && stripe_h == o.stripe_h && stripe_stride == o.stripe_stride;
  }
};
constexpr uint32_t kWfHistWords = kWfBins + 32;  // bin counts of one level + the total, padded to a 128-byte multiple
#ifndef P3D_REDO_LANES
#define P3D_REDO_LANES 64
#endif
// List entries per wave of the first work-list launch (the units rendered again: cfg2 9 995 unrelated deep pixels).
// Fewer entries per wave make that launch shorter when the frame is alone on the chip (round 2, profiles/r02/experiments/
// README.md §7: 4 per wave 130 µs, 32 per wave 160 µs), but every one of those waves holds a slot and issues for ~100 µs
// whatever the number of its active lanes, and with other frames in flight that is what counts.  Round 3, final kernels,
// one box, cfg2 with four frames in flight (1 000-step / 20-step loops) and one frame alone:
//    8 per wave  24.7 k / 23.8 k Mrays/s, 0.333 ms      32 per wave  36.9 k / 31.6 k, 0.337 ms
//   16 per wave  32.0 k / 29.5 k,         0.332 ms      64 per wave  38.9 k / 34.0 k, 0.348 ms
// P3D_REDO_LANES in the environment overrides it.
inline uint32_t redo_lanes() {
  static const uint32_t v = [] {
    const char* e = getenv("P3D_REDO_LANES");
    const int n = e ? atoi(e) : P3D_REDO_LANES;
    return (uint32_t)(n >= 1 && n <= 64 ? n : P3D_REDO_LANES);
  }();
  return v;
}
inline uint32_t round1_lanes() {  // list entries per wave of the round-1 launch (successors to re-check; P3D_ROUND1_LANES overrides)
  static const uint32_t v = [] {
    const char* e = getenv("P3D_ROUND1_LANES");
    const int n = e ? atoi(e) : 64;
    return (uint32_t)(n >= 1 && n <= 64 ? n : 64);
  }();
  return v;
}
inline uint32_t list_blocks() {  // workgroups of the round-1 work-list launch (P3D_LIST_BLOCKS overrides)
  static const uint32_t v = [] {
    const char* e = getenv("P3D_LIST_BLOCKS");
    const int n = e ? atoi(e) : 256;
    return (uint32_t)(n >= 1 ? n : 256);
  }();
  return v;
}
#ifdef P3D_ABLATION  // timing experiments only (profiles/r04/experiments): stages of the literal frame left out, frames WRONG
inline uint32_t abl_skip() {  // P3D_ABL_SKIP: 1 = no check launch, 2 = no redo launch (round 0), 4 = no round 1 launch, 8 = round 1 launched on an empty list
  static const uint32_t v = [] { const char* e = getenv("P3D_ABL_SKIP"); return e ? (uint32_t)atoi(e) : 0u; }();
  return v;
}
#else
constexpr uint32_t abl_skip() { return 0; }
#endif
constexpr uint32_t kPoolEntriesPerUnit = 8;  // compact hand-off records: pool entries per unit of the tile

// Tile of a wave of the one-lane-per-pixel Whitted kernels over a scene traversed from L2 (see p3d_render_tile_device).
// P3D_TILE_SHAPE = 88 | 84 | 44 in the environment overrides the rule (experiments).
void tile_shape(uint64_t pixels, uint32_t& w, uint32_t& h) {
  static const int forced = [] { const char* e = getenv("P3D_TILE_SHAPE"); return e ? atoi(e) : 0; }();
  int shape = forced;
  if (shape != 88 && shape != 84 && shape != 44) shape = 88;
  w = shape == 44 ? 4 : 8;
  h = shape == 88 ? 8 : 4;
  (void)pixels;
}
constexpr size_t kSchedCacheEntries = 16;
constexpr uint32_t kSchedMinTiles = 8192;  // LDS-staged scenes: with fewer tiles than ~2 per wave slot nearly all start at once anyway
constexpr uint32_t kSchedMinTilesL2 = 256;  // scenes traversed from L2: a wave lives a millisecond, the order matters from a few hundred tiles

}  // namespace

struct p3d_scene {
  int device = 0;
  std::vector<SchedEntry> sched;
  uint64_t sched_clock = 0;
  float4* d_blob = nullptr;
  uint32_t blob_f4 = 0;
  uint32_t off_nodes = 0, off_bgeom = 0, off_ogeom = 0, off_normals = 0, off_mats = 0, off_lights = 0;
  uint32_t* d_cell_start = nullptr;
  uint32_t* d_cell_items = nullptr;
  uint32_t* d_emitters = nullptr;
  uint32_t* d_sky[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool has_sky = false;
  DevScene dev{};
  bool has_bvh = false, has_grid = false;
  uint32_t bvh_max_depth = 0;
  float device_bvh_ms = 0;  // GPU time of lbvh::build, 0 for an uploaded tree
  Scratch levels, spill, deferred, wf_rays, wf_keys, wf_sorted, wf_final, out_rgb, out_hit, out_rgb8, q_in, q_out;
  // P3D_STACK_LITERAL (csrc/handoff.hpp): leftovers, per-unit records, work lists, counters
  Scratch ho_where, ho_entries, ho_meta, ho_first, ho_first_sample, ho_touched, ho_lists, ho_check, ho_counters, ho_row_chain, ho_halo_pix, ho_ucount;
  std::vector<int64_t> ho_chain_key;     // what the row_chain flags and halo pixels on the device were worked out for
  bool has_spheres = false;              // (halo_find_kernel: only a sphere test re-normalises a ray)
  uint32_t* d_halo_verdict = nullptr;    // kHoErrHalo if the memoised halo search could not start some row exactly
  float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};  // box of BVH node 0 (bins of the per-level ray queue)
  bool zero_weight_reflections = false;  // some material is transmissive AND reflective (main.cpp:282,290-300)
  unsigned long long* d_stats = nullptr;
  uint32_t* d_status = nullptr;          // kHoErr* bits raised by kernels; read and cleared by check_status()
  uint32_t last_status = 0;              // the bits check_status() found last (what the host-buffer call decides its DENSE retry on)
  p3d_debug_limits dbg{0, 0, 0, 0};      // tests only (csrc/p3d_debug.h): shrunken limits of THIS scene, all 0 = the real ones
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr, ev_p1 = nullptr;
  // p3d_scene_set_tail_stream: the dependent launches of a LITERAL frame go to a stream of their own
  hipStream_t tail_stream = nullptr;
  hipEvent_t ev_tail_go = nullptr, ev_tail_done = nullptr;
  bool tail_pending = false;  // ev_tail_done was recorded by the last frame: the next launch on this scene waits for it
};

extern "C" {

int p3d_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(P3D_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

void p3d_scene_destroy(p3d_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->d_blob) (void)hipFree(s->d_blob);
  if (s->d_cell_start) (void)hipFree(s->d_cell_start);
  if (s->d_cell_items) (void)hipFree(s->d_cell_items);
  if (s->d_emitters) (void)hipFree(s->d_emitters);
  if (s->d_stats) (void)hipFree(s->d_stats);
  for (uint32_t*& f : s->d_sky) if (f) (void)hipFree(f);
  for (SchedEntry& e : s->sched) {
    e.cost.release();
    e.sched.release();
    if (e.ready) (void)hipEventDestroy(e.ready);
  }
  s->levels.release(); s->spill.release(); s->deferred.release(); s->wf_rays.release(); s->wf_keys.release(); s->wf_sorted.release(); s->wf_final.release(); s->out_rgb.release(); s->out_hit.release();
  s->ho_where.release(); s->ho_entries.release(); s->ho_meta.release(); s->ho_first.release(); s->ho_first_sample.release(); s->ho_touched.release();
  s->ho_lists.release(); s->ho_check.release(); s->ho_counters.release(); s->ho_row_chain.release(); s->ho_halo_pix.release(); s->ho_ucount.release();
  if (s->d_status) (void)hipFree(s->d_status);
  if (s->d_halo_verdict) (void)hipFree(s->d_halo_verdict);
  s->out_rgb8.release(); s->q_in.release(); s->q_out.release();
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->ev_mid) (void)hipEventDestroy(s->ev_mid);
  if (s->ev_p1) (void)hipEventDestroy(s->ev_p1);
  if (s->ev_tail_go) (void)hipEventDestroy(s->ev_tail_go);
  if (s->ev_tail_done) (void)hipEventDestroy(s->ev_tail_done);
  delete s;
}

// device_bvh: the BVH arrays of the descriptor are ignored and a linear BVH is built on the GPU (lbvh.hpp)
static int create_impl(const p3d_scene_desc* d_in, int device, bool device_bvh, p3d_scene** out) {
  if (!d_in || !out) return fail(P3D_ERR_INVALID, "p3d_scene_create: null argument");
  p3d_scene_desc d_local = *d_in;
  if (device_bvh) {  // sizes of the device-built tree: 2 n - 1 nodes, one leaf slot per object
    d_local.n_bvh_nodes = 0; d_local.bvh_nodes = nullptr;
    d_local.n_bvh_prim_index = 0; d_local.bvh_prim_index = nullptr;
    d_local.bvh_max_depth = 0;
  }
  const p3d_scene_desc* d = &d_local;
  const uint32_t lbvh_nodes = device_bvh && d->n_prims ? 2 * d->n_prims - 1 : 0;
  const uint32_t lbvh_slots = device_bvh ? d->n_prims : 0;
  if (d->abi_version != P3D_ABI_VERSION) return fail(P3D_ERR_INVALID, "p3d_scene_create: ABI version mismatch");
  if ((d->n_prims && !d->prims) || (d->n_materials && !d->materials) || (d->n_lights && !d->lights))
    return fail(P3D_ERR_INVALID, "p3d_scene_create: null array with non-zero count");
  for (uint32_t i = 0; i < d->n_prims; ++i) {
    if (d->prims[i].material >= d->n_materials) return fail(P3D_ERR_INVALID, "p3d_scene_create: material index out of range");
    if (d->prims[i].type > P3D_PRIM_PLANE) return fail(P3D_ERR_INVALID, "p3d_scene_create: unknown primitive type");
  }
  if (d->n_bvh_nodes) {  // every index a lane may follow is checked here, not in the kernel
    if (!d->bvh_nodes || (d->n_bvh_prim_index && !d->bvh_prim_index) || d->n_bvh_prim_index != d->n_prims)
      return fail(P3D_ERR_INVALID, "p3d_scene_create: inconsistent BVH arrays");
    std::vector<uint8_t> has_parent(d->n_bvh_nodes, 0);  // a TREE: every record is the child of at most one inner node
    for (uint32_t i = 0; i < d->n_bvh_nodes; ++i) {
      const p3d_bvh_node& n = d->bvh_nodes[i];
      if (n.index > 0x0fffffffu) return fail(P3D_ERR_CAPACITY, "p3d_scene_create: BVH index exceeds 2^28");
      if (n.count_leaf & P3D_BVH_LEAF) {
        const uint64_t cnt = n.count_leaf & ~P3D_BVH_LEAF;
        if (cnt > 7) return fail(P3D_ERR_CAPACITY, "p3d_scene_create: BVH leaf with more than 7 objects (the reference's Threshold is 2)");
        if ((uint64_t)n.index + cnt > d->n_bvh_prim_index) return fail(P3D_ERR_INVALID, "p3d_scene_create: BVH leaf range out of bounds");
      } else if ((uint64_t)n.index + 1 >= d->n_bvh_nodes || n.index <= i) {
        return fail(P3D_ERR_INVALID, "p3d_scene_create: BVH child index out of bounds");
      } else {
        // (children lie behind their parent, so index 0 is nobody's child; a record with two parents - shared or
        // overlapping child pairs - would make the relabelling walk below, and every traversal, visit a DAG)
        if (has_parent[n.index] || has_parent[n.index + 1]) return fail(P3D_ERR_INVALID, "p3d_scene_create: BVH record with more than one parent (not a tree)");
        has_parent[n.index] = has_parent[n.index + 1] = 1;
      }
    }
    for (uint32_t i = 0; i < d->n_bvh_prim_index; ++i)
      if (d->bvh_prim_index[i] >= d->n_prims) return fail(P3D_ERR_INVALID, "p3d_scene_create: BVH object index out of bounds");
    if (d->bvh_max_depth == 0 || d->bvh_max_depth > 4096) return fail(P3D_ERR_INVALID, "p3d_scene_create: bad bvh_max_depth");
  }
  if (d->has_grid) {
    const p3d_grid_desc& g = d->grid;
    if (g.nx <= 0 || g.ny <= 0 || g.nz <= 0 || (uint64_t)g.nx * g.ny * g.nz != g.n_cells || !g.cell_start ||
        (g.n_items && !g.cell_items) || g.cell_start[g.n_cells] != g.n_items)
      return fail(P3D_ERR_INVALID, "p3d_scene_create: inconsistent grid arrays");
    for (uint32_t c = 0; c < g.n_cells; ++c)
      if (g.cell_start[c] > g.cell_start[c + 1]) return fail(P3D_ERR_INVALID, "p3d_scene_create: grid cell_start not monotone");
    for (uint32_t i = 0; i < g.n_items; ++i)
      if (g.cell_items[i] >= d->n_prims) return fail(P3D_ERR_INVALID, "p3d_scene_create: grid object index out of bounds");
  }
  int ndev = 0;
  P3D_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(P3D_ERR_NO_DEVICE, "p3d_scene_create: no such HIP device");
  P3D_HIP(hipSetDevice(device));

  // ---- build the float4 blob: nodes | bgeom | ogeom | normals | mats | lights ----
  auto geom_of = [&](uint32_t obj, float4 dst[3]) {
    const p3d_prim& p = d->prims[obj];
    dst[0] = make_float4(p.v[0], p.v[1], p.v[2], p.v[3]);
    dst[1] = make_float4(p.v[4], p.v[5], p.v[6], p.v[7]);
    const uint32_t tm = p.type | (p.material << 8);
    float tmf, objf;
    std::memcpy(&tmf, &tm, 4);
    std::memcpy(&objf, &obj, 4);
    dst[2] = make_float4(p.v[8], tmf, objf, 0.f);
  };
  std::vector<float4> blob;
  // every early return below frees what was allocated so far (device memory, events)
  auto s = std::unique_ptr<p3d_scene, void (*)(p3d_scene*)>(new p3d_scene(), &p3d_scene_destroy);
  s->device = device;
  // ---- node records: relabelled, never reordered as far as a ray can tell ----
  // A traversal only ever follows descriptors, so where a record lies is free; what the reference's order fixes is
  // which child is visited first, and that is untouched.  Layout: one 32-byte pad, the root, then the CHILD PAIRS (64 B,
  // 64-byte aligned: one visit = one half line; in the reference's numbering a pair starts at an odd node index,
  // i.e. it straddled two 64-byte sectors, every second one two 128-byte lines).  Pairs are laid out two to a 128-byte
  // line as "dominoes": a pair and the child pair of its bigger (by box area: likelier) child share a line, so that
  // about every second step down the tree stays in the line it is in; the root shares its line with its own child pair.
  // Pairs without a child pair (both children leaves: half of all pairs) follow behind, two to a line.
  blob.push_back(make_float4(0, 0, 0, 0));
  blob.push_back(make_float4(0, 0, 0, 0));
  s->off_nodes = (uint32_t)blob.size();
  uint32_t n_nodes_out = 0;
  bool odd_boxes = false;
  if (d->n_bvh_nodes) {
    const p3d_bvh_node* N = d->bvh_nodes;
    auto inner = [&](uint32_t i) { return !(N[i].count_leaf & P3D_BVH_LEAF); };
    auto area = [&](uint32_t i) {
      const double x = (double)N[i].bmax[0] - N[i].bmin[0], y = (double)N[i].bmax[1] - N[i].bmin[1], z = (double)N[i].bmax[2] - N[i].bmin[2];
      const double a = x * y + y * z + z * x;
      return a == a ? a : 0.0;
    };
    std::vector<uint32_t> slot_of(d->n_bvh_nodes, 0xffffffffu);  // inner node -> slot of its child pair
    std::vector<uint32_t> singles, heads;
    uint32_t next_slot = 0;
    if (inner(0)) {
      slot_of[0] = next_slot++;
      for (int k = 1; k >= 0; --k) if (inner(N[0].index + k)) heads.push_back(N[0].index + k);  // left subtree first
    }
    while (!heads.empty()) {
      const uint32_t h = heads.back();
      heads.pop_back();
      const uint32_t l = N[h].index, r = l + 1;
      const bool li = inner(l), ri = inner(r);
      if (!li && !ri) { singles.push_back(h); continue; }
      const uint32_t second = (li && ri) ? (area(r) > area(l) ? r : l) : (li ? l : r);
      slot_of[h] = next_slot++;       // an odd slot: the first half of a line
      slot_of[second] = next_slot++;  // ... and the pair most rays take next in its second half
      for (int k = 1; k >= 0; --k) if (inner(N[second].index + k)) heads.push_back(N[second].index + k);
      const uint32_t other = second == l ? r : l;
      if (inner(other)) heads.push_back(other);
    }
    for (uint32_t h : singles) slot_of[h] = next_slot++;
    n_nodes_out = 1 + 2 * next_slot;
    if (n_nodes_out > 0x0fffffffu) return fail(P3D_ERR_CAPACITY, "p3d_scene_create: BVH index exceeds 2^28");
    blob.resize(blob.size() + (size_t)2 * n_nodes_out, make_float4(0, 0, 0, 0));
    auto put = [&](uint32_t at, uint32_t old) {
      const p3d_bvh_node& n = N[old];
      const uint32_t desc = inner(old) ? 1u + 2u * slot_of[old] : (kDescLeaf | ((n.count_leaf & 7u) << 28) | n.index);
      float descf;
      std::memcpy(&descf, &desc, 4);
      blob[s->off_nodes + 2 * (size_t)at] = make_float4(n.bmin[0], n.bmin[1], n.bmin[2], descf);
      blob[s->off_nodes + 2 * (size_t)at + 1] = make_float4(n.bmax[0], n.bmax[1], n.bmax[2], 0.f);
      for (int k = 0; k < 3; ++k)  // the slab fast paths assume finite boxes with min <= max (device_core.hpp); only uploaded records matter
        if (!(std::fabs(n.bmin[k]) < INFINITY) || !(std::fabs(n.bmax[k]) < INFINITY) || !(n.bmin[k] <= n.bmax[k])) odd_boxes = true;
    };
    put(0, 0);
    for (uint32_t i = 0; i < d->n_bvh_nodes; ++i)
      if (slot_of[i] != 0xffffffffu) {  // (records no descriptor leads to are not uploaded)
        put(1 + 2 * slot_of[i], N[i].index);
        put(2 + 2 * slot_of[i], N[i].index + 1);
      }
  }
  blob.resize(blob.size() + (size_t)2 * lbvh_nodes, make_float4(0, 0, 0, 0));  // filled in by lbvh::build
  s->off_bgeom = (uint32_t)blob.size();
  blob.resize(blob.size() + (size_t)3 * lbvh_slots, make_float4(0, 0, 0, 0));
  for (uint32_t i = 0; i < d->n_bvh_prim_index; ++i) {
    float4 g[3];
    geom_of(d->bvh_prim_index[i], g);
    blob.insert(blob.end(), g, g + 3);
  }
  s->off_normals = (uint32_t)blob.size();
  for (uint32_t i = 0; i < d->n_prims; ++i) blob.push_back(make_float4(d->prims[i].n[0], d->prims[i].n[1], d->prims[i].n[2], 0.f));
  s->off_mats = (uint32_t)blob.size();
  auto plain = [](float c) { return c >= 0.0f && c <= 1e15f; };
  bool lights_plain = true;
  for (uint32_t i = 0; i < d->n_lights; ++i)
    lights_plain = lights_plain && plain(d->lights[i].color[0]) && plain(d->lights[i].color[1]) && plain(d->lights[i].color[2]);
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    const p3d_material& m = d->materials[i];
    blob.push_back(make_float4(m.diff_color[0], m.diff_color[1], m.diff_color[2], m.diffuse));
    blob.push_back(make_float4(m.spec_color[0], m.spec_color[1], m.spec_color[2], m.specular));
    blob.push_back(make_float4(m.shine, m.transmittance, m.refr_index, m.reflection));
    // .w: the material's specular term is provably multiplied by an exact zero - Ks == 0 - and provably finite and
    // non-negative whatever the geometry (0 <= shine < inf, specular colour and every light colour in [0, 1e15]): the kernels
    // then leave the pow(H.N, shine) of main.cpp:224 out for Blinn cosines <= 1 (whitted_level.inc), same bits
    blob.push_back(make_float4(m.emission[0], m.emission[1], m.emission[2], (lights_plain && m.specular == 0.0f && m.shine >= 0.0f && m.shine < INFINITY &&
                                                                              plain(m.spec_color[0]) && plain(m.spec_color[1]) && plain(m.spec_color[2])) ? 1.0f : 0.0f));
  }
  s->off_lights = (uint32_t)blob.size();
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    const p3d_light& l = d->lights[i];
    blob.push_back(make_float4(l.position[0], l.position[1], l.position[2], 0.f));
    blob.push_back(make_float4(l.color[0], l.color[1], l.color[2], 0.f));
  }
  // object-order geometry last: the kernels that walk the BVH read the BVH-ordered copy only, and an LDS-staged launch of
  // theirs leaves this array out (stage range, p3d_render_tile_device)
  s->off_ogeom = (uint32_t)blob.size();
  for (uint32_t i = 0; i < d->n_prims; ++i) {
    float4 g[3];
    geom_of(i, g);
    blob.insert(blob.end(), g, g + 3);
  }
  if (blob.empty()) blob.push_back(make_float4(0, 0, 0, 0));
  s->blob_f4 = (uint32_t)blob.size();
  P3D_HIP(hipMalloc((void**)&s->d_blob, blob.size() * sizeof(float4)));
  P3D_HIP(hipMemcpy(s->d_blob, blob.data(), blob.size() * sizeof(float4), hipMemcpyHostToDevice));

  // emissive spheres in object order: the light loop of Radiance (main.cpp:407-415)
  std::vector<uint32_t> emitters;
  for (uint32_t i = 0; i < d->n_prims; ++i) {
    const p3d_material& m = d->materials[d->prims[i].material];
    if (m.emission[0] + m.emission[1] + m.emission[2] > 0 && d->prims[i].type == P3D_PRIM_SPHERE) emitters.push_back(i);
  }
  if (!emitters.empty()) {
    P3D_HIP(hipMalloc((void**)&s->d_emitters, emitters.size() * 4));
    P3D_HIP(hipMemcpy(s->d_emitters, emitters.data(), emitters.size() * 4, hipMemcpyHostToDevice));
  }
  DevScene& v = s->dev;
  v.nodes = s->d_blob + s->off_nodes;
  v.bgeom = s->d_blob + s->off_bgeom;
  v.ogeom = s->d_blob + s->off_ogeom;
  v.normals = s->d_blob + s->off_normals;
  v.mats = s->d_blob + s->off_mats;
  v.lights = s->d_blob + s->off_lights;
  v.emitters = s->d_emitters;
  v.n_nodes = n_nodes_out;
  v.odd_boxes = odd_boxes ? 1u : 0u;
  v.n_slots = d->n_bvh_prim_index;
  v.n_objs = d->n_prims;
  v.n_mats = d->n_materials;
  v.n_lights = d->n_lights;
  v.n_emitters = (uint32_t)emitters.size();
  const p3d_camera& c = d->camera;
  v.cam.eye = to_f3(c.eye); v.cam.u = to_f3(c.u); v.cam.v = to_f3(c.v); v.cam.n = to_f3(c.n);
  v.cam.w = c.w; v.cam.h = c.h; v.cam.plane_dist = c.plane_dist; v.cam.focal_ratio = c.focal_ratio;
  v.cam.aperture = c.aperture; v.cam.res_x = c.res_x; v.cam.res_y = c.res_y;
  v.bg = to_f3(d->background);
  s->has_bvh = d->n_bvh_nodes > 0;
  if (device_bvh && d->n_prims) {
    if (d->n_prims > 0x07ffffffu) return fail(P3D_ERR_CAPACITY, "p3d_scene_create_device_bvh: too many objects");
    std::vector<float4> boxes((size_t)2 * d->n_prims);
    for (uint32_t i = 0; i < d->n_prims; ++i)
      for (int k = 0; k < 3; ++k)
        if (!(std::fabs(d->prims[i].bmin[k]) < INFINITY) || !(std::fabs(d->prims[i].bmax[k]) < INFINITY) || !(d->prims[i].bmin[k] <= d->prims[i].bmax[k]))
          v.odd_boxes = 1u;  // (the built tree's boxes are unions of these)
    for (uint32_t i = 0; i < d->n_prims; ++i) {
      boxes[2 * i] = make_float4(d->prims[i].bmin[0], d->prims[i].bmin[1], d->prims[i].bmin[2], 0.f);
      boxes[2 * i + 1] = make_float4(d->prims[i].bmax[0], d->prims[i].bmax[1], d->prims[i].bmax[2], 0.f);
    }
    float4* d_boxes = nullptr;
    P3D_HIP(hipMalloc((void**)&d_boxes, boxes.size() * sizeof(float4)));
    hipError_t e = hipMemcpy(d_boxes, boxes.data(), boxes.size() * sizeof(float4), hipMemcpyHostToDevice);
    lbvh::Result built;
    if (e == hipSuccess)
      e = lbvh::build(d_boxes, s->d_blob + s->off_ogeom, d->n_prims, s->d_blob + s->off_nodes, s->d_blob + s->off_bgeom, &built);
    (void)hipFree(d_boxes);
    if (e != hipSuccess) return fail(P3D_ERR_NO_DEVICE, std::string("device BVH build: ") + hipGetErrorString(e));
    v.n_nodes = built.n_nodes;
    v.n_slots = d->n_prims;
    s->has_bvh = true;
    s->bvh_max_depth = built.max_depth;
    s->device_bvh_ms = built.build_ms;
  }
  // The node-stack capacity (LDS + spill) is derived from the tree depth: never trust the caller's
  // number below what the node array really contains (child indices were validated above:
  // children lie behind their parent, so this walk terminates).
  uint32_t real_depth = d->n_bvh_nodes ? 1 : 0;
  if (d->n_bvh_nodes) {
    std::vector<uint32_t> level(d->n_bvh_nodes, 0);
    level[0] = 1;
    for (uint32_t i = 0; i < d->n_bvh_nodes; ++i) {
      const p3d_bvh_node& n = d->bvh_nodes[i];
      if (level[i] == 0) continue;  // unreachable record
      real_depth = std::max(real_depth, level[i]);
      if (!(n.count_leaf & P3D_BVH_LEAF)) level[n.index] = level[n.index + 1] = level[i] + 1;
    }
  }
  if (!device_bvh) s->bvh_max_depth = std::max(d->bvh_max_depth, real_depth);
  if (s->has_bvh) {  // root box, from the node array as uploaded or built
    float4 root[2];
    P3D_HIP(hipMemcpy(root, s->d_blob + s->off_nodes, sizeof(root), hipMemcpyDeviceToHost));
    s->root_min[0] = root[0].x; s->root_min[1] = root[0].y; s->root_min[2] = root[0].z;
    s->root_max[0] = root[1].x; s->root_max[1] = root[1].y; s->root_max[2] = root[1].z;
  }
  if (d->has_grid) {
    const p3d_grid_desc& g = d->grid;
    P3D_HIP(hipMalloc((void**)&s->d_cell_start, (size_t)(g.n_cells + 1) * 4));
    P3D_HIP(hipMemcpy(s->d_cell_start, g.cell_start, (size_t)(g.n_cells + 1) * 4, hipMemcpyHostToDevice));
    P3D_HIP(hipMalloc((void**)&s->d_cell_items, (size_t)std::max<uint32_t>(g.n_items, 1) * 4));
    if (g.n_items) P3D_HIP(hipMemcpy(s->d_cell_items, g.cell_items, (size_t)g.n_items * 4, hipMemcpyHostToDevice));
    v.grid.bmin = to_f3(g.bmin); v.grid.bmax = to_f3(g.bmax);
    v.grid.nx = g.nx; v.grid.ny = g.ny; v.grid.nz = g.nz;
    v.grid.cell_start = s->d_cell_start; v.grid.cell_items = s->d_cell_items;
    s->has_grid = true;
  }
  P3D_HIP(hipMalloc((void**)&s->d_stats, kNumStats * sizeof(unsigned long long)));
  P3D_HIP(hipMalloc((void**)&s->d_status, sizeof(uint32_t)));
  P3D_HIP(hipMemset(s->d_status, 0, sizeof(uint32_t)));
  P3D_HIP(hipMalloc((void**)&s->d_halo_verdict, sizeof(uint32_t)));
  P3D_HIP(hipMemset(s->d_halo_verdict, 0, sizeof(uint32_t)));
  for (uint32_t i = 0; i < d->n_prims; ++i) s->has_spheres = s->has_spheres || d->prims[i].type == P3D_PRIM_SPHERE;
  for (uint32_t i = 0; i < d->n_materials; ++i)
    if (d->materials[i].transmittance != 0 && d->materials[i].reflection > 0) s->zero_weight_reflections = true;
  P3D_HIP(hipEventCreate(&s->ev0));
  P3D_HIP(hipEventCreate(&s->ev1));
  P3D_HIP(hipEventCreate(&s->ev_mid));
  P3D_HIP(hipEventCreate(&s->ev_p1));
  P3D_HIP(hipEventCreateWithFlags(&s->ev_tail_go, hipEventDisableTiming));
  P3D_HIP(hipEventCreateWithFlags(&s->ev_tail_done, hipEventDisableTiming));
  *out = s.release();
  return P3D_OK;
}

int p3d_scene_create(const p3d_scene_desc* d, int device, p3d_scene** out) { return create_impl(d, device, false, out); }

int p3d_scene_create_device_bvh(const p3d_scene_desc* d, int device, p3d_scene** out, float* build_ms) {
  const int rc = create_impl(d, device, true, out);
  if (rc == P3D_OK && build_ms) *build_ms = (*out)->device_bvh_ms;
  return rc;
}

}  // extern "C"

extern "C" int p3d_scene_set_skybox(p3d_scene* s, const p3d_skybox_desc* sky) {
  if (!s || !sky) return fail(P3D_ERR_INVALID, "p3d_scene_set_skybox: null argument");
  for (int f = 0; f < 6; ++f) {
    const p3d_skybox_face& a = sky->face[f];
    if (!a.img || a.res_x == 0 || a.res_y == 0 || (a.bpp != 3 && a.bpp != 4) || (uint64_t)a.res_x * a.res_y > (1ull << 28))
      return fail(P3D_ERR_INVALID, "p3d_scene_set_skybox: bad face (need img, res > 0, bpp 3 or 4)");
  }
  P3D_HIP(hipSetDevice(s->device));
  for (int f = 0; f < 6; ++f) {
    const p3d_skybox_face& a = sky->face[f];
    const size_t n = (size_t)a.res_x * a.res_y;
    std::vector<uint32_t> rgba(n);  // one 4-byte texel fetch instead of three byte loads
    for (size_t i = 0; i < n; ++i)
      rgba[i] = (uint32_t)a.img[i * a.bpp] | ((uint32_t)a.img[i * a.bpp + 1] << 8) | ((uint32_t)a.img[i * a.bpp + 2] << 16);
    if (s->d_sky[f]) { (void)hipFree(s->d_sky[f]); s->d_sky[f] = nullptr; }
    P3D_HIP(hipMalloc((void**)&s->d_sky[f], n * 4));
    P3D_HIP(hipMemcpy(s->d_sky[f], rgba.data(), n * 4, hipMemcpyHostToDevice));
    s->dev.sky[f] = s->d_sky[f];
    s->dev.sky_w[f] = a.res_x;
    s->dev.sky_h[f] = a.res_y;
  }
  s->has_sky = true;
  return P3D_OK;
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
namespace {

template <int ACCEL, bool LDS, bool STATS>
hipError_t launch_one(bool pt, bool aa, bool sub4, const RenderParams& P, uint32_t blocks, size_t lds, hipStream_t st) {
  if (pt && sub4) hipLaunchKernelGGL((pt_kernel<ACCEL, LDS, STATS, 4>), dim3(blocks), dim3(kBlock), lds, st, P);
  else if (pt) hipLaunchKernelGGL((pt_kernel<ACCEL, LDS, STATS, 1>), dim3(blocks), dim3(kBlock), lds, st, P);
  // anti-aliased, four or more samples per pixel, scene traversed from L2: four lanes per pixel (4x4-pixel
  // tiles).  Only there: 100k triangles 512x512 2x2 AA 11.17 -> 6.84 ms, but a staged scene pays for one LDS
  // copy per 16 pixels instead of per 64 (balls_medium 3x3 AA 3.48 -> 5.17 ms, balls_low 2x2 0.60 -> 0.65 ms).
  else if (aa && sub4 && !LDS) {
    if constexpr (!LDS) hipLaunchKernelGGL((whitted_kernel<ACCEL, false, STATS, true, true, 4>), dim3(blocks), dim3(kBlock), lds, st, P);
  }
  // LDS-staged scene whose worst-case stack does not fit LDS: staged scene + spilling stack
  else if (LDS && P.stack_spills && aa) hipLaunchKernelGGL((whitted_kernel<ACCEL, LDS, STATS, true, true>), dim3(blocks), dim3(kBlock), lds, st, P);
  else if (LDS && P.stack_spills) hipLaunchKernelGGL((whitted_kernel<ACCEL, LDS, STATS, false, true>), dim3(blocks), dim3(kBlock), lds, st, P);
  else if (aa) hipLaunchKernelGGL((whitted_kernel<ACCEL, LDS, STATS, true>), dim3(blocks), dim3(kBlock), lds, st, P);
  else hipLaunchKernelGGL((whitted_kernel<ACCEL, LDS, STATS, false>), dim3(blocks), dim3(kBlock), lds, st, P);
  return hipGetLastError();
}
template <int ACCEL>
hipError_t launch_accel(bool pt, bool aa, bool sub4, bool lds_scene, bool stats, const RenderParams& P, uint32_t blocks, size_t lds, hipStream_t st) {
  if (lds_scene) return stats ? launch_one<ACCEL, true, true>(pt, aa, sub4, P, blocks, lds, st) : launch_one<ACCEL, true, false>(pt, aa, sub4, P, blocks, lds, st);
  return stats ? launch_one<ACCEL, false, true>(pt, aa, sub4, P, blocks, lds, st) : launch_one<ACCEL, false, false>(pt, aa, sub4, P, blocks, lds, st);
}

// P3D_STACK_LITERAL launches (BVH only).  lit 1: pass 1; lit 2: work-list launch; lit 0: the check launch over the tiles; lit 3:
// the check launch over pass 1's list; lit 4: check + repair over the tiles in one launch (whitted_kernel LIT = 3).
// ghosts: the scene has zero-weight reflection rays to trace (a transmissive AND reflective material).
template <bool LDS, bool SPILL, int LIT, bool GHOSTS>
hipError_t launch_whitted_literal(bool aa, bool stats, const RenderParams& P, uint32_t blocks, size_t lds, hipStream_t st) {
  constexpr int A = P3D_ACCEL_BVH;
  if (aa && stats) hipLaunchKernelGGL((whitted_kernel<A, LDS, true, true, SPILL, 1, LIT, GHOSTS>), dim3(blocks), dim3(kBlock), lds, st, P);
  else if (aa) hipLaunchKernelGGL((whitted_kernel<A, LDS, false, true, SPILL, 1, LIT, GHOSTS>), dim3(blocks), dim3(kBlock), lds, st, P);
  else if (stats) hipLaunchKernelGGL((whitted_kernel<A, LDS, true, false, SPILL, 1, LIT, GHOSTS>), dim3(blocks), dim3(kBlock), lds, st, P);
  else hipLaunchKernelGGL((whitted_kernel<A, LDS, false, false, SPILL, 1, LIT, GHOSTS>), dim3(blocks), dim3(kBlock), lds, st, P);
  return hipGetLastError();
}
template <bool LDS, bool SPILL>
hipError_t launch_literal_variant(int lit, bool ghosts, bool aa, bool stats, const RenderParams& P, uint32_t blocks, size_t lds, hipStream_t st) {
  if (lit == 1) return ghosts ? launch_whitted_literal<LDS, SPILL, 1, true>(aa, stats, P, blocks, lds, st) : launch_whitted_literal<LDS, SPILL, 1, false>(aa, stats, P, blocks, lds, st);
  if (lit == 2) return ghosts ? launch_whitted_literal<LDS, SPILL, 2, true>(aa, stats, P, blocks, lds, st) : launch_whitted_literal<LDS, SPILL, 2, false>(aa, stats, P, blocks, lds, st);
  if (lit == 4) {
    if constexpr (LDS) {  // (only LDS-staged scenes repair over the tiles: p3d_render_tile_device)
      return ghosts ? launch_whitted_literal<LDS, SPILL, 3, true>(aa, stats, P, blocks, lds, st) : launch_whitted_literal<LDS, SPILL, 3, false>(aa, stats, P, blocks, lds, st);
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (lit == 5) {  // round 1 as a light launch: check the entries of list B, pass the few that change on to list C
    if (stats) hipLaunchKernelGGL((handoff_check_entries_kernel<LDS, SPILL, true>), dim3(blocks), dim3(kBlock), lds, st, P);
    else hipLaunchKernelGGL((handoff_check_entries_kernel<LDS, SPILL, false>), dim3(blocks), dim3(kBlock), lds, st, P);
    return hipGetLastError();
  }
  if (lit == 3) {
    if constexpr (!LDS) {  // (only scenes traversed from global memory announce: p3d_render_tile_device)
      if (stats) hipLaunchKernelGGL((handoff_check_list_kernel<LDS, SPILL, true>), dim3(blocks), dim3(kBlock), lds, st, P);
      else hipLaunchKernelGGL((handoff_check_list_kernel<LDS, SPILL, false>), dim3(blocks), dim3(kBlock), lds, st, P);
    } else {
      return hipErrorInvalidValue;
    }
  } else if (stats) {
    hipLaunchKernelGGL((handoff_check_kernel<LDS, SPILL, true>), dim3(blocks), dim3(kBlock), lds, st, P);
  } else {
    hipLaunchKernelGGL((handoff_check_kernel<LDS, SPILL, false>), dim3(blocks), dim3(kBlock), lds, st, P);
  }
  return hipGetLastError();
}
// lit 0 here = the check kernel
hipError_t launch_literal(int lit, bool ghosts, bool aa, bool lds_scene, bool stats, const RenderParams& P, uint32_t blocks, size_t lds, hipStream_t st) {
  if (lds_scene && P.stack_spills) return launch_literal_variant<true, true>(lit, ghosts, aa, stats, P, blocks, lds, st);
  if (lds_scene) return launch_literal_variant<true, false>(lit, ghosts, aa, stats, P, blocks, lds, st);
  return launch_literal_variant<false, true>(lit, ghosts, aa, stats, P, blocks, lds, st);
}

// Looks up the schedule for the launch described by (cfg, P).  Known key: P.sched is set.  New
// key: P.tile_cost is set so that this launch (in frame order) records the costs, and *fresh
// points at the entry, to be completed by schedule_finish() right after the launch.
int schedule_lookup(p3d_scene* s, const p3d_config* cfg, bool pt, RenderParams& P, hipStream_t st, SchedEntry** fresh) {
  *fresh = nullptr;
  SchedEntry key;
  key.accel = cfg->accel; key.aa = cfg->antialiasing ? 1 : 0; key.spp = cfg->antialiasing ? cfg->spp_sqrt : 1; key.pt = pt ? 1 : 0; key.tiles_x = P.tiles_x; key.tiles_y = P.tiles_y;
  key.max_depth = P.max_depth; key.x0 = P.x0; key.y0 = P.y0; key.w = P.w; key.h = P.h;
  key.stripe_h = P.stripe_h; key.stripe_stride = P.stripe_stride;
  for (SchedEntry& c : s->sched)
    if (c.built && c.same_key(key)) {
      if (c.built_on != st) P3D_HIP(hipStreamWaitEvent(st, c.ready, 0));
      c.last_use = ++s->sched_clock;
      P.sched = (const uint32_t*)c.sched.p;
      return P3D_OK;
    }
  SchedEntry* e = nullptr;
  if (s->sched.size() < kSchedCacheEntries) {
    s->sched.emplace_back();
    e = &s->sched.back();
  } else {  // recycle the least recently used entry once the work queued with it has drained
    e = &s->sched[0];
    for (SchedEntry& c : s->sched)
      if (!c.built || c.last_use < e->last_use) e = &c;
    P3D_HIP(hipDeviceSynchronize());
  }
  e->built = false;
  const uint32_t n = P.tiles_x * P.tiles_y;
  if (int rc = e->cost.ensure((size_t)n * sizeof(uint32_t))) return rc;
  if (int rc = e->sched.ensure((size_t)n * sizeof(uint32_t))) return rc;
  if (!e->ready) P3D_HIP(hipEventCreateWithFlags(&e->ready, hipEventDisableTiming));
  e->accel = key.accel; e->aa = key.aa; e->spp = key.spp; e->pt = key.pt; e->tiles_x = key.tiles_x; e->tiles_y = key.tiles_y; e->max_depth = key.max_depth; e->x0 = key.x0; e->y0 = key.y0;
  e->w = key.w; e->h = key.h; e->stripe_h = key.stripe_h; e->stripe_stride = key.stripe_stride;
  P.tile_cost = (uint32_t*)e->cost.p;
  *fresh = e;
  return P3D_OK;
}

int schedule_finish(p3d_scene* s, SchedEntry* e, uint32_t n_tiles, hipStream_t st) {
  hipLaunchKernelGGL(sched_build_kernel, dim3(1), dim3(kSchedBuildThreads), 0, st, (const uint32_t*)e->cost.p, n_tiles, (uint32_t*)e->sched.p);
  if (hipError_t err = hipGetLastError(); err != hipSuccess)
    return fail(P3D_ERR_NO_DEVICE, std::string("schedule kernel launch: ") + hipGetErrorString(err));
  P3D_HIP(hipEventRecord(e->ready, st));
  e->built = true;
  e->built_on = st;
  e->last_use = ++s->sched_clock;
  return P3D_OK;
}

int check_accel(const p3d_scene* s, uint32_t accel) {
  if (accel == P3D_ACCEL_BVH && !s->has_bvh) return fail(P3D_ERR_INVALID, "accel = Bvh but the scene was created without BVH arrays");
  if (accel == P3D_ACCEL_GRID && !s->has_grid) return fail(P3D_ERR_INVALID, "accel = UGrid but the scene was created without a grid");
  if (accel > P3D_ACCEL_BVH) return fail(P3D_ERR_INVALID, "unknown accel");
  return P3D_OK;
}

// worst-case node-stack height: each shadow feeler that returns `true` may leave up to
// depth-1 entries behind (Q2) and the next closest-hit query adds depth-1 more
uint32_t stack_bound(const p3d_scene* s, uint32_t accel, bool whitted) {
  if (accel != P3D_ACCEL_BVH) return 1;
  const uint32_t per = s->bvh_max_depth > 1 ? s->bvh_max_depth - 1 : 1;
  return whitted ? (s->dev.n_lights + 1) * per : per;
}


// Device-detected errors (sample hand-out loop hit its trip bound, a leftover outgrew its slot, the hand-off found no
// fixed point): read and clear the status word.  Call only where the stream has been synchronised.
int check_status(p3d_scene* s) {
  uint32_t h = 0;
  P3D_HIP(hipMemcpy(&h, s->d_status, sizeof(h), hipMemcpyDeviceToHost));
  s->last_status = h;
  if (!h) return P3D_OK;
  P3D_HIP(hipMemset(s->d_status, 0, sizeof(uint32_t)));
  std::string what;
  if (h & kHoErrTrips) what += " sample hand-out loop reached its trip bound (pixels would miss samples);";
  if (h & kHoErrLeftoverCap) what += " the hit_stack leftovers of this frame do not fit their records (p3d_config.handoff_records = P3D_HANDOFF_DENSE has room for the worst case);";
  if (h & kHoErrNoFixedPoint) what += " hit_stack hand-off did not reach a fixed point;";
  if (h & kHoErrHalo) what += " a row of a stripe / sub-rectangle could not be started on the hit_stack the serial frame hands it (no pixel in front of it certifiably independent of its own incoming stack): render it with more rows in front, as part of the whole frame, or with P3D_STACK_PER_PIXEL;";
  if (h & kHoErrList) what += " a work list of the hit_stack hand-off or a ray queue segment of the per-level launches overflowed;";
  return fail(P3D_ERR_CAPACITY, "device-detected error:" + what);
}

int finish_stats(p3d_scene* s, hipStream_t st, p3d_stats* stats, bool literal) {
  P3D_HIP(hipEventRecord(s->ev1, st));
  P3D_HIP(hipEventSynchronize(s->ev1));
  float ms = 0;
  P3D_HIP(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  unsigned long long h[kNumStats];
  P3D_HIP(hipMemcpy(h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost));
  std::memset(stats, 0, sizeof(*stats));
  stats->kernel_ms = ms;
  if (literal) {
    float a = 0;
    float head = 0;
    P3D_HIP(hipEventElapsedTime(&a, s->ev_p1, s->ev_mid));
    P3D_HIP(hipEventElapsedTime(&head, s->ev0, s->ev_mid));
    stats->pass1_ms = a;
    stats->handoff_ms = ms - head;
  }
  stats->rays_primary = h[kRaysPrimary]; stats->rays_shadow = h[kRaysShadow]; stats->rays_reflect = h[kRaysReflect];
  stats->rays_refract = h[kRaysRefract]; stats->rays_bounce = h[kRaysBounce]; stats->rays_light = h[kRaysLight];
  stats->node_tests = h[kNodeTests]; stats->sphere_tests = h[kSphereTests]; stats->tri_tests = h[kTriTests];
  stats->box_tests = h[kBoxTests]; stats->plane_tests = h[kPlaneTests]; stats->shaded_hits = h[kShadedHits];
  stats->pixels = h[kPixels]; stats->max_stack = h[kMaxStack];
  if (literal) {
    uint32_t c[kHoNumCounters];
    P3D_HIP(hipMemcpy(c, s->ho_counters.p, sizeof(c), hipMemcpyDeviceToHost));
    static const bool print_handoff = getenv("P3D_PRINT_HANDOFF") != nullptr;  // (profiles/tools/ab/lists_probe.py)
    if (print_handoff) std::fprintf(stderr, "handoff: checked %u redone %u rounds %u pool %u lists A %u B %u C %u D %u check_n %u round0 %u round1 %u\n", c[kHoChecked], c[kHoRedone], c[kHoRounds], c[kHoPoolTop], c[kHoListA], c[kHoListB], c[kHoListC], c[kHoListD], c[kHoCheckN], c[kHoRound0], c[kHoRound1]);
    stats->handoff_checked = c[kHoChecked]; stats->handoff_redone = c[kHoRedone]; stats->handoff_rounds = c[kHoRounds] + (c[kHoRound0] ? 1 : 0) + (c[kHoRound1] ? 1 : 0);
  }
  return check_status(s);
}

}  // namespace

extern "C" {

int p3d_render_tile_device(p3d_scene* s, const p3d_config* cfg, const p3d_tile* tile, float* d_rgb, int32_t* d_hit,
                           uint8_t* d_rgb8, void* hip_stream, p3d_stats* stats) {
  if (!s || !cfg || !tile) return fail(P3D_ERR_INVALID, "p3d_render_tile_device: null argument");
  if (int rc = check_accel(s, cfg->accel)) return rc;
  const DevCamera& cam = s->dev.cam;
  if (cam.res_x <= 0 || cam.res_y <= 0) return fail(P3D_ERR_INVALID, "scene has no camera");
  const int sh = tile->stripe_h > 0 ? tile->stripe_h : 1, ss = tile->stripe_h > 0 ? tile->stripe_stride : 1;
  if (tile->w <= 0 || tile->h <= 0 || tile->x0 < 0 || tile->y0 < 0 || ss < 1 || tile->x0 + tile->w > cam.res_x)
    return fail(P3D_ERR_INVALID, "tile outside the image");
  {
    const int last = tile->h - 1;
    const long long ylast = (long long)tile->y0 + (long long)(last / sh) * sh * ss + (last % sh);
    if (ylast >= cam.res_y) return fail(P3D_ERR_INVALID, "tile rows outside the image");
  }
  if (cfg->integrator > P3D_PATHTRACE || cfg->sample_mode > P3D_SAMPLE_TENT) return fail(P3D_ERR_INVALID, "bad integrator / sample_mode");
  if (cfg->tile_order > P3D_TILE_ORDER_FRAME) return fail(P3D_ERR_INVALID, "bad tile_order");
  if (cfg->stack_mode > P3D_STACK_PER_PIXEL) return fail(P3D_ERR_INVALID, "bad stack_mode");
  if (cfg->chain_launch > P3D_CHAIN_PER_LEVEL) return fail(P3D_ERR_INVALID, "bad chain_launch");
  if (cfg->debug_view > P3D_DEBUG_DEPTH_MAP) return fail(P3D_ERR_INVALID, "bad debug_view");
  if (cfg->handoff_records > P3D_HANDOFF_DENSE) return fail(P3D_ERR_INVALID, "bad handoff_records");
  if (cfg->max_depth < 0 || cfg->max_depth > 1024) return fail(P3D_ERR_INVALID, "max_depth out of range");
  if (cfg->antialiasing && (cfg->spp_sqrt == 0 || cfg->spp_sqrt > 1024)) return fail(P3D_ERR_INVALID, "spp_sqrt out of range");
  if (cfg->soft_shadows && !cfg->antialiasing)
    ;  // light replication (main.cpp:725-745) is a host-side scene edit: p3d_host_scene_replicate_lights
  if (cfg->accel == P3D_ACCEL_GRID && s->dev.n_objs == 0) return fail(P3D_ERR_UNSUPPORTED, "grid over an empty scene");
  if (cfg->skybox && !s->has_sky) return fail(P3D_ERR_INVALID, "config asks for SKYBOX but no cubemap was supplied (p3d_scene_set_skybox)");
  P3D_HIP(hipSetDevice(s->device));
  hipStream_t st = (hipStream_t)hip_stream;
  s->last_status = 0;
  if (s->tail_pending) {  // the scene's scratch is in use until the previous frame's tail has run (p3d_scene_set_tail_stream)
    P3D_HIP(hipStreamWaitEvent(st, s->ev_tail_done, 0));
    s->tail_pending = false;
  }

  // main.cpp:804-812: without ANTIALIASING the frame loop always calls rayTracing
  const bool pt = cfg->integrator == P3D_PATHTRACE && cfg->antialiasing;
  // Only rayTracing over the BVH has a stack that survives a query (bvh.cpp:86,322); Radiance asks closest-hit
  // queries only, which leave it empty (bvh.cpp:256-274), the grid and the object loop have none.
  const bool literal = cfg->stack_mode == P3D_STACK_LITERAL && !pt && cfg->accel == P3D_ACCEL_BVH;
  // worst-case node-stack height (stack_bound).  LDS-staged scenes keep the WHOLE stack in LDS
  // (kernel variant without a spill path); deep trees / many lights use the global-memory variant,
  // which holds the first `cap` entries in LDS and spills the rest.
  const uint32_t bound = stack_bound(s, cfg->accel, !pt);
  // A scene of up to 26 KB is staged in LDS (beyond that the waves a CU can hold get too few: 37 KB staged
  // 1.22 ms, from L2 0.76 ms).  If scene + worst-case stack fit 20 KB and the stack has at most 24 entries, the
  // whole stack lives in LDS too (kernel without a spill path); otherwise (Whitted, several lights, deeper tree)
  // the staged scene is combined with the spilling stack: balls_medium 0.48 -> 0.40 ms, balls_box 0.42 -> 0.32 ms,
  // 96 / 128 random objects 0.49 -> 0.42 / 0.75 -> 0.66 ms against traversing those 14-25 KB from L2.
  // (staged: blob[off_nodes, stage_hi) - not the alignment pad in front of the nodes, and not the object-order geometry when
  // the kernel walks the BVH and is not the path tracer, which looks its emitters up by object)
  const uint32_t stage_lo = s->off_nodes, stage_hi = (cfg->accel == P3D_ACCEL_BVH && !pt) ? s->off_ogeom : s->blob_f4;
  const size_t stage_bytes = (size_t)(stage_hi - stage_lo) * sizeof(float4);
  const bool lds_scene = stage_bytes <= (pt ? kLdsSceneLimitBytesPt : kLdsSceneLimitBytes) && (bound <= 24 || !pt);
  const bool lds_spill = lds_scene && !pt && (bound > 24 || stage_bytes + (size_t)stack_lds_f4(false, bound) * sizeof(float4) > 20 * 1024);
  // Spilling stack (scenes traversed from L2; LDS-staged scenes whose worst case does not fit): LDS holds a window of the
  // most recent `window` entries (a power of two, device_core.hpp "Stack"), older entries sink into a per-thread column of
  // a global backing array.  8 entries = 4 KB per wave; the Whitted kernels that trade registers for waves keep 2.3 KB of
  // cold shading state behind it (cold_lds below): 6.4 KB per wave = 25 waves per CU by LDS, one more than the 24 (6 per
  // SIMD) those kernels' registers allow.
  uint32_t window = 8;
  if (const char* e = getenv("P3D_LDS_STACK_ENTRIES")) {  // experiments: LDS entries per lane (rounded up to a power of two)
    window = 1;
    while (window < (uint32_t)std::max(1, atoi(e)) && window < 1024) window *= 2;
  }
  const bool spilling = cfg->accel == P3D_ACCEL_BVH && !(lds_scene && !lds_spill);
  const uint32_t cap = cfg->accel == P3D_ACCEL_BVH ? (spilling ? window : bound) : 1;
  const uint32_t spill_entries = (spilling && bound > cap) ? bound : 0;  // rows of the backing array (entry e in row e)
  const bool want_counts = stats && cfg->collect_stats;

  RenderParams P{};
  P.sc = s->dev;
  P.blob = s->d_blob; P.blob_f4 = lds_scene ? stage_hi - stage_lo : 0; P.stage_lo = stage_lo;
  P.off_nodes = s->off_nodes; P.off_bgeom = s->off_bgeom; P.off_ogeom = s->off_ogeom;
  P.off_normals = s->off_normals; P.off_mats = s->off_mats; P.off_lights = s->off_lights;
  P.max_depth = cfg->max_depth; P.spp_sqrt = cfg->spp_sqrt; P.antialiasing = cfg->antialiasing;
  P.depth_of_field = cfg->depth_of_field; P.sample_disk = cfg->sample_disk; P.soft_shadows = cfg->soft_shadows;
  P.sample_mode = cfg->sample_mode; P.light_side = cfg->light_side; P.gamma = cfg->gamma; P.seed = cfg->seed;
  P.skybox = cfg->skybox ? 1u : 0u;
  P.debug_view = cfg->debug_view;
  P.stripe_h = tile->stripe_h > 0 ? tile->stripe_h : 0; P.stripe_stride = ss;
  P.stats = s->d_stats;
  P.status = s->d_status;
  P.debug_trip_bound = s->dbg.trip_bound;
  P.stack_cap = (int32_t)cap;
  P.stack_spills = lds_spill ? 1u : 0u;
  P.lds_scene_f4 = P.blob_f4;
  // path tracer with >= 16 samples per pixel: four lanes per pixel, 4x4-pixel tiles (pt_kernel SUB = 4)
  // ... and anti-aliased Whitted launches with >= 4 samples per pixel over a scene traversed from L2 (whitted_kernel SUB = 4),
  // unless the samples of a pixel have to hand the stack to each other in order (LITERAL)
  const bool sub4 = (pt && cfg->spp_sqrt >= kPtSub4MinSppSqrt) ||
                    (!pt && !literal && !lds_scene && cfg->antialiasing && cfg->spp_sqrt >= kWhittedSub4MinSppSqrt);
  // ... and behind the node stack: the sample ring of the four-lanes-per-pixel kernels, or the cold shading state of the
  // Whitted kernels that traverse the scene from L2 without anti-aliasing (ColdState<true>, device_core.hpp)
  const bool cold_lds = !pt && !lds_scene && !cfg->antialiasing;
  // Pixels per wave.  Four lanes per pixel: 4x4.  One lane per pixel: 8x8, or - Whitted over a scene traversed from L2,
  // where a wave is as long as the slowest of its lanes in every query - 8x4 / 4x4 when the launch has too few 8x8 tiles
  // to keep the wave slots busy for several rounds (stripes of a multi-GPU frame, small frames): quarter waves are
  // shorter and four times as many, at the price of issue slots the chip then has to spare (tile_shape()).
  uint32_t tpw = sub4 ? 4 : 8, tph = sub4 ? 4 : 8;
  if (!sub4 && !pt && !lds_scene && !cfg->antialiasing && cfg->chain_launch != P3D_CHAIN_PER_LEVEL) tile_shape((uint64_t)tile->w * tile->h, tpw, tph);
  const uint32_t tp = tph;  // rows per tile band
  P.tile_w_shift = tpw == 8 ? 3 : 2;
  P.tile_h_shift = tph == 8 ? 3 : 2;
  // (entry size: what the kernel's SPILL parameter says - the window for every kernel over a scene that is not staged and for
  // lds_spill; whole stack in LDS: eight-byte entries for the path tracer, six-byte ones for Whitted)
  const int stack_mode = (!lds_scene || lds_spill) ? kStackWindow : (pt ? kStackLds8 : kStackLds6);
  const size_t lds_bytes = (size_t)P.lds_scene_f4 * sizeof(float4) + (size_t)stack_lds_f4(stack_mode, cap) * sizeof(float4) +
                           (sub4 ? sizeof(PtPixelShared) : 0) + (cold_lds ? (size_t)kColdDwords * kBlock * sizeof(float) : 0);

  // rows per launch: whole 8-row tile bands, at most kMaxLaunchThreads threads
  const uint32_t tiles_x = ((uint32_t)tile->w + tpw - 1) / tpw;
  // per-thread global scratch: Whitted level records (+ the zero-weight reflection rays a LITERAL launch puts aside),
  // or the path tracer's two deferred dielectric branches
  const uint32_t levels = pt ? 2 * 3 : (uint32_t)cfg->max_depth;
  const uint32_t deferred = (literal && s->zero_weight_reflections) ? 2u * (uint32_t)std::max(cfg->max_depth, 1) : 0u;
  // One launch per chain level (wf_level_kernel) where the lanes of a megakernel wave die off in the reflection levels:
  // Whitted without anti-aliasing over a BVH read from L2.  Not for the zero-weight reflection rays of LITERAL frames
  // (they make the chain a tree) and not for a grid or the object loop (no stack record to carry between launches).
  bool per_level = !pt && !cfg->antialiasing && cfg->accel == P3D_ACCEL_BVH && !lds_scene && cfg->max_depth >= 1 && cfg->max_depth <= 64 &&
                   !(literal && s->zero_weight_reflections) && cfg->chain_launch == P3D_CHAIN_PER_LEVEL;
  if (cfg->chain_launch == P3D_CHAIN_PER_LEVEL && !per_level)
    return fail(P3D_ERR_UNSUPPORTED, "chain_launch = PER_LEVEL needs Whitted without anti-aliasing over a BVH too big for LDS (and no transmissive + reflective material under P3D_STACK_LITERAL)");
  // per-level launches keep their level records per pixel, not per launch thread
  size_t scratch_per_thread = (size_t)((per_level ? 0 : levels) + deferred) * sizeof(float4) + (size_t)spill_entries * sizeof(uint2);
  const uint32_t launch_threads = (uint32_t)std::min<size_t>(per_level ? (1u << 23) : kMaxLaunchThreads, kLaunchScratchBudget / std::max<size_t>(scratch_per_thread, 1));
  uint32_t bands_per_launch = std::max<uint32_t>(1, launch_threads / (tiles_x * kBlock));
  const uint32_t total_bands = ((uint32_t)tile->h + tp - 1) / tp;
  if (tile->stripe_h > 0 && sh % (int)tp == 0 && bands_per_launch >= (uint32_t)sh / tp)
    bands_per_launch = (bands_per_launch / ((uint32_t)sh / tp)) * ((uint32_t)sh / tp);  // chunks start on a stripe boundary
  bands_per_launch = std::min(bands_per_launch, total_bands);
  if (per_level && bands_per_launch < total_bands) {  // the per-level path renders the tile in one go
    if (cfg->chain_launch == P3D_CHAIN_PER_LEVEL) return fail(P3D_ERR_CAPACITY, "chain_launch = PER_LEVEL: tile too large for one launch");
    per_level = false;
    scratch_per_thread += (size_t)levels * sizeof(float4);
    const uint32_t lt = (uint32_t)std::min<size_t>(kMaxLaunchThreads, kLaunchScratchBudget / std::max<size_t>(scratch_per_thread, 1));
    bands_per_launch = std::min(std::max<uint32_t>(1, lt / (tiles_x * kBlock)), total_bands);
    if (tile->stripe_h > 0 && sh % (int)tp == 0 && bands_per_launch >= (uint32_t)sh / tp)
      bands_per_launch = (bands_per_launch / ((uint32_t)sh / tp)) * ((uint32_t)sh / tp);
  }
  const uint32_t xcd_chunk = lds_scene ? 1u : tiles_x;
  auto blocks_for = [&](uint32_t ntiles) {  // grid covering ntiles under the chunked XCD map
    const uint32_t groups = (ntiles + 8 * xcd_chunk - 1) / (8 * xcd_chunk);
    return groups * 8 * xcd_chunk;
  };
  // Cost-ordered tiles (DESIGN.md "Tile schedule"): the frame order leaves a tail of a few long-running tiles.  Since
  // round 3 also for scenes traversed from L2 (100k triangles 2048x2048 18.25 -> 16.9 ms, 1024x1024 7.2 -> 6.5 ms; in
  // round 2, with child pairs straddling cache lines, the lost L2 locality cost more than the tail: 7.2 -> 7.5-7.9 ms).
  const bool sched_ok = cfg->tile_order == P3D_TILE_ORDER_COST && cfg->max_depth > 0;
  // LITERAL: workgroups behind the tile grid of the first launch render the halo chains (8 chains of 8 pixels per wave)
  const uint32_t halo_blocks_max = literal ? ((uint32_t)tile->h * kHaloChain + kBlock - 1) / kBlock : 0;
  const uint32_t max_threads = (blocks_for(tiles_x * bands_per_launch) + halo_blocks_max) * kBlock;
  const size_t tile_units = (size_t)tile->h * ((size_t)tile->w + kHaloChain);  // upper bound of H.n_units
  // per-level launches keep one record per (level, unit); the work-list launches of a LITERAL frame behind them are the
  // megakernel and index [level][launch thread] with up to max_threads threads, whatever the size of the tile
  const size_t level_cols = per_level ? std::max<size_t>(tile_units, literal ? (size_t)max_threads : 0) : (size_t)max_threads;
  if (int rc = s->levels.ensure(std::max<size_t>(16, (size_t)levels * level_cols * sizeof(float4)))) return rc;
  // (+ halo_find_kernel: one traversal per chain row on an empty stack, window of 8 entries, the rest of a tree path here)
  const size_t halo_backing = literal && s->bvh_max_depth > 8 ? (size_t)s->bvh_max_depth * tile->h * kBlock * sizeof(uint2) : 0;
  if (int rc = s->spill.ensure(std::max<size_t>(16, std::max((size_t)spill_entries * max_threads * sizeof(uint2), halo_backing)))) return rc;
  if (int rc = s->deferred.ensure(std::max<size_t>(16, (size_t)deferred * max_threads * sizeof(float4)))) return rc;
  P.levels = (float4*)s->levels.p;
  P.spill = (uint2*)s->spill.p;
  P.deferred = (float4*)s->deferred.p;

  // ---- P3D_STACK_LITERAL: per-unit records of the hit_stack hand-off (csrc/handoff.hpp) ----
  Handoff& H = P.hand;
  uint4* ho_list[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* ho_counters = nullptr;
  size_t touched_bytes = 0;
  uint32_t counter_words = 0, wf_seg_cap = 0;
  if (literal || per_level) {
    const uint32_t per = s->bvh_max_depth > 1 ? s->bvh_max_depth - 1 : 1;
    // Rows whose predecessor in the frame is not the end of the tile row above start a chain of their own (halo_find_kernel)
    const bool full_width = tile->x0 == 0 && tile->w == cam.res_x;
    std::vector<uint8_t> chain((size_t)tile->h, 0);
    bool any_chain = false;
    if (literal) {
      long long y_prev = -2;
      for (int r = 0; r < tile->h; ++r) {
        const long long y = (long long)tile->y0 + (long long)(r / sh) * sh * ss + (r % sh);
        chain[r] = r == 0 ? !(tile->x0 == 0 && y == 0) : !(full_width && y == y_prev + 1);
        any_chain = any_chain || chain[r];
        y_prev = y;
      }
    }
    H.halo = any_chain ? kHaloChain : 0;
    H.row_units = (uint32_t)tile->w + H.halo;
    H.rows = (uint32_t)tile->h;
    if ((uint64_t)H.rows * H.row_units >= 0xffffffffull) return fail(P3D_ERR_CAPACITY, "tile too large for the hit_stack hand-off");
    H.n_units = H.rows * H.row_units;
    // what a pixel can leave behind: the entries its last shading point's feelers left (Q2), one tree path per light
    H.cap = std::max<uint32_t>(1, std::min<uint32_t>(bound, s->dev.n_lights * per));
    if (H.cap > 0xffffu) return fail(P3D_ERR_CAPACITY, "hit_stack leftover bound exceeds 65535 entries (lights x tree depth)");
    // leftover records: a pool with 8 entries per unit on average and an offset table (compact), or the worst case of
    // every unit (dense: the per-level launches rewrite a unit's record level by level; p3d_config.handoff_records)
    H.dense = (per_level || cfg->handoff_records == P3D_HANDOFF_DENSE) ? 1u : 0u;
    uint64_t pool_entries = H.dense ? (uint64_t)2 * H.cap * H.n_units : s->dbg.leftover_pool ? s->dbg.leftover_pool : std::max<uint64_t>(1u << 16, (uint64_t)kPoolEntriesPerUnit * H.n_units);
    if (!H.dense) pool_entries = std::min<uint64_t>(pool_entries, (uint64_t)2 * H.cap * H.n_units);  // never more than dense would take
    if (pool_entries > 0xffffffffull) return fail(P3D_ERR_CAPACITY, "hit_stack hand-off records exceed 2^32 entries: render the frame in smaller tiles or use P3D_STACK_PER_PIXEL");
    H.pool_cap = (uint32_t)pool_entries;
    if (int rc = s->ho_entries.ensure((size_t)pool_entries * sizeof(uint2))) return rc;
    if (!H.dense)
      if (int rc = s->ho_where.ensure((size_t)2 * H.n_units * sizeof(uint32_t))) return rc;
    if (int rc = s->ho_meta.ensure((size_t)2 * H.n_units * 4)) return rc;  // meta | meta0 (pass 1's copy)
    if (int rc = s->ho_first.ensure((size_t)H.n_units * sizeof(float4))) return rc;
    if (int rc = s->ho_first_sample.ensure(cfg->antialiasing ? (size_t)H.n_units * 4 : 16)) return rc;
    touched_bytes = ((size_t)H.n_units / 32 + 2) * 4;
    if (int rc = s->ho_touched.ensure(touched_bytes)) return rc;
    if (int rc = s->ho_lists.ensure((size_t)4 * H.n_units * sizeof(uint4))) return rc;
    if (!H.dense && !lds_scene)  // the check launch's work list (handoff.hpp Handoff::check_list)
      if (int rc = s->ho_check.ensure((size_t)H.n_units * sizeof(uint32_t))) return rc;
    // hand-off counters in the first 128-byte line, then one line per (chain level, queue segment) of the per-level launches
    // ... and the bin counts of every level's ray queue
    counter_words = 32 + (per_level ? ((uint32_t)cfg->max_depth + 1) * (kWfSegments * kWfCounterStride + kWfHistWords) : 0);
    if (int rc = s->ho_counters.ensure((size_t)counter_words * sizeof(uint32_t))) return rc;
    if (per_level) {
      wf_seg_cap = H.n_units / 4 + 4096;  // a segment takes the rays of every 8th workgroup: twice its fair share
      if (int rc = s->wf_rays.ensure((size_t)kWfSegments * wf_seg_cap * 2 * sizeof(float4))) return rc;
      if (int rc = s->wf_keys.ensure((size_t)kWfSegments * wf_seg_cap * sizeof(uint32_t))) return rc;
      if (int rc = s->wf_sorted.ensure((size_t)H.n_units * 2 * sizeof(float4))) return rc;
      if (int rc = s->wf_final.ensure((size_t)H.n_units * sizeof(float4))) return rc;
    }
    H.entries = (uint2*)s->ho_entries.p;
    H.where = (uint32_t*)s->ho_where.p;
    H.meta = (uint32_t*)s->ho_meta.p;
    H.meta0 = H.meta + H.n_units;
    H.first = (float4*)s->ho_first.p;
    H.first_sample = (uint32_t*)s->ho_first_sample.p;
    H.touched = (uint32_t*)s->ho_touched.p;
    H.row_chain = nullptr;
    H.halo_pix = nullptr;
    if (any_chain) {
      if (int rc = s->ho_row_chain.ensure((size_t)tile->h)) return rc;
      if (int rc = s->ho_halo_pix.ensure((size_t)tile->h * kHaloChain * 4)) return rc;
      H.row_chain = (const uint8_t*)s->ho_row_chain.p;
      H.halo_pix = (const uint32_t*)s->ho_halo_pix.p;
    }
    if (literal && want_counts) {  // per-unit counters: a unit rendered again replaces its first pass (handoff.hpp)
      if (int rc = s->ho_ucount.ensure((size_t)(kNumStats + kCh0Counters) * H.n_units * sizeof(uint32_t))) return rc;
      H.ucount = (uint32_t*)s->ho_ucount.p;
      H.uch0 = H.ucount + (size_t)kNumStats * H.n_units;
    }
    H.list_cap = H.n_units;
    H.count = want_counts ? 1u : 0u;
    H.max_rounds = std::min<uint32_t>(H.n_units + 2, 4096);  // a chain of n units is exact after n rounds at the latest; measured: 2.  Beyond the cap: P3D_ERR_CAPACITY
    if (s->dbg.max_rounds) H.max_rounds = s->dbg.max_rounds;
    for (int i = 0; i < 4; ++i) ho_list[i] = (uint4*)s->ho_lists.p + (size_t)i * H.n_units;
    ho_counters = (uint32_t*)s->ho_counters.p;
    H.counters = ho_counters;
    H.pool_top = ho_counters + kHoPoolTop;
    // (scenes traversed from global memory only: over an LDS-staged scene 64 unrelated pixels per wave diverge for longer than
    // the few neighbouring lanes of a tile's wave take, cfg2 literal loop 37.3 k -> 35.1 k Mrays/s; 100k triangles 15.9 -> 15.5 ms)
    H.check_list = (H.dense || lds_scene) ? nullptr : (uint32_t*)s->ho_check.p;
    H.check_n = ho_counters + kHoCheckN;
  }
  if (stats && literal) P3D_HIP(hipEventRecord(s->ev0, st));  // kernel_ms of a LITERAL frame is the whole frame: halo search (when not memoised) and clear included
  uint32_t halo_blocks = 0;
  if (literal && H.halo) {
    halo_blocks = (H.rows * kHaloChain + kBlock - 1) / kBlock;
    // The chain flags and the pixels in front of every chain row are a function of the tile and of what shapes the
    // primary rays (the scene and its camera are fixed): worked out by two launches on this stream when any of that
    // changes, kept otherwise (the frames of a sequence find them ready).
    const uint32_t max_chain = s->dbg.halo_chain ? std::min<uint32_t>(s->dbg.halo_chain, kHaloChain) : kHaloChain;
    const std::vector<int64_t> key = {tile->x0, tile->y0, tile->w, tile->h, tile->stripe_h, tile->stripe_stride, cam.res_x, cam.res_y,
                                      cfg->antialiasing ? 1 : 0, cfg->antialiasing ? (int64_t)cfg->spp_sqrt : 1, (int64_t)cfg->seed,
                                      cfg->sample_mode, cfg->depth_of_field, cfg->sample_disk, max_chain, (int64_t)(intptr_t)st};
    if (key != s->ho_chain_key) {
      RowChainParams C{(uint8_t*)s->ho_row_chain.p, s->d_halo_verdict, tile->h, tile->x0, tile->y0, tile->w, cam.res_x, sh, ss};
      hipLaunchKernelGGL(row_chain_kernel, dim3(((uint32_t)tile->h + 255) / 256), dim3(256), 0, st, C);
      P3D_HIP(hipGetLastError());
      P.x0 = tile->x0; P.y0 = tile->y0; P.row0 = 0; P.w = tile->w; P.h = tile->h;
      const uint32_t find_window = 8;  // LDS entries of the one traversal at a time each workgroup runs; deeper ones in P.spill
      hipLaunchKernelGGL(halo_find_kernel, dim3(H.rows), dim3(kHaloFindThreads), (size_t)find_window * kBlock * sizeof(uint2), st, P,
                         (uint32_t*)s->ho_halo_pix.p, s->d_halo_verdict, max_chain, s->has_spheres ? 1u : 0u, find_window, H.rows * kBlock);
      P3D_HIP(hipGetLastError());
      s->ho_chain_key = key;
    }
  }
  if (literal || per_level || stats) {  // one clear launch at the head of the frame: statistics, counters, touched bits
    ClearParams C{};
    if (stats) { C.p[0] = (uint32_t*)s->d_stats; C.n[0] = kNumStats * 2; }
    if (literal || per_level) { C.p[1] = ho_counters; C.n[1] = counter_words; }
    if (literal) { C.p[2] = (uint32_t*)s->ho_touched.p; C.n[2] = (uint32_t)(touched_bytes / 4); }
    if (literal && H.halo) { C.halo_verdict = s->d_halo_verdict; C.status = s->d_status; }
    // (not LITERAL: the clear is there for the counters only and stays outside kernel_ms)
    const uint32_t words = std::max(C.n[0], std::max(C.n[1], C.n[2]));
    hipLaunchKernelGGL(clear_kernel, dim3(std::min<uint32_t>(256, (words + 255) / 256)), dim3(256), 0, st, C);
    P3D_HIP(hipGetLastError());
    if (stats && !literal) P3D_HIP(hipEventRecord(s->ev0, st));
  }
  if (stats && literal) P3D_HIP(hipEventRecord(s->ev_p1, st));  // pass1_ms: the speculative pass on its own
  // Round 0 of the hand-off over an LDS-staged scene: check AND repair over the tiles in one launch (whitted_kernel LIT = 3)
  // instead of a check launch that fills a list and a launch that renders the listed units again, 64 unrelated pixels per wave.
  const bool repair_tiles = literal && lds_scene && !per_level;
  const bool ghosts = s->zero_weight_reflections;
  std::vector<const uint32_t*> band_sched;  // the tile schedule pass 1 used for each band: the tile launch of round 0 takes the same order
  // Everything behind pass 1 is a chain of short dependent launches (p3d_scene_set_tail_stream): on a stream of its own it
  // does not hold up the launches the caller enqueues behind this frame on `st` - other scenes' pass 1.  Not with `stats`
  // (the frame is timed as a whole on one stream).
  bool on_tail = false;
  auto to_tail = [&]() -> int {
    if (on_tail || !literal || !s->tail_stream || stats || s->tail_stream == st) return P3D_OK;
    P3D_HIP(hipEventRecord(s->ev_tail_go, st));
    P3D_HIP(hipStreamWaitEvent(s->tail_stream, s->ev_tail_go, 0));
    st = s->tail_stream;
    on_tail = true;
    return P3D_OK;
  };
  // pass: 0 = the render launches (LITERAL: pass 1, everything on an empty stack); 1 = LITERAL only: the check launches
  for (int pass = 0; pass < (literal ? 2 : 1); ++pass) {
    if (pass == 1)
      if (int rc = to_tail()) return rc;
    if (pass == 1 && (abl_skip() & 1u)) break;
    if (pass == 1 && H.check_list) break;  // the check runs over pass 1's list, once for the whole tile (below)
    for (uint32_t band0 = 0; band0 < total_bands; band0 += bands_per_launch) {
      const uint32_t nb = std::min(bands_per_launch, total_bands - band0);
      const int row0 = (int)(band0 * tp);
      const int rows = std::min<int>((int)(nb * tp), tile->h - row0);
      // a chunk starts at local row row0; stripes make the image row a function of the LOCAL
      // row of the whole tile, so pass the tile origin and offset the outputs instead
      P.x0 = tile->x0; P.w = tile->w;
      P.h = rows;
      P.row0 = row0;
      if (P.stripe_h > 0) {
        if (row0 % sh != 0 && nb != total_bands) return fail(P3D_ERR_UNSUPPORTED, "stripe_h must divide the tile bands when a frame is split into several launches");
        P.y0 = tile->y0 + (row0 / sh) * sh * ss + (row0 % sh);
      } else {
        P.y0 = tile->y0 + row0;
      }
      P.tiles_x = tiles_x; P.tiles_y = nb;
      P.xcd_chunk = xcd_chunk;
      P.sched = nullptr;
      P.tile_cost = nullptr;
      SchedEntry* fresh = nullptr;
      if (pass == 0 && sched_ok && tiles_x * nb >= (lds_scene ? kSchedMinTiles : kSchedMinTilesL2))
        if (int rc = schedule_lookup(s, cfg, pt, P, st, &fresh)) return rc;
      if (pass == 0) band_sched.push_back(fresh ? (const uint32_t*)fresh->sched.p : P.sched);
      if (pass == 1 && repair_tiles) P.sched = band_sched[band0 / bands_per_launch];  // (built on this stream by now: schedule_finish)
      const uint32_t tile_blocks = blocks_for(tiles_x * nb);
      const uint32_t blocks = tile_blocks + (band0 == 0 ? halo_blocks : 0);  // the halo chains ride on the first launch
      P.tile_blocks = tile_blocks;
      P.level_stride = blocks * kBlock;
      const size_t off = (size_t)row0 * tile->w;
      P.rgb = d_rgb ? d_rgb + 3 * off : nullptr;
      P.hit_id = d_hit ? d_hit + off : nullptr;
      P.rgb8 = d_rgb8 ? d_rgb8 + 3 * off : nullptr;
      hipError_t e = hipSuccess;
      if (per_level && pass == 0) {
        // level 0 over the tiles (+ halo chains), then one launch per chain level over the queue the level above wrote,
        // then the fold.  The grid of a queue launch is fixed (the queue length is only known on the device): 24 waves
        // per CU, each taking every (grid / 8)-th chunk of 64 entries of its segment.
        P.wf_seg_cap = wf_seg_cap;
        P.wf_final = (float4*)s->wf_final.p;
        P.level_stride2 = H.n_units;
        uint32_t* wf_counters = ho_counters + 32;
        uint32_t* wf_hists = wf_counters + (size_t)((uint32_t)cfg->max_depth + 1) * kWfSegments * kWfCounterStride;
        const uint32_t queue_blocks = std::min<uint32_t>(kWfSegments * 768, std::max<uint32_t>(kWfSegments, (max_threads / kBlock) / kWfSegments * kWfSegments));
        P.wf_ray_out = (float4*)s->wf_rays.p; P.wf_ray_in = (float4*)s->wf_rays.p;
        P.wf_key_out = (uint32_t*)s->wf_keys.p; P.wf_key_in = (const uint32_t*)s->wf_keys.p;
        P.wf_sorted = (float4*)s->wf_sorted.p;
        P.wf_cell_origin = F3{s->root_min[0], s->root_min[1], s->root_min[2]};
        auto scale = [&](int a) { const float w = s->root_max[a] - s->root_min[a]; return w > 0 ? (float)kWfCellsPerAxis / w : 0.0f; };
        P.wf_cell_scale = F3{scale(0), scale(1), scale(2)};
        for (int level = 0; level <= cfg->max_depth && e == hipSuccess; ++level) {
          P.wf_level = (uint32_t)level;
          P.wf_n_out = wf_counters + (size_t)level * kWfSegments * kWfCounterStride;
          P.wf_hist = wf_hists + (size_t)level * kWfHistWords;
          P.wf_total = level ? wf_hists + (size_t)(level - 1) * kWfHistWords + kWfBins : nullptr;
          const uint32_t g = level == 0 ? blocks : queue_blocks;
          P.level_stride = g * kBlock;
          // (wf_level_kernel keeps its shading state in registers: no cold area behind its stack window)
          const size_t wf_lds = lds_bytes - (cold_lds ? (size_t)kColdDwords * kBlock * sizeof(float) : 0);
          if (want_counts && literal) hipLaunchKernelGGL((wf_level_kernel<true, 1>), dim3(g), dim3(kBlock), wf_lds, st, P);
          else if (want_counts) hipLaunchKernelGGL((wf_level_kernel<true, 0>), dim3(g), dim3(kBlock), wf_lds, st, P);
          else if (literal) hipLaunchKernelGGL((wf_level_kernel<false, 1>), dim3(g), dim3(kBlock), wf_lds, st, P);
          else hipLaunchKernelGGL((wf_level_kernel<false, 0>), dim3(g), dim3(kBlock), wf_lds, st, P);
          e = hipGetLastError();
          if (e == hipSuccess && level < cfg->max_depth) {  // put the child rays in bin order for the next level
            hipLaunchKernelGGL(wf_scan_kernel, dim3(1), dim3(1024), 0, st, P.wf_hist);
            P.wf_n_in = P.wf_n_out;
            hipLaunchKernelGGL(wf_scatter_kernel, dim3(kWfSegments * 128), dim3(256), 0, st, P, (const uint32_t*)(P.wf_hist + kWfBins));
            e = hipGetLastError();
          }
        }
        if (e == hipSuccess) {
          hipLaunchKernelGGL(wf_fold_kernel, dim3(((uint32_t)tile->w * (uint32_t)tile->h + 255) / 256), dim3(256), 0, st, P);
          e = hipGetLastError();
        }
        P.level_stride = blocks * kBlock;
      } else if (literal) {
        if (pass == 1) { H.list_out = ho_list[repair_tiles ? 1 : 0]; H.n_out = ho_counters + kHoListA + (repair_tiles ? 1 : 0); }
        e = launch_literal(pass == 0 ? 1 : (repair_tiles ? 4 : 0), ghosts, cfg->antialiasing != 0, lds_scene, want_counts, P, blocks, lds_bytes, st);
      } else {
        switch (cfg->accel) {
          case P3D_ACCEL_BVH: e = launch_accel<P3D_ACCEL_BVH>(pt, cfg->antialiasing != 0, sub4, lds_scene, want_counts, P, blocks, lds_bytes, st); break;
          case P3D_ACCEL_GRID: e = launch_accel<P3D_ACCEL_GRID>(pt, cfg->antialiasing != 0, sub4, lds_scene, want_counts, P, blocks, lds_bytes, st); break;
          default: e = launch_accel<P3D_ACCEL_NONE>(pt, cfg->antialiasing != 0, sub4, lds_scene, want_counts, P, blocks, lds_bytes, st); break;
        }
      }
      if (e != hipSuccess) {
        if (fresh) fresh->built = false;
        return fail(P3D_ERR_NO_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
      }
      if (fresh)
        if (int rc = schedule_finish(s, fresh, tiles_x * nb, st)) return rc;
    }
    if (literal && stats && pass == 0) P3D_HIP(hipEventRecord(s->ev_mid, st));
  }
  if (int rc = to_tail()) return rc;
  if (literal) {
    // The rounds behind pass 1, over the whole tile.  Round 0: the units whose first closest hit changes under the predecessor's
    // pass-1 leftover are rendered again - over the tiles, by the launch above (LDS-staged scenes: it wrote list B directly), or
    // from list A, which the check launch fills (scenes traversed from global memory) - and put the successors of the units whose
    // own leftover changed on list B.  Round 1: a light launch re-traces the first closest hits of list B's units on the new
    // leftovers and passes the rare one that changes on to list C; one persistent workgroup renders list C again and iterates
    // whatever is left after that - almost never anything - to the fixed point (C -> D -> C ...).
    P.x0 = tile->x0; P.w = tile->w; P.h = tile->h; P.row0 = 0; P.y0 = tile->y0;
    P.rgb = d_rgb; P.hit_id = d_hit; P.rgb8 = d_rgb8;
    P.sched = nullptr; P.tile_cost = nullptr;
    // workgroups of a work-list launch: far fewer units than pixels are expected (grid-stride loop for the rest)
    // (one workgroup per 64 pixels at most: with few list entries per wave a list of 1 % of the pixels still gets a wave
    // per chunk; workgroups without a chunk leave at once)
    const uint32_t wide = std::max<uint32_t>(1, std::min<uint32_t>(max_threads / kBlock, std::max<uint32_t>(64, H.n_units / kBlock)));
    if (H.check_list && !(abl_skip() & 1u)) {  // the check of round 0 over the units pass 1 announced: writes list A
      H.list_out = ho_list[0]; H.n_out = ho_counters + kHoListA;
      P.level_stride = wide * kBlock;
      P.tile_blocks = wide;
      const hipError_t e = launch_literal(3, ghosts, cfg->antialiasing != 0, lds_scene, want_counts, P, wide, lds_bytes, st);
      if (e != hipSuccess) return fail(P3D_ERR_NO_DEVICE, std::string("hand-off check launch: ") + hipGetErrorString(e));
    }
    for (int round = 0; round < 3; ++round) {
      if ((round == 0 && (abl_skip() & 2u)) || (round > 0 && (abl_skip() & 4u))) continue;
      if (round == 0 && repair_tiles) continue;  // the tile launch above was round 0 and wrote list B
      H.list_in = ho_list[round]; H.n_in = ho_counters + kHoListA + round;
      H.list_out = ho_list[round + 1]; H.n_out = ho_counters + kHoListA + round + 1;
      H.persistent = round == 2 ? 1u : 0u;
      // (the persistent workgroup starts with round 1's own repairs, list C, before it goes on to the rounds that follow)
      H.round_base = (uint32_t)(round == 2 ? 1 : round);
      // round 0 renders unrelated deep pixels again; entries per wave: see P3D_REDO_LANES above
      H.lanes = round == 0 ? redo_lanes() : (round == 1 ? round1_lanes() : kBlock);
      // (round >= 1 works through the successors of units whose leftover changed, a few thousand list entries at most: a small
      // grid with a grid-stride loop - 16 384 workgroups that find nothing take 25 us to come and go, a lone frame waits for them)
      // (round 0 over list A - scenes traversed from global memory - rarely has anything on it: 4 096 workgroups and a
      // grid-stride loop instead of one workgroup per 64 pixels that comes only to find the list empty, 16 us for a 2048x2048 frame)
      const uint32_t blocks = round == 2 ? 1u : (round == 1 ? std::min(wide, list_blocks()) : std::min<uint32_t>(wide, 4096u));
      P.level_stride = blocks * kBlock;
      P.tile_blocks = blocks;
      const uint32_t real_cap = H.list_cap;
      if (abl_skip() & 8u) H.list_cap = 0;  // (ablation: the launch happens, every workgroup finds an empty list)
      const hipError_t e = launch_literal(round == 1 ? 5 : 2, ghosts, cfg->antialiasing != 0, lds_scene, want_counts, P, blocks, lds_bytes, st);
      H.list_cap = real_cap;
      if (e != hipSuccess) return fail(P3D_ERR_NO_DEVICE, std::string("hand-off kernel launch: ") + hipGetErrorString(e));
    }
    if (want_counts) {
      hipLaunchKernelGGL(ucount_reduce_kernel, dim3(256), dim3(256), 0, st, H, (uint32_t)tile->w, s->d_stats);
      P3D_HIP(hipGetLastError());
    }
  }
  if (on_tail) {
    P3D_HIP(hipEventRecord(s->ev_tail_done, st));
    s->tail_pending = true;
  }
  if (stats) return finish_stats(s, st, stats, literal);
  return P3D_OK;
}

int p3d_scene_set_tail_stream(p3d_scene* s, void* hip_stream) {
  if (!s) return fail(P3D_ERR_INVALID, "p3d_scene_set_tail_stream: null scene");
  P3D_HIP(hipSetDevice(s->device));
  if (s->tail_pending) {  // (a frame may still be running on the old tail stream)
    P3D_HIP(hipEventSynchronize(s->ev_tail_done));
    s->tail_pending = false;
  }
  s->tail_stream = (hipStream_t)hip_stream;
  return P3D_OK;
}

int p3d_scene_join(p3d_scene* s, void* hip_stream, int host_wait) {
  if (!s) return fail(P3D_ERR_INVALID, "p3d_scene_join: null scene");
  if (!s->tail_pending) return P3D_OK;  // nothing of this scene runs anywhere but on the streams the caller gave it
  P3D_HIP(hipSetDevice(s->device));
  if (host_wait) P3D_HIP(hipEventSynchronize(s->ev_tail_done));
  else P3D_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, s->ev_tail_done, 0));
  return P3D_OK;
}

int p3d_scene_status(p3d_scene* s) {
  if (!s) return fail(P3D_ERR_INVALID, "p3d_scene_status: null argument");
  P3D_HIP(hipSetDevice(s->device));
  P3D_HIP(hipDeviceSynchronize());
  return check_status(s);
}

// csrc/p3d_debug.h (not part of include/p3d.h): refused unless the process was started with P3D_TEST_HOOKS=1
int p3d_debug_scene_limits(p3d_scene* s, const p3d_debug_limits* limits) {
  static const bool enabled = [] { const char* e = getenv("P3D_TEST_HOOKS"); return e && e[0] == '1' && e[1] == 0; }();
  if (!enabled) return fail(P3D_ERR_UNSUPPORTED, "p3d_debug_scene_limits: test hooks are off (start the process with P3D_TEST_HOOKS=1)");
  if (!s) return fail(P3D_ERR_INVALID, "p3d_debug_scene_limits: null scene");
  s->dbg = limits ? *limits : p3d_debug_limits{0, 0, 0, 0};
  return P3D_OK;
}

int p3d_render_tile(p3d_scene* s, const p3d_config* cfg, const p3d_tile* tile, float* rgb, int32_t* hit_id, uint8_t* rgb8,
                    p3d_stats* stats) {
  if (!s || !cfg || !tile) return fail(P3D_ERR_INVALID, "p3d_render_tile: null argument");
  if (tile->w <= 0 || tile->h <= 0) return fail(P3D_ERR_INVALID, "empty tile");
  P3D_HIP(hipSetDevice(s->device));
  const size_t n = (size_t)tile->w * tile->h;
  if (rgb) if (int rc = s->out_rgb.ensure(n * 3 * sizeof(float))) return rc;
  if (hit_id) if (int rc = s->out_hit.ensure(n * sizeof(int32_t))) return rc;
  if (rgb8) if (int rc = s->out_rgb8.ensure(n * 3)) return rc;
  p3d_stats local;
  int rc = p3d_render_tile_device(s, cfg, tile, rgb ? (float*)s->out_rgb.p : nullptr, hit_id ? (int32_t*)s->out_hit.p : nullptr,
                                  rgb8 ? (uint8_t*)s->out_rgb8.p : nullptr, nullptr, stats ? stats : &local);
  bool retried = false;
  if (rc == P3D_ERR_CAPACITY && cfg->handoff_records == P3D_HANDOFF_COMPACT && (s->last_status & kHoErrLeftoverCap)) {
    p3d_config dense = *cfg;  // the leftover pool was too small for this frame: once more with room for the worst case
    dense.handoff_records = P3D_HANDOFF_DENSE;
    rc = p3d_render_tile_device(s, &dense, tile, rgb ? (float*)s->out_rgb.p : nullptr, hit_id ? (int32_t*)s->out_hit.p : nullptr,
                                rgb8 ? (uint8_t*)s->out_rgb8.p : nullptr, nullptr, stats ? stats : &local);
    retried = true;
  }
  if (stats && rc == P3D_OK) stats->handoff_dense_retry = retried ? 1 : 0;  // the frame was rendered twice (and the dense records allocated)
  if (rc) return rc;
  if (rgb) P3D_HIP(hipMemcpy(rgb, s->out_rgb.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (hit_id) P3D_HIP(hipMemcpy(hit_id, s->out_hit.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (rgb8) P3D_HIP(hipMemcpy(rgb8, s->out_rgb8.p, n * 3, hipMemcpyDeviceToHost));
  return P3D_OK;  // device-detected errors were turned into a return code by finish_stats()
}

static int trace_common(p3d_scene* s, uint32_t accel, uint32_t n, const float* origin, const float* direction, int32_t* hit_id,
                        float* t_host, float* hit_point, uint8_t* occluded, bool any) {
  if (!s || !origin || !direction || (any ? !occluded : !hit_id)) return fail(P3D_ERR_INVALID, "p3d_trace: null argument");
  if (int rc = check_accel(s, accel)) return rc;
  if (accel == P3D_ACCEL_GRID && s->dev.n_objs == 0) return fail(P3D_ERR_UNSUPPORTED, "grid over an empty scene");
  if (n == 0) return P3D_OK;
  P3D_HIP(hipSetDevice(s->device));
  if (int rc = p3d_scene_join(s, nullptr, 1)) return rc;  // (the queries use the scene's scratch on the null stream)
  const size_t in_bytes = (size_t)n * 6 * sizeof(float);
  const size_t out_bytes = (size_t)n * (sizeof(int32_t) + 4 * sizeof(float) + 1) + 256;
  if (int rc = s->q_in.ensure(in_bytes)) return rc;
  if (int rc = s->q_out.ensure(out_bytes)) return rc;
  float* d_o = (float*)s->q_in.p;
  float* d_d = d_o + (size_t)n * 3;
  int32_t* d_hit = (int32_t*)s->q_out.p;
  float* d_hp = (float*)(d_hit + n);
  float* d_t = d_hp + (size_t)n * 3;
  uint8_t* d_occ = (uint8_t*)(d_t + n);
  P3D_HIP(hipMemcpy(d_o, origin, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  P3D_HIP(hipMemcpy(d_d, direction, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  const uint32_t bound = accel == P3D_ACCEL_BVH ? std::max<uint32_t>(1, s->bvh_max_depth) : 1;
  const uint32_t cap = 16;  // LDS window of the spilling stack (a power of two), the rest in the backing array
  const uint32_t blocks = (n + kBlock - 1) / kBlock;
  if ((uint64_t)(bound > cap ? bound : 0) * blocks * kBlock > 0xffffffffull)  // entries are addressed with 32-bit offsets (device_core.hpp Stack)
    return fail(P3D_ERR_CAPACITY, "p3d_trace: too many rays for one call over a tree this deep (split the batch)");
  if (int rc = s->spill.ensure(std::max<size_t>(16, (size_t)(bound > cap ? bound : 0) * blocks * kBlock * sizeof(uint2)))) return rc;
  TraceParams P{};
  P.sc = s->dev; P.n = n; P.origin = d_o; P.direction = d_d; P.hit_id = d_hit; P.hit_point = d_hp; P.occluded = d_occ;
  P.t = t_host ? d_t : nullptr; P.spill = (uint2*)s->spill.p; P.spill_stride = blocks * kBlock; P.stack_cap = (int32_t)cap;
  const size_t lds = (size_t)cap * kBlock * sizeof(uint2);
#define P3D_TRACE(A)                                                                                     \
  do {                                                                                                   \
    if (any) hipLaunchKernelGGL((trace_kernel<A, true>), dim3(blocks), dim3(kBlock), lds, 0, P);         \
    else hipLaunchKernelGGL((trace_kernel<A, false>), dim3(blocks), dim3(kBlock), lds, 0, P);            \
  } while (0)
  if (accel == P3D_ACCEL_BVH) P3D_TRACE(P3D_ACCEL_BVH);
  else if (accel == P3D_ACCEL_GRID) P3D_TRACE(P3D_ACCEL_GRID);
  else P3D_TRACE(P3D_ACCEL_NONE);
#undef P3D_TRACE
  P3D_HIP(hipGetLastError());
  P3D_HIP(hipDeviceSynchronize());
  if (any) {
    P3D_HIP(hipMemcpy(occluded, d_occ, n, hipMemcpyDeviceToHost));
  } else {
    P3D_HIP(hipMemcpy(hit_id, d_hit, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (hit_point) P3D_HIP(hipMemcpy(hit_point, d_hp, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (t_host) P3D_HIP(hipMemcpy(t_host, d_t, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  }
  return P3D_OK;
}

int p3d_trace_closest(p3d_scene* s, uint32_t accel, uint32_t n, const float* origin, const float* direction, int32_t* hit_id,
                      float* t, float* hit_point) {
  return trace_common(s, accel, n, origin, direction, hit_id, t, hit_point, nullptr, false);
}
int p3d_trace_any(p3d_scene* s, uint32_t accel, uint32_t n, const float* origin, const float* direction, uint8_t* occluded) {
  return trace_common(s, accel, n, origin, direction, nullptr, nullptr, nullptr, occluded, true);
}

static int object_query(p3d_scene* s, int what, uint32_t object, uint32_t n, const float* a, float* b, uint8_t* hit, float* t) {
  if (!s || !a || !b || (what == 0 && (!hit || !t))) return fail(P3D_ERR_INVALID, "p3d object query: null argument");
  if (what != 2 && object >= s->dev.n_objs) return fail(P3D_ERR_INVALID, "p3d object query: no such object");
  if (what == 2 && !s->has_sky) return fail(P3D_ERR_INVALID, "p3d_skybox_color: no cubemap was supplied (p3d_scene_set_skybox)");
  if (n == 0) return P3D_OK;
  P3D_HIP(hipSetDevice(s->device));
  const size_t vec = (size_t)n * 3 * sizeof(float);
  if (int rc = s->q_in.ensure(vec)) return rc;
  if (int rc = s->q_out.ensure(vec + (size_t)n * (sizeof(float) + 1) + 64)) return rc;
  float* d_a = (float*)s->q_in.p;
  float* d_b = (float*)s->q_out.p;
  float* d_t = d_b + (size_t)n * 3;
  uint8_t* d_hit = (uint8_t*)(d_t + n);
  P3D_HIP(hipMemcpy(d_a, a, vec, hipMemcpyHostToDevice));
  if (what == 0) {
    P3D_HIP(hipMemcpy(d_b, b, vec, hipMemcpyHostToDevice));
    P3D_HIP(hipMemcpy(d_t, t, (size_t)n * sizeof(float), hipMemcpyHostToDevice));  // untouched where the test fails
  }
  ObjectQueryParams Q{};
  Q.sc = s->dev; Q.object = object; Q.n = n; Q.a = d_a; Q.b = d_b; Q.hit = d_hit; Q.t = d_t;
  const uint32_t blocks = (n + kBlock - 1) / kBlock;
  if (what == 0) hipLaunchKernelGGL((object_query_kernel<0>), dim3(blocks), dim3(kBlock), 0, 0, Q);
  else if (what == 1) hipLaunchKernelGGL((object_query_kernel<1>), dim3(blocks), dim3(kBlock), 0, 0, Q);
  else hipLaunchKernelGGL((object_query_kernel<2>), dim3(blocks), dim3(kBlock), 0, 0, Q);
  P3D_HIP(hipGetLastError());
  P3D_HIP(hipDeviceSynchronize());
  P3D_HIP(hipMemcpy(b, d_b, vec, hipMemcpyDeviceToHost));
  if (what == 0) {
    P3D_HIP(hipMemcpy(hit, d_hit, n, hipMemcpyDeviceToHost));
    P3D_HIP(hipMemcpy(t, d_t, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  }
  return P3D_OK;
}
int p3d_object_intercepts(p3d_scene* s, uint32_t object, uint32_t n, const float* origin, float* direction, uint8_t* hit, float* t) {
  return object_query(s, 0, object, n, origin, direction, hit, t);
}
int p3d_object_normal(p3d_scene* s, uint32_t object, uint32_t n, const float* point, float* normal) {
  return object_query(s, 1, object, n, point, normal, nullptr, nullptr);
}
int p3d_skybox_color(p3d_scene* s, uint32_t n, const float* direction, float* rgb) {
  return object_query(s, 2, 0, n, direction, rgb, nullptr, nullptr);
}

#ifdef P3D_PT_PROFILE
int p3d_debug_set_pt_prof(void* device_ptr) {
  unsigned long long* p = static_cast<unsigned long long*>(device_ptr);
  return hipMemcpyToSymbol(HIP_SYMBOL(p3d::g_pt_prof), &p, sizeof(p)) == hipSuccess ? P3D_OK : P3D_ERR_NO_DEVICE;
}
#endif
#ifdef P3D_TIMELINE
int p3d_debug_set_timeline(void* device_ptr) {
  unsigned long long* p = static_cast<unsigned long long*>(device_ptr);
  return hipMemcpyToSymbol(HIP_SYMBOL(p3d::g_timeline), &p, sizeof(p)) == hipSuccess ? P3D_OK : P3D_ERR_NO_DEVICE;
}
int p3d_debug_copy_sched(p3d_scene* s, uint32_t* host_sched, size_t n_sched, uint32_t* host_cost, size_t n_cost) {
  if (s->sched.empty()) return P3D_ERR_INVALID;
  (void)hipDeviceSynchronize();
  if (hipMemcpy(host_sched, s->sched[0].sched.p, n_sched * 4, hipMemcpyDeviceToHost) != hipSuccess) return P3D_ERR_NO_DEVICE;
  return hipMemcpy(host_cost, s->sched[0].cost.p, n_cost * 4, hipMemcpyDeviceToHost) == hipSuccess ? P3D_OK : P3D_ERR_NO_DEVICE;
}
#endif

}  // extern "C"
