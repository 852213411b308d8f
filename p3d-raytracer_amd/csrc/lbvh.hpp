// lbvh.hpp — optional BVH construction ON the device (second half).
//
// The reference-exact tree (bvh.cpp:89-196, host/accel_build.cpp) is the default: its shape is
// observable through the any-hit quirk Q1.  This builder is the alternative for callers who only
// need correct closest hits and do not want to wait for a host build: a linear BVH —
//   1. Morton code of every object's bounding-box centre inside the scene bounds (30 bits), made
//      unique by appending the object index: 64-bit key
//   2. radix sort of the keys (hipCUB)
//   3. Karras' "Maximizing parallelism in the construction of BVHs" hierarchy: internal node i
//      covers the key range that shares the longest common prefix around position i
//   4. bottom-up box refit (second child to arrive continues to the parent)
//   5. emission in the layout the traversal kernels read (device_core.hpp): node 0 = root, the two
//      children of internal node i at 1 + 2 i and 2 + 2 i, one or two objects per leaf, leaf
//      geometry gathered in leaf order
// Object boxes are the GetBoundingBox() values of p3d_prim (planes: the [-1,1]^3 default, Q12).
#pragma once

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "device_core.hpp"

namespace p3d {
namespace lbvh {

constexpr int kThreads = 256;

// order-preserving float <-> uint map for atomicMin/atomicMax
__device__ __forceinline__ uint32_t f2o(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// boxes: [2 * i] = {bmin, -}, [2 * i + 1] = {bmax, -}.  bounds[0..2] = min of centres, [3..5] = max (ordered uints)
__global__ void centre_bounds(const float4* boxes, uint32_t n, uint32_t* bounds) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float4 lo = boxes[2 * i], hi = boxes[2 * i + 1];
  const float c[3] = {(lo.x + hi.x) * 0.5f, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f};
  for (int k = 0; k < 3; ++k) {
    atomicMin(&bounds[k], f2o(c[k]));
    atomicMax(&bounds[3 + k], f2o(c[k]));
  }
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ void morton_keys(const float4* boxes, uint32_t n, const uint32_t* bounds, unsigned long long* keys) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float4 lo = boxes[2 * i], hi = boxes[2 * i + 1];
  const float c[3] = {(lo.x + hi.x) * 0.5f, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    const float mn = o2f(bounds[k]), mx = o2f(bounds[3 + k]);
    const float ext = mx - mn;
    float u = ext > 0 ? (c[k] - mn) / ext : 0.0f;
    u = !(u > 0.0f) ? 0.0f : (u > 1.0f ? 1.0f : u);  // NaN -> 0
    const float s = u * 1024.0f;
    q[k] = s >= 1023.0f ? 1023u : (uint32_t)s;
  }
  const uint32_t m = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
  keys[i] = ((unsigned long long)m << 32) | i;
}

// Karras 2012.  Internal nodes 0 .. n-2, leaves 0 .. n-1 (in sorted order).  A child reference is
// (index << 1) | is_leaf.  parent[] is indexed by that same reference value.
__device__ __forceinline__ int common_prefix(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return __clzll((long long)(keys[i] ^ keys[j]));  // keys are unique: never 64
}

__global__ void build_internal(const unsigned long long* keys, uint32_t n, uint2* children, uint32_t* parent) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  const int N = (int)n;
  if (i >= N - 1) return;
  const int d = common_prefix(keys, N, i, i + 1) > common_prefix(keys, N, i, i - 1) ? 1 : -1;
  const int dmin = common_prefix(keys, N, i, i - d);
  int lmax = 2;
  while (common_prefix(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (common_prefix(keys, N, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = common_prefix(keys, N, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (common_prefix(keys, N, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const uint32_t left = lo == gamma ? ((uint32_t)gamma << 1) | 1u : ((uint32_t)gamma << 1);
  const uint32_t right = hi == gamma + 1 ? ((uint32_t)(gamma + 1) << 1) | 1u : ((uint32_t)(gamma + 1) << 1);
  children[i] = make_uint2(left, right);
  parent[left] = (uint32_t)i;
  parent[right] = (uint32_t)i;
  if (i == 0) parent[0] = 0xffffffffu;  // internal node 0 is the root (reference value 0 << 1)
}

// a box another workgroup has just written (behind its __threadfence): read past this CU's L1
__device__ __forceinline__ float4 fresh(const float4* p) {
  const float* f = reinterpret_cast<const float*>(p);
  return make_float4(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.f);
}

// node_box: [2 * ref] / [2 * ref + 1] = min / max of the node with that reference value
__global__ void refit(const unsigned long long* keys, const float4* boxes, uint32_t n, const uint2* children,
                      const uint32_t* parent, uint32_t* visits, float4* node_box, uint32_t* leaf_depth_max) {
  const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  const uint32_t obj = (uint32_t)(keys[k] & 0xffffffffull);
  uint32_t ref = (k << 1) | 1u;
  node_box[2 * ref] = boxes[2 * obj];
  node_box[2 * ref + 1] = boxes[2 * obj + 1];
  uint32_t depth = 1;
  if (n == 1) { atomicMax(leaf_depth_max, depth); return; }
  uint32_t p = parent[ref];
  while (true) {
    __threadfence();  // this thread's box is visible before it announces itself
    if (atomicAdd(&visits[p], 1u) == 0) break;  // first child to arrive: the sibling will do the parent
    __threadfence();
    const uint2 ch = children[p];
    const float4 a0 = fresh(node_box + 2 * ch.x), a1 = fresh(node_box + 2 * ch.x + 1), b0 = fresh(node_box + 2 * ch.y),
                 b1 = fresh(node_box + 2 * ch.y + 1);
    ref = p << 1;
    node_box[2 * ref] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), 0.f);
    node_box[2 * ref + 1] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), 0.f);
    if (p == 0) break;
    p = parent[ref];
  }
  // depth of this leaf = number of nodes on its path from the root (bvh_max_depth convention)
  for (uint32_t r = (k << 1) | 1u; parent[r] != 0xffffffffu; r = parent[r] << 1) ++depth;
  atomicMax(leaf_depth_max, depth + 0u);
}

// An internal node whose two children are both leaves is emitted as ONE leaf of two objects (their
// slots are adjacent: the split lies between them), as the reference's trees have (Threshold 2,
// bvh.cpp:83): half the leaf-level box tests for one more object test; its own pair of child
// records stays unused.
__device__ __forceinline__ void emit_record(float4* nodes, uint32_t at, const float4* node_box, const uint2* children, uint32_t ref) {
  uint32_t desc;
  if (ref & 1u) {
    desc = kDescLeaf | (1u << 28) | (ref >> 1);
  } else {
    const uint2 ch = children[ref >> 1];
    desc = (ch.x & ch.y & 1u) ? (kDescLeaf | (2u << 28) | (ch.x >> 1)) : (1u + 2u * (ref >> 1));
  }
  const float4 lo = node_box[2 * ref], hi = node_box[2 * ref + 1];
  nodes[2 * at] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(desc));
  nodes[2 * at + 1] = make_float4(hi.x, hi.y, hi.z, 0.f);
}

__global__ void emit(const unsigned long long* keys, uint32_t n, const uint2* children, const float4* node_box,
                     const float4* ogeom, float4* nodes, float4* bgeom) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) {  // leaf slot i holds the object with the i-th smallest key
    const uint32_t obj = (uint32_t)(keys[i] & 0xffffffffull);
    for (int q = 0; q < 3; ++q) bgeom[3 * i + q] = ogeom[3 * obj + q];
  }
  if (n == 1) {
    if (i == 0) emit_record(nodes, 0, node_box, children, 1u);  // the root is the only leaf
    return;
  }
  if (i < n - 1) {
    const uint2 ch = children[i];
    emit_record(nodes, 1 + 2 * i, node_box, children, ch.x);
    emit_record(nodes, 2 + 2 * i, node_box, children, ch.y);
    if (i == 0) emit_record(nodes, 0, node_box, children, 0u);
  }
}

struct Result {
  uint32_t n_nodes = 0, max_depth = 0;
  float build_ms = 0;
};

#define P3D_LBVH_HIP(call)                 \
  do {                                     \
    const hipError_t e_ = (call);          \
    if (e_ != hipSuccess) return e_;       \
  } while (0)

// d_boxes: 2 n float4 (GetBoundingBox of every object).  d_nodes: room for 2 (2 n - 1) float4, d_bgeom: 3 n float4.
inline hipError_t build(const float4* d_boxes, const float4* d_ogeom, uint32_t n, float4* d_nodes, float4* d_bgeom, Result* out) {
  *out = Result{};
  if (n == 0) return hipSuccess;
  const uint32_t blocks = (n + kThreads - 1) / kThreads;
  uint32_t* d_bounds = nullptr;
  unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
  uint2* d_children = nullptr;
  uint32_t *d_parent = nullptr, *d_visits = nullptr, *d_depth = nullptr;
  float4* d_node_box = nullptr;
  void* d_temp = nullptr;
  size_t temp_bytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t rc = hipSuccess;
  auto body = [&]() -> hipError_t {
    P3D_LBVH_HIP(hipMalloc((void**)&d_bounds, 6 * sizeof(uint32_t)));
    P3D_LBVH_HIP(hipMalloc((void**)&d_keys, (size_t)n * 8));
    P3D_LBVH_HIP(hipMalloc((void**)&d_sorted, (size_t)n * 8));
    P3D_LBVH_HIP(hipMalloc((void**)&d_children, (size_t)n * sizeof(uint2)));
    P3D_LBVH_HIP(hipMalloc((void**)&d_parent, (size_t)2 * n * sizeof(uint32_t)));
    P3D_LBVH_HIP(hipMalloc((void**)&d_visits, (size_t)n * sizeof(uint32_t)));
    P3D_LBVH_HIP(hipMalloc((void**)&d_depth, sizeof(uint32_t)));
    P3D_LBVH_HIP(hipMalloc((void**)&d_node_box, (size_t)4 * n * sizeof(float4)));
    P3D_LBVH_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_sorted, (int)n));
    P3D_LBVH_HIP(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16));
    P3D_LBVH_HIP(hipEventCreate(&e0));
    P3D_LBVH_HIP(hipEventCreate(&e1));
    const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    P3D_LBVH_HIP(hipMemcpy(d_bounds, init, sizeof(init), hipMemcpyHostToDevice));
    P3D_LBVH_HIP(hipEventRecord(e0, 0));
    P3D_LBVH_HIP(hipMemsetAsync(d_visits, 0, (size_t)n * sizeof(uint32_t), 0));
    P3D_LBVH_HIP(hipMemsetAsync(d_depth, 0, sizeof(uint32_t), 0));
    hipLaunchKernelGGL(centre_bounds, dim3(blocks), dim3(kThreads), 0, 0, d_boxes, n, d_bounds);
    hipLaunchKernelGGL(morton_keys, dim3(blocks), dim3(kThreads), 0, 0, d_boxes, n, d_bounds, d_keys);
    P3D_LBVH_HIP(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_sorted, (int)n));
    if (n > 1) hipLaunchKernelGGL(build_internal, dim3(blocks), dim3(kThreads), 0, 0, d_sorted, n, d_children, d_parent);
    hipLaunchKernelGGL(refit, dim3(blocks), dim3(kThreads), 0, 0, d_sorted, d_boxes, n, d_children, d_parent, d_visits, d_node_box, d_depth);
    hipLaunchKernelGGL(emit, dim3(blocks), dim3(kThreads), 0, 0, d_sorted, n, d_children, d_node_box, d_ogeom, d_nodes, d_bgeom);
    P3D_LBVH_HIP(hipGetLastError());
    P3D_LBVH_HIP(hipEventRecord(e1, 0));
    P3D_LBVH_HIP(hipEventSynchronize(e1));
    P3D_LBVH_HIP(hipEventElapsedTime(&out->build_ms, e0, e1));
    P3D_LBVH_HIP(hipMemcpy(&out->max_depth, d_depth, sizeof(uint32_t), hipMemcpyDeviceToHost));
    out->n_nodes = 2 * n - 1;
    return hipSuccess;
  };
  rc = body();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  for (void* p : {(void*)d_bounds, (void*)d_keys, (void*)d_sorted, (void*)d_children, (void*)d_parent, (void*)d_visits,
                  (void*)d_depth, (void*)d_node_box, d_temp})
    if (p) (void)hipFree(p);
  return rc;
}

}  // namespace lbvh
}  // namespace p3d
