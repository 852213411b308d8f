// accel_build.cpp — BVH and uniform-grid construction (host).
#include "accel_build.hpp"

#include "p3d_error.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <thread>

namespace p3d {

namespace {
inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }  // maths.h:46-49
inline void put3(float dst[3], const Vector& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
}  // namespace

// ---------------------------------------------------------------------------
// BVH — follows bvh.cpp:89-196.
//   * root box = union of all object boxes (bvh.cpp:95-104)
//   * a range of <= Threshold objects becomes a leaf (bvh.cpp:115-117)
//   * axis = longest extent of the NODE box, ties x >= y >= z (bvh.cpp:124-128)
//   * the range is std::sort-ed by bbox centre on that axis (bvh.cpp:30-42,134)
//   * split coordinate = box middle; if that leaves one side empty, the mean of the
//     centroids; if still empty, first+Threshold (bvh.cpp:136-161)
//   * both children are appended next to each other, left subtree built first
//     (bvh.cpp:185-194)
// Bounding boxes / centroids are cached per object: the values are what the virtual
// calls of the reference return, computed once.
// ---------------------------------------------------------------------------
// bvh.cpp:198-276 / 278-340 as single queries on the device scene
bool BVH::intersect_bvh(Ray ray, Object** hit_obj, Vector& hit_point) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "BVH::intersect_bvh: no device scene bound (BVH::bindDevice)"); return false; }
  const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
  int32_t id = -1;
  float hp[3] = {0, 0, 0};
  if (p3d_trace_closest(dev_, P3D_ACCEL_BVH, 1, o, d, &id, nullptr, hp) != P3D_OK || id < 0) return false;
  if (hit_obj && (size_t)id < scene_order_.size()) *hit_obj = scene_order_[id];
  hit_point = Vector(hp[0], hp[1], hp[2]);
  return true;
}
bool BVH::bool_intersect_bvh(Ray ray) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "BVH::bool_intersect_bvh: no device scene bound (BVH::bindDevice)"); return false; }
  const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
  uint8_t occ = 0;
  return p3d_trace_any(dev_, P3D_ACCEL_BVH, 1, o, d, &occ) == P3D_OK && occ != 0;
}

// ... and as batches: n rays, one launch (p3d_trace_closest / p3d_trace_any take them as they are)
namespace {
void pack_rays(const Ray* rays, size_t n, std::vector<float>& o, std::vector<float>& d) {
  o.resize(3 * n);
  d.resize(3 * n);
  for (size_t i = 0; i < n; ++i) {
    o[3 * i] = rays[i].origin.x; o[3 * i + 1] = rays[i].origin.y; o[3 * i + 2] = rays[i].origin.z;
    d[3 * i] = rays[i].direction.x; d[3 * i + 1] = rays[i].direction.y; d[3 * i + 2] = rays[i].direction.z;
  }
}
}  // namespace
bool BVH::intersect_bvh(const Ray* rays, size_t n, bool* hit, Object** hit_obj, Vector* hit_point) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "BVH::intersect_bvh: no device scene bound (BVH::bindDevice)"); return false; }
  if (n == 0) return true;
  if (!rays || !hit || n > 0xffffffffull) { fail(P3D_ERR_INVALID, "BVH::intersect_bvh: bad batch"); return false; }
  std::vector<float> o, d, hp(3 * n);
  std::vector<int32_t> id(n, -1);
  pack_rays(rays, n, o, d);
  if (p3d_trace_closest(dev_, P3D_ACCEL_BVH, (uint32_t)n, o.data(), d.data(), id.data(), nullptr, hp.data()) != P3D_OK) return false;
  for (size_t i = 0; i < n; ++i) {
    hit[i] = id[i] >= 0;
    if (!hit[i]) continue;
    if (hit_obj && (size_t)id[i] < scene_order_.size()) hit_obj[i] = scene_order_[id[i]];
    if (hit_point) hit_point[i] = Vector(hp[3 * i], hp[3 * i + 1], hp[3 * i + 2]);
  }
  return true;
}
bool BVH::bool_intersect_bvh(const Ray* rays, size_t n, bool* hit) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "BVH::bool_intersect_bvh: no device scene bound (BVH::bindDevice)"); return false; }
  if (n == 0) return true;
  if (!rays || !hit || n > 0xffffffffull) { fail(P3D_ERR_INVALID, "BVH::bool_intersect_bvh: bad batch"); return false; }
  std::vector<float> o, d;
  std::vector<uint8_t> occ(n, 0);
  pack_rays(rays, n, o, d);
  if (p3d_trace_any(dev_, P3D_ACCEL_BVH, (uint32_t)n, o.data(), d.data(), occ.data()) != P3D_OK) return false;
  for (size_t i = 0; i < n; ++i) hit[i] = occ[i] != 0;
  return true;
}

// grid.cpp:71-151 / 154-208 as single queries on the device scene
bool Grid::Traverse(Ray& ray, Object** hitobject, Vector& hitpoint) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "Grid::Traverse: no device scene bound (Grid::bindDevice)"); return false; }
  const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
  int32_t id = -1;
  float hp[3] = {0, 0, 0};
  if (p3d_trace_closest(dev_, P3D_ACCEL_GRID, 1, o, d, &id, nullptr, hp) != P3D_OK || id < 0) return false;
  if (hitobject && (size_t)id < objects_.size()) *hitobject = objects_[id];
  hitpoint = Vector(hp[0], hp[1], hp[2]);
  return true;
}
bool Grid::Traverse(Ray& ray) {
  if (!dev_) { fail(P3D_ERR_NO_DEVICE, "Grid::Traverse: no device scene bound (Grid::bindDevice)"); return false; }
  const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
  uint8_t occ = 0;
  return p3d_trace_any(dev_, P3D_ACCEL_GRID, 1, o, d, &occ) == P3D_OK && occ != 0;
}

void BVH::build(const std::vector<Object*>& objects) {
  objs_ = objects;
  scene_order_ = objects;
  const size_t n = objs_.size();
  order_.resize(n);
  nodes_.clear();
  nodes_.reserve(2 * n + 1);
  max_depth_ = 0;
  box_.resize(n);
  centroid_.resize(n);
  keys_.resize(n);

  AABB all = AABB::empty();
  for (size_t i = 0; i < n; ++i) {
    box_[i] = objs_[i]->GetBoundingBox();
    centroid_[i] = objs_[i]->getCentroid();
    all.extend(box_[i]);
    order_[i] = static_cast<uint32_t>(i);
  }
  p3d_bvh_node root{};
  put3(root.bmin, all.min);
  put3(root.bmax, all.max);
  nodes_.push_back(root);

  const unsigned hw = std::thread::hardware_concurrency();
  Top top{root, 0, static_cast<int>(n), 1, nullptr, nullptr, {}, 0};
  // Subtrees own disjoint ranges of order_/keys_, so they are built side by side (each split
  // hands its left half to a new thread while there is fork budget); only the node numbering
  // is sequential (children are appended when their parent splits, the whole left subtree
  // before the right one, bvh.cpp:185-194) and is done afterwards by emit_top().
  int forks = 0;
  if (n >= kParallelMin && hw > 1)
    while ((2u << forks) <= std::min(hw, 32u)) ++forks;
  grow_top(top, forks);
  emit_top(top, 0);
  box_.clear(); box_.shrink_to_fit();
  centroid_.clear(); centroid_.shrink_to_fit();
  keys_.clear(); keys_.shrink_to_fit();
}

// One split decision (bvh.cpp:119-161): sorts order_[first,last) and returns the cut slot.
int BVH::partition(const p3d_bvh_node& box, int first, int last, p3d_bvh_node& l, p3d_bvh_node& r) {
  const int count = last - first;
  const Vector lo(box.bmin[0], box.bmin[1], box.bmin[2]);
  const Vector hi(box.bmax[0], box.bmax[1], box.bmax[2]);
  const Vector extent = hi - lo;
  int axis = 2;
  if (extent.x >= extent.y && extent.x >= extent.z) axis = 0;
  else if (extent.y >= extent.x && extent.y >= extent.z) axis = 1;

  // The comparator's value for every object of the range, next to its id: std::sort sees the
  // same sequence of comparison outcomes as when it sorts the objects themselves, so the
  // permutation (ties included) is the same.
  for (int s = first; s < last; ++s) {
    const uint32_t id = order_[s];
    keys_[s].key = (box_[id].max.getIndex(axis) + box_[id].min.getIndex(axis)) * 0.5f;
    keys_[s].id = id;
  }
  std::sort(keys_.begin() + first, keys_.begin() + last, [](const SortKey& a, const SortKey& b) { return a.key < b.key; });
  for (int s = first; s < last; ++s) order_[s] = keys_[s].id;

  auto centroid = [&](int slot) { return centroid_[order_[slot]].getIndex(axis); };
  auto one_side_empty = [&](float c) { return centroid(first) > c || centroid(last - 1) <= c; };

  float cut = static_cast<float>(static_cast<double>(hi.getIndex(axis) + lo.getIndex(axis)) * 0.5);
  if (one_side_empty(cut)) {
    cut = 0;
    for (int s = first; s < last; ++s) cut += centroid(s);
    cut /= count;
  }
  int mid;
  if (one_side_empty(cut)) {
    mid = first + Threshold;
  } else {
    mid = first;
    while (mid < last && !(centroid(mid) > cut)) ++mid;
  }

  AABB left = AABB::empty(), right = AABB::empty();
  for (int s = first; s < mid; ++s) left.extend(box_[order_[s]]);
  for (int s = mid; s < last; ++s) right.extend(box_[order_[s]]);
  l = p3d_bvh_node{};
  r = p3d_bvh_node{};
  put3(l.bmin, left.min); put3(l.bmax, left.max);
  put3(r.bmin, right.min); put3(r.bmax, right.max);
  return mid;
}

// Recursive build of one (sub)tree into `nodes` (indices local to that vector); returns its
// deepest level.
uint32_t BVH::split(std::vector<p3d_bvh_node>& nodes, int first, int last, uint32_t node, uint32_t level) {
  const int count = last - first;
  if (count <= Threshold) {
    nodes[node].index = static_cast<uint32_t>(first);
    nodes[node].count_leaf = P3D_BVH_LEAF | static_cast<uint32_t>(count);
    return level;
  }
  p3d_bvh_node l, r;
  const int mid = partition(nodes[node], first, last, l, r);
  const uint32_t child = static_cast<uint32_t>(nodes.size());
  nodes[node].index = child;
  nodes[node].count_leaf = 0;
  nodes.push_back(l);
  nodes.push_back(r);
  const uint32_t dl = split(nodes, first, mid, child, level + 1);
  const uint32_t dr = split(nodes, mid, last, child + 1, level + 1);
  return dl > dr ? dl : dr;
}

// Split `t` and fork while `forks` > 0; below that, build the subtree into t.sub.
void BVH::grow_top(Top& t, int forks) {
  if (forks <= 0 || t.last - t.first <= static_cast<int>(kParallelMin / 4)) {
    t.sub.reserve(2 * static_cast<size_t>(t.last - t.first) + 1);
    t.sub.push_back(t.box);
    t.depth = split(t.sub, t.first, t.last, 0, t.level);
    return;
  }
  p3d_bvh_node l, r;
  const int mid = partition(t.box, t.first, t.last, l, r);
  t.left.reset(new Top{l, t.first, mid, t.level + 1, nullptr, nullptr, {}, 0});
  t.right.reset(new Top{r, mid, t.last, t.level + 1, nullptr, nullptr, {}, 0});
  std::thread other([&] { grow_top(*t.left, forks - 1); });
  grow_top(*t.right, forks - 1);
  other.join();
}

// Number the nodes as the recursive build does.
void BVH::emit_top(Top& t, uint32_t node) {
  if (t.left) {
    const uint32_t child = static_cast<uint32_t>(nodes_.size());
    nodes_[node].index = child;
    nodes_[node].count_leaf = 0;
    nodes_.push_back(t.left->box);
    nodes_.push_back(t.right->box);
    emit_top(*t.left, child);
    emit_top(*t.right, child + 1);
    return;
  }
  if (t.depth > max_depth_) max_depth_ = t.depth;
  const uint32_t shift = static_cast<uint32_t>(nodes_.size()) - 1;  // sub[1] lands at nodes_.size()
  auto place = [&](p3d_bvh_node nd) {
    if (!(nd.count_leaf & P3D_BVH_LEAF)) nd.index += shift;
    return nd;
  };
  nodes_[node] = place(t.sub[0]);
  for (size_t i = 1; i < t.sub.size(); ++i) nodes_.push_back(place(t.sub[i]));
  t.sub.clear();
  t.sub.shrink_to_fit();
}

// ---------------------------------------------------------------------------
// Grid — follows grid.cpp:3-68 and the bounds of grid.cpp:211-259.
//   * bounds = union of object boxes grown by 1e-4
//   * powf(n / volume, 1/3): `1/3` is integer 0, so the density factor is 1 and the
//     cell counts depend on the extents only: n_axis = trunc(m * w_axis) + 1   (Q11)
//   * each object goes into every cell its box overlaps; cell lists keep object order
// ---------------------------------------------------------------------------
void Grid::Build() {
  const float kEps = 0.0001f;
  Vector p0(FLT_MAX, FLT_MAX, FLT_MAX), p1(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (Object* o : objects_) {
    const AABB b = o->GetBoundingBox();
    if (b.min.x < p0.x) p0.x = b.min.x;
    if (b.min.y < p0.y) p0.y = b.min.y;
    if (b.min.z < p0.z) p0.z = b.min.z;
    if (b.max.x > p1.x) p1.x = b.max.x;
    if (b.max.y > p1.y) p1.y = b.max.y;
    if (b.max.z > p1.z) p1.z = b.max.z;
  }
  p0.x -= kEps; p0.y -= kEps; p0.z -= kEps;
  p1.x += kEps; p1.y += kEps; p1.z += kEps;
  bbox = AABB(p0, p1);

  const Vector w = p1 - p0;
  const int num_obj = getNumObjects();
  const float s = powf(num_obj / (w.x * w.y * w.z), 1 / 3);  // == 1, kept for NaN/inf fidelity
  nx = static_cast<int>(truncf(m * w.x * s) + 1);
  ny = static_cast<int>(truncf(m * w.y * s) + 1);
  nz = static_cast<int>(truncf(m * w.z * s) + 1);

  const size_t n_cells = static_cast<size_t>(nx) * ny * nz;
  struct Span { int x0, x1, y0, y1, z0, z1; };
  std::vector<Span> spans(num_obj);
  std::vector<uint32_t> fill(n_cells + 1, 0);
  for (int j = 0; j < num_obj; ++j) {
    const AABB b = objects_[j]->GetBoundingBox();
    Span sp;
    sp.x0 = static_cast<int>(clampd((b.min.x - p0.x) * nx / (p1.x - p0.x), 0, nx - 1));
    sp.y0 = static_cast<int>(clampd((b.min.y - p0.y) * ny / (p1.y - p0.y), 0, ny - 1));
    sp.z0 = static_cast<int>(clampd((b.min.z - p0.z) * nz / (p1.z - p0.z), 0, nz - 1));
    sp.x1 = static_cast<int>(clampd((b.max.x - p0.x) * nx / (p1.x - p0.x), 0, nx - 1));
    sp.y1 = static_cast<int>(clampd((b.max.y - p0.y) * ny / (p1.y - p0.y), 0, ny - 1));
    sp.z1 = static_cast<int>(clampd((b.max.z - p0.z) * nz / (p1.z - p0.z), 0, nz - 1));
    spans[j] = sp;
    for (int iz = sp.z0; iz <= sp.z1; ++iz)
      for (int iy = sp.y0; iy <= sp.y1; ++iy)
        for (int ix = sp.x0; ix <= sp.x1; ++ix) ++fill[static_cast<size_t>(ix) + nx * iy + nx * ny * iz + 1];
  }
  // counting sort into CSR; visiting objects in index order keeps each cell's list in
  // insertion order, as push_back does in the reference
  cell_start_.assign(n_cells + 1, 0);
  for (size_t c = 0; c < n_cells; ++c) cell_start_[c + 1] = cell_start_[c] + fill[c + 1];
  cell_items_.assign(cell_start_[n_cells], 0);
  std::vector<uint32_t> cursor(cell_start_.begin(), cell_start_.end() - 1);
  for (int j = 0; j < num_obj; ++j) {
    const Span& sp = spans[j];
    for (int iz = sp.z0; iz <= sp.z1; ++iz)
      for (int iy = sp.y0; iy <= sp.y1; ++iy)
        for (int ix = sp.x0; ix <= sp.x1; ++ix)
          cell_items_[cursor[static_cast<size_t>(ix) + nx * iy + nx * ny * iz]++] = static_cast<uint32_t>(j);
  }
}

p3d_grid_desc Grid::describe() const {
  p3d_grid_desc d{};
  put3(d.bmin, bbox.min);
  put3(d.bmax, bbox.max);
  d.nx = nx; d.ny = ny; d.nz = nz;
  d.n_cells = static_cast<uint32_t>(cell_start_.empty() ? 0 : cell_start_.size() - 1);
  d.n_items = static_cast<uint32_t>(cell_items_.size());
  d.cell_start = cell_start_.data();
  d.cell_items = cell_items_.data();
  return d;
}

}  // namespace p3d
