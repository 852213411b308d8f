// accel_build.cpp — BVH and uniform-grid construction (host).
#include "accel_build.hpp"

#include <algorithm>
#include <cmath>

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
void BVH::build(const std::vector<Object*>& objects) {
  objs_ = objects;
  const size_t n = objs_.size();
  order_.resize(n);
  nodes_.clear();
  nodes_.reserve(2 * n + 1);
  max_depth_ = 0;

  AABB all = AABB::empty();
  for (size_t i = 0; i < n; ++i) {
    all.extend(objs_[i]->GetBoundingBox());
    order_[i] = static_cast<uint32_t>(i);
  }
  p3d_bvh_node root{};
  put3(root.bmin, all.min);
  put3(root.bmax, all.max);
  nodes_.push_back(root);
  split(0, static_cast<int>(n), 0, 1);
}

void BVH::split(int first, int last, uint32_t node, uint32_t level) {
  if (level > max_depth_) max_depth_ = level;
  const int count = last - first;
  if (count <= Threshold) {
    nodes_[node].index = static_cast<uint32_t>(first);
    nodes_[node].count_leaf = P3D_BVH_LEAF | static_cast<uint32_t>(count);
    return;
  }
  const Vector lo(nodes_[node].bmin[0], nodes_[node].bmin[1], nodes_[node].bmin[2]);
  const Vector hi(nodes_[node].bmax[0], nodes_[node].bmax[1], nodes_[node].bmax[2]);
  const Vector extent = hi - lo;
  int axis = 2;
  if (extent.x >= extent.y && extent.x >= extent.z) axis = 0;
  else if (extent.y >= extent.x && extent.y >= extent.z) axis = 1;

  auto centre_of_box = [&](uint32_t id) {
    const AABB b = objs_[id]->GetBoundingBox();
    return (b.max.getIndex(axis) + b.min.getIndex(axis)) * 0.5f;
  };
  std::sort(order_.begin() + first, order_.begin() + last,
            [&](uint32_t a, uint32_t b) { return centre_of_box(a) < centre_of_box(b); });

  auto centroid = [&](int slot) { return objs_[order_[slot]]->getCentroid().getIndex(axis); };
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
  for (int s = first; s < mid; ++s) left.extend(objs_[order_[s]]->GetBoundingBox());
  for (int s = mid; s < last; ++s) right.extend(objs_[order_[s]]->GetBoundingBox());

  const uint32_t child = static_cast<uint32_t>(nodes_.size());
  nodes_[node].index = child;
  nodes_[node].count_leaf = 0;
  p3d_bvh_node l{}, r{};
  put3(l.bmin, left.min); put3(l.bmax, left.max);
  put3(r.bmin, right.min); put3(r.bmax, right.max);
  nodes_.push_back(l);
  nodes_.push_back(r);
  split(first, mid, child, level + 1);
  split(mid, last, child + 1, level + 1);
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
