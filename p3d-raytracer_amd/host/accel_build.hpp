// accel_build.hpp — host-side construction of the two acceleration structures.
//
// The reference builds both on the host at the top of renderScene() (main.cpp:701-720)
// and so do we; only TRAVERSAL moved to the device.  The tree / cell layout must be the
// reference's exactly, because traversal order is observable through its any-hit quirks
// (SURVEY.md §8(a) Q1, Q10): same split rule, same std::sort (libstdc++ introsort, ties
// resolved the same way), children adjacent in push order.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "p3d.h"
#include "scene_model.hpp"

namespace p3d {

// bvh.cpp:28-196 (build only; intersect_bvh / bool_intersect_bvh run on the device)
class BVH {
 public:
  void build(const std::vector<Object*>& objects);
  const std::vector<p3d_bvh_node>& flatNodes() const { return nodes_; }
  const std::vector<uint32_t>& primOrder() const { return order_; }  // permuted objs (bvh.cpp:84)
  uint32_t maxDepth() const { return max_depth_; }

 private:
  struct SortKey { float key; uint32_t id; };
  struct Top {  // a node of the part of the tree that is split before the subtree builds start
    p3d_bvh_node box;
    int first, last;
    uint32_t level;
    std::unique_ptr<Top> left, right;  // both null: a subtree, built into `sub`
    std::vector<p3d_bvh_node> sub;     // its nodes, indices local
    uint32_t depth;
  };
  int partition(const p3d_bvh_node& box, int first, int last, p3d_bvh_node& l, p3d_bvh_node& r);
  uint32_t split(std::vector<p3d_bvh_node>& nodes, int first, int last, uint32_t node, uint32_t level);
  void grow_top(Top& t, int forks);
  void emit_top(Top& t, uint32_t node);
  static constexpr int Threshold = 2;  // bvh.cpp:83
  static constexpr size_t kParallelMin = 8192;   // below this the build is a single recursion
  std::vector<Object*> objs_;
  std::vector<uint32_t> order_;
  std::vector<AABB> box_;          // build-time caches of the virtual calls, dropped after build()
  std::vector<Vector> centroid_;
  std::vector<SortKey> keys_;
  std::vector<p3d_bvh_node> nodes_;
  uint32_t max_depth_ = 0;
};

// grid.h:13-43 / grid.cpp:3-68, 211-259 (build only)
class Grid {
 public:
  int getNumObjects() const { return static_cast<int>(objects_.size()); }
  void addObject(Object* o) { objects_.push_back(o); }
  Object* getObject(unsigned i) const { return objects_.at(i); }
  void Build();
  // CSR form of vector<vector<Object*>> cells
  const std::vector<uint32_t>& cellStart() const { return cell_start_; }
  const std::vector<uint32_t>& cellItems() const { return cell_items_; }
  p3d_grid_desc describe() const;

 private:
  std::vector<Object*> objects_;
  std::vector<uint32_t> cell_start_, cell_items_;
  int nx = 0, ny = 0, nz = 0;
  float m = 2.0f;  // grid.h:33
  AABB bbox;
};

}  // namespace p3d
