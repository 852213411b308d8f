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

// bvh.cpp:28-196: build on the host; intersect_bvh / bool_intersect_bvh (bvh.cpp:198-340) forward ONE query to the
// device scene (p3d_trace_closest / p3d_trace_any with accel = Bvh).  Every forwarded query starts on an empty
// traversal stack: the member hit_stack that outlives a query (bvh.cpp:86) exists inside a frame render only.
class BVH {
 public:
  void build(const std::vector<Object*>& objects);
  void bindDevice(::p3d_scene* dev) { dev_ = dev; }
  bool intersect_bvh(Ray ray, Object** hit_obj, Vector& hit_point);
  bool bool_intersect_bvh(Ray ray);
  // The same two queries for n rays in ONE launch (the single-ray forms above cost a launch each, about 10 us): hit[i] = the
  // return value for rays[i]; hit_obj / hit_point (may be null) are filled where hit[i].  false + p3d_last_error() if the call failed.
  bool intersect_bvh(const Ray* rays, size_t n, bool* hit, Object** hit_obj, Vector* hit_point);
  bool bool_intersect_bvh(const Ray* rays, size_t n, bool* hit);
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
  std::vector<Object*> scene_order_;  // the objects as passed to build(): hit ID -> Object*
  ::p3d_scene* dev_ = nullptr;
};

// grid.h:13-43 / grid.cpp:3-68, 211-259: build on the host; Traverse x2 (grid.cpp:71-208) forward ONE query to the
// device scene (p3d_trace_closest / p3d_trace_any with accel = UGrid; the any-hit form includes the object loop that
// follows the grid query in the reference's shadow code, main.cpp:197-217).
class Grid {
 public:
  void bindDevice(::p3d_scene* dev) { dev_ = dev; }
  bool Traverse(Ray& ray, Object** hitobject, Vector& hitpoint);
  bool Traverse(Ray& ray);
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
  ::p3d_scene* dev_ = nullptr;
};

// C++ callers that hold a p3d_host_scene (include/p3d.h) reach the classes inside it through this
struct HostClasses {
  Scene* scene;
  BVH* bvh;    // null until p3d_host_scene_desc(build_bvh = 1)
  Grid* grid;  // null until p3d_host_scene_desc(build_grid = 1)
};
}  // namespace p3d
struct p3d_host_scene;
namespace p3d {
HostClasses host_classes(::p3d_host_scene* hs);
}  // namespace p3d
