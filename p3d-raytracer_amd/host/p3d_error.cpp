#include "p3d_error.hpp"

#include "p3d.h"

namespace p3d {
namespace {
thread_local std::string g_last_error;
}
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const char* last_error_cstr() { return g_last_error.c_str(); }
}  // namespace p3d

extern "C" {
const char* p3d_last_error(void) { return p3d::last_error_cstr(); }
uint32_t p3d_abi_version(void) { return P3D_ABI_VERSION; }

// constants.h:6-45 as shipped; SKYBOX has no runtime switch (miss shading = bclr)
void p3d_config_default(p3d_config* c) {
  if (!c) return;
  *c = p3d_config{};
  c->integrator = P3D_PATHTRACE;   // PATHTRACING true
  c->accel = P3D_ACCEL_BVH;        // acl_str = Bvh
  c->max_depth = 20;               // MAX_DEPTH
  c->spp_sqrt = 20;                // SPP
  c->antialiasing = 1;             // ANTIALIASING true
  c->depth_of_field = 1;           // DEPTH_OF_FIELD true
  c->sample_disk = 1;              // SAMPLE_DISK true
  c->soft_shadows = 0;             // SOFT_SHADOWS false
  c->sample_mode = P3D_SAMPLE_JITTER;
  c->light_side = .5f;             // LIGHT_SIDE
  c->gamma = 1.0f;                 // GAMMA
  c->collect_stats = 0;
  c->skybox = 0;                   // SKYBOX is true as shipped, but the cubemap must be supplied first
  c->tile_order = P3D_TILE_ORDER_COST;  // scheduling only
  c->seed = 0x5EED;
  c->stack_mode = P3D_STACK_LITERAL;    // one hit_stack for the whole frame, as the reference's BVH member (bvh.cpp:86)
}
}
