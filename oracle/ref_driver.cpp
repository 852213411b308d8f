/*
 * ref_driver.cpp — thin extern "C" driver around the pieces of the REFERENCE that
 * build here as they lie (compiled from /root/reference/Raytracing by
 * oracle/Makefile into oracle/_ref/libp3dref.so; no reference source is copied
 * into this repo).  TEST INFRASTRUCTURE: used only to pin the oracle's L0 /
 * camera / sampler restatement (tests/golden/make_ref_vectors.py).
 *
 * Covered reference code: vector.cpp (all of it), sampler.cpp:5-11,
 * camera.h:34-115, ray.h:16-18, maths.h:26-92, color.h:39-44.
 * NOT coverable: anything including scene.h (needs <IL/il.h>, absent).
 */
#include <cstdint>
#include <cstdlib>

#include "camera.h"  // reference header (via -I): pulls vector.h, ray.h, sampler.h
#include "color.h"
#include "maths.h"

namespace {
// Counter-based rand() interposition is NOT used: sample_unit_disk is driven
// through libc srand()/rand() and the oracle's rng_mode 1 consumes the same stream.
Camera* g_cam = nullptr;
}

extern "C" {

void ref_vec_normalize(float* v) {
  Vector a(v[0], v[1], v[2]);
  a.normalize();
  v[0] = a.x; v[1] = a.y; v[2] = a.z;
}
float ref_vec_length(const float* v) { return Vector(v[0], v[1], v[2]).length(); }
float ref_vec_dot(const float* a, const float* b) {
  return Vector(a[0], a[1], a[2]) * Vector(b[0], b[1], b[2]);
}
void ref_vec_cross(const float* a, const float* b, float* out) {
  Vector r = Vector(a[0], a[1], a[2]) % Vector(b[0], b[1], b[2]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_vec_div(const float* a, float f, float* out) {
  Vector r = Vector(a[0], a[1], a[2]) / f;
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
// ray.h:16-18 applied k times to the same Ray (shows the in-place mutation)
void ref_ray_get_direction(float* d, int k) {
  Ray r(Vector(0, 0, 0), Vector(d[0], d[1], d[2]));
  for (int i = 0; i < k; i++) r.getDirection();
  d[0] = r.direction.x; d[1] = r.direction.y; d[2] = r.direction.z;
}
void ref_camera_create(const float* from, const float* at, const float* up, float angle,
                       float hither, float yon, int rx, int ry, float aperture_ratio,
                       float focal_ratio) {
  delete g_cam;
  g_cam = new Camera(Vector(from[0], from[1], from[2]), Vector(at[0], at[1], at[2]),
                     Vector(up[0], up[1], up[2]), angle, hither, yon, rx, ry, aperture_ratio,
                     focal_ratio);
}
void ref_camera_primary(float px, float py, float* o, float* d) {
  Vector p(px, py, 0);
  Ray r = g_cam->PrimaryRay(p);
  o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
  d[0] = r.direction.x; d[1] = r.direction.y; d[2] = r.direction.z;
}
void ref_camera_primary_lens(float lx, float ly, float px, float py, float* o, float* d) {
  Vector l(lx, ly, 0), p(px, py, 0);
  Ray r = g_cam->PrimaryRay(l, p);
  o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
  d[0] = r.direction.x; d[1] = r.direction.y; d[2] = r.direction.z;
}
float ref_camera_aperture() { return g_cam->GetAperture(); }
float ref_camera_plane_dist() { return g_cam->GetPlaneDist(); }
void ref_srand(unsigned s) { set_rand_seed((int)s); }
float ref_rand_float() { return rand_float(); }
void ref_sample_unit_disk(float* out) {
  Vector p = sample_unit_disk();
  out[0] = p.x; out[1] = p.y;
}
unsigned char ref_u8fromfloat(float x) { return u8fromfloat(x); }
float ref_u8tofloat(unsigned char x) { return u8tofloat(x); }
double ref_clamp(double x, double lo, double hi) { return clamp(x, lo, hi); }
void ref_color_clamp(float* c) {
  Color k = Color(c[0], c[1], c[2]).clamp();
  c[0] = k.r(); c[1] = k.g(); c[2] = k.b();
}
}
