/*
 * p3d_oracle.cpp — CPU oracle for the per-pixel ray-trace hot path of
 * fmbnicola/P3D-RayTracer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg load it.  The
 * product (p3d-raytracer_amd/) shares no code with it.
 *
 * What it is: a from-scratch, scalar, recursive restatement of the reference's
 * algorithm, written by reading the reference; every function cites the
 * reference lines it follows (paths relative to /root/reference/Raytracing/).
 * It keeps the reference's observable quirks: the
 * in-place re-normalising Ray::getDirection, the any-hit BVH "pop everything"
 * loop, the member hit_stack that leaks between queries, integer-division
 * Kr = 0, powf(x, 1/3) = 1 grid sizing, mixed float/double arithmetic, ...
 *
 * PINNING STATUS:
 *   - Vector / Camera / sampler / maths primitives: pinned against the compiled
 *     reference (oracle/_ref: vector.cpp, sampler.cpp, camera.h, maths.h,
 *     color.h build as they lie).
 *   - Shapes, AABB, BVH, Grid, rayTracing, Radiance: the reference cannot be
 *     built here under this round's rules (scene.h:6 needs DevIL's <IL/il.h>,
 *     absent; no stand-ins allowed) and the reference has no tests or golden
 *     vectors -> PARITY UNPINNED by an executable reference.  Secondary
 *     evidence: tests/golden/survey_probe/ holds crops + checksums of images the
 *     survey stage produced from the reference's sources in this container
 *     (provenance in tests/golden/README.md); the oracle
 *     reproduces them (bit-for-bit for the Whitted configs).
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off (no -ffast-math, no -march=native);
 * float expressions evaluate in float (FLT_EVAL_METHOD 0), double literals
 * promote exactly where the reference's do.
 */
#include "p3d_oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// L0 math: vector.h / vector.cpp / color.h / ray.h
// ---------------------------------------------------------------------------
struct V3 {
  float x = 0, y = 0, z = 0;
};
inline V3 v3(float x, float y, float z) {
  V3 r;
  r.x = x; r.y = y; r.z = z;
  return r;
}
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }  // vector.cpp:38
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }  // vector.cpp:44
inline V3 operator*(V3 a, float f) { return v3(a.x * f, a.y * f, a.z * f); }      // vector.cpp:50
inline V3 operator/(V3 a, float f) { return v3(a.x / f, a.y / f, a.z / f); }      // vector.cpp:60
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // vector.cpp:55
inline V3 cross(V3 u, V3 v) {                                                     // vector.cpp:84-99
  return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }    // vector.cpp:10-13
inline float sqrd_length(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }      // vector.cpp:16-19
// vector.cpp:65-70: l = 1.0 / length() is a DOUBLE division narrowed to float; mutates.
inline V3& normalize(V3& a) {
  float l = (float)(1.0 / (double)length(a));
  a.x *= l; a.y *= l; a.z *= l;
  return a;
}
inline float get_index(V3 a, int op) { return op == 0 ? a.x : (op == 1 ? a.y : a.z); }  // vector.h:20-22

struct C3 {
  float r = 0, g = 0, b = 0;
};
inline C3 c3(float r, float g, float b) {
  C3 c;
  c.r = r; c.g = g; c.b = b;
  return c;
}
inline C3 operator*(C3 a, float c) { return c3(a.r * c, a.g * c, a.b * c); }        // color.h:48
inline C3 operator/(C3 a, float c) { return c3(a.r / c, a.g / c, a.b / c); }        // color.h:51
inline C3 operator+(C3 a, C3 b) { return c3(a.r + b.r, a.g + b.g, a.b + b.b); }     // color.h:60
inline C3 operator*(C3 a, C3 b) { return c3(a.r * b.r, a.g * b.g, a.b * b.b); }     // color.h:62
inline float clamp01(float v) {  // color.h:10 CLAMP(0.0, v, 1.0) with double literals
  return (float)(((double)v < 0.0) ? 0.0 : (((double)v > 1.0) ? 1.0 : (double)v));
}
inline C3 clamp(C3 a) { return c3(clamp01(a.r), clamp01(a.g), clamp01(a.b)); }      // color.h:39-44

// ray.h:6-19.  direction is mutable state: getDirection() re-normalises IN PLACE (Q8).
struct Ray {
  V3 o, d;
};
inline V3 get_direction(Ray& r) { return normalize(r.d); }  // ray.h:16-18

// scene.h:17-22 (argument-repeating macros; NaN-order sensitive, keep the shape)
#define ORC_MIN3(a, b, c) ((a) < (b) ? ((a) < (c) ? (a) : (c)) : ((b) < (c) ? (b) : (c)))
#define ORC_MAX3(a, b, c) ((a) > (b) ? ((a) > (c) ? (a) : (c)) : ((b) > (c) ? (b) : (c)))

const float kPI = 3.141592653589793238462f;  // camera.h:13 (a FLOAT literal)
const float kEPS = 0.0001f;                  // scene.h:31

// maths.h:31-49 — double min/max/clamp
inline double dmax(double a, double b) { return a > b ? a : b; }
inline double dclamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
// Cell index as grid.cpp:37-55 / 343-351 compute it: (int)clamp(v, 0, n-1).  clamp() passes a NaN
// through (axis-parallel ray lying in a slab plane: 0 * inf) and the reference then indexes
// cells.at(INT_MIN...) and dies with std::out_of_range.  There is no behaviour to reproduce, so the
// oracle and the kernels both map a NaN to cell 0 (what v_cvt_i32_f32 gives) and go on.
inline int cell_coord(double v, int n) {
  const double c = dclamp(v, 0, n - 1);
  if (!(c >= 0)) return 0;
  return (int)c;
}

// ---------------------------------------------------------------------------
// detmath: sin/cos built only from IEEE double + - * and floor, so that this
// oracle and the HIP kernels produce identical bits.  The reference calls libm
// (cosf/sinf on float arguments main.cpp:400, cos/sin on double main.cpp:436);
// libm results differ between platforms by an ulp, so bit parity with any one
// libm is not a property the reference has.  tests/test_oracle_units.py checks
// these against libm (<= 1 ulp after rounding to float).
// Cody-Waite reduction by pi/2 with the classic fdlibm split constants and
// minimax polynomials on [-pi/4, pi/4].
// ---------------------------------------------------------------------------
void det_sincos(double x, double* s_out, double* c_out) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;  // 33 bits of pi/2
  const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
  double kd = floor(x * two_over_pi + 0.5);
  double r = (x - kd * pio2_hi) - kd * pio2_lo;
  double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double sp = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  double s = r + (z * r) * (S1 + z * sp);
  double cp = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double c = 1.0 - (0.5 * z - z * cp);
  long long k = (long long)kd;
  switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
  }
}
inline double det_sin(double x) { double s, c; det_sincos(x, &s, &c); return s; }
inline double det_cos(double x) { double s, c; det_sincos(x, &s, &c); return c; }

// ---------------------------------------------------------------------------
// RNG.  The reference draws from libc rand() seeded with time()^2
// (maths.h:67-70, main.cpp:75-77,722): serial and unrepeatable.  rng_mode 1 keeps
// that (libc rand in program order).  rng_mode 0 gives every (pixel, sample) its
// own PCG-XSH-RR 64/32 stream; draws are consumed in the same program order.
// ---------------------------------------------------------------------------
inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Rng {
  int mode = 0;
  uint64_t state = 0, inc = 1;
  void seed_stream(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint64_t k = mix64(seed ^ mix64(((uint64_t)pixel << 32) | (uint64_t)sample));
    inc = (mix64(k) << 1) | 1ull;
    state = k * 6364136223846793005ull + inc;
  }
  uint32_t next31() {  // a value in [0, 2^31-1] like glibc rand()
    if (mode == 1) return (uint32_t)rand();
    uint64_t old = state;
    state = old * 6364136223846793005ull + inc;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    uint32_t out = (xs >> rot) | (xs << ((0u - rot) & 31u));
    return out >> 1;
  }
  // maths.h:67-70 with glibc's RAND_MAX = 2^31-1: (float)RAND_MAX rounds to 2^31.
  float rand_float() { return (float)next31() / (float)2147483647; }
  // main.cpp:75-77 (the seed argument is ignored there too)
  double erand48() { return (double)next31() / (double)2147483647; }
};

// ---------------------------------------------------------------------------
// L1 scene model: scene.h / scene.cpp / boundingBox.cpp / camera.h
// ---------------------------------------------------------------------------
struct Box {
  V3 mn, mx;
};
inline Box default_box() {  // boundingBox.cpp:6-10
  Box b;
  b.mn = v3(-1.0f, -1.0f, -1.0f);
  b.mx = v3(1.0f, 1.0f, 1.0f);
  return b;
}
inline void extend(Box& a, const Box& b) {  // boundingBox.cpp:104-112
  if (a.mn.x > b.mn.x) a.mn.x = b.mn.x;
  if (a.mn.y > b.mn.y) a.mn.y = b.mn.y;
  if (a.mn.z > b.mn.z) a.mn.z = b.mn.z;
  if (a.mx.x < b.mx.x) a.mx.x = b.mx.x;
  if (a.mx.y < b.mx.y) a.mx.y = b.mx.y;
  if (a.mx.z < b.mx.z) a.mx.z = b.mx.z;
}
inline bool is_inside(const Box& b, V3 p) {  // boundingBox.cpp:39-42 (strict)
  return (p.x > b.mn.x && p.x < b.mx.x) && (p.y > b.mn.y && p.y < b.mx.y) &&
         (p.z > b.mn.z && p.z < b.mx.z);
}
inline V3 box_centroid(const Box& b) { return (b.mn + b.mx) / 2; }  // boundingBox.cpp:100-102

struct Stats {
  uint64_t rays_primary = 0, rays_shadow = 0, rays_reflect = 0, rays_refract = 0, rays_bounce = 0,
           rays_light = 0;
  uint64_t node_tests = 0, sphere_tests = 0, tri_tests = 0, box_tests = 0, plane_tests = 0,
           shaded_hits = 0, pixels = 0, max_stack = 0, ref_ray_counter = 0;
  void add(const Stats& o) {
    rays_primary += o.rays_primary; rays_shadow += o.rays_shadow; rays_reflect += o.rays_reflect;
    rays_refract += o.rays_refract; rays_bounce += o.rays_bounce; rays_light += o.rays_light;
    node_tests += o.node_tests; sphere_tests += o.sphere_tests; tri_tests += o.tri_tests;
    box_tests += o.box_tests; plane_tests += o.plane_tests; shaded_hits += o.shaded_hits;
    pixels += o.pixels; ref_ray_counter += o.ref_ray_counter;
    if (o.max_stack > max_stack) max_stack = o.max_stack;
  }
};

// boundingBox.cpp:44-98 — slab test on the RAW direction (no normalisation).
bool aabb_intercepts(const Box& b, const Ray& ray, float& t) {
  float o_x = ray.o.x, o_y = ray.o.y, o_z = ray.o.z;
  float d_x = ray.d.x, d_y = ray.d.y, d_z = ray.d.z;
  float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
  float a = 1.0f / d_x;
  if (a >= 0) { tx_min = (b.mn.x - o_x) * a; tx_max = (b.mx.x - o_x) * a; }
  else        { tx_min = (b.mx.x - o_x) * a; tx_max = (b.mn.x - o_x) * a; }
  float bb = 1.0f / d_y;
  if (bb >= 0) { ty_min = (b.mn.y - o_y) * bb; ty_max = (b.mx.y - o_y) * bb; }
  else         { ty_min = (b.mx.y - o_y) * bb; ty_max = (b.mn.y - o_y) * bb; }
  float c = 1.0f / d_z;
  if (c >= 0) { tz_min = (b.mn.z - o_z) * c; tz_max = (b.mx.z - o_z) * c; }
  else        { tz_min = (b.mx.z - o_z) * c; tz_max = (b.mn.z - o_z) * c; }
  float t0 = ORC_MAX3(tx_min, ty_min, tz_min);  // largest entering t
  float t1 = ORC_MIN3(tx_max, ty_max, tz_max);  // smallest exiting t
  t = (t0 < 0) ? t1 : t0;
  return (t0 < t1 && (double)t1 > 0.0001);  // 0.0001 is a double literal (boundingBox.cpp:97)
}

enum { K_SPHERE = 0, K_TRI = 1, K_BOX = 2, K_PLANE = 3 };

struct Material {  // scene.h:34-71
  C3 cd; float kd; C3 cs; float ks; float shine; float T; float ior; C3 em; float refl;
};
struct Light {  // scene.h:73-81
  V3 pos; C3 col;
};
struct Object {
  int kind = 0;
  int mat = -1;
  // triangle: points, normal, Min/Max (scene.h:130-132)
  V3 p0, p1, p2, normal, tmin, tmax;
  // sphere (scene.h:159-160)
  V3 center; float radius = 0;
  // aaBox (scene.h:174-175)
  V3 bmin, bmax;
  // plane (scene.h:104-105)
  V3 PN, A;
};

Object make_triangle(V3 P0, V3 P1, V3 P2) {  // scene.cpp:12-35
  Object o;
  o.kind = K_TRI;
  o.p0 = P0; o.p1 = P1; o.p2 = P2;
  o.normal = cross(P1 - P0, P2 - P0);
  normalize(o.normal);
  float x0 = std::min(std::min(P0.x, P1.x), P2.x), y0 = std::min(std::min(P0.y, P1.y), P2.y),
        z0 = std::min(std::min(P0.z, P1.z), P2.z);
  float x1 = std::max(std::max(P0.x, P1.x), P2.x), y1 = std::max(std::max(P0.y, P1.y), P2.y),
        z1 = std::max(std::max(P0.z, P1.z), P2.z);
  o.tmin = v3(x0 - kEPS, y0 - kEPS, z0 - kEPS);  // Min -= EPSILON
  o.tmax = v3(x1 + kEPS, y1 + kEPS, z1 + kEPS);  // Max += EPSILON
  return o;
}
Object make_plane(V3 P0, V3 P1, V3 P2) {  // scene.cpp:102-113
  Object o;
  o.kind = K_PLANE;
  V3 a = P2 - P1, b = P0 - P1;
  o.PN = cross(a, b);
  normalize(o.PN);
  o.A = P0;
  return o;
}
Object make_sphere(V3 c, float r) {  // scene.h:142-144
  Object o;
  o.kind = K_SPHERE;
  o.center = c; o.radius = r;
  return o;
}
Object make_box(V3 mn, V3 mx) {  // scene.cpp:205-209
  Object o;
  o.kind = K_BOX;
  o.bmin = mn; o.bmax = mx;
  return o;
}

Box bounding_box(const Object& o) {
  Box b;
  switch (o.kind) {
    case K_SPHERE:  // scene.cpp:194-198
      b.mn = o.center - v3(o.radius, o.radius, o.radius);
      b.mx = o.center + v3(o.radius, o.radius, o.radius);
      return b;
    case K_TRI: b.mn = o.tmin; b.mx = o.tmax; return b;  // scene.cpp:37-39
    case K_BOX: b.mn = o.bmin; b.mx = o.bmax; return b;  // scene.cpp:211-213
    default: return default_box();                        // scene.h:114 (Q12)
  }
}
V3 centroid(const Object& o) {
  switch (o.kind) {
    case K_SPHERE: return o.center;                          // scene.h:147-149
    case K_TRI: return box_centroid(bounding_box(o));        // scene.h:123-125
    case K_BOX: return (o.bmax + o.bmin) / 2;                // scene.cpp:269-271
    default: return v3(0, 0, 0);                             // scene.h:112
  }
}

// Object::intercepts — takes the ray BY REFERENCE; Sphere mutates its direction (Q8).
bool intercepts(const Object& ob, Ray& ray, float& time, Stats& st) {
  switch (ob.kind) {
    case K_TRI: {  // scene.cpp:47-94
      st.tri_tests++;
      V3 P0 = ob.p0, P1 = ob.p1, P2 = ob.p2;
      float a = P0.x - P1.x, b = P0.x - P2.x, c = ray.d.x, d = P0.x - ray.o.x;
      float e = P0.y - P1.y, f = P0.y - P2.y, g = ray.d.y, h = P0.y - ray.o.y;
      float i = P0.z - P1.z, j = P0.z - P2.z, k = ray.d.z, l = P0.z - ray.o.z;
      float m = f * k - g * j, n = h * k - g * l, p = f * l - h * j;
      float q = g * i - e * k, s = e * j - f * i;
      float inv_denom = (float)(1.0 / (double)(a * m + b * q + c * s));
      float e1 = d * m - b * n - c * p;
      float beta = e1 * inv_denom;
      if ((double)beta < 0.0) return false;
      float r = e * l - h * i;
      float e2 = a * n + d * q + c * r;
      float gamma = e2 * inv_denom;
      if ((double)gamma < 0.0) return false;
      if ((double)(beta + gamma) > 1.0) return false;
      float e3 = a * p - b * r + d * s;
      float t = e3 * inv_denom;
      if ((double)t < 0.0001) return false;
      time = t;  // NaN t falls through every rejection and reports a hit (A10)
      return true;
    }
    case K_PLANE: {  // scene.cpp:116-137
      st.plane_tests++;
      float numer = dot(ray.o - ob.A, ob.PN);
      float divid = dot(ob.PN, ray.d);
      if ((double)fabsf(divid) < 0.0001) return false;
      time = -(numer / divid);
      if (time <= 0) return false;
      return true;
    }
    case K_SPHERE: {  // scene.cpp:149-186
      st.sphere_tests++;
      V3 Rd = get_direction(ray);  // normalises ray.d in place
      V3 co = ob.center - ray.o;
      float doc2 = sqrd_length(co);
      float b = dot(co, Rd);
      float c = doc2 - ob.radius * ob.radius;
      if (c > 0) {
        if (b < 0) return false;
      }
      float discriminant = (b * b - c);
      if (discriminant < 0) return false;
      if (c > 0) time = b - sqrtf(discriminant);
      else time = b + sqrtf(discriminant);
      return true;
    }
    default: {  // scene.cpp:215-227 -> AABB::intercepts
      st.box_tests++;
      Box bb;
      bb.mn = ob.bmin; bb.mx = ob.bmax;
      return aabb_intercepts(bb, ray, time);
    }
  }
}

V3 get_normal(const Object& ob, V3 point) {
  switch (ob.kind) {
    case K_TRI: return ob.normal;  // scene.cpp:41-44
    case K_PLANE: return ob.PN;    // scene.cpp:139-142
    case K_SPHERE: {               // scene.cpp:188-192
      V3 n = point - ob.center;
      return normalize(n);
    }
    default: {  // scene.cpp:229-267
      V3 center = (ob.bmax + ob.bmin) / 2;
      V3 co = point - center;
      int dir;
      if (fabsf(co.x) > fabsf(co.y)) dir = 0; else dir = 1;
      if (dir == 0 && fabsf(co.z) > fabsf(co.x)) dir = 2;
      else if (dir == 1 && fabsf(co.z) > fabsf(co.y)) dir = 2;
      switch (dir) {
        case 0: return co.x >= 0 ? v3(1, 0, 0) : v3(-1, 0, 0);
        case 1: return co.y >= 0 ? v3(0, 1, 0) : v3(0, -1, 0);
        default: return co.z >= 0 ? v3(0, 0, 1) : v3(0, 0, -1);
      }
    }
  }
}

struct Camera {  // camera.h:15-116
  V3 eye, at, up, u, v, n;
  float fovy = 0, vnear = 0, vfar = 0, plane_dist = 0, focal_ratio = 0, aperture = 0, w = 0, h = 0;
  float aperture_ratio = 0;
  int res_x = 0, res_y = 0;
  void init(V3 from, V3 At, V3 Up, float angle, float hither, float yon, int ResX, int ResY,
            float Aperture_ratio, float Focal_ratio) {  // camera.h:34-63
    eye = from; at = At; up = Up; fovy = angle; vnear = hither; vfar = yon;
    res_x = ResX; res_y = ResY; focal_ratio = Focal_ratio; aperture_ratio = Aperture_ratio;
    n = eye - at;
    plane_dist = length(n);
    n = n / plane_dist;
    u = cross(up, n);
    u = u / length(u);
    v = cross(n, u);
    h = 2 * plane_dist * tanf((kPI * angle / 180) / 2.0f);
    w = ((float)res_x / res_y) * h;
    aperture = Aperture_ratio * (w / res_x);
  }
  Ray primary(V3 ps_in) const {  // camera.h:65-82
    V3 ps;
    ps.x = w * (ps_in.x / res_x - 0.5f);
    ps.y = h * (ps_in.y / res_y - 0.5f);
    ps.z = -plane_dist;
    V3 dir = (u * ps.x + v * ps.y) + n * ps.z;
    normalize(dir);
    Ray r;
    r.o = eye; r.d = dir;
    return r;
  }
  Ray primary_lens(V3 lens, V3 ps_in) const {  // camera.h:84-115
    V3 ps;
    ps.x = w * (ps_in.x / res_x - 0.5f);
    ps.y = h * (ps_in.y / res_y - 0.5f);
    ps.z = -plane_dist;
    V3 ls = v3(lens.x * aperture, lens.y * aperture, 0);
    float px = ps.x * focal_ratio, py = ps.y * focal_ratio;
    V3 dir = (u * (px - ls.x) + v * (py - ls.y)) + n * -(focal_ratio * plane_dist);
    normalize(dir);
    Ray r;
    r.o = (eye + (u * ls.x)) + (v * ls.y);
    r.d = dir;
    return r;
  }
};

// ---------------------------------------------------------------------------
// L2: BVH (bvh.cpp) and Grid (grid.cpp / grid.h)
// ---------------------------------------------------------------------------
struct BvhNode {  // bvh.cpp:44-74
  Box bbox;
  bool leaf = false;
  unsigned n_objs = 0, index = 0;
};
struct StackItem {  // bvh.cpp:76-81
  unsigned node;
  float t;
};

struct Scene;

struct Bvh {
  const Scene* sc = nullptr;
  std::vector<int> objs;  // bvh.cpp:84 (permuted object indices)
  std::vector<BvhNode> nodes;
  bool built = false;
  void build(const Scene& s);
  void build_recursive(int left, int right, unsigned node);
};

struct Grid {
  const Scene* sc = nullptr;
  std::vector<std::vector<int>> cells;
  int nx = 0, ny = 0, nz = 0;
  float m = 2.0f;  // grid.h:33
  Box bbox;
  bool built = false;
  void build(const Scene& s);
};

struct Scene {
  std::vector<Object> objects;
  std::vector<Material> materials;
  std::vector<Light> lights;
  Camera cam;
  bool has_cam = false;
  // the `v` block as read, so that the camera can be rebuilt (resolution / lens edits)
  V3 v_from, v_at, v_up;
  float v_angle = 0, v_hither = 0, v_aperture = 0, v_focal = 0;
  C3 bg;
  bool skybox_flag = false;  // scene.cpp:610 (never read by the hot path, Q15)
  // skybox_img[6] of scene.h:218-223: decoded bytes, row 0 = BOTTOM row (IL_ORIGIN_LOWER_LEFT,
  // scene.cpp:344-345), order RIGHT, LEFT, TOP, BOTTOM, FRONT, BACK (scene.h:28)
  struct Face { std::vector<uint8_t> img; unsigned resX = 0, resY = 0, BPP = 3; } sky[6];
  bool sky_loaded = false;
  Bvh bvh;
  Grid grid;
};

void Bvh::build(const Scene& s) {  // bvh.cpp:89-111
  sc = &s;
  nodes.clear();
  objs.clear();
  BvhNode root;
  root.bbox.mn = v3(FLT_MAX, FLT_MAX, FLT_MAX);
  root.bbox.mx = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (int i = 0; i < (int)s.objects.size(); i++) {
    extend(root.bbox, bounding_box(s.objects[i]));
    objs.push_back(i);
  }
  root.leaf = false; root.index = 0; root.n_objs = 0;
  nodes.push_back(root);
  build_recursive(0, (int)objs.size(), 0);
  built = true;
}

void Bvh::build_recursive(int left_index, int right_index, unsigned node_id) {  // bvh.cpp:113-196
  const int Threshold = 2;  // bvh.cpp:83
  const std::vector<Object>& O = sc->objects;
  if ((right_index - left_index) <= Threshold) {
    nodes[node_id].leaf = true;
    nodes[node_id].index = (unsigned)left_index;
    nodes[node_id].n_objs = (unsigned)(right_index - left_index);
    return;
  }
  Box node_bb = nodes[node_id].bbox;
  int op;
  V3 len = node_bb.mx - node_bb.mn;
  if (len.x >= len.y && len.x >= len.z) op = 0;
  else if (len.y >= len.x && len.y >= len.z) op = 1;
  else op = 2;
  // bvh.cpp:30-42,134: std::sort (libstdc++ introsort, unstable) on the bbox centre
  std::sort(objs.begin() + left_index, objs.begin() + right_index, [&](int a, int b) {
    Box box = bounding_box(O[a]);
    float ca = (get_index(box.mx, op) + get_index(box.mn, op)) * 0.5f;
    box = bounding_box(O[b]);
    float cb = (get_index(box.mx, op) + get_index(box.mn, op)) * 0.5f;
    return ca < cb;
  });
  float mid_coord = (float)((double)(get_index(node_bb.mx, op) + get_index(node_bb.mn, op)) * 0.5);
  int i;
  // no objects on one side -> use the mean of the centroids (bvh.cpp:141-148)
  if (get_index(centroid(O[objs[left_index]]), op) > mid_coord ||
      get_index(centroid(O[objs[right_index - 1]]), op) <= mid_coord) {
    mid_coord = 0;
    for (i = left_index; i < right_index; i++) mid_coord += get_index(centroid(O[objs[i]]), op);
    mid_coord /= (right_index - left_index);
  }
  if (get_index(centroid(O[objs[left_index]]), op) > mid_coord ||
      get_index(centroid(O[objs[right_index - 1]]), op) <= mid_coord) {
    i = left_index + Threshold;  // bvh.cpp:150-154
  } else {
    for (i = left_index; i < right_index; i++)
      if (get_index(centroid(O[objs[i]]), op) > mid_coord) break;
  }
  Box left_bbox, right_bbox;
  left_bbox.mn = right_bbox.mn = v3(FLT_MAX, FLT_MAX, FLT_MAX);
  left_bbox.mx = right_bbox.mx = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (int j = left_index; j < i; j++) extend(left_bbox, bounding_box(O[objs[j]]));
  for (int j = i; j < right_index; j++) extend(right_bbox, bounding_box(O[objs[j]]));
  BvhNode ln, rn;
  ln.bbox = left_bbox;
  rn.bbox = right_bbox;
  unsigned first = (unsigned)nodes.size();
  nodes[node_id].leaf = false;
  nodes[node_id].index = first;
  nodes[node_id].n_objs = 0;
  nodes.push_back(ln);  // children adjacent (bvh.cpp:185-191)
  nodes.push_back(rn);
  build_recursive(left_index, i, first);
  build_recursive(i, right_index, first + 1);
}

void Grid::build(const Scene& s) {  // grid.cpp:3-68, bounds grid.cpp:211-259
  sc = &s;
  cells.clear();
  float kEpsilon = 0.0001f;
  V3 p0 = v3(FLT_MAX, FLT_MAX, FLT_MAX), p1 = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
  for (const Object& o : s.objects) {
    Box b = bounding_box(o);
    if (b.mn.x < p0.x) p0.x = b.mn.x;
    if (b.mn.y < p0.y) p0.y = b.mn.y;
    if (b.mn.z < p0.z) p0.z = b.mn.z;
    if (b.mx.x > p1.x) p1.x = b.mx.x;
    if (b.mx.y > p1.y) p1.y = b.mx.y;
    if (b.mx.z > p1.z) p1.z = b.mx.z;
  }
  p0.x -= kEpsilon; p0.y -= kEpsilon; p0.z -= kEpsilon;
  p1.x += kEpsilon; p1.y += kEpsilon; p1.z += kEpsilon;
  bbox.mn = p0; bbox.mx = p1;
  V3 w = p1 - p0;
  int num_obj = (int)s.objects.size();
  float sfac = powf(num_obj / (w.x * w.y * w.z), 1 / 3);  // 1/3 == 0 -> s == 1 (Q11)
  nx = (int)(truncf(m * w.x * sfac) + 1);
  ny = (int)(truncf(m * w.y * sfac) + 1);
  nz = (int)(truncf(m * w.z * sfac) + 1);
  int cell_num = nx * ny * nz;
  cells.resize(cell_num);
  for (int j = 0; j < num_obj; j++) {
    Box ob = bounding_box(s.objects[j]);
    int ixmin = cell_coord((ob.mn.x - p0.x) * nx / (p1.x - p0.x), nx);
    int iymin = cell_coord((ob.mn.y - p0.y) * ny / (p1.y - p0.y), ny);
    int izmin = cell_coord((ob.mn.z - p0.z) * nz / (p1.z - p0.z), nz);
    int ixmax = cell_coord((ob.mx.x - p0.x) * nx / (p1.x - p0.x), nx);
    int iymax = cell_coord((ob.mx.y - p0.y) * ny / (p1.y - p0.y), ny);
    int izmax = cell_coord((ob.mx.z - p0.z) * nz / (p1.z - p0.z), nz);
    for (int iz = izmin; iz <= izmax; iz++)
      for (int iy = iymin; iy <= iymax; iy++)
        for (int ix = ixmin; ix <= ixmax; ix++) cells.at(ix + nx * iy + nx * ny * iz).push_back(j);
  }
  built = true;
}

// ---------------------------------------------------------------------------
// Per-thread tracing context: the BVH member stack (bvh.cpp:86), RNG, counters.
// ---------------------------------------------------------------------------
// One per thread.  Its counters are bumped at every node and primitive test: each Ctx gets cache lines of its own (two
// threads' counters on one line would bounce it between their cores for the whole render).
struct alignas(128) Ctx {
  const Scene* sc;
  orc_config cfg;
  std::vector<StackItem> hit_stack;
  Rng rng;
  Stats st;
  void note_stack() {
    if (hit_stack.size() > st.max_stack) st.max_stack = hit_stack.size();
  }
};

// bvh.cpp:198-276 — closest hit; the ray is a BY-VALUE copy.
bool intersect_bvh(Ctx& cx, Ray ray, int* hit_obj, V3& hit_point, float* t_out = nullptr) {
  const Bvh& B = cx.sc->bvh;
  const std::vector<Object>& O = cx.sc->objects;
  float tmp, tmin = FLT_MAX;
  bool hit = false;
  unsigned cur = 0;
  cx.st.node_tests++;
  if (!aabb_intercepts(B.nodes[0].bbox, ray, tmp)) return false;
  while (true) {
    const BvhNode& node = B.nodes[cur];
    if (!node.leaf) {
      unsigned l = node.index, r = node.index + 1;
      float l_t, r_t;
      cx.st.node_tests += 2;
      bool l_hit = aabb_intercepts(B.nodes[l].bbox, ray, l_t);
      bool r_hit = aabb_intercepts(B.nodes[r].bbox, ray, r_t);
      if (is_inside(B.nodes[l].bbox, ray.o)) l_t = 0;
      if (is_inside(B.nodes[r].bbox, ray.o)) r_t = 0;
      if (l_hit && r_hit) {
        if (l_t < r_t) { cur = l; cx.hit_stack.push_back({r, r_t}); }
        else           { cur = r; cx.hit_stack.push_back({l, l_t}); }
        cx.note_stack();
        continue;
      } else if (l_hit) { cur = l; continue; }
      else if (r_hit)   { cur = r; continue; }
    } else {
      float curr_t;
      for (unsigned i = node.index; i < node.index + node.n_objs; i++) {
        int obj = B.objs[i];
        if (intercepts(O[obj], ray, curr_t, cx.st) && curr_t < tmin) {
          tmin = curr_t;
          *hit_obj = obj;
          hit = true;
        }
      }
    }
    bool changed = false;
    while (!cx.hit_stack.empty()) {
      StackItem popped = cx.hit_stack.back();
      cx.hit_stack.pop_back();
      if (popped.t < tmin) { cur = popped.node; changed = true; break; }
    }
    if (changed) continue;
    if (cx.hit_stack.empty()) {
      if (hit) hit_point = ray.d * tmin + ray.o;  // bvh.cpp:271 (the copy's current direction)
      if (t_out) *t_out = tmin;
      return hit;
    }
  }
}

// bvh.cpp:278-340 — any hit.  Q1: the pop loop has no break, so it empties the
// stack and resumes at the BOTTOM entry.  Q2: `return true` leaves entries behind.
bool bool_intersect_bvh(Ctx& cx, Ray ray) {
  const Bvh& B = cx.sc->bvh;
  const std::vector<Object>& O = cx.sc->objects;
  float tmp;
  unsigned cur = 0;
  cx.st.node_tests++;
  if (!aabb_intercepts(B.nodes[0].bbox, ray, tmp)) return false;
  while (true) {
    const BvhNode& node = B.nodes[cur];
    if (!node.leaf) {
      unsigned l = node.index, r = node.index + 1;
      float l_t, r_t;
      cx.st.node_tests += 2;
      bool l_hit = aabb_intercepts(B.nodes[l].bbox, ray, l_t);
      bool r_hit = aabb_intercepts(B.nodes[r].bbox, ray, r_t);
      if (l_hit && r_hit) {
        if (l_t < r_t) { cur = l; cx.hit_stack.push_back({r, r_t}); }
        else           { cur = r; cx.hit_stack.push_back({l, l_t}); }
        cx.note_stack();
        continue;
      } else if (l_hit) { cur = l; continue; }
      else if (r_hit)   { cur = r; continue; }
    } else {
      float curr_t;
      for (unsigned i = node.index; i < node.index + node.n_objs; i++)
        if (intercepts(O[B.objs[i]], ray, curr_t, cx.st)) return true;
    }
    bool changed = false;
    while (!cx.hit_stack.empty()) {
      StackItem popped = cx.hit_stack.back();
      cx.hit_stack.pop_back();
      cur = popped.node;
      changed = true;
    }
    if (changed) continue;
    if (cx.hit_stack.empty()) return false;
  }
}

struct GridWalk {
  int ix, iy, iz, ix_step, iy_step, iz_step, ix_stop, iy_stop, iz_stop;
  double dtx, dty, dtz, tx_next, ty_next, tz_next;
};

// grid.cpp:261-370
bool grid_init_traverse(const Grid& G, Ray& ray, GridWalk& k) {
  V3 o = ray.o, dir = ray.d;
  V3 bmn = G.bbox.mn, bmx = G.bbox.mx;
  int nx = G.nx, ny = G.ny, nz = G.nz;
  float tx_min = (bmn.x - o.x) / ray.d.x, ty_min = (bmn.y - o.y) / ray.d.y,
        tz_min = (bmn.z - o.z) / ray.d.z;
  float tx_max = (bmx.x - o.x) / ray.d.x, ty_max = (bmx.y - o.y) / ray.d.y,
        tz_max = (bmx.z - o.z) / ray.d.z;
  if (tx_min > tx_max) std::swap(tx_max, tx_min);
  if (ty_min > ty_max) std::swap(ty_max, ty_min);
  if (tz_min > tz_max) std::swap(tz_max, tz_min);
  float t0 = ORC_MAX3(tx_min, ty_min, tz_min);
  float t1 = ORC_MIN3(tx_max, ty_max, tz_max);
  if (t0 > t1 || t1 < 0) return false;
  k.dtx = (tx_max - tx_min) / nx;  // float / int -> float, widened on store
  k.dty = (ty_max - ty_min) / ny;
  k.dtz = (tz_max - tz_min) / nz;
  if (is_inside(G.bbox, o)) {
    k.ix = cell_coord((o.x - bmn.x) * nx / (bmx.x - bmn.x), nx);
    k.iy = cell_coord((o.y - bmn.y) * ny / (bmx.y - bmn.y), ny);
    k.iz = cell_coord((o.z - bmn.z) * nz / (bmx.z - bmn.z), nz);
  } else {
    V3 p = o + dir * t0;
    k.ix = cell_coord((p.x - bmn.x) * nx / (bmx.x - bmn.x), nx);
    k.iy = cell_coord((p.y - bmn.y) * ny / (bmx.y - bmn.y), ny);
    k.iz = cell_coord((p.z - bmn.z) * nz / (bmx.z - bmn.z), nz);
  }
  float dx = dir.x, dy = dir.y, dz = dir.z;
  if (dx > 0) { k.tx_next = tx_min + (k.ix + 1) * k.dtx; k.ix_step = +1; k.ix_stop = nx; }
  else        { k.tx_next = tx_min + (nx - k.ix) * k.dtx; k.ix_step = -1; k.ix_stop = -1; }
  if ((double)dx == 0.0) k.tx_next = FLT_MAX;
  if (dy > 0) { k.ty_next = ty_min + (k.iy + 1) * k.dty; k.iy_step = +1; k.iy_stop = ny; }
  else        { k.ty_next = ty_min + (ny - k.iy) * k.dty; k.iy_step = -1; k.iy_stop = -1; }
  if ((double)dy == 0.0) k.ty_next = FLT_MAX;
  if (dz > 0) { k.tz_next = tz_min + (k.iz + 1) * k.dtz; k.iz_step = +1; k.iz_stop = nz; }
  else        { k.tz_next = tz_min + (nz - k.iz) * k.dtz; k.iz_step = -1; k.iz_stop = -1; }
  if ((double)dz == 0.0) k.tz_next = FLT_MAX;
  return true;
}

// grid.cpp:71-151 — closest hit; the ray is taken BY REFERENCE (caller sees Q8 mutation).
bool grid_traverse(Ctx& cx, Ray& ray, int* hitobject, V3& hitpoint, float* t_out = nullptr) {
  const Grid& G = cx.sc->grid;
  const std::vector<Object>& O = cx.sc->objects;
  GridWalk k;
  if (!grid_init_traverse(G, ray, k)) return false;
  int min_obj = -1;
  float min_t = FLT_MAX, t = FLT_MAX;
  while (true) {
    const std::vector<int>& cell = G.cells.at(k.ix + G.nx * k.iy + G.nx * G.ny * k.iz);
    for (size_t i = 0; i < cell.size(); i++) {
      if (intercepts(O[cell[i]], ray, t, cx.st)) {
        if (t < min_t) { min_t = t; min_obj = cell[i]; }
      }
    }
    if (k.tx_next < k.ty_next && k.tx_next < k.tz_next) {
      if (min_obj >= 0 && (double)min_t < k.tx_next) {
        *hitobject = min_obj; hitpoint = ray.o + ray.d * min_t; if (t_out) *t_out = min_t; return true;
      }
      k.tx_next += k.dtx; k.ix += k.ix_step;
      if (k.ix == k.ix_stop) return false;
    } else if (k.ty_next < k.tz_next) {
      if (min_obj >= 0 && (double)min_t < k.ty_next) {
        *hitobject = min_obj; hitpoint = ray.o + ray.d * min_t; if (t_out) *t_out = min_t; return true;
      }
      k.ty_next += k.dty; k.iy += k.iy_step;
      if (k.iy == k.iy_stop) return false;
    } else {
      if (min_obj >= 0 && (double)min_t < k.tz_next) {
        *hitobject = min_obj; hitpoint = ray.o + ray.d * min_t; if (t_out) *t_out = min_t; return true;
      }
      k.tz_next += k.dtz; k.iz += k.iz_step;
      if (k.iz == k.iz_stop) return false;
    }
  }
}

// grid.cpp:154-208 — any hit
bool grid_traverse_any(Ctx& cx, Ray& ray) {
  const Grid& G = cx.sc->grid;
  const std::vector<Object>& O = cx.sc->objects;
  GridWalk k;
  if (!grid_init_traverse(G, ray, k)) return false;
  float t;
  while (true) {
    const std::vector<int>& cell = G.cells.at(k.ix + G.nx * k.iy + G.nx * G.ny * k.iz);
    for (size_t i = 0; i < cell.size(); i++)
      if (intercepts(O[cell[i]], ray, t, cx.st)) return true;
    if (k.tx_next < k.ty_next && k.tx_next < k.tz_next) {
      k.tx_next += k.dtx; k.ix += k.ix_step;
      if (k.ix == k.ix_stop) return false;
    } else if (k.ty_next < k.tz_next) {
      k.ty_next += k.dty; k.iy += k.iy_step;
      if (k.iy == k.iy_stop) return false;
    } else {
      k.tz_next += k.dtz; k.iz += k.iz_step;
      if (k.iz == k.iz_stop) return false;
    }
  }
}

// The three closest-hit back ends as rayTracing/Radiance select them
// (main.cpp:103-125 / 324-343 / 449-469).  Returns the object index or -1; for
// accel None, min_t is valid and the caller's ray carries the Q8 mutation.
int closest_hit(Ctx& cx, Ray& ray, float& min_t, V3& hit_p) {
  const Scene& S = *cx.sc;
  int min_obj = -1;
  float t = FLT_MAX;
  min_t = FLT_MAX;
  if (cx.cfg.accel == 1) {
    if (!grid_traverse(cx, ray, &min_obj, hit_p, &min_t)) min_obj = -1;
  } else if (cx.cfg.accel == 2) {
    if (!intersect_bvh(cx, ray, &min_obj, hit_p, &min_t)) min_obj = -1;
  } else {
    for (int i = 0; i < (int)S.objects.size(); i++) {
      if (intercepts(S.objects[i], ray, t, cx.st) && (t < min_t)) { min_obj = i; min_t = t; }
    }
  }
  return min_obj;
}


// Scene::GetSkyboxColor — scene.cpp:379-457.  Indexed by the RAW ray direction; the two
// "clamp" lines scene.cpp:448,450 are expression statements without effect.
C3 get_skybox_color(const Scene& S, const Ray& r) {
  V3 cc = r.d;
  float ma;
  int side;  // RIGHT 0, LEFT 1, TOP 2, BOTTOM 3, FRONT 4, BACK 5
  if (fabsf(cc.x) > fabsf(cc.y)) { ma = fabsf(cc.x); side = cc.x >= 0 ? 1 : 0; }
  else                           { ma = fabsf(cc.y); side = cc.y >= 0 ? 2 : 3; }
  if (fabsf(cc.z) > ma) { ma = fabsf(cc.z); side = cc.z >= 0 ? 4 : 5; }
  float sc, tc;
  switch (side) {
    case 0: sc = -cc.z; tc = cc.y; break;
    case 1: sc = cc.z; tc = cc.y; break;
    case 2: sc = -cc.x; tc = -cc.z; break;
    case 3: sc = -cc.x; tc = cc.z; break;
    case 4: sc = -cc.x; tc = cc.y; break;
    default: sc = cc.x; tc = cc.y; break;
  }
  double invMa = 1 / ma;  // int / float: a FLOAT division, widened afterwards
  float s = (float)((sc * invMa + 1) / 2);
  float t = (float)((tc * invMa + 1) / 2);
  const Scene::Face& f = S.sky[side];
  unsigned width = f.resX, height = f.resY, bpp = f.BPP;
  unsigned xp = (unsigned)(int)((width - 1) * s);
  unsigned yp = (unsigned)(int)((height - 1) * t);
  if (xp >= width) xp = width - 1;   // the reference would read out of bounds here (NaN / zero direction)
  if (yp >= height) yp = height - 1;
  size_t idx = ((size_t)yp * width + xp) * bpp;
  auto u8tofloat = [](uint8_t x) { return (float)(x / 255.99f); };  // maths.h:89-92
  return c3(u8tofloat(f.img[idx]), u8tofloat(f.img[idx + 1]), u8tofloat(f.img[idx + 2]));
}
inline C3 miss_color(const Scene& S, const orc_config& cfg, const Ray& r) {  // main.cpp:144-147 / 350-355
  return (cfg.skybox && S.sky_loaded) ? get_skybox_color(S, r) : S.bg;
}

inline V3 offset_intersection(V3 inter, V3 normal) { return inter + normal * .0001f; }  // main.cpp:82-84

// ---------------------------------------------------------------------------
// L3: Whitted — main.cpp:92-309
// ---------------------------------------------------------------------------
C3 ray_tracing(Ctx& cx, Ray ray, int depth, float ior_1, int off_x, int off_y, bool inside,
               int* primary_hit) {
  const Scene& S = *cx.sc;
  const orc_config& cfg = cx.cfg;
  V3 hit_p;
  float min_t;
  int min_obj = closest_hit(cx, ray, min_t, hit_p);
  if (primary_hit) *primary_hit = min_obj;
  if (cfg.accel == 0 && cfg.debug_view == 2) {  // DEPTH_MAP, main.cpp:127-139 (remap: main.cpp:86-88)
    float depth_grey = 1.f + (min_t - 5.f) * (0.f - 1.f) / (20.f - 5.f);
    return clamp(c3(depth_grey, depth_grey, depth_grey));
  }
  if (min_obj < 0) return miss_color(S, cfg, ray);  // main.cpp:144-147
  if (cfg.debug_view == 1) return c3(1, 0, 0);      // TEST_INTERSECT, main.cpp:156

  const Object& ob = S.objects[min_obj];
  const Material& mat = S.materials[ob.mat];
  cx.st.shaded_hits++;
  C3 col, diff, spec;
  V3 l_dir, norm, blinn;
  float fs;
  // main.cpp:164-167
  V3 interceptNotPrecise = (cfg.accel == 0) ? ray.o + ray.d * min_t : hit_p;
  V3 intercept = offset_intersection(interceptNotPrecise, get_normal(ob, interceptNotPrecise));
  norm = get_normal(ob, intercept);

  if (!inside) {  // main.cpp:172-227
    for (int i = 0; i < (int)S.lights.size(); i++) {
      const Light& light = S.lights[i];
      if (cfg.antialiasing && cfg.soft_shadows) {  // main.cpp:180-186
        float jx, jy;
        if (cfg.eval_order & 4) { jx = cx.rng.rand_float(); jy = cx.rng.rand_float(); }
        else                    { jy = cx.rng.rand_float(); jx = cx.rng.rand_float(); }
        V3 pos = v3(light.pos.x + cfg.light_side * (off_x + jx) / cfg.spp_sqrt,
                    light.pos.y + cfg.light_side * (off_y + jy) / cfg.spp_sqrt, light.pos.z);
        l_dir = pos - intercept;
        normalize(l_dir);
      } else {
        l_dir = light.pos - intercept;
        normalize(l_dir);
      }
      Ray feeler;
      feeler.o = intercept; feeler.d = l_dir;
      cx.st.rays_shadow++;
      cx.st.ref_ray_counter++;
      fs = 1;
      if (cfg.accel == 1) {
        if (grid_traverse_any(cx, feeler)) fs = 0;
      }
      if (cfg.accel == 2) {
        if (bool_intersect_bvh(cx, feeler)) fs = 0;
      } else {  // Q6: also runs after the grid query
        float t;
        for (int j = 0; j < (int)S.objects.size(); j++) {
          if (intercepts(S.objects[j], feeler, t, cx.st)) { fs = 0; break; }
        }
      }
      blinn = (l_dir + (get_direction(ray) * -1)) / 2;
      normalize(blinn);
      if (fs != 0) {  // main.cpp:222-225; max() and pow() evaluate in double
        diff = diff + (light.col * mat.cd) * (float)dmax(0, dot(norm, l_dir));
        spec = spec + (light.col * mat.cs) *
                          (float)pow(dmax(0, dot(blinn, norm)), (double)mat.shine);
      }
    }
  }
  col = col + (diff * mat.kd + spec * mat.ks);  // main.cpp:232
  if (depth <= 0) return clamp(col);

  norm = !inside ? norm : norm * -1;  // main.cpp:238
  float Kr;
  V3 v = get_direction(ray) * -1;
  V3 vn = norm * dot(v, norm);
  V3 vt = vn - v;
  C3 refrCol, reflCol;
  bool zero_weight_reflection = false;
  if (mat.T == 0) {
    Kr = mat.ks;  // main.cpp:250
  } else {
    float Rs = 1, Rp = 1;
    float n = !inside ? ior_1 / mat.ior : ior_1 / 1;  // main.cpp:256
    float cosOi = length(vn);
    float sinOt = (n)*length(vt), cosOt;
    float insqrt = (float)(1 - pow((double)sinOt, 2));  // std::pow(float,int) -> double
    if (insqrt >= 0) {
      cosOt = sqrtf(insqrt);
      V3 refractDir = normalize(vt) * sinOt + norm * (-cosOt);
      normalize(refractDir);
      V3 interceptin = offset_intersection(interceptNotPrecise, refractDir);  // main.cpp:267
      Ray refractedRay;
      refractedRay.o = interceptin; refractedRay.d = refractDir;
      cx.st.rays_refract++;
      cx.st.ref_ray_counter++;
      float newior = !inside ? mat.ior : 1;
      refrCol = ray_tracing(cx, refractedRay, depth - 1, newior, off_x, off_y, !inside, nullptr);
      // main.cpp:277-278 (pow(float,int) -> double, narrowed on store)
      Rs = (float)pow((double)fabsf((ior_1 * cosOi - newior * cosOt) / (ior_1 * cosOi + newior * cosOt)), 2);
      Rp = (float)pow((double)fabsf((ior_1 * cosOt - newior * cosOi) / (ior_1 * cosOt + newior * cosOi)), 2);
    }
    Kr = (1 / 2) * (Rs + Rp);  // main.cpp:282: integer division, Kr == 0 (Q3)
    zero_weight_reflection = true;
  }
  if (mat.refl > 0 && !(zero_weight_reflection && !cfg.trace_zero_weight)) {  // main.cpp:290-300
    V3 rdir;
    if (cfg.eval_order & 8) {
      V3 a = norm * dot(get_direction(ray) * -1, norm) * 2;
      rdir = a + get_direction(ray);
    } else {  // g++: the right operand of `+` is evaluated first (both calls mutate ray.d)
      V3 b = get_direction(ray);
      V3 a = norm * dot(get_direction(ray) * -1, norm) * 2;
      rdir = a + b;
    }
    Ray rray;
    rray.o = intercept; rray.d = rdir;
    cx.st.rays_reflect++;
    cx.st.ref_ray_counter++;
    reflCol = ray_tracing(cx, rray, depth - 1, ior_1, off_x, off_y, inside, nullptr);
  }
  col = col + (reflCol * Kr + refrCol * (1 - Kr));  // main.cpp:305
  return clamp(col);
}

// ---------------------------------------------------------------------------
// L3: path tracer — main.cpp:313-516
// ---------------------------------------------------------------------------
// math_mode 0: detmath (what the HIP kernels compute, bit for bit); 1: this host's libm
// (what the reference binary calls) — used to check the restatement against reference runs.
inline float pt_cosf(const orc_config& c, float x) { return c.math_mode ? cosf(x) : (float)det_cos((double)x); }
inline float pt_sinf(const orc_config& c, float x) { return c.math_mode ? sinf(x) : (float)det_sin((double)x); }

C3 radiance(Ctx& cx, Ray ray, int depth, float ior_1, int off_x, int off_y, bool inside,
            int* primary_hit) {
  const Scene& S = *cx.sc;
  const orc_config& cfg = cx.cfg;
  V3 hit_p;
  float min_t;
  int min_obj = closest_hit(cx, ray, min_t, hit_p);
  if (primary_hit) *primary_hit = min_obj;
  if (min_obj < 0 || depth == 0) return miss_color(S, cfg, ray);  // main.cpp:350-355
  if (cfg.debug_view == 1) return c3(1, 0, 0);                    // TEST_INTERSECT, main.cpp:359

  const Object& ob = S.objects[min_obj];
  const Material& mat = S.materials[ob.mat];
  cx.st.shaded_hits++;
  V3 interceptNotPrecise = (cfg.accel == 0) ? ray.o + ray.d * min_t : hit_p;  // main.cpp:364
  V3 norm = get_normal(ob, interceptNotPrecise);
  V3 norml = (dot(norm, ray.d) < 0) ? norm : norm * -1;
  V3 intercept_out = offset_intersection(interceptNotPrecise, norm);
  V3 intercept_in = offset_intersection(interceptNotPrecise, norm * -1);
  C3 f = mat.cd;
  float p = ORC_MAX3(f.r, f.g, f.b);
  if (--depth <= cfg.max_depth - 5) {  // main.cpp:382-388
    if (cx.rng.rand_float() < p) f = f * (1 / p);
    else return mat.em;
  }
  if (mat.kd == 1.0f) {  // ideal diffuse, main.cpp:391-480
    float r1 = 2 * kPI * cx.rng.rand_float();
    float r2 = cx.rng.rand_float();
    float r2s = sqrtf(r2);
    V3 w = norml;
    V3 u = cross(((double)fabsf(w.x) > .1) ? v3(0, 1, 0) : v3(1, 0, 0), w);
    normalize(u);
    V3 v = cross(w, u);
    V3 d = (u * pt_cosf(cfg, r1) * r2s + v * pt_sinf(cfg, r1) * r2s) + w * sqrtf(1 - r2);
    normalize(d);
    Ray new_r;
    new_r.o = intercept_out; new_r.d = d;
    C3 e;
    for (int i = 0; i < (int)S.objects.size(); i++) {  // main.cpp:407-477
      const Object& lo = S.objects[i];
      C3 emi = S.materials[lo.mat].em;
      if (emi.r + emi.g + emi.b <= 0) continue;
      if (lo.kind != K_SPHERE) continue;  // reference null-derefs here (Q13); emitters must be spheres
      V3 sw = lo.center - intercept_out;
      V3 su = cross(((double)fabsf(sw.x) > .1 ? v3(0, 1, 0) : v3(1, 0, 0)), sw);
      normalize(su);
      V3 sv = cross(sw, su);
      V3 center = lo.center;
      float rad = lo.radius;
      V3 ic = intercept_out - center;
      double cos_a_max = sqrt(1 - pow((double)rad, 2) / (double)dot(ic, ic));
      double eps1 = cx.rng.erand48();
      double eps2 = cx.rng.erand48();
      double cos_a = 1 - eps1 + eps1 * cos_a_max;
      double sin_a = sqrt(1 - cos_a * cos_a);
      double phi = (double)(2 * kPI) * eps2;
      double sphi, cphi;
      if (cfg.math_mode) { sphi = sin(phi); cphi = cos(phi); }
      else det_sincos(phi, &sphi, &cphi);
      V3 l = (su * (float)cphi * (float)sin_a + sv * (float)sphi * (float)sin_a) + sw * (float)cos_a;
      normalize(l);
      Ray feeler;
      feeler.o = intercept_out; feeler.d = l;
      cx.st.rays_light++;
      cx.st.ref_ray_counter++;
      V3 hit_p2;
      float min_t2;
      int min_obj2 = closest_hit(cx, feeler, min_t2, hit_p2);
      if (min_obj2 >= 0 && min_obj2 == i) {  // main.cpp:472-475
        double omega = (double)(2 * kPI) * (1 - cos_a_max);
        e = e + f * (emi * dot(l, norml) * (float)omega) * (1 / kPI);
      }
    }
    cx.st.rays_bounce++;
    return (mat.em + e) + f * radiance(cx, new_r, depth, ior_1, off_x, off_y, false, nullptr);
  } else if (mat.ks == 1.0f) {  // mirror, main.cpp:481-484
    Ray new_r;
    new_r.o = intercept_out;
    new_r.d = ray.d - norm * (2 * dot(norm, ray.d));
    cx.st.rays_bounce++;
    return mat.em + f * radiance(cx, new_r, depth, ior_1, off_x, off_y, false, nullptr);
  }
  // dielectric, main.cpp:486-515
  Ray reflRay;
  reflRay.o = intercept_out;
  reflRay.d = ray.d - norm * 2 * dot(norm, ray.d);
  bool into = dot(norm, norml) > 0;
  double nc = 1.0f, nt = mat.ior;
  double nnt = into ? nc / nt : nt / nc;
  double ddn = dot(ray.d, norml);
  double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
  if (cos2t < 0) {  // total internal reflection
    cx.st.rays_bounce++;
    return mat.em + f * radiance(cx, reflRay, depth, ior_1, off_x, off_y, false, nullptr);
  }
  V3 tdir = ray.d * (float)nnt - norm * (float)((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t)));
  normalize(tdir);
  double a = nt - nc, b = nt + nc;
  double R0 = (a * a) / (b * b);
  double c = 1 - (into ? -ddn : (double)dot(tdir, norm));
  double Re = R0 + (1 - R0) * c * c * c * c * c;
  double Tr = 1 - Re;
  double P = 0.25 + 0.5 * Re;
  double RP = Re / P, TP = Tr / (1 - P);
  C3 col;
  if (depth <= cfg.max_depth - 2) {  // main.cpp:509-511: Russian roulette between the two
    if (cx.rng.erand48() < P) {
      cx.st.rays_bounce++;
      col = radiance(cx, reflRay, depth, ior_1, off_x, off_y, false, nullptr) * (float)RP;
    } else {
      Ray tr;
      tr.o = intercept_out; tr.d = tdir;
      cx.st.rays_bounce++;
      col = radiance(cx, tr, depth, ior_1, off_x, off_y, false, nullptr) * (float)TP;
    }
  } else {  // main.cpp:512-513: first two bounces trace both
    Ray tr;
    tr.o = intercept_in; tr.d = tdir;
    cx.st.rays_bounce += 2;
    C3 cr, ct;
    if (cfg.eval_order & 2) {
      cr = radiance(cx, reflRay, depth, ior_1, off_x, off_y, false, nullptr) * (float)Re;
      ct = radiance(cx, tr, depth, ior_1, off_x, off_y, false, nullptr) * (float)Tr;
    } else {  // g++: transmission (right operand) first; matters for the shared RNG stream
      ct = radiance(cx, tr, depth, ior_1, off_x, off_y, false, nullptr) * (float)Tr;
      cr = radiance(cx, reflRay, depth, ior_1, off_x, off_y, false, nullptr) * (float)Re;
    }
    col = cr + ct;
  }
  return mat.em + f * col;
}

// sampler.cpp:5-11
V3 sample_unit_disk(Ctx& cx) {
  V3 p;
  do {
    float a, b;
    if (cx.cfg.eval_order & 1) { a = cx.rng.rand_float(); b = cx.rng.rand_float(); }
    else                       { b = cx.rng.rand_float(); a = cx.rng.rand_float(); }  // g++: args right to left
    p = v3(a, b, 0.0f) * 2 - v3(1.0f, 1.0f, 0.0f);
  } while ((double)dot(p, p) >= 1.0);
  return p;
}

inline uint8_t u8fromfloat(float x) {  // maths.h:81-86
  return ((x * 255.99f) >= 255.0f ? 255 : (uint8_t)(x * 255.99f));
}

// One pixel of the frame loop, main.cpp:753-820.
void render_pixel(Ctx& cx, int x, int y, float* rgb, int32_t* hit, uint8_t* rgb8) {
  const Scene& S = *cx.sc;
  const orc_config& cfg = cx.cfg;
  C3 color;
  V3 pixel, lens;
  int first_hit = -1;
  const int SPP = cfg.spp_sqrt;
  cx.st.pixels++;
  if (cfg.antialiasing) {
    for (int i = 0; i < SPP; i++) {
      for (int j = 0; j < SPP; j++) {
        if (cfg.rng_mode == 0)
          cx.rng.seed_stream(cfg.seed, (uint32_t)(y * S.cam.res_x + x), (uint32_t)(i * SPP + j));
        if (cfg.stack_mode == 0) cx.hit_stack.clear();
        if (cfg.sample_mode == 0) {  // jitter, main.cpp:763-766
          pixel.x = x + (i + cx.rng.rand_float()) / SPP;
          pixel.y = y + (j + cx.rng.rand_float()) / SPP;
        } else {  // tent, main.cpp:767-773
          double r1 = 2 * cx.rng.erand48(), dx = r1 < 1 ? sqrt(r1) - 1 : 1 - sqrt(2 - r1);
          double r2 = 2 * cx.rng.erand48(), dy = r2 < 1 ? sqrt(r2) - 1 : 1 - sqrt(2 - r2);
          pixel.x = (float)(x + (0.5 + dx) / SPP);
          pixel.y = (float)(y + (0.5 + dy) / SPP);
        }
        Ray ray;
        if (cfg.depth_of_field) {  // main.cpp:776-784
          if (cfg.sample_disk) lens = sample_unit_disk(cx);
          else {
            lens.x = (i + cx.rng.rand_float()) / SPP;
            lens.y = (j + cx.rng.rand_float()) / SPP;
          }
          ray = S.cam.primary_lens(lens, pixel);
        } else {
          ray = S.cam.primary(pixel);
        }
        cx.st.rays_primary++;
        cx.st.ref_ray_counter++;
        int h = -1;
        C3 c = cfg.integrator ? radiance(cx, ray, cfg.max_depth, 1.0f, i, j, false, &h)
                              : ray_tracing(cx, ray, cfg.max_depth, 1.0f, i, j, false, &h);
        if (i == 0 && j == 0) first_hit = h;
        color = color + c;
      }
    }
    color = color / (float)(SPP * SPP);  // main.cpp:800
  } else {  // main.cpp:804-812
    if (cfg.rng_mode == 0) cx.rng.seed_stream(cfg.seed, (uint32_t)(y * S.cam.res_x + x), 0);
    if (cfg.stack_mode == 0) cx.hit_stack.clear();
    pixel.x = (float)(x + 0.5);
    pixel.y = (float)(y + 0.5);
    Ray ray = S.cam.primary(pixel);
    cx.st.rays_primary++;
    cx.st.ref_ray_counter++;
    // the reference always calls rayTracing here, whatever PATHTRACING says (main.cpp:811)
    color = color + ray_tracing(cx, ray, cfg.max_depth, 1.0f, 0, 0, false, &first_hit);
  }
  if (rgb) { rgb[0] = color.r; rgb[1] = color.g; rgb[2] = color.b; }
  if (hit) *hit = first_hit;
  if (rgb8) {  // main.cpp:814-820
    double invGamma = 1 / cfg.gamma;
    rgb8[0] = u8fromfloat((float)pow((double)color.r, invGamma));
    rgb8[1] = u8fromfloat((float)pow((double)color.g, invGamma));
    rgb8[2] = u8fromfloat((float)pow((double)color.b, invGamma));
  }
}

// ---------------------------------------------------------------------------
// .p3f loader — scene.cpp:472-628 (iostream extraction semantics kept: a failed
// numeric read poisons the stream and ends the parse)
// ---------------------------------------------------------------------------
std::istream& operator>>(std::istream& s, V3& v) { return s >> v.x >> v.y >> v.z; }  // vector.h:41-43
std::istream& operator>>(std::istream& s, C3& c) { return s >> c.r >> c.g >> c.b; }  // color.h:80-82

bool load_p3f(Scene& S, const char* name, bool legacy_f11) {
  std::ifstream file(name, std::ios::in);
  if (!file.is_open()) return false;
  std::string cmd;
  int material = -1;
  char token[256];
  auto next_token = [&](const char* expect) {  // scene.cpp:465-470
    file >> token;
    if (strcmp(token, expect)) fprintf(stderr, "'%s' expected.\n", expect);
  };
  if (file >> cmd) {
    while (true) {
      if (cmd == "f") {  // scene.cpp:487-495: Kd..ior are read as DOUBLE then narrowed
        double Kd, Ks, Shine, T, ior;
        C3 cd, cs, em;
        file >> cd >> Kd >> cs >> Ks >> Shine >> T >> ior;
        if (legacy_f11) {
          // extension: emission present only if the next token starts a number
          std::streampos pos = file.tellg();
          std::string tok;
          bool numeric = false;
          if (file >> tok) {
            char* endp = nullptr;
            strtod(tok.c_str(), &endp);
            numeric = (endp && *endp == 0 && endp != tok.c_str());
          }
          file.clear();
          file.seekg(pos);
          if (numeric) file >> em;
        } else {
          file >> em;
        }
        Material m;
        m.cd = cd; m.kd = (float)Kd; m.cs = cs; m.ks = (float)Ks; m.shine = (float)Shine;
        m.T = (float)T; m.ior = (float)ior; m.em = em; m.refl = (float)Ks;  // scene.h:42
        S.materials.push_back(m);
        material = (int)S.materials.size() - 1;
      } else if (cmd == "s") {
        V3 center;
        float radius;
        file >> center >> radius;
        Object o = make_sphere(center, radius);
        o.mat = material;
        S.objects.push_back(o);
      } else if (cmd == "box") {
        V3 mn, mx;
        file >> mn >> mx;
        Object o = make_box(mn, mx);
        o.mat = material;
        S.objects.push_back(o);
      } else if (cmd == "p") {
        V3 P0, P1, P2;
        unsigned total_vertices;
        file >> total_vertices;
        if (total_vertices == 3) {
          file >> P0 >> P1 >> P2;
          Object o = make_triangle(P0, P1, P2);
          o.mat = material;
          S.objects.push_back(o);
        } else {
          fprintf(stderr, "Unsupported number of vertices.\n");
          break;
        }
      } else if (cmd == "pl") {
        V3 P0, P1, P2;
        file >> P0 >> P1 >> P2;
        Object o = make_plane(P0, P1, P2);
        o.mat = material;
        S.objects.push_back(o);
      } else if (cmd == "l") {
        Light l;
        file >> l.pos >> l.col;
        S.lights.push_back(l);
      } else if (cmd == "v") {  // scene.cpp:561-596
        int xres, yres;
        next_token("from"); file >> S.v_from;
        next_token("at"); file >> S.v_at;
        next_token("up"); file >> S.v_up;
        next_token("angle"); file >> S.v_angle;
        next_token("hither"); file >> S.v_hither;
        next_token("resolution"); file >> xres >> yres;
        next_token("aperture"); file >> S.v_aperture;
        next_token("focal"); file >> S.v_focal;
        S.cam.init(S.v_from, S.v_at, S.v_up, S.v_angle, S.v_hither, (float)(100.0 * S.v_hither),
                   xres, yres, S.v_aperture, S.v_focal);
        S.has_cam = true;
      } else if (cmd == "bclr") {
        file >> S.bg;
      } else if (cmd == "env") {  // scene.cpp:605-611; skybox images are out of scope
        file >> token;
        S.skybox_flag = true;
      } else if (cmd[0] == '#') {
        file.ignore(1024, '\n');
      } else {
        fprintf(stderr, "unknown command '%s'.\n", cmd.c_str());
        break;
      }
      if (!(file >> cmd)) break;
    }
  }
  return true;
}

void rebuild_camera(Scene& S, int rx, int ry) {
  S.cam.init(S.v_from, S.v_at, S.v_up, S.v_angle, S.v_hither, (float)(100.0 * S.v_hither), rx, ry,
             S.v_aperture, S.v_focal);
}

// Rows are handed out one at a time from a shared counter (threads > 1: rows differ in cost by an order of magnitude,
// and a fixed deal left the timing at the mercy of which thread got the expensive ones; the pixels do not depend on
// who renders them: own RNG stream per pixel and sample, stack emptied per pixel in the threaded modes).
void render_rows(Ctx& cx, int x0, int y0, int w, int h, std::atomic<int>& next_row, float* rgb,
                 int32_t* hit, uint8_t* rgb8, int repeat = 1) {
  const long long rows = (long long)h * repeat;  // timing runs render the tile `repeat` times with one thread pool
  // rows are handed out in blocks (one atomic per block, not per row) small enough that the last blocks still balance
  const int block = rows >= 64 * 64 ? 4 : 1;
  for (long long i0 = next_row.fetch_add(block, std::memory_order_relaxed); i0 < rows; i0 = next_row.fetch_add(block, std::memory_order_relaxed)) {
    for (long long i = i0; i < i0 + block && i < rows; i++) {
      const int r = (int)(i % h);
      int y = y0 + r;
      for (int c = 0; c < w; c++) {
        size_t k = (size_t)r * w + c;
        render_pixel(cx, x0 + c, y, rgb ? rgb + 3 * k : nullptr, hit ? hit + k : nullptr,
                     rgb8 ? rgb8 + 3 * k : nullptr);
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------
extern "C" {

void orc_config_default(orc_config* c) {  // constants.h:6-45 as shipped (SKYBOX n/a)
  memset(c, 0, sizeof(*c));
  c->integrator = 1; c->accel = 2; c->max_depth = 20; c->spp_sqrt = 20; c->antialiasing = 1;
  c->depth_of_field = 1; c->sample_disk = 1; c->soft_shadows = 0; c->sample_mode = 0;
  c->light_side = .5f; c->gamma = 1.0f; c->skybox = 0;
  c->rng_mode = 0; c->stack_mode = 0; c->trace_zero_weight = 0; c->eval_order = 0; c->threads = 1;
  c->seed = 0x5EED;
}

void* orc_scene_load(const char* path, int legacy_f11) {
  Scene* S = new Scene();
  if (!load_p3f(*S, path, legacy_f11 != 0)) { delete S; return nullptr; }
  return S;
}
void orc_scene_free(void* s) { delete (Scene*)s; }

int orc_scene_counts(void* s, int* n_objects, int* n_lights, int* n_materials, int* has_camera) {
  Scene* S = (Scene*)s;
  if (n_objects) *n_objects = (int)S->objects.size();
  if (n_lights) *n_lights = (int)S->lights.size();
  if (n_materials) *n_materials = (int)S->materials.size();
  if (has_camera) *has_camera = S->has_cam;
  return 0;
}
int orc_scene_set_resolution(void* s, int rx, int ry) {
  Scene* S = (Scene*)s;
  if (!S->has_cam) return -1;
  rebuild_camera(*S, rx > 0 ? rx : S->cam.res_x, ry > 0 ? ry : S->cam.res_y);
  return 0;
}
int orc_scene_set_lens(void* s, float aperture_ratio, float focal_ratio) {
  Scene* S = (Scene*)s;
  if (!S->has_cam) return -1;
  S->v_aperture = aperture_ratio; S->v_focal = focal_ratio;
  rebuild_camera(*S, S->cam.res_x, S->cam.res_y);
  return 0;
}
int orc_scene_replicate_lights(void* s, int SPP, float LIGHT_SIDE) {  // main.cpp:725-745
  Scene* S = (Scene*)s;
  std::vector<Light> new_lights;
  float step = LIGHT_SIDE / SPP;
  float start = -LIGHT_SIDE / 2 + step / 2;
  float end = LIGHT_SIDE / 2;
  for (const Light& light : S->lights) {
    C3 avg_col = light.col / (float)(SPP * SPP);
    for (float i = start; i < end; i += step)
      for (float j = start; j < end; j += step) {
        Light l;
        l.pos = v3(light.pos.x + i, light.pos.y + j, light.pos.z);
        l.col = avg_col;
        new_lights.push_back(l);
      }
  }
  S->lights = new_lights;
  return 0;
}
int orc_scene_object(void* s, int i, int* type, int* material, float* v9, float* n3, float* bmin3,
                     float* bmax3) {
  Scene* S = (Scene*)s;
  if (i < 0 || i >= (int)S->objects.size()) return -1;
  const Object& o = S->objects[i];
  float v[9] = {0}, n[3] = {0};
  switch (o.kind) {
    case K_SPHERE: v[0] = o.center.x; v[1] = o.center.y; v[2] = o.center.z; v[3] = o.radius; break;
    case K_TRI:
      v[0] = o.p0.x; v[1] = o.p0.y; v[2] = o.p0.z; v[3] = o.p1.x; v[4] = o.p1.y; v[5] = o.p1.z;
      v[6] = o.p2.x; v[7] = o.p2.y; v[8] = o.p2.z;
      n[0] = o.normal.x; n[1] = o.normal.y; n[2] = o.normal.z;
      break;
    case K_BOX:
      v[0] = o.bmin.x; v[1] = o.bmin.y; v[2] = o.bmin.z; v[3] = o.bmax.x; v[4] = o.bmax.y; v[5] = o.bmax.z;
      break;
    default:
      v[0] = o.PN.x; v[1] = o.PN.y; v[2] = o.PN.z; v[3] = o.A.x; v[4] = o.A.y; v[5] = o.A.z;
      n[0] = o.PN.x; n[1] = o.PN.y; n[2] = o.PN.z;
      break;
  }
  Box b = bounding_box(o);
  if (type) *type = o.kind;
  if (material) *material = o.mat;
  if (v9) memcpy(v9, v, sizeof(v));
  if (n3) memcpy(n3, n, sizeof(n));
  if (bmin3) { bmin3[0] = b.mn.x; bmin3[1] = b.mn.y; bmin3[2] = b.mn.z; }
  if (bmax3) { bmax3[0] = b.mx.x; bmax3[1] = b.mx.y; bmax3[2] = b.mx.z; }
  return 0;
}
int orc_scene_material(void* s, int i, float* m) {
  Scene* S = (Scene*)s;
  if (i < 0 || i >= (int)S->materials.size()) return -1;
  const Material& a = S->materials[i];
  float v[16] = {a.cd.r, a.cd.g, a.cd.b, a.kd, a.cs.r, a.cs.g, a.cs.b, a.ks,
                 a.shine, a.T,   a.ior,  a.refl, a.em.r, a.em.g, a.em.b, 0};
  memcpy(m, v, sizeof(v));
  return 0;
}
int orc_scene_light(void* s, int i, float* pos3, float* col3) {
  Scene* S = (Scene*)s;
  if (i < 0 || i >= (int)S->lights.size()) return -1;
  const Light& l = S->lights[i];
  pos3[0] = l.pos.x; pos3[1] = l.pos.y; pos3[2] = l.pos.z;
  col3[0] = l.col.r; col3[1] = l.col.g; col3[2] = l.col.b;
  return 0;
}
int orc_scene_camera(void* s, float* eye3, float* u3, float* v3_, float* n3, float* whdfa5, int* res2) {
  Scene* S = (Scene*)s;
  if (!S->has_cam) return -1;
  const Camera& c = S->cam;
  eye3[0] = c.eye.x; eye3[1] = c.eye.y; eye3[2] = c.eye.z;
  u3[0] = c.u.x; u3[1] = c.u.y; u3[2] = c.u.z;
  v3_[0] = c.v.x; v3_[1] = c.v.y; v3_[2] = c.v.z;
  n3[0] = c.n.x; n3[1] = c.n.y; n3[2] = c.n.z;
  whdfa5[0] = c.w; whdfa5[1] = c.h; whdfa5[2] = c.plane_dist; whdfa5[3] = c.focal_ratio; whdfa5[4] = c.aperture;
  res2[0] = c.res_x; res2[1] = c.res_y;
  return 0;
}
int orc_scene_set_skybox_face(void* s, int face, const uint8_t* img, unsigned res_x, unsigned res_y, unsigned bpp) {
  Scene* S = (Scene*)s;
  if (face < 0 || face > 5 || !img || res_x == 0 || res_y == 0 || (bpp != 3 && bpp != 4)) return -1;
  S->sky[face].img.assign(img, img + (size_t)res_x * res_y * bpp);
  S->sky[face].resX = res_x; S->sky[face].resY = res_y; S->sky[face].BPP = bpp;
  S->sky_loaded = true;
  for (int i = 0; i < 6; i++) if (S->sky[i].img.empty()) S->sky_loaded = false;
  return 0;
}
int orc_scene_background(void* s, float* rgb3) {
  Scene* S = (Scene*)s;
  rgb3[0] = S->bg.r; rgb3[1] = S->bg.g; rgb3[2] = S->bg.b;
  return 0;
}

int orc_build_bvh(void* s) {
  Scene* S = (Scene*)s;
  if (!S->bvh.built) S->bvh.build(*S);  // objs is appended, never cleared, in the reference: build once
  return 0;
}
int orc_build_grid(void* s) {
  Scene* S = (Scene*)s;
  if (!S->grid.built) S->grid.build(*S);
  return 0;
}
int orc_bvh_info(void* s, int* n_nodes, int* n_leaves, int* max_depth) {
  Scene* S = (Scene*)s;
  if (!S->bvh.built) return -1;
  int leaves = 0, maxd = 0;
  std::vector<std::pair<unsigned, int>> st;
  st.push_back({0u, 1});
  while (!st.empty()) {
    auto [i, d] = st.back();
    st.pop_back();
    if (d > maxd) maxd = d;
    const BvhNode& n = S->bvh.nodes[i];
    if (n.leaf) leaves++;
    else { st.push_back({n.index, d + 1}); st.push_back({n.index + 1, d + 1}); }
  }
  if (n_nodes) *n_nodes = (int)S->bvh.nodes.size();
  if (n_leaves) *n_leaves = leaves;
  if (max_depth) *max_depth = maxd;
  return 0;
}
int orc_bvh_nodes(void* s, float* bmin, float* bmax, uint32_t* index, uint32_t* n_objs, uint8_t* leaf) {
  Scene* S = (Scene*)s;
  if (!S->bvh.built) return -1;
  for (size_t i = 0; i < S->bvh.nodes.size(); i++) {
    const BvhNode& n = S->bvh.nodes[i];
    bmin[3 * i] = n.bbox.mn.x; bmin[3 * i + 1] = n.bbox.mn.y; bmin[3 * i + 2] = n.bbox.mn.z;
    bmax[3 * i] = n.bbox.mx.x; bmax[3 * i + 1] = n.bbox.mx.y; bmax[3 * i + 2] = n.bbox.mx.z;
    index[i] = n.index; n_objs[i] = n.n_objs; leaf[i] = n.leaf;
  }
  return 0;
}
int orc_bvh_order(void* s, int32_t* obj_index) {
  Scene* S = (Scene*)s;
  if (!S->bvh.built) return -1;
  for (size_t i = 0; i < S->bvh.objs.size(); i++) obj_index[i] = S->bvh.objs[i];
  return 0;
}
int orc_grid_info(void* s, int* nxyz3, float* bmin3, float* bmax3, int* n_items) {
  Scene* S = (Scene*)s;
  if (!S->grid.built) return -1;
  const Grid& G = S->grid;
  nxyz3[0] = G.nx; nxyz3[1] = G.ny; nxyz3[2] = G.nz;
  bmin3[0] = G.bbox.mn.x; bmin3[1] = G.bbox.mn.y; bmin3[2] = G.bbox.mn.z;
  bmax3[0] = G.bbox.mx.x; bmax3[1] = G.bbox.mx.y; bmax3[2] = G.bbox.mx.z;
  size_t n = 0;
  for (auto& c : G.cells) n += c.size();
  *n_items = (int)n;
  return 0;
}
int orc_grid_cells(void* s, uint32_t* cell_start, uint32_t* cell_items) {
  Scene* S = (Scene*)s;
  if (!S->grid.built) return -1;
  uint32_t n = 0;
  for (size_t c = 0; c < S->grid.cells.size(); c++) {
    cell_start[c] = n;
    for (int o : S->grid.cells[c]) cell_items[n++] = (uint32_t)o;
  }
  cell_start[S->grid.cells.size()] = n;
  return 0;
}

int orc_render(void* s, const orc_config* cfg, int x0, int y0, int w, int h, float* rgb,
               int32_t* hit_id, uint8_t* rgb8, orc_stats* stats) {
  return orc_render_repeat(s, cfg, x0, y0, w, h, 1, rgb, hit_id, rgb8, stats);
}

int orc_render_repeat(void* s, const orc_config* cfg, int x0, int y0, int w, int h, int repeat, float* rgb,
                      int32_t* hit_id, uint8_t* rgb8, orc_stats* stats) {
  Scene* S = (Scene*)s;
  if (repeat < 1) repeat = 1;
  if (repeat > 1 && cfg->stack_mode != 0 && cfg->accel == 2) return -3;  // the serial stack would run on from frame to frame
  if (!S->has_cam) return -1;
  if (cfg->accel == 2) orc_build_bvh(s);
  if (cfg->accel == 1) orc_build_grid(s);
  for (const Object& o : S->objects)
    if (o.mat < 0) return -2;  // object before the first `f`: the reference null-derefs (Appendix B)
  int threads = cfg->threads < 1 ? 1 : cfg->threads;
  if (threads > 1 && (cfg->rng_mode != 0 || cfg->stack_mode != 0)) return -3;
  auto t0 = std::chrono::high_resolution_clock::now();
  std::vector<Ctx> ctx(threads);
  for (int t = 0; t < threads; t++) {
    ctx[t].sc = S;
    ctx[t].cfg = *cfg;
    ctx[t].rng.mode = cfg->rng_mode;
  }
  if (cfg->rng_mode == 1) srand((unsigned)cfg->seed);  // maths.h:75-78
  std::atomic<int> next_row(0);
  if (threads == 1) {
    render_rows(ctx[0], x0, y0, w, h, next_row, rgb, hit_id, rgb8, repeat);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
      pool.emplace_back([&, t]() { render_rows(ctx[t], x0, y0, w, h, next_row, rgb, hit_id, rgb8, repeat); });
    for (auto& th : pool) th.join();
  }
  auto t1 = std::chrono::high_resolution_clock::now();
  if (stats) {
    Stats tot;
    for (auto& c : ctx) tot.add(c.st);
    stats->rays_primary = tot.rays_primary; stats->rays_shadow = tot.rays_shadow;
    stats->rays_reflect = tot.rays_reflect; stats->rays_refract = tot.rays_refract;
    stats->rays_bounce = tot.rays_bounce; stats->rays_light = tot.rays_light;
    stats->node_tests = tot.node_tests; stats->sphere_tests = tot.sphere_tests;
    stats->tri_tests = tot.tri_tests; stats->box_tests = tot.box_tests;
    stats->plane_tests = tot.plane_tests; stats->shaded_hits = tot.shaded_hits;
    stats->pixels = tot.pixels; stats->max_stack = tot.max_stack;
    stats->ref_ray_counter = tot.ref_ray_counter;
    stats->seconds = std::chrono::duration<double>(t1 - t0).count();
  }
  return 0;
}

int orc_aabb_intercepts(const float* bmin3, const float* bmax3, const float* o3, const float* d3, float* t) {
  Box b;
  b.mn = v3(bmin3[0], bmin3[1], bmin3[2]);
  b.mx = v3(bmax3[0], bmax3[1], bmax3[2]);
  Ray r;
  r.o = v3(o3[0], o3[1], o3[2]);
  r.d = v3(d3[0], d3[1], d3[2]);
  float tt = 0;
  bool h = aabb_intercepts(b, r, tt);
  *t = tt;
  return h;
}
int orc_object_intercepts(void* s, int obj, float* o3, float* d3, float* t) {
  Scene* S = (Scene*)s;
  if (obj < 0 || obj >= (int)S->objects.size()) return -1;
  Ray r;
  r.o = v3(o3[0], o3[1], o3[2]);
  r.d = v3(d3[0], d3[1], d3[2]);
  Stats st;
  float tt = FLT_MAX;
  bool h = intercepts(S->objects[obj], r, tt, st);
  d3[0] = r.d.x; d3[1] = r.d.y; d3[2] = r.d.z;  // Q8: the caller sees the re-normalised direction
  *t = tt;
  return h;
}
int orc_object_normal(void* s, int obj, const float* p3, float* n3) {
  Scene* S = (Scene*)s;
  if (obj < 0 || obj >= (int)S->objects.size()) return -1;
  V3 n = get_normal(S->objects[obj], v3(p3[0], p3[1], p3[2]));
  n3[0] = n.x; n3[1] = n.y; n3[2] = n.z;
  return 0;
}
int orc_primary_ray(void* s, float px, float py, float* o3, float* d3) {
  Scene* S = (Scene*)s;
  if (!S->has_cam) return -1;
  Ray r = S->cam.primary(v3(px, py, 0));
  o3[0] = r.o.x; o3[1] = r.o.y; o3[2] = r.o.z;
  d3[0] = r.d.x; d3[1] = r.d.y; d3[2] = r.d.z;
  return 0;
}
int orc_primary_ray_lens(void* s, float lx, float ly, float px, float py, float* o3, float* d3) {
  Scene* S = (Scene*)s;
  if (!S->has_cam) return -1;
  Ray r = S->cam.primary_lens(v3(lx, ly, 0), v3(px, py, 0));
  o3[0] = r.o.x; o3[1] = r.o.y; o3[2] = r.o.z;
  d3[0] = r.d.x; d3[1] = r.d.y; d3[2] = r.d.z;
  return 0;
}
int orc_trace_closest(void* s, int accel, int n, const float* o, const float* d, int32_t* hit,
                      float* t, float* hit_point) {
  Scene* S = (Scene*)s;
  if (accel == 2) orc_build_bvh(s);
  if (accel == 1) orc_build_grid(s);
  Ctx cx;
  cx.sc = S;
  orc_config_default(&cx.cfg);
  cx.cfg.accel = accel;
  for (int i = 0; i < n; i++) {
    cx.hit_stack.clear();
    Ray r;
    r.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    float min_t = FLT_MAX;
    V3 hp;
    int obj = closest_hit(cx, r, min_t, hp);
    if (obj >= 0 && accel == 0) hp = r.o + r.d * min_t;  // main.cpp:164
    hit[i] = obj;
    if (obj < 0) { hp = v3(0, 0, 0); min_t = FLT_MAX; }
    if (t) t[i] = min_t;  // the traversal's tmin / min_t (bvh.cpp:246, grid.cpp:100, main.cpp:120); FLT_MAX on a miss
    if (hit_point) { hit_point[3 * i] = hp.x; hit_point[3 * i + 1] = hp.y; hit_point[3 * i + 2] = hp.z; }
  }
  return 0;
}
int orc_trace_any(void* s, int accel, int n, const float* o, const float* d, uint8_t* occluded) {
  Scene* S = (Scene*)s;
  if (accel == 2) orc_build_bvh(s);
  if (accel == 1) orc_build_grid(s);
  Ctx cx;
  cx.sc = S;
  orc_config_default(&cx.cfg);
  cx.cfg.accel = accel;
  for (int i = 0; i < n; i++) {
    cx.hit_stack.clear();
    Ray r;
    r.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    bool occ = false;
    if (accel == 1) occ = grid_traverse_any(cx, r);
    if (accel == 2) {
      occ = bool_intersect_bvh(cx, r);
    } else {  // Q6: brute force also after the grid
      float t;
      bool b = false;
      for (size_t j = 0; j < S->objects.size(); j++)
        if (intercepts(S->objects[j], r, t, cx.st)) { b = true; break; }
      occ = occ || b;
    }
    occluded[i] = occ;
  }
  return 0;
}

int orc_skybox_color(void* s, const float* d3, float* rgb3) {  // scene.cpp:379-457
  Scene* S = (Scene*)s;
  if (!S->sky_loaded) return -1;
  Ray r;
  r.o = v3(0, 0, 0);
  r.d = v3(d3[0], d3[1], d3[2]);
  C3 c = get_skybox_color(*S, r);
  rgb3[0] = c.r; rgb3[1] = c.g; rgb3[2] = c.b;
  return 0;
}

// ---- L0 unit entry points (pinned against oracle/_ref, tests/test_oracle_ref_vectors.py) ----
void orc_vec_normalize(float* v) { V3 a = v3(v[0], v[1], v[2]); normalize(a); v[0] = a.x; v[1] = a.y; v[2] = a.z; }
float orc_vec_length(const float* v) { return length(v3(v[0], v[1], v[2])); }
float orc_vec_dot(const float* a, const float* b) { return dot(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); }
void orc_vec_cross(const float* a, const float* b, float* o) {
  V3 r = cross(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2]));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void orc_vec_div(const float* a, float f, float* o) {
  V3 r = v3(a[0], a[1], a[2]) / f;
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void orc_get_direction(float* d, int k) {
  Ray r;
  r.d = v3(d[0], d[1], d[2]);
  for (int i = 0; i < k; i++) get_direction(r);
  d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
// camera from explicit `v`-block values; returns aperture and plane_dist in state2
void orc_camera_rays(const float* p15, int n, const float* px2, const float* lens2, float* ray_o,
                     float* ray_d, float* lray_o, float* lray_d, float* state2) {
  Camera c;
  c.init(v3(p15[0], p15[1], p15[2]), v3(p15[3], p15[4], p15[5]), v3(p15[6], p15[7], p15[8]), p15[9],
         p15[10], (float)(100.0 * p15[10]), (int)p15[11], (int)p15[12], p15[13], p15[14]);
  state2[0] = c.aperture; state2[1] = c.plane_dist;
  for (int i = 0; i < n; i++) {
    Ray r = c.primary(v3(px2[2 * i], px2[2 * i + 1], 0));
    ray_o[3 * i] = r.o.x; ray_o[3 * i + 1] = r.o.y; ray_o[3 * i + 2] = r.o.z;
    ray_d[3 * i] = r.d.x; ray_d[3 * i + 1] = r.d.y; ray_d[3 * i + 2] = r.d.z;
    Ray q = c.primary_lens(v3(lens2[2 * i], lens2[2 * i + 1], 0), v3(px2[2 * i], px2[2 * i + 1], 0));
    lray_o[3 * i] = q.o.x; lray_o[3 * i + 1] = q.o.y; lray_o[3 * i + 2] = q.o.z;
    lray_d[3 * i] = q.d.x; lray_d[3 * i + 1] = q.d.y; lray_d[3 * i + 2] = q.d.z;
  }
}
// rand_float / sample_unit_disk on the libc stream (rng_mode 1)
void orc_libc_rand_floats(unsigned seed, int n, float* out) {
  Rng r; r.mode = 1;
  srand(seed);
  for (int i = 0; i < n; i++) out[i] = r.rand_float();
}
void orc_libc_unit_disk(unsigned seed, int n, int eval_order, float* out2) {
  Ctx cx;
  cx.sc = nullptr;
  orc_config_default(&cx.cfg);
  cx.cfg.eval_order = eval_order;
  cx.rng.mode = 1;
  srand(seed);
  for (int i = 0; i < n; i++) { V3 p = sample_unit_disk(cx); out2[2 * i] = p.x; out2[2 * i + 1] = p.y; }
}
void orc_color_clamp(float* c) { C3 k = clamp(c3(c[0], c[1], c[2])); c[0] = k.r; c[1] = k.g; c[2] = k.b; }
float orc_u8tofloat(uint8_t x) { return (float)(x / 255.99f); }  // maths.h:89-92

double orc_det_sin(double x) { return det_sin(x); }
double orc_det_cos(double x) { return det_cos(x); }
int orc_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out) {
  Rng r;
  r.seed_stream(seed, pixel, sample);
  for (int i = 0; i < n; i++) out[i] = r.next31();
  return 0;
}
uint8_t orc_u8fromfloat(float x) { return u8fromfloat(x); }

}  // extern "C"
