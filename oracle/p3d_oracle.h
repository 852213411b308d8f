/*
 * p3d_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Nothing under p3d-raytracer_amd/ includes, links or calls it.
 * See p3d_oracle.cpp for the restatement itself and its pinning status.
 */
#ifndef P3D_ORACLE_H
#define P3D_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_config {
  int32_t integrator;   /* 0 Whitted (rayTracing), 1 path tracer (Radiance)   constants.h:36 */
  int32_t accel;        /* 0 None, 1 UGrid, 2 Bvh                             constants.h:41,44 */
  int32_t max_depth;    /* MAX_DEPTH */
  int32_t spp_sqrt;     /* SPP */
  int32_t antialiasing; /* ANTIALIASING */
  int32_t depth_of_field;
  int32_t sample_disk;
  int32_t soft_shadows;
  int32_t sample_mode;  /* 0 jitter, 1 tent */
  float light_side;
  float gamma;
  int32_t skybox;       /* SKYBOX constants.h:30: miss colour = cubemap texel (needs orc_scene_set_skybox_face x6) */
  /* --- semantics switches (see DESIGN.md "Sequential state") --- */
  int32_t rng_mode;      /* 0: one PCG32 stream per (pixel, sample) keyed by `seed`
                            1: libc rand() in program order, srand((unsigned)seed) once per
                               render call (what the reference binary does, main.cpp:722) */
  int32_t stack_mode;    /* 0: BVH hit_stack emptied at every primary sample (parallel semantics)
                            1: one member stack for the whole frame (bvh.cpp:86, literal) */
  int32_t trace_zero_weight; /* 1: trace the reflection ray of transmissive materials although
                                its weight Kr is 0 (main.cpp:282,290-300, literal); 0: skip it */
  int32_t eval_order;    /* C++ leaves these operand orders unspecified; 0 = what g++ 11 -O2 does
                            (right operand / last argument first), which is what the reference
                            outputs this oracle was checked against were built with.  Set a bit
                            to get the other order:
                            bit0: sample_unit_disk draws x before y        (sampler.cpp:8)
                            bit1: dielectric two-branch traces reflection before transmission
                                  (main.cpp:512-513)
                            bit2: soft-shadow light jitter draws x before y (main.cpp:181-184)
                            bit3: reflection direction calls the left getDirection() first
                                  (main.cpp:293; both calls re-normalise the ray in place) */
  int32_t math_mode;     /* 0: detmath sin/cos (bit-identical to the HIP kernels); 1: host libm */
  int32_t threads;       /* worker threads (rows handed out dynamically); needs rng_mode 0, stack_mode 0 */
  int32_t debug_view;    /* 0 none, 1 TEST_INTERSECT (main.cpp:156,359), 2 DEPTH_MAP (main.cpp:127-139, accel None only) */
  uint64_t seed;
} orc_config;

typedef struct orc_stats {
  uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract, rays_bounce, rays_light;
  uint64_t node_tests, sphere_tests, tri_tests, box_tests, plane_tests, shaded_hits, pixels;
  uint64_t max_stack;
  uint64_t ref_ray_counter; /* the reference's own rayCounter semantics (main.cpp:42) */
  double seconds;
} orc_stats;

void orc_config_default(orc_config* c);

/* scene.cpp:472-628.  legacy_f11 != 0 additionally accepts the 11-number `f` line. */
void* orc_scene_load(const char* path, int legacy_f11);
void orc_scene_free(void* s);
int orc_scene_counts(void* s, int* n_objects, int* n_lights, int* n_materials, int* has_camera);
int orc_scene_set_resolution(void* s, int rx, int ry);
int orc_scene_set_lens(void* s, float aperture_ratio, float focal_ratio);
int orc_scene_replicate_lights(void* s, int spp_sqrt, float light_side); /* main.cpp:725-745 */
/* object i: type (0 sphere,1 triangle,2 box,3 plane), material index, 9 geometry floats,
 * normal (triangles), bbox min/max */
int orc_scene_object(void* s, int i, int* type, int* material, float* v9, float* n3, float* bmin3,
                     float* bmax3);
int orc_scene_material(void* s, int i, float* m16);
int orc_scene_light(void* s, int i, float* pos3, float* col3);
int orc_scene_camera(void* s, float* eye3, float* u3, float* v3, float* n3, float* whdfa5,
                     int* res2);
int orc_scene_background(void* s, float* rgb3);
/* decoded face bytes (what Scene::LoadSkybox stores, scene.cpp:329-377): row 0 = bottom row,
 * face order RIGHT, LEFT, TOP, BOTTOM, FRONT, BACK (scene.h:28) */
int orc_scene_set_skybox_face(void* s, int face, const uint8_t* img, unsigned res_x, unsigned res_y, unsigned bpp);

/* bvh.cpp:89-196 / grid.cpp:3-68; idempotent */
int orc_build_bvh(void* s);
int orc_build_grid(void* s);
int orc_bvh_info(void* s, int* n_nodes, int* n_leaves, int* max_depth);
int orc_bvh_nodes(void* s, float* bmin_n3, float* bmax_n3, uint32_t* index, uint32_t* n_objs,
                  uint8_t* leaf);
int orc_bvh_order(void* s, int32_t* obj_index); /* permuted objs, n_objects entries */
int orc_grid_info(void* s, int* nxyz3, float* bmin3, float* bmax3, int* n_items);
int orc_grid_cells(void* s, uint32_t* cell_start, uint32_t* cell_items);

/* main.cpp:747-820 over the tile [x0,x0+w) x [y0,y0+h) (y = 0 bottom row).  rgb is the
 * averaged colour before gamma; rgb8 after gamma + u8fromfloat.  Any output may be NULL. */
int orc_render(void* s, const orc_config* cfg, int x0, int y0, int w, int h, float* rgb,
               int32_t* hit_id, uint8_t* rgb8, orc_stats* stats);
/* The same tile rendered `repeat` times by ONE pool of cfg->threads threads (timing runs of bench.py's cpu_baseline:
 * a sample long enough to time without paying for thread start-up per frame).  Counters and seconds cover all
 * repetitions; not for the serial hit_stack (stack_mode 1 over the BVH), which would run on from frame to frame. */
int orc_render_repeat(void* s, const orc_config* cfg, int x0, int y0, int w, int h, int repeat, float* rgb,
                      int32_t* hit_id, uint8_t* rgb8, orc_stats* stats);

/* unit-level entry points */
int orc_aabb_intercepts(const float* bmin3, const float* bmax3, const float* o3, const float* d3,
                        float* t); /* boundingBox.cpp:44-98 */
int orc_object_intercepts(void* s, int obj, float* o3, float* d3_inout, float* t);
int orc_object_normal(void* s, int obj, const float* p3, float* n3);
int orc_primary_ray(void* s, float px, float py, float* o3, float* d3);               /* camera.h:65-82 */
int orc_primary_ray_lens(void* s, float lx, float ly, float px, float py, float* o3, float* d3); /* camera.h:84-115 */
/* Batched queries, each ray on an empty stack: bvh.cpp:198-276 / grid.cpp:71-151 / main.cpp:116-124 */
int orc_trace_closest(void* s, int accel, int n, const float* o, const float* d, int32_t* hit,
                      float* t, float* hit_point);
int orc_trace_any(void* s, int accel, int n, const float* o, const float* d, uint8_t* occluded);
int orc_skybox_color(void* s, const float* d3, float* rgb3); /* Scene::GetSkyboxColor, scene.cpp:379-457 */
/* L0 restatements (vector.cpp, ray.h:16-18, camera.h:34-115, maths.h:67-92, sampler.cpp:5-11, color.h:39-44) */
void orc_vec_normalize(float* v3);
float orc_vec_length(const float* v3);
float orc_vec_dot(const float* a3, const float* b3);
void orc_vec_cross(const float* a3, const float* b3, float* out3);
void orc_vec_div(const float* a3, float f, float* out3);
void orc_get_direction(float* d3, int k);
void orc_camera_rays(const float* p15, int n, const float* px2, const float* lens2, float* ray_o,
                     float* ray_d, float* lray_o, float* lray_d, float* state2);
void orc_libc_rand_floats(unsigned seed, int n, float* out);
void orc_libc_unit_disk(unsigned seed, int n, int eval_order, float* out2);
void orc_color_clamp(float* c3);
float orc_u8tofloat(uint8_t x);
/* deterministic sin/cos used by both sides (see p3d_oracle.cpp "detmath") */
double orc_det_sin(double x);
double orc_det_cos(double x);
/* rng stream probe: first n 31-bit draws of the (pixel, sample) stream */
int orc_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out);
uint8_t orc_u8fromfloat(float x); /* maths.h:81-86 */

#ifdef __cplusplus
}
#endif
#endif
