/*
 * p3d.h — C-ABI of the MI355X-native per-pixel ray-trace hot path.
 *
 * This is the drop-in boundary.  The reference (fmbnicola/P3D-RayTracer) has no
 * FFI: the seam is the in-process call that renderScene() (Raytracing/main.cpp:694)
 * makes once per pixel sample,
 *     rayTracing(ray, MAX_DEPTH, 1.0, i, j)            main.cpp:795,811
 *     Radiance  (ray, MAX_DEPTH, 1.0, i, j, seed)      main.cpp:792
 * against process globals (Scene* scene, Grid grid, BVH bvh: main.cpp:67-69) and the
 * compile-time option set of constants.h:6-45.  A per-ray FFI is meaningless on a
 * GPU, so one call here replaces the whole pixel x sample loop body of
 * main.cpp:747-820 for a tile of the image: primary-ray generation
 * (camera.h:65-115), closest-hit / any-hit traversal (bvh.cpp:198-340,
 * grid.cpp:71-208, brute force main.cpp:116-124), the shape tests
 * (scene.cpp:47-94,116-137,149-186,215-227; boundingBox.cpp:44-98), Whitted
 * shading (main.cpp:92-309) or the path tracer (main.cpp:313-516), sample
 * averaging (main.cpp:800), gamma (main.cpp:814-815) and the u8 pack
 * (maths.h:81-86).
 *
 * Plain C types only: no C++ classes, no torch types.  All functions return 0 on
 * success and a negative p3d_status on failure; they never throw and never
 * exit().  There is NO CPU fallback behind this ABI: without a HIP device every
 * render/trace entry point fails with P3D_ERR_NO_DEVICE.
 */
#ifndef P3D_H
#define P3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_ABI_VERSION 4u /* round 4: p3d_stats.handoff_dense_retry; the test hooks left this header (csrc/p3d_debug.h) */

typedef enum p3d_status {
  P3D_OK = 0,
  P3D_ERR_INVALID = -1,     /* bad argument / inconsistent descriptor           */
  P3D_ERR_NO_DEVICE = -2,   /* no HIP device, or HIP runtime error (see last_error) */
  P3D_ERR_UNSUPPORTED = -3, /* option combination not implemented on the device  */
  P3D_ERR_CAPACITY = -4,    /* traversal stack / scene exceeds device limits     */
  P3D_ERR_IO = -5           /* .p3f file could not be opened                     */
} p3d_status;

/* Object kinds of scene.h:101-179 (Sphere, Triangle, aaBox, Plane). */
typedef enum p3d_prim_type {
  P3D_PRIM_SPHERE = 0,
  P3D_PRIM_TRIANGLE = 1,
  P3D_PRIM_BOX = 2,
  P3D_PRIM_PLANE = 3
} p3d_prim_type;

/* constants.h:41 — enum accel_struct {None, UGrid, Bvh}; same numeric values. */
typedef enum p3d_accel { P3D_ACCEL_NONE = 0, P3D_ACCEL_GRID = 1, P3D_ACCEL_BVH = 2 } p3d_accel;
/* constants.h:42 — enum sample_mode {jitter, tent}. */
typedef enum p3d_sample_mode { P3D_SAMPLE_JITTER = 0, P3D_SAMPLE_TENT = 1 } p3d_sample_mode;
/* constants.h:36 — PATHTRACING false/true. */
typedef enum p3d_integrator { P3D_WHITTED = 0, P3D_PATHTRACE = 1 } p3d_integrator;

/*
 * One scene object, in Scene::objects push order (scene.cpp:296-299).  The index
 * of a record in the array IS the hit ID reported by the renderer.
 *   sphere   : v[0..2] = center, v[3] = radius                   (scene.h:139-162)
 *   triangle : v[0..8] = P0,P1,P2 ; n = unit normal as the ctor computes it
 *              (scene.cpp:17-18) ; bmin/bmax = Min/Max -/+ EPSILON (scene.cpp:21-34)
 *   box      : v[0..2] = min, v[3..5] = max                      (scene.cpp:205-209)
 *   plane    : v[0..2] = PN (unit normal), v[3..5] = A            (scene.cpp:102-113)
 * bmin/bmax hold GetBoundingBox() for every kind (sphere: center -/+ r,
 * scene.cpp:194-198; plane: the default [-1,1]^3 box, scene.h:114).
 */
typedef struct p3d_prim {
  float v[9];
  uint32_t type;     /* p3d_prim_type */
  uint32_t material; /* index into p3d_scene_desc.materials */
  uint32_t reserved0;
  float n[3];
  uint32_t reserved1;
  float bmin[3];
  uint32_t reserved2;
  float bmax[3];
  uint32_t reserved3;
} p3d_prim; /* 96 bytes */

/* Material (scene.h:34-71).  reflection == specular: m_Refl = Ks (scene.h:42). */
typedef struct p3d_material {
  float diff_color[3];
  float diffuse;      /* Kd */
  float spec_color[3];
  float specular;     /* Ks */
  float shine;
  float transmittance; /* T */
  float refr_index;    /* ior */
  float reflection;    /* = Ks */
  float emission[3];
  float reserved;
} p3d_material; /* 64 bytes */

/* Light (scene.h:73-81). */
typedef struct p3d_light {
  float position[3];
  float reserved0;
  float color[3];
  float reserved1;
} p3d_light; /* 32 bytes */

/* Camera state after the constructor of camera.h:34-63 has run. */
typedef struct p3d_camera {
  float eye[3];
  float plane_dist;
  float u[3];
  float w;           /* view-window width  */
  float v[3];
  float h;           /* view-window height */
  float n[3];
  float focal_ratio;
  float aperture;    /* Aperture_ratio * (w / res_x), camera.h:59 */
  int32_t res_x;
  int32_t res_y;
  int32_t reserved;
} p3d_camera; /* 80 bytes */

/*
 * BVH node, 32 bytes, in the order BVH::build_recursive pushes them
 * (bvh.cpp:185-194): node 0 is the root and the two children of an inner node
 * are adjacent at [index, index+1].
 *   inner: count_leaf == 0,                 index = left child
 *   leaf : count_leaf = 0x80000000 | n_objs, index = first entry in bvh_prim_index
 */
typedef struct p3d_bvh_node {
  float bmin[3];
  uint32_t index;
  float bmax[3];
  uint32_t count_leaf;
} p3d_bvh_node;
#define P3D_BVH_LEAF 0x80000000u

/* Uniform grid as Grid::Build lays it out (grid.cpp:3-68): cell (ix,iy,iz) is
 * entry ix + nx*iy + nx*ny*iz; its objects are cell_items[cell_start[c] ..
 * cell_start[c+1]) in insertion (= object) order. */
typedef struct p3d_grid_desc {
  float bmin[3];
  int32_t nx;
  float bmax[3];
  int32_t ny;
  int32_t nz;
  uint32_t n_cells;          /* nx*ny*nz */
  uint32_t n_items;          /* cell_start[n_cells] */
  uint32_t reserved;
  const uint32_t* cell_start; /* n_cells + 1 entries */
  const uint32_t* cell_items; /* object indices */
} p3d_grid_desc;

/* Everything the device needs; all arrays are host memory owned by the caller
 * and are copied by p3d_scene_create. */
typedef struct p3d_scene_desc {
  uint32_t abi_version; /* P3D_ABI_VERSION */
  uint32_t n_prims;
  uint32_t n_materials;
  uint32_t n_lights;
  const p3d_prim* prims;
  const p3d_material* materials;
  const p3d_light* lights;
  p3d_camera camera;
  float background[3];       /* bclr, Scene::GetBackgroundColor (scene.h:189) */
  uint32_t n_bvh_nodes;      /* 0 = no BVH supplied */
  const p3d_bvh_node* bvh_nodes;
  const uint32_t* bvh_prim_index; /* permuted BVH::objs (bvh.cpp:84), object indices */
  uint32_t n_bvh_prim_index;
  uint32_t bvh_max_depth;    /* levels, root = 1 */
  uint32_t has_grid;
  uint32_t reserved;
  p3d_grid_desc grid;
} p3d_scene_desc;

/* Runtime form of the compile-time options of constants.h:6-45.  p3d_config_default()
 * fills in the reference's shipped values, except SKYBOX: it defaults to 0 (miss = bclr)
 * because the cubemap has to be supplied separately (p3d_scene_set_skybox). */
/* Order in which the 8x8-pixel tiles of a launch are handed to the GPU.  COST (default, 0):
 * tiles whose pixels spawn reflection / refraction chains go first, so the long-running
 * tiles do not end up alone at the end of the frame; the order comes from a small estimate
 * pass that is memoised per (scene, max_depth, accel, tile).  FRAME: image order. */
#define P3D_TILE_ORDER_COST 0u
#define P3D_TILE_ORDER_FRAME 1u

/*
 * What becomes of BVH::hit_stack (bvh.cpp:86), which the reference keeps as ONE member for the
 * whole frame: an any-hit query that returns `true` (bvh.cpp:322) leaves its entries behind, and
 * they are still there when the NEXT pixel's primary ray is traced (serial pixel loop,
 * main.cpp:747-751); that closest-hit query drains them (bvh.cpp:256-274), which can re-normalise
 * its ray copy once more (ray.h:16-18) and move the hit point by an ulp.
 *   LITERAL (default, 0): every pixel starts with exactly the entries the reference's serial loop
 *     would hand it (x fastest, rows ascending, samples of a pixel in order).  The GPU renders all
 *     pixels speculatively on an empty stack, keeps every pixel's leftover, and re-renders the
 *     pixels whose first closest hit changes under the predecessor's leftover, round by round
 *     until nothing changes (DESIGN.md "hit_stack hand-off").  Zero-weight reflection rays of
 *     transmissive materials (Kr = 1/2*(Rs+Rp) = 0, main.cpp:282,290-300) are traced for their
 *     effect on the stack, as the reference traces them.  Frames are bit-identical to the
 *     reference's order of execution.  Only Whitted + accel = Bvh have such a stack.
 *   PER_PIXEL (1): the stack is emptied at every primary sample and zero-weight rays are skipped:
 *     one launch, no hand-off; hit IDs as LITERAL, colours differ by a few 1e-5 on sphere scenes.
 */
#define P3D_STACK_LITERAL 0u
#define P3D_STACK_PER_PIXEL 1u

/*
 * How the chain of rayTracing calls of a pixel (main.cpp:247-300) is put on the GPU.  MEGAKERNEL: one lane follows its
 * pixel from the primary ray to the end of the chain.  PER_LEVEL: one launch per chain level; the surviving child rays
 * are compacted into a queue and put in order (origin cell, direction octant) between levels, so that every wave of a
 * reflection level starts with 64 live rays.  Only for Whitted frames without anti-aliasing over a BVH too big for LDS.
 * AUTO (default) = MEGAKERNEL: on the scenes measured the per-level launches issue 13 % fewer instructions but take
 * longer (100k triangles 2048x2048: 25.3 ms against 19.6 ms) — what keeps lanes idle there is the spread of traversal
 * lengths inside one query, not dead pixels.  Same queries in the same per-pixel order either way: the
 * frames are bit-identical.
 */
#define P3D_CHAIN_AUTO 0u
#define P3D_CHAIN_MEGAKERNEL 1u
#define P3D_CHAIN_PER_LEVEL 2u

/* p3d_config.handoff_records.  COMPACT (default): leftovers share a pool of 8 entries (64 bytes) per pixel of the tile on
 * average - most pixels leave nothing, a few leave up to lights x tree depth entries; a frame that needs more fails with
 * P3D_ERR_CAPACITY (the host-buffer call p3d_render_tile then renders it again with DENSE records by itself; after a
 * device-buffer call ask p3d_scene_status and repeat the call with DENSE).  DENSE: room for the worst case of every pixel
 * (2 x lights x tree depth x 8 bytes per pixel: 2.7 GB for the 100k-triangle frame at 2048 x 2048), cannot run out. */
#define P3D_HANDOFF_COMPACT 0u
#define P3D_HANDOFF_DENSE 1u

/* p3d_config.debug_view */
#define P3D_DEBUG_NONE 0u
#define P3D_DEBUG_TEST_INTERSECT 1u /* TEST_INTERSECT (constants.h:18): every hit is Color(1,0,0) (main.cpp:156, 359) */
#define P3D_DEBUG_DEPTH_MAP 2u      /* DEPTH_MAP (constants.h:33): rayTracing without an acceleration structure returns
                                       the grey value remap(5, 20, 1, 0, min_t), clamped (main.cpp:86-88, 127-139);
                                       ignored with a grid or BVH and by the path tracer, as in the reference */

typedef struct p3d_config {
  uint32_t integrator;    /* PATHTRACING        constants.h:36  */
  uint32_t accel;         /* acl_str            constants.h:44  */
  int32_t max_depth;      /* MAX_DEPTH          constants.h:6   */
  uint32_t spp_sqrt;      /* SPP (sqrt of spp)  constants.h:12  */
  uint32_t antialiasing;  /* ANTIALIASING       constants.h:24  */
  uint32_t depth_of_field;/* DEPTH_OF_FIELD     constants.h:27  */
  uint32_t sample_disk;   /* SAMPLE_DISK        constants.h:21  */
  uint32_t soft_shadows;  /* SOFT_SHADOWS       constants.h:9   */
  uint32_t sample_mode;   /* s_mode             constants.h:45  */
  float light_side;       /* LIGHT_SIDE         constants.h:15  */
  float gamma;            /* GAMMA              constants.h:38  */
  uint32_t collect_stats; /* 1: fill the test/ray counters of p3d_stats (slower kernel) */
  uint32_t skybox;        /* SKYBOX             constants.h:30: a miss returns the cubemap texel
                             (main.cpp:145,351) instead of bclr; needs p3d_scene_set_skybox */
  uint32_t tile_order;    /* P3D_TILE_ORDER_*: scheduling only, never changes a result */
  uint64_t seed;          /* replaces set_rand_seed(time*time), main.cpp:722:
                             every (pixel, sample) draws from its own stream */
  uint32_t stack_mode;    /* P3D_STACK_*: BVH::hit_stack across pixels (bvh.cpp:86) */
  uint32_t chain_launch;  /* P3D_CHAIN_*: how the reflect / refract chain is launched; never changes a result */
  uint32_t debug_view;    /* P3D_DEBUG_*: the reference's two debug switches (constants.h:18,33), 0 as shipped */
  uint32_t handoff_records; /* P3D_HANDOFF_*: storage of the per-pixel hit_stack leftovers under P3D_STACK_LITERAL; never changes a result */
} p3d_config;

/*
 * Region of the image a call renders.  Rows are numbered as the reference
 * numbers them: y = 0 is the BOTTOM image row (main.cpp:747, camera.h:71).
 * Local row r of the output buffers maps to image row
 *     y = y0 + (r / stripe_h) * stripe_h * stripe_stride + (r % stripe_h)
 * so that one rank of an N-rank job renders every N-th stripe of stripe_h rows
 * (stripe_stride = N, y0 = rank * stripe_h); stripe_stride = 1 (or stripe_h = 0)
 * is a plain rectangle.  Output buffers are w*h, local row 0 first.
 * Under P3D_STACK_LITERAL a tile that is not the whole frame still renders the pixels of the serial frame: for every
 * row whose predecessor in the frame lies outside the tile the call first finds, among the frame pixels in front of
 * that row, one whose result provably does not depend on the hit_stack it finds (no primitive of the scene nearer than
 * its own first hit, ray direction stable under re-normalisation) and renders the pixels from there to the row for
 * what they leave on the stack.  If no such pixel is found among the 16 nearest candidates the call FAILS with
 * P3D_ERR_CAPACITY (it never returns a frame that is only probably right); stripes and sub-rectangles of every rank
 * count are therefore bit-identical to the whole frame, or the caller is told.
 */
typedef struct p3d_tile {
  int32_t x0, y0, w, h;
  int32_t stripe_h;
  int32_t stripe_stride;
} p3d_tile;

/* Counters of one call.  A "ray" is one traversal query (closest-hit or
 * any-hit).  The test counters feed the algorithmic-bytes figure of DESIGN.md.
 * They count what the FINAL frame traced, query by query, as the reference's serial loop would: under
 * P3D_STACK_LITERAL a pixel that was rendered again counts once, with its last render, a pixel whose first closest hit
 * was only re-traced on its predecessor's leftover counts that query's tests as re-traced (the stale entries it
 * visits), and the zero-weight reflection rays are rays.  max_stack is the deepest stack of anything that was traced,
 * speculative passes included.  What the hand-off cost on top is in the handoff_* fields and handoff_ms. */
typedef struct p3d_stats {
  uint64_t rays_primary;
  uint64_t rays_shadow;
  uint64_t rays_reflect;
  uint64_t rays_refract;
  uint64_t rays_bounce;  /* path-tracer continuation rays */
  uint64_t rays_light;   /* path-tracer light-visibility rays */
  uint64_t node_tests;   /* AABB::intercepts calls on BVH nodes */
  uint64_t sphere_tests;
  uint64_t tri_tests;
  uint64_t box_tests;
  uint64_t plane_tests;
  uint64_t shaded_hits;
  uint64_t pixels;
  uint64_t max_stack;    /* deepest traversal stack seen (entries) */
  double kernel_ms;      /* HIP-event time of the kernel(s) of this call */
  /* P3D_STACK_LITERAL only: the hand-off of BVH::hit_stack from pixel to pixel */
  uint64_t handoff_checked;  /* pixels whose first closest hit was re-traced on the predecessor's leftover */
  uint64_t handoff_redone;   /* pixels rendered again because that hit changed */
  uint64_t handoff_rounds;   /* rounds until no leftover changed any more */
  double pass1_ms;           /* of kernel_ms: the speculative pass over all pixels (the launches of pass 1 alone) */
  double handoff_ms;         /* of kernel_ms: check, redo and fixed-point launches */
  uint64_t handoff_dense_retry; /* p3d_render_tile only: 1 = the COMPACT leftover pool was too small for this frame and the
                                   call rendered it a second time with P3D_HANDOFF_DENSE records (twice the time, and the
                                   dense records allocated: lights x tree depth entries per pixel) */
} p3d_stats;

/* Device-resident scene, one per HIP device.  A p3d_scene also owns per-launch scratch and the
 * memoised tile schedules, so calls on ONE scene must not overlap in time from several host
 * threads (as the reference's renderScene() is single-threaded); different scenes are independent. */
typedef struct p3d_scene p3d_scene;

/* ---- library ---- */
uint32_t p3d_abi_version(void);
const char* p3d_last_error(void);          /* thread-local message of the last failure */
int p3d_device_count(void);                /* >= 0, or negative p3d_status */
void p3d_config_default(p3d_config* cfg);  /* constants.h:6-45 as shipped */

/* ---- scene on the device ---- */
/* Uploads the flattened scene to HBM of HIP device `device` (arrays copied). */
int p3d_scene_create(const p3d_scene_desc* desc, int device, p3d_scene** out);
/*
 * Same, but the BVH is built ON the GPU (linear BVH: Morton codes of the bounding-box centres,
 * radix sort, Karras hierarchy, bottom-up refit) instead of being uploaded; the descriptor's bvh_*
 * arrays are ignored, prims[].bmin/bmax must hold GetBoundingBox().  NOT the reference's tree
 * (bvh.cpp:89-196): closest-hit queries find the same nearest intersection (exact-t ties aside),
 * but the any-hit quirk of bvh.cpp:329-334 depends on the tree shape, so Whitted shadow feelers
 * can differ from a scene created with the reference-exact tree.  For callers who want correct
 * closest hits without waiting for the host build.  *build_ms (may be NULL): GPU time of the build.
 */
int p3d_scene_create_device_bvh(const p3d_scene_desc* desc, int device, p3d_scene** out, float* build_ms);
void p3d_scene_destroy(p3d_scene* scene);

/*
 * Cubemap for miss shading = the skybox_img[6] array that Scene::LoadSkybox fills
 * (scene.cpp:329-377, scene.h:218-223).  The caller decodes the six images (the reference
 * uses DevIL for that; JPEG decoding is not part of this library) and hands over raw bytes:
 * bpp 3 (RGB) or 4 (RGBA), row 0 = BOTTOM image row (IL_ORIGIN_LOWER_LEFT, scene.cpp:344-345),
 * face order RIGHT, LEFT, TOP, BOTTOM, FRONT, BACK (enum CubeMap, scene.h:28).  The texel
 * lookup itself (Scene::GetSkyboxColor, scene.cpp:379-457) runs in the kernels.
 */
typedef struct p3d_skybox_face {
  const uint8_t* img;
  uint32_t res_x, res_y, bpp;
  uint32_t reserved;
} p3d_skybox_face;
typedef struct p3d_skybox_desc {
  p3d_skybox_face face[6];
} p3d_skybox_desc;
int p3d_scene_set_skybox(p3d_scene* scene, const p3d_skybox_desc* sky);

/* ---- the hot path ---- */
/*
 * Renders one tile.  Replaces the loop body main.cpp:753-820 for every pixel of
 * the tile.  Host-buffer form: synchronous; any output pointer may be NULL.
 *   rgb    : w*h*3 float, linear colour after sample averaging (main.cpp:800),
 *            BEFORE gamma
 *   hit_id : w*h int32, object index hit by the pixel's first primary ray, -1 = miss
 *   rgb8   : w*h*3 uint8, after gamma + u8fromfloat (main.cpp:814-820) = img_Data
 */
int p3d_render_tile(p3d_scene* scene, const p3d_config* cfg, const p3d_tile* tile,
                    float* rgb, int32_t* hit_id, uint8_t* rgb8, p3d_stats* stats);
/*
 * Device-buffer form: the output pointers are HBM addresses on the scene's
 * device; the kernel is enqueued on `hip_stream` (a hipStream_t, NULL = default
 * stream) and the call returns without synchronising unless `stats` is non-NULL
 * (then it waits for the kernel and fills kernel_ms and, with
 * cfg->collect_stats, the counters).
 * All launches on ONE scene share its scratch (level records, stack spill area, hit_stack hand-off records, work lists,
 * counters): enqueue them on one stream, or order the streams with events; two renders of the same scene in flight at
 * once on different streams would overwrite each other's records.  Different scenes are independent.
 */
int p3d_render_tile_device(p3d_scene* scene, const p3d_config* cfg, const p3d_tile* tile,
                           float* d_rgb, int32_t* d_hit_id, uint8_t* d_rgb8,
                           void* hip_stream, p3d_stats* stats);

/*
 * The launches behind pass 1 of a P3D_STACK_LITERAL frame - the hit_stack hand-off rounds: a chain of short launches that
 * depend on each other and carry a few per cent of the frame's work - can be given a stream of their own, so that what the
 * caller enqueues behind the frame on `hip_stream` (other scenes' frames) is not held up by them: with a tail stream set,
 * p3d_render_tile_device without `stats` enqueues clear + pass 1 on `hip_stream` and everything behind them on the tail stream
 * (ordered by an event).  The frame's outputs are complete when the TAIL stream has passed the call: p3d_scene_join makes
 * `hip_stream` wait for that (host_wait = 0) or the calling thread (host_wait != 0); the next call on the same scene waits by
 * itself.  NULL switches it off.  No counterpart in the reference (one frame at a time on one core); what `bench.py` uses to
 * keep the chip busy with several frames in flight.
 */
int p3d_scene_set_tail_stream(p3d_scene* scene, void* tail_hip_stream);
int p3d_scene_join(p3d_scene* scene, void* hip_stream, int host_wait);

/*
 * Errors a kernel detects while it runs (a hit_stack leftover that outgrew its record, a work list of the hit_stack
 * hand-off that overflowed or did not run empty within its round bound, a row of a stripe or sub-rectangle whose
 * incoming hit_stack could not be established, a sample hand-out loop that reached its trip bound and would write pixels
 * with samples missing) raise a flag on the device.  The host-buffer
 * call and every call with `stats` turn it into P3D_ERR_CAPACITY themselves; after device-buffer calls without
 * `stats` ask here: waits for the scene's device, returns P3D_OK or P3D_ERR_CAPACITY and clears the flag.
 */
int p3d_scene_status(p3d_scene* scene);
/*
 * Batched traversal queries — device counterparts of BVH::intersect_bvh
 * (bvh.cpp:198), Grid::Traverse (grid.cpp:71) and the brute-force loop
 * (main.cpp:116-124) for closest hit, and of BVH::bool_intersect_bvh
 * (bvh.cpp:278), Grid::Traverse(ray) (grid.cpp:154) and main.cpp:208-216 for any
 * hit.  Every ray starts with an empty traversal stack.  Host buffers:
 *   origin, direction : n*3 float (direction used as given, not normalised)
 *   hit_id : n int32 (-1 = miss) ; t : n float, the traversal's tmin / min_t (bvh.cpp:246, grid.cpp:100,
 *   main.cpp:120), FLT_MAX on a miss (may be NULL) ; hit_point : n*3 float (may be NULL)
 *   occluded : n uint8
 */
int p3d_trace_closest(p3d_scene* scene, uint32_t accel, uint32_t n, const float* origin,
                      const float* direction, int32_t* hit_id, float* t, float* hit_point);
int p3d_trace_any(p3d_scene* scene, uint32_t accel, uint32_t n, const float* origin,
                  const float* direction, uint8_t* occluded);
/*
 * Per-object queries — device counterparts of the virtual Object::intercepts(Ray&, float&)
 * (scene.cpp:47-94,116-137,149-186,215-227) and Object::getNormal(Vector) (scene.cpp:41-44,
 * 139-142,188-192,229-267) for object `object` (index in Scene::objects) and n rays / points.
 *   direction : n*3 float, IN and OUT — Sphere::intercepts normalises the ray in place
 *               (scene.cpp:156, ray.h:16-18); the other kinds leave it as it was
 *   hit : n uint8 ; t : n float (written where hit, as the reference writes its out-parameter)
 */
int p3d_object_intercepts(p3d_scene* scene, uint32_t object, uint32_t n, const float* origin,
                          float* direction, uint8_t* hit, float* t);
int p3d_object_normal(p3d_scene* scene, uint32_t object, uint32_t n, const float* point, float* normal);
/* Scene::GetSkyboxColor (scene.cpp:379-457) for n ray directions: rgb n*3 float.  Needs p3d_scene_set_skybox. */
int p3d_skybox_color(p3d_scene* scene, uint32_t n, const float* direction, float* rgb);

/* ---- host side: .p3f loader and acceleration-structure builders ---- */
/*
 * Host scene = Scene::load_p3f (scene.cpp:472-628) + the accel builds that
 * renderScene() does first (main.cpp:701-720): BVH::build (bvh.cpp:89-196) and
 * Grid::Build (grid.cpp:3-68).  These run on the host (as they do in the
 * reference) and produce the flat p3d_scene_desc that p3d_scene_create uploads.
 */
typedef struct p3d_host_scene p3d_host_scene;

#define P3D_LOAD_LEGACY_F11 1u /* accept the older 11-number `f` line (no emission);
                                  the shipped parser breaks on it */

int p3d_host_scene_load(const char* p3f_path, uint32_t flags, p3d_host_scene** out);
void p3d_host_scene_destroy(p3d_host_scene* hs);
/* Re-runs the Camera constructor (camera.h:34-63) with another resolution —
 * what editing the `resolution` line of the .p3f would do.  rx,ry <= 0 keeps. */
int p3d_host_scene_set_resolution(p3d_host_scene* hs, int32_t res_x, int32_t res_y);
/* Same for the `aperture` / `focal` entries of the `v` block. */
int p3d_host_scene_set_lens(p3d_host_scene* hs, float aperture_ratio, float focal_ratio);
/* Replaces every light by SPP x SPP jittered copies (main.cpp:725-745); used for
 * SOFT_SHADOWS without ANTIALIASING. */
int p3d_host_scene_replicate_lights(p3d_host_scene* hs, uint32_t spp_sqrt, float light_side);
/* Builds (once) the requested structures and returns the descriptor; the
 * pointer stays valid until the host scene is destroyed or modified. */
int p3d_host_scene_desc(p3d_host_scene* hs, int build_bvh, int build_grid,
                        const p3d_scene_desc** out);
/* Points the query methods of the host classes (C++: Object::intercepts / getNormal, BVH::intersect_bvh /
 * bool_intersect_bvh, Grid::Traverse, Scene::GetSkyboxColor — p3d-raytracer_amd/host/scene_model.hpp) at the
 * device scene created from this host scene's descriptor, and uploads the cubemap the loader read for an `env`
 * line (p3d_host_scene_has_skybox) to it.  NULL unbinds.
 * The forwards exist so that code written against the reference's classes compiles and gives the reference's
 * answers; each call is a kernel launch for ONE ray (about 10 us).  Anything that asks more than a handful of
 * questions should ask them in one batch: p3d_trace_closest / p3d_trace_any / p3d_object_intercepts /
 * p3d_object_normal / p3d_skybox_color take n rays per call, and so do the array overloads of BVH::intersect_bvh /
 * BVH::bool_intersect_bvh on the host class (accel_build.hpp). */
int p3d_host_scene_bind_device(p3d_host_scene* hs, p3d_scene* scene);
/* 1 if a cubemap is loaded: the `.p3f` had an `env <dir>` line (scene.cpp:605-610) and the folder was found (relative to
 * the working directory, as in the reference, or next to the scene file), or p3d_host_scene_load_skybox was called. */
int p3d_host_scene_has_skybox(p3d_host_scene* hs);
/* Scene::LoadSkybox (scene.cpp:329-377): <sky_dir>/{right,left,top,bottom,front,back}.jpg, decoded by the library
 * (baseline JPEG - sequential DCT, Huffman, 8 bit, grey or YCbCr with 1x1 / 2x1 / 2x2 luma sampling - following the IJG
 * library's default path: the six shipped faces come out byte for byte as PIL's libjpeg-turbo decodes them; DevIL's own
 * version is unpinned in the reference).  A face that is not there as .jpg is read from a binary .ppm of that name.
 * P3D_ERR_IO names the face that could not be read (the reference exit(0)s, scene.cpp:354-357).  The faces go to the
 * device scene with p3d_host_scene_bind_device. */
int p3d_host_scene_load_skybox(p3d_host_scene* hs, const char* sky_dir);
/* Scene::skybox_img[face] (scene.h:218-223; RIGHT 0, LEFT 1, TOP 2, BOTTOM 3, FRONT 4, BACK 5): the decoded RGB bytes,
 * bottom row first (IL_ORIGIN_LOWER_LEFT, scene.cpp:344-345), owned by the host scene. */
int p3d_host_scene_skybox_face(p3d_host_scene* hs, int face, const uint8_t** img, uint32_t* res_x, uint32_t* res_y);

#ifdef __cplusplus
}
#endif
#endif /* P3D_H */
