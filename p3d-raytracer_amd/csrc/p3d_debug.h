/* p3d_debug.h — test hooks of libp3d.so.  NOT part of the drop-in boundary (include/p3d.h) and not installed with it:
 * nothing a maintainer of the reference would call.  The GPU tests use them to prove that the device-side error paths
 * fire (tests/test_gpu_parity.py).  The entry point is refused (P3D_ERR_UNSUPPORTED) unless the process was started
 * with P3D_TEST_HOOKS=1 in its environment, and what it sets belongs to ONE scene: no process-wide state. */
#ifndef P3D_DEBUG_H
#define P3D_DEBUG_H
#include "p3d.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct p3d_debug_limits {
  uint32_t trip_bound;     /* trip bound of the four-lanes-per-pixel sample loops (0 = the real bound) */
  uint32_t max_rounds;     /* round bound of the hit_stack hand-off (0 = the real bound, min(units + 2, 4096)) */
  uint32_t halo_chain;     /* frame pixels the search in front of a row of a stripe / sub-rectangle may collect (0 = 16) */
  uint32_t leftover_pool;  /* entries of the pool of COMPACT hit_stack leftover records (0 = 8 per pixel, at least 65536) */
} p3d_debug_limits;
/* NULL limits = all real again. */
int p3d_debug_scene_limits(p3d_scene* scene, const p3d_debug_limits* limits);
#ifdef __cplusplus
}
#endif
#endif
