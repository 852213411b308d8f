#!/usr/bin/env python3
"""Where the Whitted kernel spends its wave time.

  make -C p3d-raytracer_amd debuglibs
  P3D_LIB=$PWD/build/variants/libp3d_ptprof.so python profiles/tools/whitted_regions.py tests/golden/scenes/balls_low.p3f 1024 4

s_memtime deltas, entries and active lanes per region of whitted_kernel's chain loop, summed over all waves.
The mark behind the closest-hit query is taken by the first lanes that leave the traversal loop, so
the second line is the time they then wait for the slowest lane of the wave (divergence).

CAVEAT (round 2): do not read the split between traversal and shading off this tool.  Every mark is an
`s_waitcnt 0` + `s_memtime`: outstanding stores (level records) are charged to whatever region comes next, and the
compiler moves code across the marks.  On balls_low it reports 5 % for the shadow feelers' traversals; a build that
answers the feelers without traversing runs in HALF the time (ablation builds).  Use it for
entries / active lanes per region; for time, build an ablation variant and time the kernel."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import p3d_amd as p3d  # noqa: E402

scene, res, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if scene == "tri100k":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "scenes"))
    import make_tri100k
    scene = "/tmp/tri100k_prof.p3f"
    if not os.path.exists(scene):
        make_tri100k.generate(scene, res=1024)
hs = p3d.HostScene(scene)
hs.set_resolution(res, res)
dev = p3d.DeviceScene(hs, bvh=True)
cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=depth, tile_order=p3d.TILE_ORDER_FRAME)
prof = torch.zeros(36, dtype=torch.int64, device="cuda")
if not hasattr(dev._L, "p3d_debug_set_pt_prof"):
    sys.exit("P3D_LIB must point at build/variants/libp3d_ptprof.so (make -C p3d-raytracer_amd debuglibs)")
assert dev._L.p3d_debug_set_pt_prof(C.c_void_p(prof.data_ptr())) == 0
rgb = torch.empty(res * res * 3, dtype=torch.float32, device="cuda")
tile = p3d.Tile(0, 0, res, res, 0, 1)
for _ in range(2):
    prof.zero_()
    torch.cuda.synchronize()
    dev.render_device(cfg, tile, rgb.data_ptr())
    torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(12, 3).astype(np.float64)
names = ["prologue: staging, primary ray", "closest hit", "closest hit: lanes done, waiting for the slowest", "light: direction, feeler set-up",
         "any hit (shadow feeler)", "light: Blinn-Phong, pow", "child ray (reflect / refract), level record", "fold (reads the level records back)", "output",
         "hit: normal, offset point", "before the light loop", ""]
print("%-44s %8s %10s %9s" % ("region", "time %", "entries", "lanes/64"))
for i, n in enumerate(names):
    if p[i, 2] > 0 or p[i, 0] > 0:
        print("%-44s %8.1f %10d %9.2f" % (n, 100 * p[i, 0] / p[:, 0].sum(), p[i, 2], p[i, 1] / max(p[i, 2], 1) / 64))
