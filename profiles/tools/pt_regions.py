#!/usr/bin/env python3
"""Where the path tracer's bounce loop spends its wave time.

  make -C p3d-raytracer_amd debuglibs
  P3D_LIB=$PWD/build/variants/libp3d_ptprof.so python profiles/tools/pt_regions.py scenes/cornell.p3f 512 4

s_memtime deltas, entries and active lanes per region of pt_kernel's loop, summed over all waves."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import p3d_amd as p3d  # noqa: E402

scene, res, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
hs = p3d.HostScene(scene)
hs.set_resolution(res, res)
dev = p3d.DeviceScene(hs, bvh=True)
cfg = p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=spp)
prof = torch.zeros(30, dtype=torch.int64, device="cuda")
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
p = prof.cpu().numpy().reshape(10, 3).astype(np.float64)
names = ["loop head", "new sample / resume deferred branch", "closest hit (bounce ray)", "hit/miss, material, normal, roulette",
         "diffuse: basis + light sample", "closest hit (light sample)", "diffuse: accumulate + next ray", "mirror", "dielectric", "epilogue"]
print("%-40s %8s %10s %9s" % ("region", "time %", "entries", "lanes/64"))
for i, n in enumerate(names):
    if p[i, 2] > 0:
        print("%-40s %8.1f %10d %9.2f" % (n, 100 * p[i, 0] / p[:, 0].sum(), p[i, 2], p[i, 1] / p[i, 2] / 64))
