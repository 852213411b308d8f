#!/usr/bin/env python3
"""Per-workgroup clock trace of one frame (the evidence behind DESIGN.md "Tile schedule").

  make -C p3d-raytracer_amd debuglibs
  P3D_LIB=$PWD/build/variants/libp3d_timeline.so python profiles/tools/wave_timeline.py \\
      tests/golden/scenes/balls_low.p3f 1024 whitted 4 [frame|cost]
  ... scenes/cornell.p3f 1024 pt 16

Every workgroup of the instrumented library stores wall_clock64() (100 MHz) at its start and end
and its HW id.  Prints the frame span, the wave-duration distribution, resident waves per tenth of
the frame and how the first workgroups were placed."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import p3d_amd as p3d  # noqa: E402

scene, res, kind, arg = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
order = sys.argv[5] if len(sys.argv) > 5 else "frame"
hs = p3d.HostScene(scene)
hs.set_resolution(res, res)
dev = p3d.DeviceScene(hs, bvh=True)
tile_order = p3d.TILE_ORDER_COST if order == "cost" else p3d.TILE_ORDER_FRAME
if kind == "pt":
    cfg = p3d.pathtrace_config(accel=p3d.ACCEL_BVH, spp_sqrt=arg, tile_order=tile_order)
else:
    cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=arg, tile_order=tile_order)
nblk = ((res + 3) // 4) ** 2 if kind == "pt" else ((res + 7) // 8) ** 2  # upper bound (the path tracer uses 4x4-pixel tiles from 16 spp)
tl = torch.zeros(3 * (nblk + 64), dtype=torch.int64, device="cuda")
if not hasattr(dev._L, "p3d_debug_set_timeline"):
    sys.exit("P3D_LIB must point at build/variants/libp3d_timeline.so (make -C p3d-raytracer_amd debuglibs)")
assert dev._L.p3d_debug_set_timeline(C.c_void_p(tl.data_ptr())) == 0
rgb = torch.empty(res * res * 3, dtype=torch.float32, device="cuda")
tile = p3d.Tile(0, 0, res, res, 0, 1)
for _ in range(4):  # with order == cost: the first launch records, the later ones are scheduled
    tl.zero_()
    torch.cuda.synchronize()
    dev.render_device(cfg, tile, rgb.data_ptr())
    torch.cuda.synchronize()
t = tl.cpu().numpy().reshape(-1, 3)[:nblk]
t = t[t[:, 0] != 0]
t0 = (t[:, 0] - t[:, 0].min()) * 0.01
t1 = (t[:, 1] - t[:, 0].min()) * 0.01
dur = t1 - t0
print("%s %dx%d %s %d, %s order: span %.1f us, last workgroup starts at %.1f us" % (os.path.basename(scene), res, res, kind, arg, order, t1.max(), t0.max()))
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; mean resident waves %.0f"
      % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max(), dur.sum() / t1.max()))
edges = np.linspace(0, t1.max(), 11)
print("resident waves per tenth of the frame:",
      [int((np.minimum(t1, b) - np.maximum(t0, a)).clip(min=0).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])])
hw = t[:, 2]
xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
cu = xcc * 10000 + ((hwid >> 13) & 7) * 1000 + ((hwid >> 12) & 1) * 100 + ((hwid >> 8) & 0xF)
first = min(1024, len(cu))
print("first %d workgroups on %d distinct CUs, %d distinct SIMDs" % (first, len(np.unique(cu[:first])), len(np.unique(cu[:first] * 10 + ((hwid[:first] >> 4) & 3)))))
