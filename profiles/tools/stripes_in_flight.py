"""Rank 0's stripes of the BASELINE configs[3] frame on ONE GPU with 1 / 2 / 3 frames in flight (one device scene and stream each):
what a rank of a multi-GPU run gains from rendering frame k + 1 while frame k's slowest tiles finish (bench.py keeps two in flight
per rank).  usage (GPU box): python3 profiles/tools/stripes_in_flight.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_tri100k, p3d_amd as p3d
scene = "/tmp/tri100k_probe_%d.p3f" % os.getuid()
if not os.path.exists(scene): make_tri100k.generate(scene, res=1024)
res = 2048
hs = p3d.HostScene(scene); hs.set_resolution(res, res)
for stack in ("literal", "per_pixel"):
    cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=6, stack_mode=p3d.STACK_LITERAL if stack == "literal" else p3d.STACK_PER_PIXEL)
    for world in (8, 4, 2, 1):
        for nfl in (1, 2, 3):
            devs = [p3d.DeviceScene(hs, bvh=True) for _ in range(nfl)]
            streams = [torch.cuda.Stream() for _ in range(nfl)]
            tile = p3d.stripe_tile((res, res), 0, world, 8) if world > 1 else devs[0].full_tile()
            bufs = [torch.empty(tile.w * tile.h * 16, dtype=torch.uint8, device="cuda") for _ in range(nfl)]
            def frame(i):
                k = i % nfl
                devs[k].render_device(cfg, tile, d_rgb=bufs[k].data_ptr(), d_hit=bufs[k].data_ptr() + tile.w * tile.h * 12, stream=streams[k].cuda_stream)
            for i in range(2 * nfl): frame(i)
            torch.cuda.synchronize()
            n = 24
            t0 = time.perf_counter()
            for i in range(n): frame(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
            print("%-9s world %d rank 0, %d frame(s) in flight: %.3f ms per frame" % (stack, world, nfl, dt), flush=True)
            del devs
