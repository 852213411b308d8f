#!/usr/bin/env python3
"""Every rank's stripes of the BASELINE configs[3] frame on ONE GPU with 1 / 2 frames in flight (one device scene and stream each):
what a rank of a multi-GPU run gains from rendering frame k + 1 while frame k's slowest tiles finish - bench.py keeps two in
flight per rank - and what the SLOWEST rank of 2 / 4 / 8 makes of it (that rank paces a strong-scaled run).

usage (GPU box): python3 profiles/tools/stripes_in_flight.py [out.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_tri100k  # noqa: E402
import p3d_amd as p3d  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else None
scene = "/tmp/tri100k_probe_%d.p3f" % os.getuid()
if not os.path.exists(scene):
    make_tri100k.generate(scene, res=1024)
res = 2048
hs = p3d.HostScene(scene)
hs.set_resolution(res, res)
STREAM2 = torch.cuda.Stream()
report = {"workload": "100k random triangles 2048x2048, Whitted MAX_DEPTH=6, BVH (BASELINE configs[3])", "stripe_rows": 8, "modes": {}}
for stack in ("literal", "per_pixel"):
    cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=6, stack_mode=p3d.STACK_LITERAL if stack == "literal" else p3d.STACK_PER_PIXEL)
    # (the default stream + one pool stream, as bench.py does: HIP spreads streams over four hardware queues and two pool streams
    # can share one, in which case their frames simply queue up)
    streams = [torch.cuda.current_stream(), STREAM2]

    def ms_per_frame(tile, nfl, n=16):
        # (fresh device scenes per tile: a scene memoises 16 tile schedules and one set of row starts, and this script renders more tiles than that)
        devs = [p3d.DeviceScene(hs, bvh=True) for _ in range(nfl)]
        bufs = [torch.empty(tile.w * tile.h * 16, dtype=torch.uint8, device="cuda") for _ in range(nfl)]

        def frame(i):
            k = i % nfl
            devs[k].render_device(cfg, tile, d_rgb=bufs[k].data_ptr(), d_hit=bufs[k].data_ptr() + tile.w * tile.h * 12, stream=streams[k].cuda_stream)
        for i in range(3 * nfl):
            frame(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            frame(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    whole = {nfl: ms_per_frame(p3d.Tile(0, 0, res, res, 0, 1), nfl, 8) for nfl in (1, 2)}
    entry = {"whole_frame_ms": {str(k): round(v, 3) for k, v in whole.items()}, "worlds": {}}
    for world in (2, 4, 8):
        per = {1: [], 2: []}
        for rank in range(world):
            tile = p3d.stripe_tile((res, res), rank, world, 8)
            for nfl in (1, 2):
                per[nfl].append(ms_per_frame(tile, nfl))
        share = whole[1] / world
        entry["worlds"][str(world)] = {"even_share_ms": round(share, 3),
                                       "one_in_flight": {"ranks_ms": [round(v, 3) for v in per[1]], "slowest_ms": round(max(per[1]), 3), "linear_frac": round(share / max(per[1]), 3)},
                                       "two_in_flight": {"ranks_ms": [round(v, 3) for v in per[2]], "slowest_ms": round(max(per[2]), 3), "linear_frac": round(share / max(per[2]), 3)}}
        print("%-9s world %d: even share %.2f ms | one frame at a time: slowest rank %.2f (%.2f of linear) | two in flight: %.2f (%.2f of linear)" %
              (stack, world, share, max(per[1]), share / max(per[1]), max(per[2]), share / max(per[2])), flush=True)
    report["modes"][stack] = entry
if out_path:
    json.dump(report, open(out_path, "w"), indent=1)
