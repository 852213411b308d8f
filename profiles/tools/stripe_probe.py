#!/usr/bin/env python3
"""What one rank's share of the BASELINE configs[3] frame costs on ONE GPU: the 8-row stripes
of every rank of 2 / 4 / 8 rendered as that rank would render them (p3d.stripe_tile), literal and per-pixel stack,
megakernel and one-launch-per-chain-level, against an even share of the whole frame.

usage (GPU box): python3 profiles/tools/stripe_probe.py [out.json]     -> JSON (also printed as a table)

The slowest rank bounds a strong-scaled frame: `linear_frac` = (whole frame / world) / slowest rank's stripes."""
import json
import os
import sys

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
res, depth = 2048, 6
hs = p3d.HostScene(scene)
hs.set_resolution(res, res)
dev = p3d.DeviceScene(hs, bvh=True)
buf = torch.empty(res * res * 16, dtype=torch.uint8, device="cuda")


def run(cfg, tile, k):
    ms = []
    for _ in range(k):
        st = p3d.Stats()
        dev.render_device(cfg, tile, d_rgb=buf.data_ptr(), d_hit=buf.data_ptr() + tile.w * tile.h * 12, stats=st)
        ms.append((st.kernel_ms, st.pass1_ms, st.handoff_ms))
    return ms


report = {"workload": "100k random triangles 2048x2048, Whitted MAX_DEPTH=6, BVH (BASELINE configs[3])", "stripe_rows": 8, "modes": {}}
for stack, chain in (("literal", "megakernel"), ("per_pixel", "megakernel"), ("literal", "per_level")):
    cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=depth,
                             stack_mode=p3d.STACK_LITERAL if stack == "literal" else p3d.STACK_PER_PIXEL,
                             chain_launch=p3d.CHAIN_PER_LEVEL if chain == "per_level" else p3d.CHAIN_MEGAKERNEL)
    full = run(cfg, dev.full_tile(), 5)
    full_ms = min(m[0] for m in full[1:])
    entry = {"whole_frame_ms": round(full_ms, 3), "worlds": {}}
    for world in (2, 4, 8):
        ranks = []
        for rank in range(world):
            ms = run(cfg, p3d.stripe_tile((res, res), rank, world, 8), 5)
            best = min(ms[1:], key=lambda m: m[0])  # (the first call records the tile schedule and finds the row starts)
            ranks.append({"rank": rank, "first_call_ms": round(ms[0][0], 3), "ms": round(best[0], 3), "pass1_ms": round(best[1], 3), "handoff_ms": round(best[2], 3)})
        slowest = max(r["ms"] for r in ranks)
        entry["worlds"][str(world)] = {"even_share_ms": round(full_ms / world, 3), "rank0_ms": ranks[0]["ms"], "slowest_rank_ms": slowest,
                                       "linear_frac": round(full_ms / world / slowest, 3), "ranks": ranks}
        print("%-9s %-10s whole %.2f ms | world %d: even share %.2f, rank 0 %.2f, slowest %.2f -> %.2f of linear" %
              (stack, chain, full_ms, world, full_ms / world, ranks[0]["ms"], slowest, full_ms / world / slowest), flush=True)
    report["modes"]["%s/%s" % (stack, chain)] = entry
if out_path:
    json.dump(report, open(out_path, "w"), indent=1)
