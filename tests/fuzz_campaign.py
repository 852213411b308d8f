#!/usr/bin/env python3
"""Long differential-fuzz run (GPU kernels vs CPU oracle), outside the pytest suite.

  python tests/fuzz_campaign.py [--whitted N] [--pt M] [--stripes K] [--start S]

Same generator and same bars as test_fuzz_random_scenes_* in test_gpu_parity.py (check_whitted: the default
P3D_STACK_LITERAL frame over the BVH bit-identical to the oracle's serial order, then the per-pixel stack with every
counter), over many more seeds and over scene sizes on both sides of the LDS-staging limit.  Prints one line per failing
(seed, accel) and a summary; exit code 1 if anything failed.  Test infrastructure: uses oracle/."""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

if torch.cuda.is_available():  # (torch's HIP runtime first, as tests/conftest.py does: check_whitted renders on torch streams too)
    torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import p3d_amd as p3d  # noqa: E402
from fuzz_scenes import random_scene  # noqa: E402
from oracle import binding as ob  # noqa: E402
from test_gpu_parity import check_whitted, oracle_cfg_like  # noqa: E402

COUNTERS = ("rays", "node_tests", "sphere_tests", "tri_tests", "box_tests", "plane_tests")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--whitted", type=int, default=200)
    ap.add_argument("--pt", type=int, default=40)
    ap.add_argument("--stripes", type=int, default=0,
                    help="scenes for the stripe / sub-rectangle phase: 192x160 frames (enough tiles for the cost-ordered schedule "
                         "of scenes traversed from L2), the literal full frame against the oracle, then every stripe "
                         "set of 2 and 4 ranks and a random sub-rectangle against the full frame, each rendered twice (recording "
                         "launch, scheduled launch)")
    ap.add_argument("--start", type=int, default=0)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp()
    fails = 0
    worst = 0.0
    for k in range(a.start, a.start + a.whitted):
        size = (1, 4, 16, 60)[k % 4]  # 16 objects ... ~1000 objects (past the 16 KB LDS limit)
        path = random_scene(50000 + k, os.path.join(tmp, "w.p3f"), n_spheres=6 * size, n_tris=8 * size, n_boxes=2 * size,
                            n_planes=1 if k % 7 == 0 else 0, n_lights=1 + k % 4, res=(80, 64))
        hs, sc = p3d.HostScene(path), ob.Scene(path)
        dev = p3d.DeviceScene(hs, bvh=True, grid=True)
        for accel in (p3d.ACCEL_NONE, p3d.ACCEL_GRID, p3d.ACCEL_BVH):
            if accel == p3d.ACCEL_NONE and size > 16:
                continue
            kw = dict(antialiasing=1, spp_sqrt=2, soft_shadows=1, depth_of_field=k % 2, sample_disk=(k // 2) % 2,
                      sample_mode=(k // 4) % 2, seed=k) if k % 5 == 1 else {}
            cfg = p3d.whitted_config(accel=accel, max_depth=k % 8, **kw)
            try:
                check_whitted(dev, sc, cfg, tol=5e-6, counters=("rays_primary", "rays_shadow", "rays_reflect", "rays_refract") + COUNTERS[1:])
            except AssertionError as e:
                fails += 1
                print("FAIL whitted seed %d accel %d size %d depth %d: %s" % (k, accel, size, k % 8, str(e)[:200]), flush=True)
        if k % 20 == 19:
            print("whitted %d done, failures %d" % (k + 1 - a.start, fails), flush=True)
    worst_pt = 0.0
    for k in range(a.start, a.start + a.pt):
        size = (1, 3, 10)[k % 3]
        path = random_scene(70000 + k, os.path.join(tmp, "p.p3f"), n_spheres=6 * size, n_tris=8 * size, n_boxes=2 * size,
                            n_lights=0, emitters=1 + k % 3, res=(48, 48))
        hs, sc = p3d.HostScene(path), ob.Scene(path)
        dev = p3d.DeviceScene(hs, bvh=True, grid=True)
        for accel in (p3d.ACCEL_GRID, p3d.ACCEL_BVH):
            cfg = p3d.pathtrace_config(accel=accel, spp_sqrt=2 + k % 6, max_depth=6 + k % 20, dof=k % 2, seed=k, collect_stats=1,
                                       sample_mode=(k // 2) % 2)
            rgb, hit, st = dev.render(cfg)
            o_rgb, o_hit, o_st = sc.render(oracle_cfg_like(cfg))
            m = np.isfinite(o_rgb).all(-1)
            scale = max(1.0, float(np.abs(o_rgb[m]).max())) if m.any() else 1.0
            d = float(np.abs(rgb[m] - o_rgb[m]).max()) / scale if m.any() else 0.0
            worst_pt = max(worst_pt, d)
            bad = []
            if not (hit == o_hit).all(): bad.append("hit ids")
            if not (np.isfinite(rgb).all(-1) == m).all(): bad.append("finite mask")
            if d > 1e-4: bad.append("rgb %.3g" % d)
            if (st.rays_primary, st.rays_bounce, st.rays_light) != (o_st.rays_primary, o_st.rays_bounce, o_st.rays_light): bad.append("ray counts")
            if bad:
                fails += 1
                print("FAIL pt seed %d accel %d size %d: %s" % (k, accel, size, ", ".join(bad)), flush=True)
        if k % 10 == 9:
            print("pt %d done, worst relative |rgb diff| %.3g, failures %d" % (k + 1 - a.start, worst_pt, fails), flush=True)
    uncertified = 0
    for k in range(a.start, a.start + a.stripes):
        size = (2, 16, 60, 120)[k % 4]
        res = (192, 160)
        path = random_scene(90000 + k, os.path.join(tmp, "s.p3f"), n_spheres=(6 * size) if k % 3 else 0, n_tris=8 * size, n_boxes=2 * size,
                            n_lights=1 + k % 3, res=res)
        hs, sc = p3d.HostScene(path), ob.Scene(path)
        dev = p3d.DeviceScene(hs, bvh=True)
        kw = dict(antialiasing=1, spp_sqrt=2, soft_shadows=1, seed=k) if k % 6 == 5 else {}
        cfg = p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=1 + k % 6, **kw)
        try:
            full, full_hit, _ = check_whitted(dev, sc, cfg, tol=5e-6)
        except AssertionError as e:
            fails += 1
            print("FAIL stripes seed %d full frame: %s" % (k, str(e)[:200]), flush=True)
            continue
        rng = np.random.default_rng(k)
        x0, y0 = int(rng.integers(0, 100)), int(rng.integers(0, 90))
        tiles = [(p3d.stripe_tile(res, r, w, 8), p3d.stripe_rows(res, r, w, 8), slice(None)) for w in (2, 4) for r in range(w)]
        tw, th = int(rng.integers(9, 90)), int(rng.integers(9, 70))
        tiles.append((p3d.Tile(x0, y0, tw, th, 0, 1), np.arange(y0, y0 + th), slice(x0, x0 + tw)))
        for t, rows, cols in tiles:
            for rep in range(2):
                try:
                    rgb, hit, _ = dev.render(cfg, tile=t)
                except p3d.P3DError as e:
                    if "could not be started" in str(e):
                        uncertified += 1  # loud, not wrong
                        break
                    raise
                want, want_hit = full[rows][:, cols], full_hit[rows][:, cols]
                same = (rgb.view(np.uint32) == want.view(np.uint32)) | (np.isnan(rgb) & np.isnan(want))
                if not (same.all() and (hit == want_hit).all()):
                    fails += 1
                    print("FAIL stripes seed %d tile (%d,%d,%d,%d,%d,%d) pass %d: %d pixels differ" % (
                        k, t.x0, t.y0, t.w, t.h, t.stripe_h, t.stripe_stride, rep, int((~same.all(-1)).sum())), flush=True)
        if k % 10 == 9:
            print("stripes %d done, failures %d, calls that failed for an uncertifiable row start %d" % (k + 1 - a.start, fails, uncertified), flush=True)
    print("SUMMARY whitted %d pt %d stripes %d failures %d uncertified-row failures %d worst whitted %.3g worst pt %.3g" % (
        a.whitted, a.pt, a.stripes, fails, uncertified, worst, worst_pt))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
