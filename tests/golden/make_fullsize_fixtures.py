#!/usr/bin/env python3
"""Full-size fixtures for BASELINE configs[2..4], rendered by the CPU oracle in the build container.

    python tests/golden/make_fullsize_fixtures.py [cfg4] [cfg3] [cfg5]

What the GPU suite cannot afford to ask the oracle at test time (minutes of CPU work) is recorded here once, with
the oracle of this tree, and committed as data under tests/golden/fullsize/:

  cfg4.npz  100k random triangles (scenes/make_tri100k.py), 2048x2048, Whitted MAX_DEPTH 6, BVH, the reference's
            serial hit_stack (oracle stack_mode=1, one thread): SHA-256 of the float RGB frame and of the hit IDs,
            every 8th pixel of every 8th row, four 64x64 crops, every ray / test counter.
  cfg3.npz  scenes/cornell.p3f 1024x1024, path tracer 256 spp, MAX_DEPTH 20, BVH, seed 0x5EED: image rows 0, 8, 16 ...
            (what a tile with stripe_h = 1, stripe_stride = 8 renders) at full width; kept: every 8th pixel of those
            rows (float RGB + hit IDs) and the counters summed over the rows.
  cfg5.npz  the same with the thin lens (aperture 10, focal 1, SAMPLE_DISK) at 4096 spp.

Provenance: these are outputs of oracle/p3d_oracle.cpp, i.e. of the restatement, not of the reference; they pin the
HIP path to the oracle at sizes the test suite cannot re-render, nothing more (pinning status of the oracle itself:
tests/golden/README.md).
"""
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
OUT = os.path.join(HERE, "fullsize")
COUNTERS = ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "rays_bounce", "rays_light", "node_tests",
            "sphere_tests", "tri_tests", "box_tests", "plane_tests", "shaded_hits", "pixels")
CROPS4 = [(992, 992), (300, 1500), (1500, 420), (64, 1984)]  # centre, two off-centre windows, a corner (mostly sky)


def cfg4():
    from oracle import binding as ob
    import make_tri100k
    path = "/tmp/p3d_fixture_tri100k.p3f"
    make_tri100k.generate(path, res=1024)
    sc = ob.Scene(path)
    sc.set_resolution(2048, 2048)
    cfg = ob.whitted_config(2, 6, stack_mode=1, trace_zero_weight=1, threads=1)
    t0 = time.time()
    rgb, hit, st = sc.render(cfg)
    print("cfg4: %.1f s, %d rays" % (time.time() - t0, st.rays), flush=True)
    np.savez_compressed(
        os.path.join(OUT, "cfg4.npz"), res=np.array([2048, 2048], np.int32),
        sha256_rgb=hashlib.sha256(np.ascontiguousarray(rgb).tobytes()).hexdigest(),
        sha256_hit=hashlib.sha256(np.ascontiguousarray(hit).tobytes()).hexdigest(),
        sub8_rgb=rgb[::8, ::8].copy(), sub8_hit=hit[::8, ::8].copy(), crop_xy=np.array(CROPS4, np.int32),
        crops=np.stack([rgb[y:y + 64, x:x + 64] for x, y in CROPS4]),
        counters=np.array([getattr(st, k) for k in COUNTERS], np.uint64), counter_names=np.array(COUNTERS),
        max_stack=np.uint64(st.max_stack))


def _pt_rows(args):
    lens, spp_sqrt, rows = args
    from oracle import binding as ob
    sc = ob.Scene(os.path.join(ROOT, "scenes", "cornell.p3f"))
    sc.set_resolution(1024, 1024)
    if lens:
        sc.set_lens(*lens)
    cfg = ob.default_config(integrator=1, accel=2, max_depth=20, spp_sqrt=spp_sqrt, antialiasing=1,
                            depth_of_field=1 if lens else 0, sample_disk=1, soft_shadows=0, seed=0x5EED, rng_mode=0,
                            stack_mode=0, trace_zero_weight=0, math_mode=0, threads=1)
    out = []
    for y in rows:
        rgb, hit, st = sc.render(cfg, 0, int(y), 1024, 1)
        out.append((int(y), rgb[0, ::8].copy(), hit[0, ::8].copy(), [getattr(st, k) for k in COUNTERS]))
    return out


def pt(name, lens, spp_sqrt):
    rows = list(range(0, 1024, 8))
    jobs = [(lens, spp_sqrt, rows[i::16]) for i in range(16)]
    t0 = time.time()
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        parts = [r for chunk in pool.map(_pt_rows, jobs) for r in chunk]
    parts.sort(key=lambda r: r[0])
    assert [r[0] for r in parts] == rows
    counters = np.sum(np.array([r[3] for r in parts], np.uint64), axis=0)
    print("%s: %.1f s, %d rays" % (name, time.time() - t0, int(counters[:6].sum())), flush=True)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), res=np.array([1024, 1024], np.int32), rows=np.array(rows, np.int32),
        spp_sqrt=np.int32(spp_sqrt), lens=np.array(lens if lens else [0.0, 0.0], np.float32), seed=np.uint64(0x5EED),
        sub8_rgb=np.stack([r[1] for r in parts]), sub8_hit=np.stack([r[2] for r in parts]),
        counters=counters, counter_names=np.array(COUNTERS))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    want = sys.argv[1:] or ["cfg4", "cfg3", "cfg5"]
    if "cfg4" in want:
        cfg4()
    if "cfg3" in want:
        pt("cfg3", None, 16)
    if "cfg5" in want:
        pt("cfg5", (10.0, 1.0), 64)
