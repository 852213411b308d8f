#!/usr/bin/env python3
"""Turn the survey stage's reference renders into small committed fixtures.

PROVENANCE (read tests/golden/README.md): the inputs are the float-RGB frames that
the SURVEY stage wrote to /tmp/oracle_probe/*.bin in this container by compiling
the reference's own hot-path sources with g++ 11.4 -O2.
This round may not rebuild that binary (it needs a stand-in for DevIL's
<IL/il.h>), so these frames are kept as DATA: expected outputs of the reference
for the named scene + option set.  This script only crops / hashes them.

Output per frame (tests/golden/survey_probe/<name>.npz):
    res            (2,)  int32     frame size
    sha256         str             SHA-256 of the full float32 RGB frame bytes (row y=0 first)
    chan_sum       (3,)  float64   per-channel sum over the full frame
    sub8           (H/8, W/8, 3)   every 8th pixel of every 8th row
    crop_xy        (k, 2) int32    x0,y0 of each crop
    crops          (k, 64, 64, 3)  float32 windows (silhouettes / shadow edges / reflections)
"""
import hashlib
import json
import os
import sys

import numpy as np

SRC = "/tmp/oracle_probe"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "survey_probe")

# name -> (file, scene fixture, description of the option set the survey harness used)
FRAMES = {
    "cfg1_none": ("out_low_0.bin", "balls_low.p3f", dict(integrator=0, accel=0, max_depth=1, res=512)),
    "cfg1_grid": ("out_low_1.bin", "balls_low.p3f", dict(integrator=0, accel=1, max_depth=1, res=512)),
    "cfg1_bvh": ("out_low_2.bin", "balls_low.p3f", dict(integrator=0, accel=2, max_depth=1, res=512)),
    "cfg2_none": ("out_1024_d4_0.bin", "balls_low.p3f", dict(integrator=0, accel=0, max_depth=4, res=1024)),
    "cfg2_bvh": ("out_1024_d4_2.bin", "balls_low.p3f", dict(integrator=0, accel=2, max_depth=4, res=1024)),
    "tri100k_bvh_d6": ("out_tri100k.bin", "scenes/make_tri100k.py", dict(integrator=0, accel=2, max_depth=6, res=512)),
    "pt_path_balls_none": ("out_pt_path_balls_0.bin", "path_balls.p3f",
                           dict(integrator=1, accel=0, max_depth=20, spp_sqrt=4, srand=12345, res=512)),
    "pt_path_balls_bvh": ("out_pt_path_balls_2.bin", "path_balls.p3f",
                          dict(integrator=1, accel=2, max_depth=20, spp_sqrt=4, srand=12345, res=512)),
    "pt_path_mirror_none": ("out_pt_path_mirror_0.bin", "path_mirror.p3f",
                            dict(integrator=1, accel=0, max_depth=20, spp_sqrt=4, srand=12345, res=512)),
    "pt_path_mirror_bvh": ("out_pt_path_mirror_2.bin", "path_mirror.p3f",
                           dict(integrator=1, accel=2, max_depth=20, spp_sqrt=4, srand=12345, res=512)),
}


def load(path):
    raw = np.fromfile(path, np.uint8)
    rx, ry = np.frombuffer(raw[:8].tobytes(), np.int32)
    return np.frombuffer(raw[8:].tobytes(), np.float32).reshape(ry, rx, 3)


def pick_crops(img, k=3, size=64):
    """Deterministic choice: the windows (on a size/2 lattice) with the largest local variation."""
    h, w, _ = img.shape
    g = np.abs(np.diff(img, axis=0)).sum(-1)[:, :-1] + np.abs(np.diff(img, axis=1)).sum(-1)[:-1, :]
    best = []
    for y0 in range(0, h - size, size // 2):
        for x0 in range(0, w - size, size // 2):
            best.append((float(g[y0:y0 + size - 1, x0:x0 + size - 1].sum()), x0, y0))
    best.sort(reverse=True)
    chosen = []
    for s, x0, y0 in best:
        if all(abs(x0 - a) >= size or abs(y0 - b) >= size for a, b in chosen):
            chosen.append((x0, y0))
        if len(chosen) == k:
            break
    return chosen


def main():
    if not os.path.isdir(SRC):
        sys.exit("no %s here: the committed fixtures cannot be regenerated in this container" % SRC)
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for name, (fn, scene, opts) in FRAMES.items():
        img = load(os.path.join(SRC, fn))
        xy = pick_crops(img)
        crops = np.stack([img[y0:y0 + 64, x0:x0 + 64] for x0, y0 in xy])
        sha = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), res=np.array(img.shape[1::-1], np.int32),
                            sha256=sha, chan_sum=img.astype(np.float64).sum((0, 1)),
                            sub8=img[::8, ::8].copy(), crop_xy=np.array(xy, np.int32), crops=crops)
        manifest[name] = dict(source=fn, scene=scene, options=opts, sha256=sha)
        print(name, img.shape, sha[:16], xy)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
