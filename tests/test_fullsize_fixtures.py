"""tests/golden/fullsize/ (make_fullsize_fixtures.py) against the oracle of this tree, on the part the CPU suite can afford:
a fixture that no longer is what the oracle renders would silently turn the GPU tests that read it into tests of nothing."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import binding as ob

FULL = os.path.join(GOLDEN, "fullsize")
NAMES = ("rays_primary", "rays_bounce", "rays_light", "node_tests", "sphere_tests", "tri_tests", "shaded_hits")


@pytest.mark.parametrize("name,row_index", [("cfg3", 64), ("cfg3", 3), ("cfg5", None)])
def test_path_tracer_fixture_rows_are_what_the_oracle_renders(name, row_index):
    g = np.load(os.path.join(FULL, name + ".npz"))
    lens = tuple(float(v) for v in g["lens"]) if float(g["lens"][0]) != 0 else None
    sc = ob.Scene(os.path.join(ROOT, "scenes", "cornell.p3f"))
    sc.set_resolution(1024, 1024)
    if lens:
        sc.set_lens(*lens)
    cfg = ob.default_config(integrator=1, accel=2, max_depth=20, spp_sqrt=int(g["spp_sqrt"]), antialiasing=1,
                            depth_of_field=1 if lens else 0, sample_disk=1, soft_shadows=0, seed=int(g["seed"]), rng_mode=0,
                            stack_mode=0, trace_zero_weight=0, math_mode=0, threads=1)
    if row_index is None:  # 4096 spp: eight pixels of one row are what a second of CPU time buys
        y = int(g["rows"][70])
        for k in (40, 41, 90):
            rgb, hit, _ = sc.render(cfg, 8 * k, y, 1, 1)
            assert hit[0, 0] == g["sub8_hit"][70, k]
            assert (rgb[0, 0].view(np.uint32) == g["sub8_rgb"][70, k].view(np.uint32)).all()
        return
    y = int(g["rows"][row_index])
    rgb, hit, st = sc.render(cfg, 0, y, 1024, 1)
    assert (hit[0, ::8] == g["sub8_hit"][row_index]).all()
    assert (rgb[0, ::8].view(np.uint32) == g["sub8_rgb"][row_index].view(np.uint32)).all()


def test_cfg4_fixture_counters_are_consistent():
    g = np.load(os.path.join(FULL, "cfg4.npz"))
    c = dict(zip([str(k) for k in g["counter_names"]], [int(v) for v in g["counters"]]))
    assert c["rays_primary"] == c["pixels"] == 2048 * 2048
    assert c["rays_shadow"] == 2 * c["shaded_hits"]                       # two lights, one feeler each per shaded hit
    assert c["rays_reflect"] + c["rays_primary"] >= c["shaded_hits"]      # every shaded hit is the end of one traced ray
    assert g["sub8_rgb"].shape == (256, 256, 3) and g["crops"].shape == (4, 64, 64, 3)
    # the sub-sampled frame and the crops overlap where a crop pixel lies on the 8-grid
    for (x0, y0), crop in zip(g["crop_xy"], g["crops"]):
        dx, dy = (-int(x0)) % 8, (-int(y0)) % 8  # first pixel of the crop that lies on the grid
        assert (crop[dy, dx].view(np.uint32) == g["sub8_rgb"][(y0 + dy) // 8, (x0 + dx) // 8].view(np.uint32)).all()
