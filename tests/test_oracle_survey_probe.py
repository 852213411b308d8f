"""Oracle vs. frames rendered by the REFERENCE's own sources during the survey stage
(tests/golden/survey_probe/, provenance in tests/golden/README.md).

In its literal mode (one member hit_stack for the whole frame, libc rand() in program
order, host libm) the oracle must reproduce those frames BIT FOR BIT: full-frame
SHA-256, crops and sub-sampled frame.  The same runs pin the ray counts and the
tests-per-ray figures that SURVEY.md §6 / BASELINE.md quote from the reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path
from oracle import binding as ob

PROBE = os.path.join(GOLDEN, "survey_probe")
MANIFEST = json.load(open(os.path.join(PROBE, "manifest.json")))


def literal_cfg(opts):
    if opts["integrator"] == 0:
        return ob.whitted_config(opts["accel"], opts["max_depth"], stack_mode=1, rng_mode=1,
                                 trace_zero_weight=1, math_mode=1)
    return ob.default_config(integrator=1, accel=opts["accel"], max_depth=opts["max_depth"],
                             spp_sqrt=opts["spp_sqrt"], antialiasing=1, depth_of_field=0, rng_mode=1,
                             stack_mode=1, trace_zero_weight=1, math_mode=1, seed=opts["srand"])


def check_frame(name, rgb):
    g = np.load(os.path.join(PROBE, name + ".npz"))
    assert tuple(g["res"]) == rgb.shape[1::-1]
    assert (rgb[::8, ::8].view(np.uint32) == g["sub8"].view(np.uint32)).all()
    for (x0, y0), crop in zip(g["crop_xy"], g["crops"]):
        assert (rgb[y0:y0 + 64, x0:x0 + 64].view(np.uint32) == crop.view(np.uint32)).all()
    assert hashlib.sha256(np.ascontiguousarray(rgb).tobytes()).hexdigest() == str(g["sha256"])


# (fixture name, expected reference rayCounter, expected (node, tri, sphere) tests per ray or None)
# counts: SURVEY.md §6 table and BASELINE.md §2 ([probe] runs of the reference)
WHITTED = [
    ("cfg1_none", 1195867, (0.0, 2.00, 8.89)),
    ("cfg1_grid", 1195863, None),
    ("cfg1_bvh", 1195863, None),
    ("cfg2_none", 4944907, None),
    ("cfg2_bvh", 4944908, (10.36, 0.45, 1.06)),
]


@pytest.mark.parametrize("name,rays,per_ray", WHITTED)
def test_whitted_frames_bit_exact(name, rays, per_ray):
    opts = MANIFEST[name]["options"]
    sc = ob.Scene(scene_path(MANIFEST[name]["scene"]))
    sc.set_resolution(opts["res"], opts["res"])
    rgb, hit, st = sc.render(literal_cfg(opts))
    check_frame(name, rgb)
    assert st.ref_ray_counter == rays
    assert st.rays == rays  # no zero-weight rays in balls_low: both ray definitions agree
    if per_ray:
        assert round(st.node_tests / rays, 2) == per_ray[0]
        assert round(st.tri_tests / rays, 2) == per_ray[1]
        assert round(st.sphere_tests / rays, 2) == per_ray[2]
    if name == "cfg2_bvh":  # 21.2 % primary / 72.8 % shadow / 6.0 % reflection
        assert (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract) == (1048576, 3600366, 295966, 0)


def test_bvh_shape_matches_reference_counts(tri100k_path):
    sc = ob.Scene(scene_path("balls_low.p3f"))
    assert sc.bvh_info() == dict(nodes=15, leaves=8, max_depth=5)
    big = ob.Scene(tri100k_path)
    assert big.counts()["objects"] == 100000
    assert big.bvh_info() == dict(nodes=125701, leaves=62851, max_depth=21)


@pytest.mark.slow
def test_tri100k_frame_bit_exact(tri100k_path):
    opts = MANIFEST["tri100k_bvh_d6"]["options"]
    sc = ob.Scene(tri100k_path)
    rgb, hit, st = sc.render(literal_cfg(opts))
    check_frame("tri100k_bvh_d6", rgb)
    assert st.ref_ray_counter == 2887776
    assert round(st.node_tests / st.rays, 1) == 78.5 and round(st.tri_tests / st.rays, 2) == 5.49


@pytest.mark.slow
@pytest.mark.parametrize("name,counted", [("pt_path_balls_none", None), ("pt_path_balls_bvh", None),
                                          ("pt_path_mirror_none", None), ("pt_path_mirror_bvh", None)])
def test_path_tracer_frames_bit_exact(name, counted):
    """Radiance (main.cpp:313-516) on the libc rand() stream srand(12345), 16 spp, depth 20."""
    opts = MANIFEST[name]["options"]
    sc = ob.Scene(scene_path(MANIFEST[name]["scene"]))
    rgb, hit, st = sc.render(literal_cfg(opts))
    check_frame(name, rgb)


def test_loader_reproduces_the_shipped_parser_failure():
    """scene.cpp:489-492 always reads 14 numbers after `f`; an 11-number `f` poisons the stream
    (balls_medium -> objects=0 lights=3)."""
    sc = ob.Scene(scene_path("balls_medium.p3f"))
    c = sc.counts()
    assert (c["objects"], c["lights"]) == (0, 3)
    ext = ob.Scene(scene_path("balls_medium.p3f"), legacy_f11=True)
    assert ext.counts()["objects"] == 91 + 2 and ext.counts()["lights"] == 3
    low = ob.Scene(scene_path("balls_low.p3f")).counts()
    assert (low["objects"], low["lights"], low["materials"]) == (12, 3, 2)
