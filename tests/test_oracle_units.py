"""Properties of the oracle itself (CPU): the pieces the GPU parity tests lean on."""
import math
import os

import numpy as np
import pytest

from conftest import scene_path
from oracle import binding as ob


def test_detmath_agrees_with_libm_after_float_rounding():
    """det_sin/det_cos replace libm in the path tracer on BOTH sides; they must be libm-quality:
    <= 2e-16 absolute in double, and the float-rounded value equals libm's cosf/sinf-quality result."""
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(0, 2 * math.pi, 20000), rng.uniform(-50, 50, 5000), [0.0, math.pi / 2, math.pi, 2 * math.pi]])
    worst = 0.0
    mism = 0
    for x in xs:
        x = float(np.float32(x))
        s, c = ob.det_sin(x), ob.det_cos(x)
        worst = max(worst, abs(s - math.sin(x)), abs(c - math.cos(x)))
        mism += int(np.float32(s) != np.float32(math.sin(x))) + int(np.float32(c) != np.float32(math.cos(x)))
    assert worst < 4e-16
    assert mism == 0


def test_rng_streams_are_deterministic_and_distinct():
    a = ob.rng_stream(0x5EED, 1234, 7, 64)
    assert (a == ob.rng_stream(0x5EED, 1234, 7, 64)).all()
    assert (a < 2 ** 31).all()
    for other in (ob.rng_stream(0x5EED, 1234, 8, 64), ob.rng_stream(0x5EED, 1235, 7, 64), ob.rng_stream(0x5EEE, 1234, 7, 64)):
        assert (a != other).mean() > 0.95
    big = np.concatenate([ob.rng_stream(1, p, 0, 256) for p in range(64)]).astype(np.float64) / 2 ** 31
    assert abs(big.mean() - 0.5) < 0.01 and abs(big.var() - 1 / 12) < 0.005


def test_render_is_independent_of_tiling_and_threads():
    sc = ob.Scene(scene_path("balls_low.p3f"))
    sc.set_resolution(96, 96)
    cfg = ob.whitted_config(2, 3)
    full, hit, st = sc.render(cfg)
    cfg8 = ob.whitted_config(2, 3, threads=8)
    full8, hit8, st8 = sc.render(cfg8)
    assert (full.view(np.uint32) == full8.view(np.uint32)).all() and (hit == hit8).all() and st.rays == st8.rays
    part, phit, _ = sc.render(cfg, 16, 40, 50, 30)
    assert (part.view(np.uint32) == full[40:70, 16:66].view(np.uint32)).all()
    # the path tracer with per-(pixel,sample) streams is tile- and thread-invariant as well
    pt = ob.default_config(integrator=1, accel=2, spp_sqrt=2, antialiasing=1, depth_of_field=0, seed=9)
    pc = ob.Scene(scene_path("path_balls.p3f"))
    pc.set_resolution(48, 48)
    a, _, _ = pc.render(pt)
    pt.threads = 4
    b, _, _ = pc.render(pt)
    c, _, _ = pc.render(pt, 8, 8, 16, 16)
    assert (a.view(np.uint32) == b.view(np.uint32)).all() and (c.view(np.uint32) == a[8:24, 8:24].view(np.uint32)).all()


@pytest.mark.parametrize("scene,depth", [("balls_low.p3f", 4), ("path_glass.p3f", 5)])
def test_parallel_semantics_stay_within_tolerance_of_literal(scene, depth):
    """DESIGN.md 'Sequential state': emptying hit_stack per primary sample and skipping the
    zero-weight reflection ray must stay within 1e-4 of the reference-literal frame, with
    identical primary hit IDs."""
    sc = ob.Scene(scene_path(scene))
    sc.set_resolution(256, 256)
    lit, lhit, _ = sc.render(ob.whitted_config(2, depth, stack_mode=1, trace_zero_weight=1))
    par, phit, _ = sc.render(ob.whitted_config(2, depth, stack_mode=0, trace_zero_weight=0, threads=8))
    assert (lhit == phit).all()
    assert np.abs(lit - par).max() <= 1e-4


def test_accel_structures_agree_on_primary_hits():
    """Closest hit is order-invariant up to exact ties: the three back ends must find the same object
    (AABB-epsilon culling may differ on a handful of silhouette pixels)."""
    sc = ob.Scene(scene_path("balls_low.p3f"))
    sc.set_resolution(200, 200)
    hits = [sc.render(ob.whitted_config(a, 0))[1] for a in (0, 1, 2)]
    assert (hits[0] != hits[1]).mean() < 2e-3 and (hits[0] != hits[2]).mean() < 2e-3


def test_aabb_slab_test_edge_cases():
    """boundingBox.cpp:44-98: strict t0 < t1, t1 > 0.0001 (double literal), t = t1 when the origin is inside."""
    hit, t = ob.aabb_intercepts([-1, -1, -1], [1, 1, 1], [0, 0, -5], [0, 0, 1])
    assert hit and t == 4.0
    hit, t = ob.aabb_intercepts([-1, -1, -1], [1, 1, 1], [0, 0, 0], [0, 0, 1])
    assert hit and t == 1.0                       # origin inside: exit distance
    assert not ob.aabb_intercepts([-1, -1, -1], [1, 1, 1], [0, 0, 5], [0, 0, 1])[0]     # behind
    assert not ob.aabb_intercepts([-1, -1, 0], [1, 1, 0], [0, 0, -5], [0, 0, 1])[0]     # flat box: t0 == t1 rejected
    assert not ob.aabb_intercepts([-1, -1, -1], [1, 1, 1], [0, 0, -1.0001], [0, 0, -1])[0]
    # t1 exactly 0.0001f is below the double literal 0.0001 -> rejected; next float up is accepted
    f = np.float32(0.0001)
    assert not ob.aabb_intercepts([-1, -1, -1], [1, 1, float(f)], [0, 0, 0], [0, 0, 1])[0]
    assert ob.aabb_intercepts([-1, -1, -1], [1, 1, float(np.nextafter(f, np.float32(1)))], [0, 0, 0], [0, 0, 1])[0]


def test_sphere_test_mutates_the_ray_direction():
    sc = ob.Scene(scene_path("balls_low.p3f"))
    hit, t, d = sc.object_intercepts(2, [2.1, 1.3, 1.7], [-4.2, -2.6, -3.4])   # unnormalised direction
    assert hit and abs(np.linalg.norm(d) - 1) < 1e-6 and abs(t - (np.sqrt(2.1 ** 2 + 1.3 ** 2 + 1.7 ** 2) - 0.5)) < 1e-5
    hit, t, d2 = sc.object_intercepts(0, [2.1, 1.3, 1.7], [-4.2, -2.6, -3.4])  # triangle: direction untouched
    assert (d2 == np.array([-4.2, -2.6, -3.4], np.float32)).all()


def test_skybox_lookup_against_a_numpy_restatement():
    """Scene::GetSkyboxColor (scene.cpp:379-457) re-derived independently in numpy on the primary rays
    of an empty scene: face choice (+x -> LEFT, -x -> RIGHT, +y -> TOP, -y -> BOTTOM, +z -> FRONT,
    -z -> BACK; |x| > |y| strict, |z| > max strict), the (s,t) axes per face, nearest texel by
    truncation, u8 / 255.99."""
    sc = ob.Scene(scene_path("balls_medium.p3f"))   # 0 objects with the shipped parser
    sc.set_resolution(96, 96)
    rng = np.random.default_rng(4)
    faces = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (w, h) in ((32, 32), (48, 24), (20, 40), (32, 32), (64, 8), (9, 9))]
    sc.set_skybox(faces)
    rgb, hit, _ = sc.render(ob.whitted_config(0, 0, skybox=1))
    assert (hit == -1).all()
    exp = np.zeros_like(rgb)
    for y in range(96):
        for x in range(96):
            _, d = sc.primary_ray(x + 0.5, y + 0.5)
            ax, ay, az = abs(d[0]), abs(d[1]), abs(d[2])
            if ax > ay:
                ma, side = ax, (1 if d[0] >= 0 else 0)
            else:
                ma, side = ay, (2 if d[1] >= 0 else 3)
            if az > ma:
                ma, side = az, (4 if d[2] >= 0 else 5)
            s_c, t_c = [(-d[2], d[1]), (d[2], d[1]), (-d[0], -d[2]), (-d[0], d[2]), (-d[0], d[1]), (d[0], d[1])][side]
            inv = float(np.float32(1.0) / np.float32(ma))
            s = np.float32((float(s_c) * inv + 1) / 2)
            t = np.float32((float(t_c) * inv + 1) / 2)
            f = faces[side]
            xp = int(np.float32(f.shape[1] - 1) * s)
            yp = int(np.float32(f.shape[0] - 1) * t)
            exp[y, x] = (f[yp, xp].astype(np.float32) / np.float32(255.99)).astype(np.float32)
    assert (rgb.view(np.uint32) == exp.view(np.uint32)).all()
