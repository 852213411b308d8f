"""Oracle L0 / camera / sampler restatement vs. golden vectors produced by the
COMPILED REFERENCE (oracle/_ref: vector.cpp, sampler.cpp, camera.h, ray.h, maths.h,
color.h built as they lie; generator tests/golden/make_ref_vectors.py).  Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import binding as ob


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "ref_vectors", "l0.npz"))


@pytest.fixture(scope="module")
def L():
    lib = ob.lib()
    lib.orc_vec_length.restype = C.c_float
    lib.orc_vec_dot.restype = C.c_float
    lib.orc_u8tofloat.restype = C.c_float
    lib.orc_u8tofloat.argtypes = [C.c_uint8]
    lib.orc_vec_div.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
    return lib


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_vector_ops_bit_exact(g, L):
    a, b, f = g["vec_a"], g["vec_b"], g["scal_f"]
    n = len(a)
    norm = a.copy()
    length = np.zeros(n, np.float32)
    dot = np.zeros(n, np.float32)
    cross = np.zeros((n, 3), np.float32)
    div = np.zeros((n, 3), np.float32)
    for i in range(n):
        L.orc_vec_normalize(fp(norm[i]))
        length[i] = L.orc_vec_length(fp(a[i]))
        dot[i] = L.orc_vec_dot(fp(a[i]), fp(b[i]))
        L.orc_vec_cross(fp(a[i]), fp(b[i]), fp(cross[i]))
        L.orc_vec_div(fp(a[i]), C.c_float(f[i]), fp(div[i]))
    assert (bits(norm) == bits(g["normalize"])).all()      # vector.cpp:65-70
    assert (bits(length) == bits(g["length"])).all()       # vector.cpp:10-13
    assert (bits(dot) == bits(g["dot"])).all()             # vector.cpp:55-58
    assert (bits(cross) == bits(g["cross"])).all()         # vector.cpp:84-99
    assert (bits(div) == bits(g["div"])).all()             # vector.cpp:60-63


def test_get_direction_mutates_and_is_not_idempotent(g, L):
    """ray.h:16-18: every call re-normalises in place (Q8); k calls must match k reference calls."""
    a = g["vec_a"]
    for k, key in ((1, "getdir1"), (2, "getdir2"), (3, "getdir3")):
        d = a.copy()
        for i in range(len(d)):
            L.orc_get_direction(fp(d[i]), k)
        assert (bits(d) == bits(g[key])).all()
    # the golden data itself shows that a second normalisation can still move the vector
    assert (bits(g["getdir1"]) != bits(g["getdir2"])).any()


def test_camera_rays_bit_exact(g, L):
    cams = g["cams"]
    for c in range(len(cams)):
        px = np.ascontiguousarray(g["cam_px"][c])
        lens = np.ascontiguousarray(g["cam_lens"][c])
        m = len(px)
        ro, rd, lo, ld = (np.zeros((m, 3), np.float32) for _ in range(4))
        st = np.zeros(2, np.float32)
        p15 = np.ascontiguousarray(cams[c])
        L.orc_camera_rays(fp(p15), m, fp(px), fp(lens), fp(ro), fp(rd), fp(lo), fp(ld), fp(st))
        assert (bits(st) == bits(g["cam_state"][c])).all()      # camera.h:34-63
        assert (bits(ro) == bits(g["cam_ray_o"][c])).all()      # camera.h:65-82
        assert (bits(rd) == bits(g["cam_ray_d"][c])).all()
        assert (bits(lo) == bits(g["cam_lray_o"][c])).all()     # camera.h:84-115
        assert (bits(ld) == bits(g["cam_lray_d"][c])).all()


def test_rand_float_and_unit_disk_on_libc_stream(g, L):
    """maths.h:67-70 and sampler.cpp:5-11 (incl. g++'s right-to-left argument order)."""
    for s, seed in enumerate(g["rand_seeds"]):
        rf = np.zeros(64, np.float32)
        L.orc_libc_rand_floats(int(seed), 64, fp(rf))
        assert (bits(rf) == bits(g["rand_float"][s])).all()
        disk = np.zeros((64, 2), np.float32)
        L.orc_libc_unit_disk(int(seed), 64, 0, fp(disk))
        assert (bits(disk) == bits(g["unit_disk"][s])).all()
        other = np.zeros((64, 2), np.float32)
        L.orc_libc_unit_disk(int(seed), 64, 1, fp(other))
        assert (bits(other) != bits(g["unit_disk"][s])).any()  # the other operand order is observable


def test_u8_pack_and_clamp(g, L):
    u8 = np.array([L.orc_u8fromfloat(float(x)) for x in g["u8_in"]], np.uint8)
    assert (u8 == g["u8_out"]).all()                                  # maths.h:81-86
    tof = np.array([L.orc_u8tofloat(i) for i in range(256)], np.float32)
    assert (bits(tof) == bits(g["u8tofloat"])).all()                  # maths.h:89-92
    cl = g["clamp_in"].copy()
    for i in range(len(cl)):
        L.orc_color_clamp(fp(cl[i]))
    assert (bits(cl) == bits(g["clamp_out"])).all()                   # color.h:39-44
