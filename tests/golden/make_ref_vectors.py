#!/usr/bin/env python3
"""Generate golden vectors from the pieces of the reference that build as they lie.

Runs oracle/_ref/libp3dref.so (built by oracle/Makefile from
/root/reference/Raytracing/{vector.cpp,sampler.cpp,camera.h,ray.h,maths.h,color.h};
see oracle/ref_driver.cpp) on seeded inputs and stores inputs + outputs in
tests/golden/ref_vectors/l0.npz.  Needs /root/reference (builder container only);
the committed .npz is what the tests read.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libp3dref.so")


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    L = C.CDLL(LIB)
    L.ref_vec_length.restype = C.c_float
    L.ref_vec_dot.restype = C.c_float
    L.ref_rand_float.restype = C.c_float
    L.ref_camera_aperture.restype = C.c_float
    L.ref_camera_plane_dist.restype = C.c_float
    L.ref_u8fromfloat.restype = C.c_ubyte
    L.ref_u8fromfloat.argtypes = [C.c_float]
    L.ref_u8tofloat.restype = C.c_float
    L.ref_u8tofloat.argtypes = [C.c_ubyte]
    L.ref_vec_div.argtypes = [C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
    L.ref_camera_create.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float] * 3 + [C.c_int] * 2 + [C.c_float] * 2
    L.ref_camera_primary.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ref_camera_primary_lens.argtypes = [C.c_float] * 4 + [C.POINTER(C.c_float)] * 2

    rng = np.random.default_rng(20261003)
    n = 4096
    # vectors with a wide range of magnitudes, plus near-unit ones (the re-normalisation case)
    a = (rng.standard_normal((n, 3)) * np.exp(rng.uniform(-6, 6, (n, 1)))).astype(np.float32)
    a[: n // 2] = (a[: n // 2] / np.linalg.norm(a[: n // 2], axis=1, keepdims=True)).astype(np.float32)
    b = rng.standard_normal((n, 3)).astype(np.float32)
    f = rng.uniform(0.1, 10, n).astype(np.float32)
    out = dict(vec_a=a, vec_b=b, scal_f=f)
    norm = a.copy()
    length = np.zeros(n, np.float32)
    dot = np.zeros(n, np.float32)
    cross = np.zeros((n, 3), np.float32)
    div = np.zeros((n, 3), np.float32)
    gd1, gd2, gd3 = a.copy(), a.copy(), a.copy()
    for i in range(n):
        L.ref_vec_normalize(fp(norm[i]))
        length[i] = L.ref_vec_length(fp(a[i]))
        dot[i] = L.ref_vec_dot(fp(a[i]), fp(b[i]))
        L.ref_vec_cross(fp(a[i]), fp(b[i]), fp(cross[i]))
        L.ref_vec_div(fp(a[i]), C.c_float(f[i]), fp(div[i]))
        L.ref_ray_get_direction(fp(gd1[i]), 1)
        L.ref_ray_get_direction(fp(gd2[i]), 2)
        L.ref_ray_get_direction(fp(gd3[i]), 3)
    out.update(normalize=norm, length=length, dot=dot, cross=cross, div=div, getdir1=gd1, getdir2=gd2,
               getdir3=gd3)

    # cameras: the `v` blocks of balls_low / path_dof / tri100k + an oblique one, two resolutions
    cams = np.array([
        # from            at         up        angle hither rx   ry   aperture focal
        [2.1, 1.3, 1.7, 0, 0, 0, 0, 0, 1, 45, 0.01, 512, 512, 0, 1],
        [2.1, 1.3, 1.7, 0, 0, 0, 0, 0, 1, 45, 0.01, 1024, 1024, 0, 1],
        [7.5, 4.5, 2, 0, 0, 0, 0, 1, 0, 45, 1.0, 512, 512, 10, 1.0],
        [0, 0, 4.5, 0, 0, 0, 0, 1, 0, 35, 0.01, 2048, 2048, 0, 1],
        [-20, -5.7, 4, 4, 0, -4, 0, 1, 0, 30, 0.01, 640, 360, 4, 1.25],
    ], np.float32)
    m = 256
    px = rng.uniform(0, 1, (len(cams), m, 2)).astype(np.float32)
    lens = rng.uniform(-1, 1, (len(cams), m, 2)).astype(np.float32)
    ray_o = np.zeros((len(cams), m, 3), np.float32)
    ray_d = np.zeros((len(cams), m, 3), np.float32)
    lray_o = np.zeros((len(cams), m, 3), np.float32)
    lray_d = np.zeros((len(cams), m, 3), np.float32)
    cam_state = np.zeros((len(cams), 2), np.float32)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the Camera constructor printf()s (camera.h:61-62)
    try:
        for c, row in enumerate(cams):
            frm, at, up = row[0:3].copy(), row[3:6].copy(), row[6:9].copy()
            rx, ry = int(row[11]), int(row[12])
            L.ref_camera_create(fp(frm), fp(at), fp(up), float(row[9]), float(row[10]),
                                float(np.float32(100.0 * float(row[10]))), rx, ry, float(row[13]), float(row[14]))
            cam_state[c] = (L.ref_camera_aperture(), L.ref_camera_plane_dist())
            px[c, :, 0] *= rx
            px[c, :, 1] *= ry
            for k in range(m):
                L.ref_camera_primary(float(px[c, k, 0]), float(px[c, k, 1]), fp(ray_o[c, k]), fp(ray_d[c, k]))
                L.ref_camera_primary_lens(float(lens[c, k, 0]), float(lens[c, k, 1]), float(px[c, k, 0]),
                                          float(px[c, k, 1]), fp(lray_o[c, k]), fp(lray_d[c, k]))
    finally:
        os.dup2(saved, 1)
    out.update(cams=cams, cam_px=px, cam_lens=lens, cam_ray_o=ray_o, cam_ray_d=ray_d, cam_lray_o=lray_o,
               cam_lray_d=lray_d, cam_state=cam_state)

    # rand_float / sample_unit_disk on libc rand(): the oracle's rng_mode 1 consumes the same stream
    seeds = np.array([1, 12345, 20261003], np.uint32)
    rf = np.zeros((len(seeds), 64), np.float32)
    disk = np.zeros((len(seeds), 64, 2), np.float32)
    for s, seed in enumerate(seeds):
        L.ref_srand(int(seed))
        for k in range(64):
            rf[s, k] = L.ref_rand_float()
        L.ref_srand(int(seed))
        for k in range(64):
            L.ref_sample_unit_disk(fp(disk[s, k]))
    out.update(rand_seeds=seeds, rand_float=rf, unit_disk=disk)

    # u8fromfloat / u8tofloat / Color::clamp
    xs = np.concatenate([np.linspace(0, 1.1, 2048), rng.uniform(0, 1, 2048), [0.99609, 0.996094, 0.9961, 1.0]]).astype(np.float32)
    u8 = np.array([L.ref_u8fromfloat(float(x)) for x in xs], np.uint8)
    tof = np.array([L.ref_u8tofloat(i) for i in range(256)], np.float32)
    cl_in = rng.uniform(-0.5, 1.5, (512, 3)).astype(np.float32)
    cl = cl_in.copy()
    for i in range(len(cl)):
        L.ref_color_clamp(fp(cl[i]))
    out.update(u8_in=xs, u8_out=u8, u8tofloat=tof, clamp_in=cl_in, clamp_out=cl)
    os.makedirs(os.path.join(HERE, "ref_vectors"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "ref_vectors", "l0.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_vectors", "l0.npz"))


if __name__ == "__main__":
    main()
