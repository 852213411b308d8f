"""ctypes binding of the CPU oracle (oracle/libp3doracle.so).

TEST INFRASTRUCTURE.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libp3doracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libp3dref.so")


class OrcConfig(C.Structure):
    _fields_ = [
        ("integrator", C.c_int32), ("accel", C.c_int32), ("max_depth", C.c_int32),
        ("spp_sqrt", C.c_int32), ("antialiasing", C.c_int32), ("depth_of_field", C.c_int32),
        ("sample_disk", C.c_int32), ("soft_shadows", C.c_int32), ("sample_mode", C.c_int32),
        ("light_side", C.c_float), ("gamma", C.c_float), ("skybox", C.c_int32),
        ("rng_mode", C.c_int32), ("stack_mode", C.c_int32), ("trace_zero_weight", C.c_int32),
        ("eval_order", C.c_int32), ("math_mode", C.c_int32), ("threads", C.c_int32),
        ("debug_view", C.c_int32), ("seed", C.c_uint64),
    ]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "rays_bounce", "rays_light",
        "node_tests", "sphere_tests", "tri_tests", "box_tests", "plane_tests", "shaded_hits",
        "pixels", "max_stack", "ref_ray_counter")] + [("seconds", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    @property
    def rays(self):
        return (self.rays_primary + self.rays_shadow + self.rays_reflect + self.rays_refract
                + self.rays_bounce + self.rays_light)


def build(force=False):
    """Compile the oracle (and oracle/_ref when the reference is present)."""
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "p3d_oracle.cpp"))):
        subprocess.check_call(["make", "-C", HERE, "libp3doracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/Raytracing") and (force or not os.path.exists(REF_LIB_PATH)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_load.restype = C.c_void_p
        L.orc_scene_load.argtypes = [C.c_char_p, C.c_int]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_det_sin.restype = C.c_double
        L.orc_det_sin.argtypes = [C.c_double]
        L.orc_det_cos.restype = C.c_double
        L.orc_det_cos.argtypes = [C.c_double]
        L.orc_u8fromfloat.restype = C.c_uint8
        L.orc_u8fromfloat.argtypes = [C.c_float]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_config(**kw):
    c = OrcConfig()
    lib().orc_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def whitted_config(accel, max_depth, **kw):
    """SURVEY.md §8(d) Whitted configs: no AA, no soft shadows, no DOF."""
    base = dict(integrator=0, accel=accel, max_depth=max_depth, spp_sqrt=1, antialiasing=0,
                depth_of_field=0, soft_shadows=0)
    base.update(kw)
    return default_config(**base)


class Scene:
    def __init__(self, path, legacy_f11=False):
        self._L = lib()
        self._h = self._L.orc_scene_load(os.fsencode(path), int(legacy_f11))
        if not self._h:
            raise IOError("oracle: cannot open %s" % path)
        self.path = path

    def close(self):
        if self._h:
            self._L.orc_scene_free(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def h(self):
        return C.c_void_p(self._h)

    def counts(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._L.orc_scene_counts(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(objects=a.value, lights=b.value, materials=c.value, has_camera=bool(d.value))

    def set_resolution(self, rx, ry):
        if self._L.orc_scene_set_resolution(self.h, rx, ry) != 0:
            raise RuntimeError("no camera")

    def set_lens(self, aperture_ratio, focal_ratio):
        if self._L.orc_scene_set_lens(self.h, C.c_float(aperture_ratio), C.c_float(focal_ratio)) != 0:
            raise RuntimeError("no camera")

    def replicate_lights(self, spp_sqrt, light_side):
        self._L.orc_scene_replicate_lights(self.h, spp_sqrt, C.c_float(light_side))

    def resolution(self):
        return tuple(int(v) for v in self.camera()["res"])

    def camera(self):
        eye, u, v, n = (np.zeros(3, np.float32) for _ in range(4))
        whdfa = np.zeros(5, np.float32)
        res = np.zeros(2, np.int32)
        if self._L.orc_scene_camera(self.h, _fp(eye), _fp(u), _fp(v), _fp(n), _fp(whdfa),
                                    res.ctypes.data_as(C.POINTER(C.c_int))) != 0:
            raise RuntimeError("no camera")
        return dict(eye=eye, u=u, v=v, n=n, w=whdfa[0], h=whdfa[1], plane_dist=whdfa[2],
                    focal_ratio=whdfa[3], aperture=whdfa[4], res=res)

    def set_skybox(self, faces):
        """faces: 6 uint8 arrays (h, w, 3|4), row 0 = bottom row, order RIGHT LEFT TOP BOTTOM FRONT BACK."""
        for i, f in enumerate(faces):
            f = np.ascontiguousarray(f, np.uint8)
            rc = self._L.orc_scene_set_skybox_face(self.h, i, f.ctypes.data_as(C.POINTER(C.c_uint8)), f.shape[1], f.shape[0], f.shape[2])
            if rc != 0:
                raise ValueError("bad skybox face %d" % i)

    def background(self):
        b = np.zeros(3, np.float32)
        self._L.orc_scene_background(self.h, _fp(b))
        return b

    def object(self, i):
        t, m = C.c_int(), C.c_int()
        v = np.zeros(9, np.float32)
        n, mn, mx = (np.zeros(3, np.float32) for _ in range(3))
        if self._L.orc_scene_object(self.h, i, C.byref(t), C.byref(m), _fp(v), _fp(n), _fp(mn), _fp(mx)) != 0:
            raise IndexError(i)
        return dict(type=t.value, material=m.value, v=v, n=n, bmin=mn, bmax=mx)

    def material(self, i):
        m = np.zeros(16, np.float32)
        if self._L.orc_scene_material(self.h, i, _fp(m)) != 0:
            raise IndexError(i)
        return m

    def light(self, i):
        p, c = np.zeros(3, np.float32), np.zeros(3, np.float32)
        if self._L.orc_scene_light(self.h, i, _fp(p), _fp(c)) != 0:
            raise IndexError(i)
        return p, c

    def build_bvh(self):
        self._L.orc_build_bvh(self.h)

    def build_grid(self):
        self._L.orc_build_grid(self.h)

    def bvh_info(self):
        self.build_bvh()
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._L.orc_bvh_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(nodes=a.value, leaves=b.value, max_depth=c.value)

    def bvh_nodes(self):
        n = self.bvh_info()["nodes"]
        bmin = np.zeros((n, 3), np.float32)
        bmax = np.zeros((n, 3), np.float32)
        index = np.zeros(n, np.uint32)
        nobj = np.zeros(n, np.uint32)
        leaf = np.zeros(n, np.uint8)
        self._L.orc_bvh_nodes(self.h, _fp(bmin), _fp(bmax), index.ctypes.data_as(C.POINTER(C.c_uint32)),
                              nobj.ctypes.data_as(C.POINTER(C.c_uint32)),
                              leaf.ctypes.data_as(C.POINTER(C.c_uint8)))
        order = np.zeros(self.counts()["objects"], np.int32)
        self._L.orc_bvh_order(self.h, order.ctypes.data_as(C.POINTER(C.c_int32)))
        return dict(bmin=bmin, bmax=bmax, index=index, n_objs=nobj, leaf=leaf, order=order)

    def grid(self):
        self.build_grid()
        nxyz = np.zeros(3, np.int32)
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        ni = C.c_int()
        self._L.orc_grid_info(self.h, nxyz.ctypes.data_as(C.POINTER(C.c_int)), _fp(mn), _fp(mx), C.byref(ni))
        ncell = int(nxyz[0]) * int(nxyz[1]) * int(nxyz[2])
        start = np.zeros(ncell + 1, np.uint32)
        items = np.zeros(max(ni.value, 1), np.uint32)
        self._L.orc_grid_cells(self.h, start.ctypes.data_as(C.POINTER(C.c_uint32)),
                               items.ctypes.data_as(C.POINTER(C.c_uint32)))
        return dict(n=nxyz, bmin=mn, bmax=mx, cell_start=start, cell_items=items[:ni.value])

    def render_repeat(self, cfg, repeat, x0=0, y0=0, w=None, h=None):
        """The tile rendered `repeat` times by one thread pool (orc_render_repeat): timing runs. -> stats over all repetitions."""
        rx, ry = self.resolution()
        w = int(rx - x0 if w is None else w)
        h = int(ry - y0 if h is None else h)
        rgb = np.zeros((h, w, 3), np.float32)
        hit = np.zeros((h, w), np.int32)
        st = OrcStats()
        rc = self._L.orc_render_repeat(self.h, C.byref(cfg), int(x0), int(y0), w, h, int(repeat), _fp(rgb),
                                       hit.ctypes.data_as(C.POINTER(C.c_int32)), None, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render_repeat failed rc=%d" % rc)
        return st

    def render(self, cfg, x0=0, y0=0, w=None, h=None, want_rgb8=False):
        rx, ry = self.resolution()
        w = int(rx - x0 if w is None else w)
        h = int(ry - y0 if h is None else h)
        x0, y0 = int(x0), int(y0)
        rgb = np.zeros((h, w, 3), np.float32)
        hit = np.zeros((h, w), np.int32)
        rgb8 = np.zeros((h, w, 3), np.uint8) if want_rgb8 else None
        st = OrcStats()
        rc = self._L.orc_render(self.h, C.byref(cfg), x0, y0, w, h, _fp(rgb),
                                hit.ctypes.data_as(C.POINTER(C.c_int32)),
                                rgb8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_rgb8 else None,
                                C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render failed rc=%d" % rc)
        if want_rgb8:
            return rgb, hit, rgb8, st
        return rgb, hit, st

    def trace_closest(self, accel, o, d):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        hp = np.zeros((n, 3), np.float32)
        self._L.orc_trace_closest(self.h, accel, n, _fp(o), _fp(d), hit.ctypes.data_as(C.POINTER(C.c_int32)),
                                  _fp(t), _fp(hp))
        return hit, t, hp

    def trace_any(self, accel, o, d):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        occ = np.zeros(n, np.uint8)
        self._L.orc_trace_any(self.h, accel, n, _fp(o), _fp(d), occ.ctypes.data_as(C.POINTER(C.c_uint8)))
        return occ

    def primary_ray(self, px, py):
        o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._L.orc_primary_ray(self.h, C.c_float(px), C.c_float(py), _fp(o), _fp(d))
        return o, d

    def primary_ray_lens(self, lx, ly, px, py):
        o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._L.orc_primary_ray_lens(self.h, C.c_float(lx), C.c_float(ly), C.c_float(px), C.c_float(py),
                                     _fp(o), _fp(d))
        return o, d

    def object_intercepts(self, obj, o, d):
        o = np.array(o, np.float32)
        d = np.array(d, np.float32)
        t = C.c_float()
        h = self._L.orc_object_intercepts(self.h, obj, _fp(o), _fp(d), C.byref(t))
        return bool(h), t.value, d

    def skybox_color(self, d):
        d = np.array(d, np.float32)
        c = np.zeros(3, np.float32)
        if self._L.orc_skybox_color(self.h, _fp(d), _fp(c)) != 0:
            raise RuntimeError("no skybox loaded")
        return c

    def object_normal(self, obj, p):
        p = np.array(p, np.float32)
        n = np.zeros(3, np.float32)
        self._L.orc_object_normal(self.h, obj, _fp(p), _fp(n))
        return n


def aabb_intercepts(bmin, bmax, o, d):
    bmin, bmax, o, d = (np.array(a, np.float32) for a in (bmin, bmax, o, d))
    t = C.c_float()
    h = lib().orc_aabb_intercepts(_fp(bmin), _fp(bmax), _fp(o), _fp(d), C.byref(t))
    return bool(h), t.value


def rng_stream(seed, pixel, sample, n):
    out = np.zeros(n, np.uint32)
    lib().orc_rng_stream(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), n,
                         out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def det_sin(x):
    return lib().orc_det_sin(float(x))


def det_cos(x):
    return lib().orc_det_cos(float(x))
