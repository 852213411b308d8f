"""p3d-raytracer_amd — Python face of libp3d.so (the C-ABI of include/p3d.h).

Thin ctypes plumbing only: the product is the shared library (host C++ loader / BVH /
grid builders + gfx950 HIP kernels).  Nothing here computes pixels, and nothing here
touches oracle/: if libp3d.so is missing, or there is no HIP device, calls fail loudly.

The directory name has a hyphen, so import it through the root-level shim:

    import p3d_amd as p3d        # repo root on sys.path
    hs = p3d.HostScene("scene.p3f")
    dev = p3d.DeviceScene(hs, bvh=True)
    rgb, hit, stats = dev.render(p3d.whitted_config(accel=p3d.ACCEL_BVH, max_depth=4))
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P3D_LIB") or os.path.join(HERE, "libp3d.so")  # P3D_LIB: kernel-variant experiments

ACCEL_NONE, ACCEL_GRID, ACCEL_BVH = 0, 1, 2
WHITTED, PATHTRACE = 0, 1
TILE_ORDER_COST, TILE_ORDER_FRAME = 0, 1
STACK_LITERAL, STACK_PER_PIXEL = 0, 1
CHAIN_AUTO, CHAIN_MEGAKERNEL, CHAIN_PER_LEVEL = 0, 1, 2
SAMPLE_JITTER, SAMPLE_TENT = 0, 1
LOAD_LEGACY_F11 = 1
DEBUG_NONE, DEBUG_TEST_INTERSECT, DEBUG_DEPTH_MAP = 0, 1, 2
HANDOFF_COMPACT, HANDOFF_DENSE = 0, 1

EXPORTS = [
    "p3d_abi_version", "p3d_last_error", "p3d_device_count", "p3d_config_default",
    "p3d_scene_create", "p3d_scene_create_device_bvh", "p3d_scene_destroy", "p3d_scene_set_skybox", "p3d_render_tile", "p3d_render_tile_device",
    "p3d_scene_status", "p3d_scene_set_tail_stream", "p3d_scene_join", "p3d_object_intercepts", "p3d_object_normal", "p3d_skybox_color",
    "p3d_trace_closest", "p3d_trace_any", "p3d_host_scene_load", "p3d_host_scene_destroy",
    "p3d_host_scene_set_resolution", "p3d_host_scene_set_lens", "p3d_host_scene_replicate_lights",
    "p3d_host_scene_desc", "p3d_host_scene_bind_device", "p3d_host_scene_has_skybox", "p3d_host_scene_load_skybox", "p3d_host_scene_skybox_face",
]


class P3DError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("p3d error %d: %s" % (code, msg))
        self.code = code


class Prim(C.Structure):
    _fields_ = [("v", C.c_float * 9), ("type", C.c_uint32), ("material", C.c_uint32), ("reserved0", C.c_uint32),
                ("n", C.c_float * 3), ("reserved1", C.c_uint32), ("bmin", C.c_float * 3), ("reserved2", C.c_uint32),
                ("bmax", C.c_float * 3), ("reserved3", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("diff_color", C.c_float * 3), ("diffuse", C.c_float), ("spec_color", C.c_float * 3),
                ("specular", C.c_float), ("shine", C.c_float), ("transmittance", C.c_float),
                ("refr_index", C.c_float), ("reflection", C.c_float), ("emission", C.c_float * 3),
                ("reserved", C.c_float)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("reserved0", C.c_float), ("color", C.c_float * 3),
                ("reserved1", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("plane_dist", C.c_float), ("u", C.c_float * 3), ("w", C.c_float),
                ("v", C.c_float * 3), ("h", C.c_float), ("n", C.c_float * 3), ("focal_ratio", C.c_float),
                ("aperture", C.c_float), ("res_x", C.c_int32), ("res_y", C.c_int32), ("reserved", C.c_int32)]


class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("index", C.c_uint32), ("bmax", C.c_float * 3), ("count_leaf", C.c_uint32)]


class GridDesc(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("nx", C.c_int32), ("bmax", C.c_float * 3), ("ny", C.c_int32),
                ("nz", C.c_int32), ("n_cells", C.c_uint32), ("n_items", C.c_uint32), ("reserved", C.c_uint32),
                ("cell_start", C.POINTER(C.c_uint32)), ("cell_items", C.POINTER(C.c_uint32))]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_prims", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_lights", C.c_uint32), ("prims", C.POINTER(Prim)), ("materials", C.POINTER(Material)),
                ("lights", C.POINTER(Light)), ("camera", Camera), ("background", C.c_float * 3),
                ("n_bvh_nodes", C.c_uint32), ("bvh_nodes", C.POINTER(BvhNode)),
                ("bvh_prim_index", C.POINTER(C.c_uint32)), ("n_bvh_prim_index", C.c_uint32),
                ("bvh_max_depth", C.c_uint32), ("has_grid", C.c_uint32), ("reserved", C.c_uint32),
                ("grid", GridDesc)]


class Config(C.Structure):
    _fields_ = [("integrator", C.c_uint32), ("accel", C.c_uint32), ("max_depth", C.c_int32),
                ("spp_sqrt", C.c_uint32), ("antialiasing", C.c_uint32), ("depth_of_field", C.c_uint32),
                ("sample_disk", C.c_uint32), ("soft_shadows", C.c_uint32), ("sample_mode", C.c_uint32),
                ("light_side", C.c_float), ("gamma", C.c_float), ("collect_stats", C.c_uint32),
                ("skybox", C.c_uint32), ("tile_order", C.c_uint32), ("seed", C.c_uint64),
                ("stack_mode", C.c_uint32), ("chain_launch", C.c_uint32), ("debug_view", C.c_uint32), ("handoff_records", C.c_uint32)]


class SkyboxFace(C.Structure):
    _fields_ = [("img", C.c_void_p), ("res_x", C.c_uint32), ("res_y", C.c_uint32), ("bpp", C.c_uint32),
                ("reserved", C.c_uint32)]


class SkyboxDesc(C.Structure):
    _fields_ = [("face", SkyboxFace * 6)]


class Tile(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("stripe_h", C.c_int32), ("stripe_stride", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "rays_bounce", "rays_light",
        "node_tests", "sphere_tests", "tri_tests", "box_tests", "plane_tests", "shaded_hits", "pixels",
        "max_stack")] + [("kernel_ms", C.c_double)] + [(n, C.c_uint64) for n in (
        "handoff_checked", "handoff_redone", "handoff_rounds")] + [("pass1_ms", C.c_double), ("handoff_ms", C.c_double),
                                                                ("handoff_dense_retry", C.c_uint64)]

    @property
    def rays(self):
        return (self.rays_primary + self.rays_shadow + self.rays_reflect + self.rays_refract
                + self.rays_bounce + self.rays_light)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}

    def algorithmic_bytes(self):
        """DESIGN.md / SURVEY.md §8(d): bytes the algorithm has to look at."""
        return (32 * self.node_tests + 16 * self.sphere_tests + 48 * self.tri_tests + 24 * self.box_tests
                + 24 * self.plane_tests + 64 * self.shaded_hits + 16 * self.pixels)


def build(force=False):
    """Compile libp3d.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".hpp"))]
    srcs += [os.path.join(HERE, "host", f) for f in os.listdir(os.path.join(HERE, "host")) if f.endswith((".cpp", ".hpp"))]
    srcs.append(os.path.join(HERE, "..", "include", "p3d.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            raise RuntimeError("libp3d.so is missing/stale and hipcc is not available to build it")
        subprocess.check_call(["make", "-C", HERE, "libp3d.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load libp3d.so.  Raises if it is absent: there is no Python or CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C p3d-raytracer_amd`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.p3d_last_error.restype = C.c_char_p
        L.p3d_abi_version.restype = C.c_uint32
        L.p3d_host_scene_load.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.p3d_host_scene_destroy.argtypes = [C.c_void_p]
        L.p3d_host_scene_set_resolution.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.p3d_host_scene_set_lens.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.p3d_host_scene_replicate_lights.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        L.p3d_host_scene_desc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(SceneDesc))]
        L.p3d_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.p3d_host_scene_bind_device.argtypes = [C.c_void_p, C.c_void_p]
        L.p3d_host_scene_has_skybox.argtypes = [C.c_void_p]
        L.p3d_host_scene_load_skybox.argtypes = [C.c_void_p, C.c_char_p]
        L.p3d_host_scene_skybox_face.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.p3d_scene_destroy.argtypes = [C.c_void_p]
        L.p3d_scene_set_skybox.argtypes = [C.c_void_p, C.POINTER(SkyboxDesc)]
        L.p3d_render_tile.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(Tile), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(Stats)]
        L.p3d_render_tile_device.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(Tile), C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.p3d_trace_closest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.p3d_trace_any.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.p3d_object_intercepts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.p3d_object_normal.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.p3d_skybox_color.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.p3d_scene_status.argtypes = [C.c_void_p]
        L.p3d_scene_set_tail_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.p3d_scene_join.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.p3d_debug_scene_limits.argtypes = [C.c_void_p, C.c_void_p]  # csrc/p3d_debug.h: not part of include/p3d.h
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise P3DError(rc, lib().p3d_last_error().decode("utf-8", "replace"))


def device_count():
    n = lib().p3d_device_count()
    return max(n, 0)


def default_config(**kw):
    c = Config()
    lib().p3d_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def whitted_config(accel=ACCEL_BVH, max_depth=4, **kw):
    """SURVEY.md §8(d) Whitted configs: no AA / soft shadows / DOF, one ray per pixel."""
    base = dict(integrator=WHITTED, accel=accel, max_depth=max_depth, spp_sqrt=1, antialiasing=0,
                depth_of_field=0, soft_shadows=0)
    base.update(kw)
    return default_config(**base)


def pathtrace_config(accel=ACCEL_BVH, spp_sqrt=16, max_depth=20, dof=0, **kw):
    base = dict(integrator=PATHTRACE, accel=accel, max_depth=max_depth, spp_sqrt=spp_sqrt, antialiasing=1,
                depth_of_field=dof, sample_disk=1, soft_shadows=0)
    base.update(kw)
    return default_config(**base)


class HostScene:
    """Scene::load_p3f + BVH::build / Grid::Build on the host (C++ inside libp3d.so)."""

    def __init__(self, path, legacy_f11=False):
        self._L = lib()
        h = C.c_void_p()
        _check(self._L.p3d_host_scene_load(os.fsencode(path), LOAD_LEGACY_F11 if legacy_f11 else 0, C.byref(h)))
        self._h = h
        self.path = path

    def close(self):
        if getattr(self, "_h", None):
            self._L.p3d_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_resolution(self, rx, ry):
        _check(self._L.p3d_host_scene_set_resolution(self._h, int(rx), int(ry)))

    def set_lens(self, aperture_ratio, focal_ratio):
        _check(self._L.p3d_host_scene_set_lens(self._h, float(aperture_ratio), float(focal_ratio)))

    def replicate_lights(self, spp_sqrt, light_side):
        _check(self._L.p3d_host_scene_replicate_lights(self._h, int(spp_sqrt), float(light_side)))

    def has_skybox(self):
        """A cubemap is loaded: the folder of the scene's `env` line was found, or load_skybox was called (p3d_host_scene_has_skybox)."""
        return bool(self._L.p3d_host_scene_has_skybox(self._h))

    def load_skybox(self, sky_dir):
        """Scene::LoadSkybox (scene.cpp:329-377): the six faces of `sky_dir` as .jpg (decoded by the library) or .ppm."""
        _check(self._L.p3d_host_scene_load_skybox(self._h, os.fsencode(sky_dir)))

    def skybox_face(self, face):
        """Scene::skybox_img[face] as the library decoded it: (res_y, res_x, 3) uint8, bottom row first."""
        img = C.POINTER(C.c_uint8)()
        w, h = C.c_uint32(), C.c_uint32()
        _check(self._L.p3d_host_scene_skybox_face(self._h, int(face), C.byref(img), C.byref(w), C.byref(h)))
        return np.ctypeslib.as_array(img, shape=(h.value, w.value, 3)).copy()

    def desc(self, bvh=False, grid=False):
        p = C.POINTER(SceneDesc)()
        _check(self._L.p3d_host_scene_desc(self._h, int(bvh), int(grid), C.byref(p)))
        return p.contents

    # numpy views of the flattened arrays (copies), for host-logic tests
    def arrays(self, bvh=False, grid=False):
        d = self.desc(bvh, grid)
        out = dict(n_prims=d.n_prims, n_materials=d.n_materials, n_lights=d.n_lights,
                   res=(d.camera.res_x, d.camera.res_y), background=np.array(d.background[:], np.float32))
        prims = [d.prims[i] for i in range(d.n_prims)] if d.n_prims else []
        out["prim_v"] = np.array([list(p.v) for p in prims], np.float32).reshape(-1, 9)
        out["prim_type"] = np.array([p.type for p in prims], np.uint32)
        out["prim_material"] = np.array([p.material for p in prims], np.uint32)
        out["prim_n"] = np.array([list(p.n) for p in prims], np.float32).reshape(-1, 3)
        out["prim_bmin"] = np.array([list(p.bmin) for p in prims], np.float32).reshape(-1, 3)
        out["prim_bmax"] = np.array([list(p.bmax) for p in prims], np.float32).reshape(-1, 3)
        mats = [d.materials[i] for i in range(d.n_materials)]
        out["materials"] = np.array([[*m.diff_color, m.diffuse, *m.spec_color, m.specular, m.shine, m.transmittance,
                                      m.refr_index, m.reflection, *m.emission, 0.0] for m in mats],
                                    np.float32).reshape(-1, 16)
        out["lights"] = np.array([[*d.lights[i].position, *d.lights[i].color] for i in range(d.n_lights)],
                                 np.float32).reshape(-1, 6)
        c = d.camera
        out["camera"] = dict(eye=np.array(c.eye[:], np.float32), u=np.array(c.u[:], np.float32),
                             v=np.array(c.v[:], np.float32), n=np.array(c.n[:], np.float32), w=c.w, h=c.h,
                             plane_dist=c.plane_dist, focal_ratio=c.focal_ratio, aperture=c.aperture)
        if bvh:
            n = d.n_bvh_nodes
            buf = np.ctypeslib.as_array(C.cast(d.bvh_nodes, C.POINTER(C.c_uint32)), shape=(n, 8)).copy()
            out["bvh_bmin"] = buf[:, 0:3].view(np.float32)
            out["bvh_index"] = buf[:, 3].copy()
            out["bvh_bmax"] = buf[:, 4:7].view(np.float32)
            out["bvh_count_leaf"] = buf[:, 7].copy()
            out["bvh_order"] = (np.ctypeslib.as_array(d.bvh_prim_index, shape=(d.n_bvh_prim_index,)).copy()
                                if d.n_bvh_prim_index else np.zeros(0, np.uint32))
            out["bvh_max_depth"] = d.bvh_max_depth
        if grid:
            g = d.grid
            out["grid_n"] = (g.nx, g.ny, g.nz)
            out["grid_bmin"] = np.array(g.bmin[:], np.float32)
            out["grid_bmax"] = np.array(g.bmax[:], np.float32)
            out["grid_cell_start"] = np.ctypeslib.as_array(g.cell_start, shape=(g.n_cells + 1,)).copy()
            out["grid_cell_items"] = (np.ctypeslib.as_array(g.cell_items, shape=(g.n_items,)).copy()
                                      if g.n_items else np.zeros(0, np.uint32))
        return out


class DeviceScene:
    """p3d_scene_create: the flattened scene resident in HBM of one MI355X."""

    def __init__(self, host_scene, bvh=True, grid=False, device=0):
        """bvh: True = the reference-exact tree built on the host (BVH::build), "device" = a linear BVH built
        on the GPU (p3d_scene_create_device_bvh: correct closest hits, not the reference's tree), False = none."""
        self._L = lib()
        self.host = host_scene
        self.device_bvh_ms = None
        h = C.c_void_p()
        if bvh == "device":
            d = host_scene.desc(False, grid)
            ms = C.c_float(0)
            _check(self._L.p3d_scene_create_device_bvh(C.byref(d), int(device), C.byref(h), C.byref(ms)))
            self.device_bvh_ms = ms.value
        else:
            d = host_scene.desc(bool(bvh), grid)
            _check(self._L.p3d_scene_create(C.byref(d), int(device), C.byref(h)))
        self.res = (d.camera.res_x, d.camera.res_y)
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.p3d_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_skybox(self, faces):
        """faces: 6 uint8 arrays (h, w, 3|4), row 0 = BOTTOM row, order RIGHT LEFT TOP BOTTOM FRONT BACK
        (what Scene::LoadSkybox keeps, scene.cpp:329-377).  See load_skybox_dir for JPEG folders."""
        keep = [np.ascontiguousarray(f, np.uint8) for f in faces]
        d = SkyboxDesc()
        for i, f in enumerate(keep):
            d.face[i].img = f.ctypes.data
            d.face[i].res_x, d.face[i].res_y, d.face[i].bpp = f.shape[1], f.shape[0], f.shape[2]
        _check(self._L.p3d_scene_set_skybox(self._h, C.byref(d)))

    def bind_host(self):
        """p3d_host_scene_bind_device: the host classes' query forwards answer from this device scene, and the cubemap the
        loader read for an `env` line is uploaded."""
        _check(self._L.p3d_host_scene_bind_device(self.host._h, self._h))

    def full_tile(self):
        return Tile(0, 0, self.res[0], self.res[1], 0, 1)

    def render(self, cfg, tile=None, want_rgb8=False, stats=True):
        """Host-buffer form (p3d_render_tile): returns numpy arrays."""
        t = tile or self.full_tile()
        rgb = np.zeros((t.h, t.w, 3), np.float32)
        hit = np.zeros((t.h, t.w), np.int32)
        rgb8 = np.zeros((t.h, t.w, 3), np.uint8) if want_rgb8 else None
        st = Stats()
        _check(self._L.p3d_render_tile(self._h, C.byref(cfg), C.byref(t), rgb.ctypes.data, hit.ctypes.data,
                                       rgb8.ctypes.data if want_rgb8 else None, C.byref(st) if stats else None))
        if want_rgb8:
            return rgb, hit, rgb8, st
        return rgb, hit, st

    def status(self):
        """p3d_scene_status: waits for the device, returns P3D_OK (0) or the code of a device-detected error of the
        asynchronous render_device calls since the last check (message: last_error())."""
        return int(self._L.p3d_scene_status(self._h))

    def set_tail_stream(self, stream=None):
        """p3d_scene_set_tail_stream: the launches behind pass 1 of a literal frame (the hit_stack hand-off) go to `stream` (a
        torch.cuda.Stream, a raw hipStream_t or None = off); render_device calls without stats then return their outputs when
        that stream has passed them - see join()."""
        raw = getattr(stream, "cuda_stream", stream)
        _check(self._L.p3d_scene_set_tail_stream(self._h, C.c_void_p(raw) if raw else None))

    def join(self, stream=None, host_wait=False):
        """p3d_scene_join: make `stream` (or, with host_wait, the calling thread) wait for the scene's last frame, tail included."""
        raw = getattr(stream, "cuda_stream", stream)
        _check(self._L.p3d_scene_join(self._h, C.c_void_p(raw) if raw else None, 1 if host_wait else 0))

    def debug_limits(self, trip_bound=0, max_rounds=0, halo_chain=0, leftover_pool=0):
        """Tests only (csrc/p3d_debug.h, not part of include/p3d.h): shrink limits of THIS scene so that the device-side
        error paths fire; all 0 = the real limits.  Refused unless the process was started with P3D_TEST_HOOKS=1."""
        lim = (C.c_uint32 * 4)(trip_bound, max_rounds, halo_chain, leftover_pool)
        _check(self._L.p3d_debug_scene_limits(self._h, C.cast(lim, C.c_void_p)))

    def render_device(self, cfg, tile, d_rgb=0, d_hit=0, d_rgb8=0, stream=0, stats=None):
        """Device-buffer form: raw HBM addresses (e.g. torch.Tensor.data_ptr()) and a hipStream_t."""
        _check(self._L.p3d_render_tile_device(self._h, C.byref(cfg), C.byref(tile), C.c_void_p(d_rgb or None),
                                              C.c_void_p(d_hit or None), C.c_void_p(d_rgb8 or None),
                                              C.c_void_p(stream or None), C.byref(stats) if stats is not None else None))

    def trace_closest(self, accel, origin, direction, want_t=False):
        o = np.ascontiguousarray(origin, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        hp = np.zeros((n, 3), np.float32)
        t = np.zeros(n, np.float32) if want_t else None
        _check(self._L.p3d_trace_closest(self._h, int(accel), n, o.ctypes.data, d.ctypes.data, hit.ctypes.data,
                                         t.ctypes.data if want_t else None, hp.ctypes.data))
        return (hit, hp, t) if want_t else (hit, hp)

    def object_intercepts(self, obj, origin, direction):
        """Object::intercepts for n rays: -> (hit, t, direction as the test left it)."""
        o = np.ascontiguousarray(origin, np.float32)
        d = np.array(direction, np.float32, order="C")
        n = o.shape[0]
        hit = np.zeros(n, np.uint8)
        t = np.zeros(n, np.float32)
        _check(self._L.p3d_object_intercepts(self._h, int(obj), n, C.c_void_p(o.ctypes.data), C.c_void_p(d.ctypes.data),
                                             C.c_void_p(hit.ctypes.data), C.c_void_p(t.ctypes.data)))
        return hit.astype(bool), t, d

    def object_normal(self, obj, points):
        p = np.ascontiguousarray(points, np.float32)
        out = np.zeros_like(p)
        _check(self._L.p3d_object_normal(self._h, int(obj), p.shape[0], C.c_void_p(p.ctypes.data), C.c_void_p(out.ctypes.data)))
        return out

    def skybox_color(self, directions):
        d = np.ascontiguousarray(directions, np.float32)
        out = np.zeros_like(d)
        _check(self._L.p3d_skybox_color(self._h, d.shape[0], C.c_void_p(d.ctypes.data), C.c_void_p(out.ctypes.data)))
        return out

    def trace_any(self, accel, origin, direction):
        o = np.ascontiguousarray(origin, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        n = o.shape[0]
        occ = np.zeros(n, np.uint8)
        _check(self._L.p3d_trace_any(self._h, int(accel), n, o.ctypes.data, d.ctypes.data, occ.ctypes.data))
        return occ


def stripe_tile(res, rank, world, stripe_h=16):
    """Tile of rank `rank` in an N-rank job: every world-th stripe of stripe_h rows (DESIGN.md multi-GPU).
    Requires res_y % (stripe_h * world) == 0 so that every rank renders the same number of rows."""
    rx, ry = res
    if ry % (stripe_h * world) != 0:
        raise ValueError("res_y=%d must be a multiple of stripe_h*world=%d" % (ry, stripe_h * world))
    return Tile(0, rank * stripe_h, rx, ry // world, stripe_h, world)


def stripe_rows(res, rank, world, stripe_h=16):
    """Image rows (bottom-up numbering) that `stripe_tile` renders, in local-row order."""
    t = stripe_tile(res, rank, world, stripe_h)
    r = np.arange(t.h)
    return t.y0 + (r // stripe_h) * stripe_h * world + (r % stripe_h)


# ---- multi-GPU helpers (one process per GPU; torch.distributed moves the bytes) ----
def packed_bytes(n_pixels):
    """Per-rank framebuffer layout used for the gather: [rgb float32 x3 | hit int32] = 16 B/pixel."""
    return n_pixels * 16


def assemble_frame(gathered, res, world, stripe_h, frame_rgb=None, frame_hit=None, batch=1):
    """De-interleave the per-rank stripe buffers (torch uint8 tensors, `batch` consecutive frames each
    laid out as packed_bytes) into full frames: stripe s of rank r holds frame rows
    (s*world + r)*stripe_h ... +stripe_h.  Returns (rgb, hit) of shape (ry, rx, 3) / (ry, rx), with a
    leading batch axis when batch > 1."""
    import torch
    rx, ry = res
    n_str = ry // (stripe_h * world)
    n_local = rx * (ry // world)
    dev = gathered[0].device
    if frame_rgb is None:
        frame_rgb = torch.empty((batch, ry, rx, 3), dtype=torch.float32, device=dev)
    if frame_hit is None:
        frame_hit = torch.empty((batch, ry, rx), dtype=torch.int32, device=dev)
    per_frame = [g.view(batch, n_local * 16) for g in gathered]
    parts = [g[:, : n_local * 12].contiguous().view(torch.float32).view(batch, n_str, stripe_h, rx, 3) for g in per_frame]
    frame_rgb.view(batch, n_str, world, stripe_h, rx, 3).copy_(torch.stack(parts, dim=2))
    hits = [g[:, n_local * 12:].contiguous().view(torch.int32).view(batch, n_str, stripe_h, rx) for g in per_frame]
    frame_hit.view(batch, n_str, world, stripe_h, rx).copy_(torch.stack(hits, dim=2))
    if batch == 1:
        return frame_rgb.view(ry, rx, 3), frame_hit.view(ry, rx)
    return frame_rgb, frame_hit


_GATHER_MODE = {"mode": "gather"}


def gather_frame(local_buf, res, rank, world, stripe_h, dst=0, gathered=None, async_op=False):
    """One collective for a frame (or a batch of frames, rendered back to back into one buffer): every
    rank's packed stripe buffer to rank `dst` (RCCL on GPUs, gloo in the CPU tests).  Returns (work handle or None, gather list or None).
    `gather` is the natural all-to-one; should a backend build not provide it, the first failure
    switches this process group to `all_gather` (same bytes into rank `dst`, the other ranks simply
    ignore what they receive) - every rank hits the same error at the same call, so they switch
    together."""
    import torch
    import torch.distributed as dist
    if _GATHER_MODE["mode"] == "gather":
        if rank == dst and gathered is None:
            gathered = [torch.empty_like(local_buf) for _ in range(world)]
        try:
            work = dist.gather(local_buf, gathered if rank == dst else None, dst=dst, async_op=async_op)
            return work, gathered
        except (RuntimeError, NotImplementedError):
            _GATHER_MODE["mode"] = "all_gather"
    if gathered is None:
        gathered = [torch.empty_like(local_buf) for _ in range(world)]
    work = dist.all_gather(gathered, local_buf, async_op=async_op)
    return work, gathered


def assemble_frame8(gathered, res, world, stripe_h, frame8=None, batch=1):
    """Same de-interleave for u8 frames (img_Data, 3 B/pixel): the payload bench.py gathers by default."""
    import torch
    rx, ry = res
    n_str = ry // (stripe_h * world)
    if frame8 is None:
        frame8 = torch.empty((batch, ry, rx, 3), dtype=torch.uint8, device=gathered[0].device)
    parts = [g.view(batch, n_str, stripe_h, rx, 3) for g in gathered]
    frame8.view(batch, n_str, world, stripe_h, rx, 3).copy_(torch.stack(parts, dim=2))
    return frame8.view(ry, rx, 3) if batch == 1 else frame8


SKYBOX_FACE_FILES = ("right", "left", "top", "bottom", "front", "back")  # scene.cpp:333


def load_skybox_dir(sky_dir, ext=".jpg"):
    """Decode <dir>/{right,left,top,bottom,front,back}.jpg the way Scene::LoadSkybox asks DevIL to
    (RGB bytes, lower-left origin).  Decoding is host-binding work (PIL here, DevIL in the reference);
    the texel values are whatever the decoder produces."""
    from PIL import Image
    faces = []
    for name in SKYBOX_FACE_FILES:
        img = Image.open(os.path.join(sky_dir, name + ext)).convert("RGB")
        faces.append(np.ascontiguousarray(np.asarray(img)[::-1]))
    return faces
