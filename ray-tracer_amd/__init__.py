"""ctypes bindings of the MI355X render path.

Two in-tree shared libraries (built by ``ray-tracer_amd/Makefile``; see ``__graft_entry__.build``):

* ``lib/librtamd.so``       -- the GPU library, C ABI of ``include/rtamd.h`` (hand-written HIP, gfx950)
* ``lib/librtamd_host.so``  -- the host-side scene API (C++ mirror of the reference's World/Camera/
  shapes/loaders) that flattens a scene into an ``rtamd_scene``

Import with ``importlib.import_module("ray-tracer_amd")`` (the directory name is not an
identifier).  There is no CPU fallback: without the libraries the import raises, without a GPU
``Context()`` raises.
"""
import ctypes as C
import os

import numpy as np

# A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; when it is installed
# it has to be loaded before librtamd.so pulls in the system one, or device-code registration of the
# second arrival fails ("unknown error" at the first kernel query).  Callers use torch anyway for device
# buffers, streams and torch.distributed.
try:
    import torch  # noqa: F401
except ImportError:  # a host without PyTorch: the ROCm runtime of the system is the only one
    pass

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("RTAMD_LIB_DIR") or os.path.join(_HERE, "lib")  # override: development only

MODE_STRICT = 0
MODE_FAST = 1


class RtamdError(RuntimeError):
    pass


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {_HERE}` (hipcc, --offload-arch=gfx950)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_gpu = _load("librtamd.so")
_host = _load("librtamd_host.so")


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "primary_rays", "shadow_rays", "reflect_rays", "refract_rays",
                                          "container_passes", "shade_points", "light_samples", "csg_overflows")] + \
               [("kernel_ms", C.c_double)] + \
               [(n, C.c_uint64) for n in ("tests_sphere", "tests_plane", "tests_cube", "tests_slab2", "tests_slab3", "tests_cylinder",
                                          "tests_cone", "tests_caps", "tests_tri_first", "tests_tri_second", "tests_box",
                                          "tests_run_transform", "team_shade_points")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Rows(C.Structure):
    _fields_ = [("tile_rows", C.c_int32), ("part", C.c_int32), ("n_parts", C.c_int32), ("span", C.c_int32)]


# every symbol include/rtamd.h declares
GPU_SYMBOLS = ["rtamd_ctx_create", "rtamd_ctx_destroy", "rtamd_last_error", "rtamd_scene_upload", "rtamd_part_rows",
               "rtamd_prepare_rows", "rtamd_render_rows", "rtamd_render_rows_static", "rtamd_render", "rtamd_host_alloc", "rtamd_host_free", "rtamd_color_at", "rtamd_get_stats",
               "rtamd_reset_stats", "rtamd_count_work", "rtamd_rows_layout", "rtamd_multi_create", "rtamd_multi_destroy",
               "rtamd_multi_last_error", "rtamd_multi_upload", "rtamd_multi_render", "rtamd_multi_frame", "rtamd_multi_get_stats",
               "rtamd_multi_reset_stats"]

_gpu.rtamd_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
_gpu.rtamd_ctx_create.restype = C.c_int
_gpu.rtamd_ctx_destroy.argtypes = [C.c_void_p]
_gpu.rtamd_ctx_destroy.restype = None
_gpu.rtamd_last_error.argtypes = [C.c_void_p]
_gpu.rtamd_last_error.restype = C.c_char_p
_gpu.rtamd_scene_upload.argtypes = [C.c_void_p, C.c_void_p]
_gpu.rtamd_scene_upload.restype = C.c_int
_gpu.rtamd_part_rows.argtypes = [C.c_void_p, C.POINTER(Rows)]
_gpu.rtamd_part_rows.restype = C.c_int64
_gpu.rtamd_render_rows.argtypes = [C.c_void_p, C.POINTER(Rows), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
_gpu.rtamd_render_rows.restype = C.c_int
_gpu.rtamd_prepare_rows.argtypes = [C.c_void_p, C.POINTER(Rows), C.c_int]
_gpu.rtamd_prepare_rows.restype = C.c_int
_gpu.rtamd_render_rows_static.argtypes = _gpu.rtamd_render_rows.argtypes
_gpu.rtamd_render_rows_static.restype = C.c_int
_gpu.rtamd_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
_gpu.rtamd_render.restype = C.c_int
_gpu.rtamd_host_alloc.argtypes = [C.c_size_t]
_gpu.rtamd_host_alloc.restype = C.c_void_p
_gpu.rtamd_host_free.argtypes = [C.c_void_p]
_gpu.rtamd_host_free.restype = None
_gpu.rtamd_color_at.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
_gpu.rtamd_color_at.restype = C.c_int
_gpu.rtamd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
_gpu.rtamd_get_stats.restype = C.c_int
_gpu.rtamd_reset_stats.argtypes = [C.c_void_p]
_gpu.rtamd_reset_stats.restype = C.c_int
_gpu.rtamd_count_work.argtypes = [C.c_void_p, C.c_int]
_gpu.rtamd_count_work.restype = C.c_int



class Layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("first_row", "chunk_rows", "stride_rows", "n_chunks", "tail_rows")]


_gpu.rtamd_rows_layout.argtypes = [C.c_int, C.POINTER(Rows), C.POINTER(Layout)]
_gpu.rtamd_rows_layout.restype = C.c_int
_gpu.rtamd_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
_gpu.rtamd_multi_create.restype = C.c_int
_gpu.rtamd_multi_destroy.argtypes = [C.c_void_p]
_gpu.rtamd_multi_destroy.restype = None
_gpu.rtamd_multi_last_error.argtypes = [C.c_void_p]
_gpu.rtamd_multi_last_error.restype = C.c_char_p
_gpu.rtamd_multi_upload.argtypes = [C.c_void_p, C.c_void_p]
_gpu.rtamd_multi_upload.restype = C.c_int
_gpu.rtamd_multi_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
_gpu.rtamd_multi_render.restype = C.c_int
_gpu.rtamd_multi_frame.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
_gpu.rtamd_multi_frame.restype = C.c_int
_gpu.rtamd_multi_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_void_p, C.c_void_p]
_gpu.rtamd_multi_get_stats.restype = C.c_int
_gpu.rtamd_multi_reset_stats.argtypes = [C.c_void_p]
_gpu.rtamd_multi_reset_stats.restype = C.c_int

_host.rth_last_error.restype = C.c_char_p
for _n in ("rth_load_yaml", "rth_load_yaml_str", "rth_load_sg"):
    getattr(_host, _n).argtypes = [C.c_char_p]
    getattr(_host, _n).restype = C.c_void_p
_host.rth_builtin_scene.argtypes = [C.c_char_p, C.c_char_p]
_host.rth_builtin_scene.restype = C.c_void_p
_host.rth_save_sg.argtypes = [C.c_void_p, C.c_char_p]
_host.rth_free_scene.argtypes = [C.c_void_p]
_host.rth_free_scene.restype = None
_host.rth_scene_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
_host.rth_set_resolution.argtypes = [C.c_void_p, C.c_int, C.c_int]
_host.rth_flatten.argtypes = [C.c_void_p, C.c_int]
_host.rth_flatten.restype = C.c_void_p
_host.rth_flat_view.argtypes = [C.c_void_p]
_host.rth_flat_view.restype = C.c_void_p
_host.rth_free_flat.argtypes = [C.c_void_p]
_host.rth_free_flat.restype = None
_host.rth_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
_host.rth_canvas_save.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
_host.rth_canvas_quantize.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
_host.rth_canvas_ppm.argtypes = [C.c_void_p, C.c_int, C.c_int]
_host.rth_canvas_ppm.restype = C.c_void_p
_host.rth_free_str.argtypes = [C.c_void_p]
_host.rth_free_str.restype = None
_host.rth_obj_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 7
_host.rth_obj_triangle.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
_host.rth_save_png8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]


class Material(C.Structure):
    """``rth_material`` of include/rtamd_host.h (src/material.rs:8-18 plus the optional Pattern)."""
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double), ("refractive_index", C.c_double),
                ("pattern", C.c_int32), ("pad_", C.c_int32), ("pat_a", C.c_double * 3), ("pat_b", C.c_double * 3),
                ("pat_transform", C.c_double * 16)]


# every symbol include/rtamd_host.h declares
HOST_SYMBOLS = ["rth_last_error", "rth_material_default", "rth_world_new", "rth_world_free", "rth_add_sphere", "rth_add_plane", "rth_add_cube",
                "rth_add_cylinder", "rth_add_cone", "rth_add_triangle", "rth_add_smooth_triangle", "rth_group_begin", "rth_group_end",
                "rth_csg_begin", "rth_csg_end", "rth_add_light", "rth_set_camera", "rth_world_scene", "rth_load_yaml", "rth_load_yaml_str",
                "rth_load_sg", "rth_builtin_scene", "rth_save_sg", "rth_free_scene", "rth_scene_info", "rth_set_resolution", "rth_flatten",
                "rth_flat_view", "rth_free_flat", "rth_render", "rth_canvas_save", "rth_canvas_quantize", "rth_save_png8", "rth_canvas_ppm",
                "rth_free_str", "rth_obj_info", "rth_obj_triangle"]

_host.rth_material_default.argtypes = [C.POINTER(Material)]
_host.rth_material_default.restype = None
_host.rth_world_new.restype = C.c_void_p
_host.rth_world_free.argtypes = [C.c_void_p]
_host.rth_world_free.restype = None
_SHAPE = [C.c_void_p, C.c_void_p, C.POINTER(Material), C.c_int]  # world, transform[16], material, shadow
for _n in ("rth_add_sphere", "rth_add_plane", "rth_add_cube", "rth_group_begin"):
    getattr(_host, _n).argtypes = _SHAPE
_host.rth_add_cylinder.argtypes = _SHAPE + [C.c_double, C.c_double, C.c_int]
_host.rth_add_cone.argtypes = _SHAPE + [C.c_double, C.c_double, C.c_int]
_host.rth_add_triangle.argtypes = _SHAPE + [C.c_void_p]
_host.rth_add_smooth_triangle.argtypes = _SHAPE + [C.c_void_p, C.c_void_p]
_host.rth_csg_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Material), C.c_int]
_host.rth_group_end.argtypes = [C.c_void_p]
_host.rth_csg_end.argtypes = [C.c_void_p]
_host.rth_add_light.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
_host.rth_set_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
_host.rth_world_scene.argtypes = [C.c_void_p]
_host.rth_world_scene.restype = C.c_void_p


def _host_err():
    return RtamdError(_host.rth_last_error().decode())


class Flat:
    """A flattened scene (owns the arrays an ``rtamd_scene`` points into)."""

    def __init__(self, handle):
        self._h = handle

    @property
    def view(self):
        return _host.rth_flat_view(self._h)

    @property
    def scene(self):
        """The ``rtamd_scene`` as a ctypes structure (pointers stay valid while this object lives)."""
        return abi.SceneStruct.from_address(self.view)

    def __del__(self, _free=_host.rth_free_flat):  # bound now: module globals are gone at interpreter exit
        if getattr(self, "_h", None):
            _free(self._h)
            self._h = None


class Scene:
    """World + Camera of the host API."""

    def __init__(self, handle):
        if not handle:
            raise _host_err()
        self._h = handle

    @staticmethod
    def from_yaml(path):
        return Scene(_host.rth_load_yaml(os.fsencode(path)))

    @staticmethod
    def from_yaml_str(text):
        return Scene(_host.rth_load_yaml_str(text.encode()))

    @staticmethod
    def from_sg(path):
        return Scene(_host.rth_load_sg(os.fsencode(path)))

    @staticmethod
    def builtin(name, arg=None):
        return Scene(_host.rth_builtin_scene(name.encode(), os.fsencode(arg) if arg else None))

    def info(self):
        v = [C.c_int() for _ in range(4)]
        if _host.rth_scene_info(self._h, *[C.byref(x) for x in v]):
            raise _host_err()
        return dict(hsize=v[0].value, vsize=v[1].value, n_objects=v[2].value, n_lights=v[3].value)

    def set_resolution(self, hsize, vsize):
        if _host.rth_set_resolution(self._h, hsize, vsize):
            raise _host_err()
        return self

    def save_sg(self, path):
        if _host.rth_save_sg(self._h, os.fsencode(path)):
            raise _host_err()

    def flatten(self, max_remaining=4):
        h = _host.rth_flatten(self._h, max_remaining)
        if not h:
            raise _host_err()
        return Flat(h)

    def render(self, mode=MODE_FAST, want_u8=False, want_f64=True):
        """Camera::render on the GPU -> (vsize, hsize, 3) float64 [, uint8], Stats; with ``want_f64=False`` the 8-bit
        channels alone (3 bytes per pixel over the PCIe link instead of 27) -> uint8, Stats."""
        i = self.info()
        rgb = np.empty((i["vsize"], i["hsize"], 3), dtype=np.float64) if want_f64 else None
        u8 = np.empty((i["vsize"], i["hsize"], 3), dtype=np.uint8) if want_u8 or not want_f64 else None
        st = Stats()
        if _host.rth_render(self._h, mode, rgb.ctypes.data if want_f64 else None, u8.ctypes.data if u8 is not None else None, C.byref(st)):
            raise _host_err()
        if not want_f64:
            return u8, st
        return (rgb, u8, st) if want_u8 else (rgb, st)

    def __del__(self, _free=_host.rth_free_scene):
        if getattr(self, "_h", None) and getattr(self, "_borrowed_from", None) is None:  # (a builder's scene is the builder's to free)
            _free(self._h)
        self._h = None


CSG_OPS = {"union": 0, "intersection": 1, "difference": 2}
PATTERNS = {"none": 0, "stripe": 1, "gradient": 2, "ring": 3, "checkers": 4, "test": 5}


def default_material():
    m = Material()
    _host.rth_material_default(C.byref(m))
    return m


def _m16(m):
    if m is None:
        return None
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(16)
    return a.ctypes.data_as(C.c_void_p), a  # (the array is returned so that it outlives the call)


class WorldBuilder:
    """The scene builder of include/rtamd_host.h: one call per node of the reference's object tree (World.objects,
    Group.children, Csg.left / right), then ``scene()`` -> the ``Scene`` that flattens and renders like a loaded one."""

    def __init__(self):
        self._h = _host.rth_world_new()
        if not self._h:
            raise _host_err()

    def _chk(self, rc):
        if rc:
            raise _host_err()
        return self

    def _args(self, transform, material, shadow):
        t = _m16(transform)
        return (t[0] if t else None), (C.byref(material) if material is not None else None), (1 if shadow else 0), t

    def add(self, kind, transform=None, material=None, shadow=True, mn=None, mx=None, closed=False, points=None, normals=None):
        t, m, sh, keep = self._args(transform, material, shadow)
        if kind in ("sphere", "plane", "cube"):
            return self._chk(getattr(_host, "rth_add_" + kind)(self._h, t, m, sh))
        if kind in ("cylinder", "cone"):
            lo = -np.inf if mn is None else mn
            hi = np.inf if mx is None else mx
            return self._chk(getattr(_host, "rth_add_" + kind)(self._h, t, m, sh, lo, hi, 1 if closed else 0))
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(12)
        if kind == "triangle":
            return self._chk(_host.rth_add_triangle(self._h, t, m, sh, p.ctypes.data))
        if kind == "smooth_triangle":
            n = np.ascontiguousarray(normals, dtype=np.float64).reshape(12)
            return self._chk(_host.rth_add_smooth_triangle(self._h, t, m, sh, p.ctypes.data, n.ctypes.data))
        raise RtamdError(f"unknown shape kind {kind}")

    def group_begin(self, transform=None, material=None, shadow=True):
        t, m, sh, keep = self._args(transform, material, shadow)
        return self._chk(_host.rth_group_begin(self._h, t, m, sh))

    def group_end(self):
        return self._chk(_host.rth_group_end(self._h))

    def csg_begin(self, op, transform=None, material=None, shadow=True):
        t, m, sh, keep = self._args(transform, material, shadow)
        return self._chk(_host.rth_csg_begin(self._h, CSG_OPS[op] if isinstance(op, str) else op, t, m, sh))

    def csg_end(self):
        return self._chk(_host.rth_csg_end(self._h))

    def light(self, position, intensity=(1, 1, 1)):
        p, i = np.asarray(position, dtype=np.float64), np.asarray(intensity, dtype=np.float64)
        return self._chk(_host.rth_add_light(self._h, p.ctypes.data, i.ctypes.data))

    def camera(self, hsize, vsize, fov, transform=None):
        t = _m16(transform)
        return self._chk(_host.rth_set_camera(self._h, hsize, vsize, fov, t[0] if t else None))

    def scene(self):
        """The built World + Camera as a ``Scene`` (borrowed from this builder, which it keeps alive)."""
        h = _host.rth_world_scene(self._h)
        if not h:
            raise _host_err()
        s = Scene(h)
        s._borrowed_from = self  # rth_free_scene must not see it
        return s

    def __del__(self, _free=_host.rth_world_free):
        if getattr(self, "_h", None):
            _free(self._h)
            self._h = None


class Context:
    """One GPU (``rtamd_ctx``)."""

    def __init__(self, device=0):
        p = C.c_void_p()
        rc = _gpu.rtamd_ctx_create(device, C.byref(p))
        if rc:
            raise RtamdError(f"rtamd_ctx_create({device}) = {rc}: {_gpu.rtamd_last_error(None).decode()}")
        self._h = p
        self._flat = None
        self._size = None

    def _chk(self, rc):
        if rc:
            raise RtamdError(f"rtamd error {rc}: {_gpu.rtamd_last_error(self._h).decode()}")

    def upload(self, flat):
        self._chk(_gpu.rtamd_scene_upload(self._h, flat.view))
        self._flat = flat
        self._size = (flat.scene.camera.hsize, flat.scene.camera.vsize)
        return self

    def upload_struct(self, scene_struct):
        """Upload an ``abi.SceneStruct`` built or edited in Python; returns the status code (no exception)."""
        rc = _gpu.rtamd_scene_upload(self._h, C.byref(scene_struct))
        if rc == 0:
            self._flat = None
            self._size = (scene_struct.camera.hsize, scene_struct.camera.vsize)
        return rc

    def last_error(self):
        return _gpu.rtamd_last_error(self._h).decode()

    def part_rows(self, tile_rows, part, n_parts, span=1):
        r = Rows(tile_rows, part, n_parts, span)
        return _gpu.rtamd_part_rows(self._h, C.byref(r))

    def prepare_rows(self, tile_rows, part, n_parts, mode, span=1):
        """rtamd_prepare_rows: the host-side work of the first render of this partition, ahead of time."""
        r = Rows(tile_rows, part, n_parts, span)
        self._chk(_gpu.rtamd_prepare_rows(self._h, C.byref(r), mode))

    def render_rows(self, tile_rows, part, n_parts, mode, d_rgb=0, d_rgb8=0, stream=0, span=1, static=False):
        """Asynchronous; d_rgb / d_rgb8 are device addresses (e.g. ``tensor.data_ptr()``).  Renders the tiles of
        parts ``part .. part + span - 1`` of ``n_parts``.  ``static``: through rtamd_render_rows_static, the entry
        point that allocates nothing."""
        r = Rows(tile_rows, part, n_parts, span)
        fn = _gpu.rtamd_render_rows_static if static else _gpu.rtamd_render_rows
        self._chk(fn(self._h, C.byref(r), mode, C.c_void_p(d_rgb or None), C.c_void_p(d_rgb8 or None), C.c_void_p(stream or None)))

    def render(self, hsize, vsize, mode=MODE_FAST, want_u8=False, pinned=False, want_f64=True, into=None):
        """Whole frame into host arrays; (hsize, vsize) must be the resolution of the uploaded scene.
        ``pinned``: the arrays live in page-locked memory from rtamd_host_alloc (freed with the arrays);
        ``want_f64=False``: the 8-bit channels alone; ``into``: (f64 array or None, u8 array or None) to render into."""
        if getattr(self, "_size", None) is not None and self._size != (hsize, vsize):
            raise RtamdError(f"the uploaded scene is {self._size[0]}x{self._size[1]}, not {hsize}x{vsize}")
        new = host_empty if pinned else np.empty
        if into is not None:
            rgb, u8 = into
        else:
            rgb = new((vsize, hsize, 3), dtype=np.float64) if want_f64 else None
            u8 = new((vsize, hsize, 3), dtype=np.uint8) if want_u8 or not want_f64 else None
        self._chk(_gpu.rtamd_render(self._h, mode, rgb.ctypes.data if rgb is not None else None, u8.ctypes.data if u8 is not None else None))
        if rgb is None:
            return u8
        return (rgb, u8) if (want_u8 or (into is not None and u8 is not None)) else rgb

    def color_at(self, rays, remaining, mode=MODE_STRICT):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.empty((rays.shape[0], 3), dtype=np.float64)
        self._chk(_gpu.rtamd_color_at(self._h, mode, rays.ctypes.data, rays.shape[0], remaining, out.ctypes.data))
        return out

    def stats(self):
        st = Stats()
        self._chk(_gpu.rtamd_get_stats(self._h, C.byref(st)))
        return st

    def reset_stats(self):
        self._chk(_gpu.rtamd_reset_stats(self._h))

    def count_work(self, on=True):
        """Renders that follow also count the tests their traversals execute (Stats.tests_*)."""
        self._chk(_gpu.rtamd_count_work(self._h, 1 if on else 0))

    def close(self, _destroy=_gpu.rtamd_ctx_destroy):
        if getattr(self, "_h", None):
            _destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def rows_layout(vsize, tile_rows, part, n_parts, span=1):
    """rtamd_rows_layout: where the rows of a rows-spec lie in the frame (pure arithmetic, no GPU)."""
    r, out = Rows(tile_rows, part, n_parts, span), Layout()
    if _gpu.rtamd_rows_layout(vsize, C.byref(r), C.byref(out)):
        raise RtamdError("bad rows specification")
    return {n: getattr(out, n) for n, _ in out._fields_}


WANT_F64, WANT_U8 = 1, 2


class MultiContext:
    """Several GPUs, one frame (``rtamd_multi``): one context per entry of ``devices``, frame assembled on the first."""

    def __init__(self, devices):
        arr = (C.c_int * len(devices))(*devices)
        p = C.c_void_p()
        rc = _gpu.rtamd_multi_create(arr, len(devices), C.byref(p))
        if rc:
            raise RtamdError(f"rtamd_multi_create({list(devices)}) = {rc}: {_gpu.rtamd_multi_last_error(None).decode()}")
        self._h, self.n = p, len(devices)
        self._flat = None

    def _chk(self, rc):
        if rc:
            raise RtamdError(f"rtamd error {rc}: {_gpu.rtamd_multi_last_error(self._h).decode()}")

    def upload(self, flat):
        self._chk(_gpu.rtamd_multi_upload(self._h, flat.view))
        self._flat = flat
        self._size = (flat.scene.camera.hsize, flat.scene.camera.vsize)
        return self

    def render(self, mode=MODE_FAST, tile_rows=8, first_parts=1, other_parts=1, want_u8=False):
        """Whole frame assembled on the first GPU and copied into host arrays."""
        w, h = self._size
        rgb = np.empty((h, w, 3), dtype=np.float64)
        u8 = np.empty((h, w, 3), dtype=np.uint8) if want_u8 else None
        self._chk(_gpu.rtamd_multi_render(self._h, mode, tile_rows, first_parts, other_parts, rgb.ctypes.data,
                                          u8.ctypes.data if want_u8 else None, 0))
        return (rgb, u8) if want_u8 else rgb

    def render_on_device(self, mode=MODE_FAST, tile_rows=8, first_parts=1, other_parts=1, want=WANT_F64):
        """The frame stays on the first GPU; returns its device addresses (f64, u8; 0 for a kind not asked for)."""
        self._chk(_gpu.rtamd_multi_render(self._h, mode, tile_rows, first_parts, other_parts, None, None, want))
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(_gpu.rtamd_multi_frame(self._h, C.byref(a), C.byref(b)))
        return (a.value or 0) if want & WANT_F64 else 0, (b.value or 0) if want & WANT_U8 else 0

    def stats(self):
        """(Stats summed over the ranks, render ms per rank, ms per rank of putting its rows in place)."""
        st = Stats()
        r, p = np.zeros(self.n), np.zeros(self.n)
        self._chk(_gpu.rtamd_multi_get_stats(self._h, C.byref(st), r.ctypes.data, p.ctypes.data))
        return st, r, p

    def reset_stats(self):
        self._chk(_gpu.rtamd_multi_reset_stats(self._h))

    def close(self, _destroy=_gpu.rtamd_multi_destroy):
        if getattr(self, "_h", None):
            _destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class _Pinned:
    """Frees one rtamd_host_alloc block when the last array made from it is gone."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self, _free=_gpu.rtamd_host_free):
        if self.ptr:
            _free(self.ptr)
            self.ptr = None


def host_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (rtamd_host_alloc)."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(1, count * dtype.itemsize)
    ptr = _gpu.rtamd_host_alloc(nbytes)
    if not ptr:
        raise RtamdError(f"rtamd_host_alloc({nbytes}) failed")
    buf = (C.c_char * nbytes).from_address(ptr)
    buf._owner = _Pinned(ptr)  # the buffer is the array's base: the block lives as long as any view of it
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def quantize(rgb):
    """`From<Color> for (usize, usize, usize)` of the reference (src/tuple.rs:226-237) on an array."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.empty(rgb.shape, dtype=np.uint8)
    _host.rth_canvas_quantize(rgb.ctypes.data, rgb.size // 3, out.ctypes.data)
    return out


def canvas_to_ppm(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    p = _host.rth_canvas_ppm(rgb.ctypes.data, rgb.shape[1], rgb.shape[0])
    if not p:
        raise _host_err()
    s = C.string_at(p).decode()
    _host.rth_free_str(p)
    return s


def canvas_save(rgb, path, ppm=False):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    if _host.rth_canvas_save(rgb.ctypes.data, rgb.shape[1], rgb.shape[0], os.fsencode(path), 1 if ppm else 0):
        raise _host_err()


def obj_info(text):
    v = [C.c_int() for _ in range(7)]
    if _host.rth_obj_info(text.encode(), *[C.byref(x) for x in v]):
        raise _host_err()
    k = ("n_vertices", "n_normals", "n_default_tris", "n_groups", "n_group_tris", "lines_ignored", "n_smooth")
    return dict(zip(k, [x.value for x in v]))


def obj_triangle(text, k):
    out = np.zeros(21)
    smooth = _host.rth_obj_triangle(text.encode(), k, out.ctypes.data)
    if smooth < 0:
        raise _host_err()
    return dict(smooth=bool(smooth), p=out[:9].reshape(3, 3), n=out[9:].reshape(3, 4))
