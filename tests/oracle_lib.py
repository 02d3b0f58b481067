"""ctypes access to the CPU oracle (oracle/build/liboracle.so) -- test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "oracle", "build", "liboracle.so")


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "shade_points", "leaf_tests", "inversions")]


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{_PATH} missing: run `make -C oracle`")
        _lib = C.CDLL(_PATH)
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_load_sg.argtypes = [C.c_char_p]
        _lib.orc_load_sg.restype = C.c_void_p
        _lib.orc_free_scene.argtypes = [C.c_void_p]
        _lib.orc_free_scene.restype = None
        _lib.orc_scene_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.orc_render_pixels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_int, C.POINTER(OrcStats)]
        _lib.orc_render_pixels.restype = C.c_int
        _lib.orc_color_at.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        _lib.orc_color_at.restype = C.c_int
        _lib.orc_first_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_first_hit.restype = C.c_int
    return _lib


class OracleScene:
    def __init__(self, sg_path):
        self._h = lib().orc_load_sg(os.fsencode(sg_path))
        if not self._h:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())

    def size(self):
        h, v = C.c_int(), C.c_int()
        lib().orc_scene_size(self._h, C.byref(h), C.byref(v))
        return h.value, v.value

    def render_pixels(self, pix, hsize=0, vsize=0, faithful=False, threads=1):
        """pix: flat pixel indices (x + y*hsize) -> (n,3) float64, (n,3) uint8, stats dict."""
        pix = np.ascontiguousarray(pix, dtype=np.int64)
        rgb = np.empty((pix.size, 3), dtype=np.float64)
        u8 = np.empty((pix.size, 3), dtype=np.uint8)
        st = OrcStats()
        rc = lib().orc_render_pixels(self._h, 1 if faithful else 0, hsize, vsize, pix.ctypes.data, pix.size,
                                     rgb.ctypes.data, u8.ctypes.data, threads, C.byref(st))
        if rc:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        return rgb, u8, {n: getattr(st, n) for n, _ in st._fields_}

    def color_at(self, rays, remaining, faithful=False):
        """rays: (n,6) origin+direction -> (n,3) colours of World::color_at(ray, remaining)."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.empty((rays.shape[0], 3), dtype=np.float64)
        if lib().orc_color_at(self._h, 1 if faithful else 0, rays.ctypes.data, rays.shape[0], remaining, out.ctypes.data):
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        return out

    def first_hit(self, ray):
        """Diagnostic: prepare_computations of the ray's hit as a dict, or None for a miss."""
        ray = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        out = np.zeros(16)
        rc = lib().orc_first_hit(self._h, ray.ctypes.data, out.ctypes.data)
        if rc < 0:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        if rc == 0:
            return None
        return dict(t=out[0], point=out[1:4].copy(), over_point=out[4:7].copy(), normalv=out[7:10].copy(), reflectv=out[10:13].copy(),
                    inside=bool(out[13]), kind=int(out[14]), shadowed=int(out[15]))

    def render(self, hsize=0, vsize=0, faithful=False, threads=1):
        if hsize <= 0:
            hsize, vsize = self.size()
        rgb, u8, st = self.render_pixels(np.arange(hsize * vsize), hsize, vsize, faithful, threads)
        return rgb.reshape(vsize, hsize, 3), u8.reshape(vsize, hsize, 3), st

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free_scene(self._h)
            self._h = None
