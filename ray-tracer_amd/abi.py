"""ctypes mirrors of the plain-old-data structures of ``include/rtamd.h``.

Used to inspect (or hand-build) an ``rtamd_scene`` from Python: ``Scene.flatten().scene`` gives the
structure the host library produced, ``Context.upload_struct`` takes one built here.
"""
import ctypes as C

import numpy as np

SPHERE, PLANE, CUBE, CYLINDER, CONE, MESH = range(6)
PAT_NONE, PAT_STRIPE, PAT_GRADIENT, PAT_RING, PAT_CHECKERS, PAT_TEST = range(6)
CSG_UNION, CSG_INTERSECTION, CSG_DIFFERENCE = range(3)
LEAF_CLOSED, LEAF_SHADOW, LEAF_SMOOTH = 1, 2, 4
VISIT_CSG, VISIT_FIRST, VISIT_CLASS_END = 1, 2, 4
MAX_LIGHTS, MAX_CHAIN, MAX_CSG_LEAVES, CSG_LIST_CAP, MAX_MESH_TRIS, MAX_REMAINING = 16, 8, 64, 24, 4096, 8
MAX_RAY_LEVELS = 80
OK, E_INVALID, E_HIP, E_NO_SCENE, E_LIMIT, E_OVERFLOW = 0, -1, -2, -3, -4, -5


class Leaf(C.Structure):
    _fields_ = [("w2o", C.c_double * 12), ("p0", C.c_double), ("p1", C.c_double), ("kind", C.c_int32),
                ("flags", C.c_uint32), ("material", C.c_int32), ("eq_class", C.c_int32), ("chain_start", C.c_int32),
                ("chain_len", C.c_int32), ("tri_start", C.c_int32), ("tri_count", C.c_int32), ("ray_start", C.c_int32), ("ray_len", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double), ("pat_a", C.c_double * 3), ("pat_b", C.c_double * 3),
                ("pat_inv", C.c_double * 12), ("pattern_kind", C.c_int32), ("pad_", C.c_int32)]


class CsgNode(C.Structure):
    _fields_ = [("left_mask", C.c_uint64), ("op", C.c_int32), ("leaf_lo", C.c_int32), ("leaf_mid", C.c_int32),
                ("leaf_hi", C.c_int32), ("pad_", C.c_int32)]


class CsgRoot(C.Structure):
    _fields_ = [("node_start", C.c_int32), ("node_count", C.c_int32), ("leaf_lo", C.c_int32), ("leaf_hi", C.c_int32)]


class Visit(C.Structure):
    _fields_ = [("unit", C.c_int32), ("eq_class", C.c_int32), ("flags", C.c_uint32), ("class_material", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("pixel_size", C.c_double), ("half_width", C.c_double),
                ("half_height", C.c_double), ("inv", C.c_double * 12)]


class Light(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("intensity", C.c_double * 3)]


class SceneStruct(C.Structure):
    _fields_ = [("camera", Camera), ("max_remaining", C.c_int32), ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
                ("n_leaves", C.c_int32), ("leaves", C.POINTER(Leaf)), ("n_chain", C.c_int32), ("chain", C.POINTER(C.c_double)),
                ("n_tris", C.c_int32), ("tri_geom", C.POINTER(C.c_double)), ("tri_normals", C.POINTER(C.c_double)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(Material)), ("n_csg_nodes", C.c_int32),
                ("csg_nodes", C.POINTER(CsgNode)), ("n_csg_roots", C.c_int32), ("csg_roots", C.POINTER(CsgRoot)),
                ("n_visits", C.c_int32), ("visits", C.POINTER(Visit)), ("n_classes", C.c_int32)]


# sizes the header states
assert C.sizeof(Leaf) == 152 and C.sizeof(Material) == 232


def doubles(ptr, n):
    """Copy n doubles from a ctypes pointer into a numpy array."""
    if n == 0:
        return np.zeros(0)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def clone_scene(src):
    """Deep copy of an rtamd_scene into Python-owned ctypes arrays: (SceneStruct, keepalive list).

    The copy can be edited freely (tests of the upload validation do)."""
    keep = []

    def arr(ptr, n, typ):
        a = (typ * max(n, 1))()
        if n:
            C.memmove(a, ptr, n * C.sizeof(typ))
        keep.append(a)
        return C.cast(a, C.POINTER(typ))

    d = SceneStruct()
    C.memmove(C.byref(d), C.byref(src), C.sizeof(SceneStruct))
    d.lights = arr(src.lights, src.n_lights, Light)
    d.leaves = arr(src.leaves, src.n_leaves, Leaf)
    d.chain = arr(src.chain, src.n_chain * 12, C.c_double)
    d.tri_geom = arr(src.tri_geom, src.n_tris * 9, C.c_double)
    d.tri_normals = arr(src.tri_normals, src.n_tris * 9, C.c_double)
    d.materials = arr(src.materials, src.n_materials, Material)
    d.csg_nodes = arr(src.csg_nodes, src.n_csg_nodes, CsgNode)
    d.csg_roots = arr(src.csg_roots, src.n_csg_roots, CsgRoot)
    d.visits = arr(src.visits, src.n_visits, Visit)
    return d, keep


def scene_bytes(s):
    """Every byte an rtamd_scene says, as one dict of byte strings (counts, camera, each array): two scenes that
    flatten alike compare equal.  Pointers are left out; padding inside the records is included, as upload reads it."""
    out = {"camera": bytes(s.camera), "counts": repr((s.max_remaining, s.n_lights, s.n_leaves, s.n_chain, s.n_tris, s.n_materials,
                                                       s.n_csg_nodes, s.n_csg_roots, s.n_visits, s.n_classes)).encode()}

    def arr(name, ptr, n, typ):
        out[name] = C.string_at(ptr, n * C.sizeof(typ)) if n else b""

    arr("lights", s.lights, s.n_lights, Light)
    arr("leaves", s.leaves, s.n_leaves, Leaf)
    arr("chain", s.chain, s.n_chain * 12, C.c_double)
    arr("tri_geom", s.tri_geom, s.n_tris * 9, C.c_double)
    arr("tri_normals", s.tri_normals, s.n_tris * 9, C.c_double)
    arr("materials", s.materials, s.n_materials, Material)
    arr("csg_nodes", s.csg_nodes, s.n_csg_nodes, CsgNode)
    arr("csg_roots", s.csg_roots, s.n_csg_roots, CsgRoot)
    arr("visits", s.visits, s.n_visits, Visit)
    return out
