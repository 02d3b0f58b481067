"""The scene builder of include/rtamd_host.h (rth_world_* / rth_add_* / rth_group_* / rth_csg_*): a host that holds the
reference's object tree -- World.objects, Group.children, Csg.left / right (src/scene.rs:6-9, src/shapes/group.rs:8-11,
src/shapes/csg.rs:5-15) -- makes one call per node and gets the flat `rtamd_scene` without writing a flattener.  Every
golden scene is rebuilt through those calls here (ctypes: the calls a Rust or C host would make) and must flatten to
the bytes the loader's scene flattens to."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import sgtext
from conftest import ROOT, sg_path

SCENES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "scenes", "*.sg")))


@pytest.mark.parametrize("name", SCENES)
def test_scene_built_call_by_call_flattens_to_the_bytes_of_the_loaded_one(rt, name):
    abi = rt.abi
    loaded = rt.Scene.from_sg(sg_path(name)).flatten()  # (the Flat objects own the arrays: they stay alive while their bytes are read)
    want = abi.scene_bytes(loaded.scene)
    builder = sgtext.replay(sg_path(name), rt)
    built = builder.scene().flatten()
    got = abi.scene_bytes(built.scene)
    assert got.keys() == want.keys()
    for k in want:
        assert got[k] == want[k], f"{name}: {k} differs"
    assert want["counts"] != b"" and len(want["leaves"]) > 0


def test_builder_refuses_what_the_tree_cannot_be(rt):
    b = rt.WorldBuilder()
    with pytest.raises(rt.RtamdError, match="open"):
        b.group_end()
    b.csg_begin("difference")
    b.add("cube")
    with pytest.raises(rt.RtamdError, match="two children"):
        b.csg_end()
    with pytest.raises(rt.RtamdError, match="still open"):
        b.scene()
    b.add("sphere")
    with pytest.raises(rt.RtamdError, match="two children"):
        b.add("sphere")
    b.csg_end()
    with pytest.raises(rt.RtamdError, match="camera"):
        b.scene().flatten()
    b.camera(10, 10, 1.0)
    keep = b.scene().flatten()
    flat = keep.scene
    assert flat.n_leaves == 2 and flat.n_csg_roots == 1 and flat.n_csg_nodes == 1 and flat.csg_nodes[0].op == rt.abi.CSG_DIFFERENCE
    # a singular transform is reported by flatten (the reference panics in inverse().unwrap(), src/shape.rs:40)
    c = rt.WorldBuilder()
    c.camera(4, 4, 1.0)
    c.add("sphere", np.diag([1.0, 0.0, 1.0, 1.0]))
    with pytest.raises(rt.RtamdError, match="singular"):
        c.scene().flatten()
    m = rt.default_material()
    m.pattern = 9
    with pytest.raises(rt.RtamdError, match="pattern"):
        c.add("cube", None, m)


def test_group_material_is_inherited_as_refresh_parents_does(rt):
    """src/shapes/group.rs:76-89: a child that still has Material::new() takes the group's material; one that has its own keeps it."""
    red = rt.default_material()
    red.color[:] = [1, 0, 0]
    blue = rt.default_material()
    blue.color[:] = [0, 0, 1]
    b = rt.WorldBuilder()
    b.camera(8, 8, 1.0)
    b.group_begin(sgtext.translation(1, 2, 3), red)
    b.add("sphere")
    b.add("cube", None, blue)
    b.group_end()
    keep = b.scene().flatten()
    s = keep.scene
    assert s.n_leaves == 2 and s.n_materials == 2
    assert list(s.materials[s.leaves[0].material].color) == [1, 0, 0] and list(s.materials[s.leaves[1].material].color) == [0, 0, 1]
    assert s.leaves[0].chain_len == 2 and s.leaves[0].w2o[3] == -1.0 and s.leaves[0].w2o[7] == -2.0


def test_ray_levels_follow_the_objects_a_ray_passes_not_the_parents_a_normal_sees(rt):
    """A ray is transformed at every group and CSG object above a leaf (src/shape.rs:39-42 from src/shapes/group.rs:37-45 and
    src/shapes/csg.rs:90-97); normals and patterns go through parent_transforms, which Group::refresh_parents (group.rs:76-89)
    does not carry past a CSG object.  The flat scene keeps both lists."""
    b = rt.WorldBuilder()
    b.camera(8, 8, 1.0)
    b.group_begin(sgtext.translation(1, 0, 0))
    b.csg_begin("difference", sgtext.scaling(2, 2, 2))
    b.add("sphere", sgtext.translation(0, 3, 0))
    b.group_begin(sgtext.translation(0, 0, 5))
    b.add("cube")
    b.group_end()
    b.csg_end()
    b.add("cone", sgtext.scaling(1, 4, 1))
    b.group_end()
    s = b.scene().flatten().scene
    sphere, cube, cone = s.leaves[0], s.leaves[1], s.leaves[2]
    lv = lambda L, k: np.array(s.chain[(L.ray_start + k) * 12:(L.ray_start + k + 1) * 12]).reshape(3, 4)
    assert (sphere.chain_len, sphere.ray_len) == (1, 3) and (cube.chain_len, cube.ray_len) == (1, 4) and (cone.chain_len, cone.ray_len) == (2, 2)
    assert (cone.ray_start, cone.ray_len) == (cone.chain_start, cone.chain_len)
    assert lv(sphere, 0)[0, 3] == -1.0 and lv(sphere, 1)[0, 0] == 0.5 and lv(sphere, 2)[1, 3] == -3.0
    assert lv(cube, 0)[0, 3] == -1.0 and lv(cube, 1)[1, 1] == 0.5 and lv(cube, 2)[2, 3] == -5.0 and np.array_equal(lv(cube, 3), np.eye(4)[:3])
    assert lv(cone, 0)[0, 3] == -1.0 and lv(cone, 1)[1, 1] == 0.25


def test_every_declaration_of_the_host_header_is_exported_and_compiles_as_c99(rt, tmp_path):
    hdr = open(os.path.join(ROOT, "include", "rtamd_host.h")).read()
    declared = set(re.findall(r"\b(rth_\w+)\s*\(", hdr))
    assert declared == set(rt.HOST_SYMBOLS), declared ^ set(rt.HOST_SYMBOLS)
    for sym in declared:
        assert hasattr(rt._host, sym), sym
    src = tmp_path / "t.c"
    src.write_text('#include "rtamd_host.h"\nint main(void) { rth_material m; rth_material_default(&m); return sizeof(rth_material) == 264 ? 0 : 1; }\n')
    import subprocess
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", rt.LIB_DIR, "-lrtamd_host", "-lrtamd", f"-Wl,-rpath,{rt.LIB_DIR}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    assert C.sizeof(rt.Material) == 264
