"""Host-side logic that runs without a GPU: loaders, canvas output, scene flattening, the C ABI's
symbols, and the absence of any CPU fallback."""
import ctypes as C
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ALL_SCENES, ROOT, sg_path

# ------------------------------------------------------------------ OBJ (inputs and expectations of src/obj.rs:143-314)
QUAD = "v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n"


def test_obj_ignores_unrecognised_lines(rt):
    text = ("There was a young lady name Bright\nwho traveled much faster than ight.\nShe set out one day\n"
            "in a relative way,\nand came back the previous night.\n")
    assert rt.obj_info(text)["lines_ignored"] == 5


def test_obj_vertices_and_triangle_faces(rt):
    text = "v -1 1 0\nv -1.0000 0.5000 0.0000\nv 1 0 0\nv 1 1 0\n\nf 1 2 3\nf 1 3 4\n"
    info = rt.obj_info(text)
    assert info["n_vertices"] == 4 and info["n_default_tris"] == 2
    t1, t2 = rt.obj_triangle(text, 0), rt.obj_triangle(text, 1)
    assert np.array_equal(t1["p"], [[-1, 1, 0], [-1, 0.5, 0], [1, 0, 0]])
    assert np.array_equal(t2["p"], [[-1, 1, 0], [1, 0, 0], [1, 1, 0]])
    assert not t1["smooth"]


def test_obj_fan_triangulation(rt):
    text = QUAD + "v 0 2 0\n\nf 1 2 3 4 5\n"
    assert rt.obj_info(text)["n_default_tris"] == 3
    v = np.array([[-1, 1, 0], [-1, 0, 0], [1, 0, 0], [1, 1, 0], [0, 2, 0]], dtype=float)
    for k, (a, b, c) in enumerate([(0, 1, 2), (0, 2, 3), (0, 3, 4)]):
        assert np.array_equal(rt.obj_triangle(text, k)["p"], v[[a, b, c]])


def test_obj_named_groups(rt):
    info = rt.obj_info(QUAD + "\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4\n")
    assert info["n_groups"] == 2 and info["n_group_tris"] == 2 and info["n_default_tris"] == 0


def test_obj_normals_and_smooth_faces(rt):
    assert rt.obj_info("vn 0 0 1\nvn 0.707 0 -0.707\nvn 1 2 3\n")["n_normals"] == 3
    # two spaces after `v`; texture indices are ignored
    text = "v  0 1 0\nv -1 0 0\nv 1 0 0\n\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\n\nf 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2\n"
    info = rt.obj_info(text)
    assert info["n_default_tris"] == 2 and info["n_smooth"] == 2
    t1, t2 = rt.obj_triangle(text, 0), rt.obj_triangle(text, 1)
    assert t1["smooth"] and np.array_equal(t1["p"], [[0, 1, 0], [-1, 0, 0], [1, 0, 0]])
    assert np.array_equal(t1["n"][:, :3], [[0, 1, 0], [-1, 0, 0], [1, 0, 0]])
    assert np.array_equal(t1["n"][:, 3], [1, 1, 1])  # normals are stored as points (src/obj.rs:111-115)
    assert np.array_equal(t1["p"], t2["p"]) and np.array_equal(t1["n"], t2["n"])


# ------------------------------------------------------------------ canvas (src/tuple.rs:226-237, 343-371, tests :579-620)
def test_ppm_header_pixels_wrapping_and_newline(rt):
    c = np.zeros((3, 5, 3))
    ppm = rt.canvas_to_ppm(c)
    assert ppm.split("\n")[:3] == ["P3", "5 3", "255"] and ppm.endswith("\n")
    c[0, 0] = (1.5, 0, 0)
    c[1, 2] = (0, 0.5, 0)
    c[2, 4] = (-0.5, 0, 1)
    assert rt.canvas_to_ppm(c).split("\n")[3:6] == ["255 0 0 0 0 0 0 0 0 0 0 0 0 0 0", "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0",
                                                    "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"]
    c = np.empty((2, 10, 3))
    c[:] = (1.0, 0.8, 0.6)
    lines = rt.canvas_to_ppm(c).split("\n")[3:7]
    assert lines == ["255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204",
                     "153 255 204 153 255 204 153 255 204 153 255 204 153"] * 2
    assert max(len(x) for x in lines) <= 70


def test_quantisation_rounds_half_away_and_saturates(rt):
    x = np.array([[0.5 / 255, 1.5 / 255, 2.5 / 255], [-3.0, 7.0, float("nan")], [float("inf"), -float("inf"), 0.999],
                  [254.5 / 255, 0.4999 / 255, 1.0]])
    q = rt.quantize(x)
    assert q.tolist() == [[1, 2, 3], [0, 255, 0], [255, 0, 255], [255, 0, 255]]


def test_png_writer_round_trips(rt, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    c = rng.random((7, 13, 3))
    p = str(tmp_path / "c.png")
    rt.canvas_save(c, p)
    assert np.array_equal(np.array(Image.open(p).convert("RGB")), rt.quantize(c))


# ------------------------------------------------------------------ YAML subset (src/bin/yaml_renderer.rs:21-357)
YAML = """
- add: camera
  width: 20
  height: 10
  field-of-view: 0.5
  from: [0, 1, -6]
  to: [0, 1, 0]
  up: [0, 1, 0]
- add: light
  at: [-5, 5, -5]
  intensity: [1, 1, 0.5]
- define: base-material
  value:
    color: [1, 0, 0]
    diffuse: 0.5
    reflective: 0.25
- define: shiny-material     # a comment
  extend: base-material
  value:
    shininess: 300
- define: up-transform
  value:
    - [translate, 0, 1, 0]
- define: big-transform
  value:
    - up-transform
    - [scale, 2, 2, 2]
- add: sphere
  material: shiny-material
  transform:
    - big-transform
- add: plane
  material:
    pattern:
      type: checkers
      colors:
        - [0, 0, 0]
        - [1, 1, 1]
      transform:
        - [scale, 0.5, 0.5, 0.5]
    ambient: 1
  shadow: false
- add: cylinder
  min: 0
  max: 2
  closed: true
  transform:
    - [rotate-x, 1.5]
    - [translate, 3, 0, 0]
- add: group
  transform:
    - [translate, 0, 0, 2]
  children:
    - add: cube
    - add: cone
      min: -1
      max: 0
      closed: false
"""


def test_yaml_scene_flattens_as_the_reference_would_build_it(rt):
    from importlib import import_module
    abi = import_module("ray-tracer_amd.abi")
    sc = rt.Scene.from_yaml_str(YAML)
    assert sc.info() == dict(hsize=20, vsize=10, n_objects=4, n_lights=1)
    flat = sc.flatten()
    s = flat.scene
    assert s.n_leaves == 5 and s.n_lights == 1 and s.max_remaining == 4
    assert [s.leaves[i].kind for i in range(5)] == [abi.SPHERE, abi.PLANE, abi.CYLINDER, abi.CUBE, abi.CONE]
    assert list(s.lights[0].intensity) == [1, 1, 0.5]
    # `extend:` copies the base, `value:` overrides; ints are accepted as floats
    m = s.materials[s.leaves[0].material]
    assert (list(m.color), m.diffuse, m.reflective, m.shininess, m.ambient) == ([1, 0, 0], 0.5, 0.25, 300.0, 0.1)
    # a named transform inside a list replaces what was accumulated, later ops pre-multiply:
    # scale(2) * translate(0,1,0); the leaf stores rows 0..2 of the inverse
    w2o = np.array(s.leaves[0].w2o).reshape(3, 4)
    assert np.allclose(w2o, [[0.5, 0, 0, 0], [0, 0.5, 0, -1.0], [0, 0, 0.5, 0]])
    pm = s.materials[s.leaves[1].material]
    assert pm.pattern_kind == abi.PAT_CHECKERS and pm.ambient == 1.0 and list(pm.pat_b) == [1, 1, 1]
    assert np.allclose(np.array(pm.pat_inv).reshape(3, 4)[:, :3], 2 * np.eye(3))
    assert not s.leaves[1].flags & abi.LEAF_SHADOW and s.leaves[0].flags & abi.LEAF_SHADOW
    cyl = s.leaves[2]
    assert (cyl.p0, cyl.p1) == (0.0, 2.0) and cyl.flags & abi.LEAF_CLOSED
    # children of a group: two chain levels (group, own), composed world->object = own^-1 * group^-1
    cube, cone = s.leaves[3], s.leaves[4]
    assert cube.chain_len == 2 and np.allclose(np.array(cube.w2o).reshape(3, 4)[:, 3], [0, 0, -2])
    assert (cone.p0, cone.p1) == (-1.0, 0.0) and not cone.flags & abi.LEAF_CLOSED


def test_yaml_errors_are_reported_not_fatal(rt):
    with pytest.raises(rt.RtamdError):
        rt.Scene.from_yaml_str("- add: sphere\n  material: nowhere-material\n")
    with pytest.raises(rt.RtamdError):
        rt.Scene.from_yaml_str("- add: camera\n  width: 10\n")
    # a singular transform is found on the host, before anything reaches the device (src/shape.rs:40 panics)
    bad = YAML + "- add: cube\n  transform:\n    - [scale, 0, 1, 1]\n"
    with pytest.raises(rt.RtamdError):
        rt.Scene.from_yaml_str(bad).flatten()


# ------------------------------------------------------------------ scene-graph files and flattening invariants
@pytest.mark.parametrize("name", [n for n in ALL_SCENES if n != "teapot_room"])
def test_scene_graph_file_round_trips(rt, name, tmp_path):
    p = str(tmp_path / "again.sg")
    rt.Scene.from_sg(sg_path(name)).save_sg(p)
    assert open(p).read() == open(sg_path(name)).read()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_flatten_invariants(rt, name):
    from importlib import import_module
    abi = import_module("ray-tracer_amd.abi")
    flat = rt.Scene.from_sg(sg_path(name)).flatten()
    s = flat.scene
    leaves = [s.leaves[i] for i in range(s.n_leaves)]
    visits = [s.visits[i] for i in range(s.n_visits)]
    in_csg = set()
    for r in range(s.n_csg_roots):
        root = s.csg_roots[r]
        assert 0 < root.leaf_hi - root.leaf_lo <= abi.MAX_CSG_LEAVES
        in_csg.update(range(root.leaf_lo, root.leaf_hi))
        last = s.csg_nodes[root.node_start + root.node_count - 1]  # post-order: the root comes last
        assert (last.leaf_lo, last.leaf_hi) == (root.leaf_lo, root.leaf_hi)
    tri_next = 0
    first_of_class = {}
    for i, L in enumerate(leaves):
        assert 0 <= L.material < s.n_materials and 1 <= L.chain_len <= abi.MAX_CHAIN
        # the levels a ray passes on its way to the leaf: their product is w2o (to rounding), the last one the leaf's own inverse
        # (= the last of its chain); groups alone above the leaf: the chain itself; below a CSG object the chain is the leaf's own
        assert 1 <= L.ray_len <= abi.MAX_RAY_LEVELS and 0 <= L.ray_start and L.ray_start + L.ray_len <= s.n_chain
        lv = [np.vstack([np.array(s.chain[(L.ray_start + k) * 12:(L.ray_start + k + 1) * 12]).reshape(3, 4), [0, 0, 0, 1]]) for k in range(L.ray_len)]
        prod = np.eye(4)
        for m in lv:
            prod = m @ prod
        assert np.allclose(prod[:3], np.array(L.w2o).reshape(3, 4), rtol=1e-12, atol=1e-12)
        own = np.array(s.chain[(L.chain_start + L.chain_len - 1) * 12:(L.chain_start + L.chain_len) * 12])
        assert np.array_equal(lv[-1][:3].ravel(), own)
        if i in in_csg:
            assert L.chain_len == 1
        else:
            assert (L.ray_start, L.ray_len) == (L.chain_start, L.chain_len)
        # classes are dense and numbered by first appearance; equal leaves share kind and material values
        assert 0 <= L.eq_class <= len(first_of_class) and L.eq_class < s.n_classes
        rep = leaves[first_of_class.setdefault(L.eq_class, i)]
        assert rep.kind == L.kind
        if L.kind == abi.MESH:
            assert 1 <= L.tri_count <= abi.MAX_MESH_TRIS and L.tri_start == tri_next  # triangles stay in DFS order
            tri_next += L.tri_count
        else:
            assert L.tri_count == 0
    assert tri_next == s.n_tris
    # every plain leaf is visited exactly once, every CSG root once per class among its leaves;
    # classes are contiguous and closed by CLASS_END; FIRST marks one visit per unit
    seen_leaf, seen_root_first, closed = [], [], set()
    for k, v in enumerate(visits):
        assert v.eq_class not in closed
        if v.flags & abi.VISIT_CSG:
            root = s.csg_roots[v.unit]
            assert v.eq_class in {leaves[i].eq_class for i in range(root.leaf_lo, root.leaf_hi)}
            if v.flags & abi.VISIT_FIRST:
                seen_root_first.append(v.unit)
        else:
            assert v.flags & abi.VISIT_FIRST and leaves[v.unit].eq_class == v.eq_class and v.unit not in in_csg
            seen_leaf.append(v.unit)
        assert v.class_material == leaves[first_of_class[v.eq_class]].material
        if v.flags & abi.VISIT_CLASS_END:
            closed.add(v.eq_class)
        else:
            assert visits[k + 1].eq_class == v.eq_class
    assert sorted(seen_leaf) == [i for i in range(s.n_leaves) if i not in in_csg]
    assert sorted(seen_root_first) == list(range(s.n_csg_roots))
    assert len(closed) == len({v.eq_class for v in visits}) <= s.n_classes


def test_teapot_triangles_form_bounded_mesh_leaves(rt):
    from importlib import import_module
    abi = import_module("ray-tracer_amd.abi")
    s_flat = rt.Scene.from_sg(sg_path("teapot_room")).flatten()
    s = s_flat.scene
    mesh = [s.leaves[i] for i in range(s.n_leaves) if s.leaves[i].kind == abi.MESH]
    assert sum(m.tri_count for m in mesh) == 6320 == s.n_tris
    assert all(m.flags & abi.LEAF_SMOOTH for m in mesh) and len({m.eq_class for m in mesh}) == 1
    assert s.n_leaves - len(mesh) == 12


# ------------------------------------------------------------------ the C ABI
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtamd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    declared = _declared_functions()
    assert declared == sorted(rt.GPU_SYMBOLS)
    lib = C.CDLL(os.path.join(rt.LIB_DIR, "librtamd.so"))
    for name in declared:
        assert hasattr(lib, name), name


def test_gpu_library_is_device_code_for_gfx950_and_independent_of_the_oracle(rt):
    so = os.path.join(rt.LIB_DIR, "librtamd.so")
    for lib in (so, os.path.join(rt.LIB_DIR, "librtamd_host.so")):
        needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
        assert "oracle" not in needed
    blob = open(so, "rb").read()
    assert b"gfx950" in blob and b"rtamd_render_fast_plain" in blob and b"rtamd_render_strict_csg" in blob
    # nothing in the package or the bench's product path names the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ray-tracer_amd")):
        if os.path.basename(dirpath) in ("build", "lib", "bin", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".inc", ".h")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(dirpath, f), errors="ignore").read().lower(), f


def test_no_gpu_means_an_error_not_a_fallback(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtamdError):
        rt.Context(0)
    with pytest.raises(rt.RtamdError):
        rt.Scene.from_sg(sg_path("default_world")).render()


def test_cli_reports_failure_without_gpu(rt, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "ray-tracer_amd", "bin", "yaml_renderer")
    p = tmp_path / "tiny.yaml"
    p.write_text(YAML)
    r = subprocess.run([exe, str(p)], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode != 0 and not (tmp_path / "tiny.png").exists()


# ------------------------------------------------------------------ bench helpers
def test_bench_flop_model_matches_the_per_scene_constants():
    import bench
    expect = {"cover": 1028, "reflect_refract": 518, "cylinders": 831, "tables": 1008, "glass_spheres": 138, "groups": 4964,
              "teapot_room": 518 - 62 + 33 + 6320 * 51}
    for name, flops in expect.items():
        assert bench.flops_per_ray(sg_path(name)) == flops, name


def test_bench_refuses_to_run_without_a_gpu():
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")


def test_rows_layout_is_the_rows_rule_as_runs(rt):
    """rtamd_rows_layout (what a host that moves the rows itself copies by) against the rule rows.py states row by row."""
    rows_mod = importlib.import_module("ray-tracer_amd.rows")
    for vsize in (0, 1, 7, 8, 77, 100, 2160):
        for tile_rows, part, n_parts, span in ((8, 0, 1, 1), (8, 1, 2, 1), (8, 7, 8, 1), (3, 2, 5, 2), (100, 0, 2, 1), (8, 0, 9, 2), (8, 2, 9, 1),
                                               (5, 3, 7, 4), (1, 0, 3, 3)):
            lay = rt.rows_layout(vsize, tile_rows, part, n_parts, span)
            rows = []
            for c in range(lay["n_chunks"]):
                start = lay["first_row"] + c * lay["stride_rows"]
                rows += list(range(start, start + lay["chunk_rows"]))
            start = lay["first_row"] + lay["n_chunks"] * lay["stride_rows"]
            rows += list(range(start, start + lay["tail_rows"]))
            assert rows == rows_mod.part_row_indices(vsize, tile_rows, part, n_parts, span).tolist(), (vsize, tile_rows, part, n_parts, span)
    with pytest.raises(rt.RtamdError):
        rt.rows_layout(10, 8, 2, 2)


def test_header_is_plain_c_and_the_c_example_builds(tmp_path):
    """include/rtamd.h must compile as C99 on its own (bindings in any language start from it), and examples/render_c.c
    -- a whole host in C -- against it, warnings as errors."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = tmp_path / "probe.c"
    probe.write_text('#include "rtamd.h"\nint main(void) { rtamd_scene s; rtamd_stats t; rtamd_layout l; (void)s, (void)t, (void)l; return (int)sizeof(rtamd_rows) - 16; }\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(probe)], check=True)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-c", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "render_c.c"), "-o", str(tmp_path / "render_c.o")], check=True)


def test_c99_host_builds_the_csg_scene_call_by_call_to_the_bytes_of_the_librarys_own():
    """examples/build_world_c.c without a GPU ("-": nothing is rendered): the scene it hands over node by node through
    rth_add_* / rth_csg_begin / rth_csg_end flattens to the bytes of rth_builtin_scene("csg")."""
    import subprocess
    exe = os.path.join(ROOT, "ray-tracer_amd", "bin", "build_world_c")
    r = subprocess.run([exe, "100", "50", "-"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "11 leaves, 6 materials, 2 csg nodes in 2 objects" in r.stdout and "own csg scene: yes" in r.stdout


def _rust_structs(text):
    """{name: [(field, rust type), ...]} of every #[repr(C)] struct of a Rust source."""
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", text, re.S):
        out[m.group(1)] = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+):\s*([^,\n]+),?", m.group(2))]
    return out


def test_rust_ffi_crate_mirrors_the_structs_field_for_field(rt):
    """rtamd-sys/src/lib.rs cannot be compiled here (no Rust toolchain in the image), so its #[repr(C)] structs are held
    against the ctypes mirrors the whole test-suite renders through (which upload validation and the GPU tests prove
    against the C headers): same fields, same order, same types; and every function of both headers is declared."""
    abi = rt.abi
    text = open(os.path.join(ROOT, "rtamd-sys", "src", "lib.rs")).read()
    rust = _rust_structs(text)
    ctype_of = {"f64": C.c_double, "i32": C.c_int32, "u32": C.c_uint32, "u64": C.c_uint64, "i64": C.c_int64}
    pairs = {"rtamd_leaf": abi.Leaf, "rtamd_material": abi.Material, "rtamd_csg_node": abi.CsgNode, "rtamd_csg_root": abi.CsgRoot,
             "rtamd_visit": abi.Visit, "rtamd_camera": abi.Camera, "rtamd_light": abi.Light, "rtamd_scene": abi.SceneStruct,
             "rtamd_rows": rt.Rows, "rtamd_stats": rt.Stats, "rtamd_layout": rt.Layout, "rth_material": rt.Material}
    assert set(rust) == set(pairs)
    for name, cls in pairs.items():
        fields = rust[name]
        assert [f for f, _ in fields] == [f for f, _ in cls._fields_], name
        for (fname, rtype), (_, ctype) in zip(fields, cls._fields_):
            m = re.fullmatch(r"\[(\w+); (\d+)\]", rtype)
            if m:
                assert ctype._length_ == int(m.group(2)) and ctype._type_ is ctype_of[m.group(1)], (name, fname)
            elif rtype.startswith("*const "):
                assert issubclass(ctype, C._Pointer), (name, fname)
                inner = rtype[len("*const "):]
                assert ctype._type_ is (ctype_of[inner] if inner in ctype_of else pairs[inner]), (name, fname)
            elif rtype in pairs:
                assert ctype is pairs[rtype], (name, fname)
            else:
                assert ctype is ctype_of[rtype], (name, fname)
    declared = set(re.findall(r"pub fn (\w+)\(", text))
    assert declared == set(rt.GPU_SYMBOLS) | set(rt.HOST_SYMBOLS), declared ^ (set(rt.GPU_SYMBOLS) | set(rt.HOST_SYMBOLS))
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "rtamd-sys", f))
