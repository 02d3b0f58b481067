"""The command-line renderer end to end on the GPU (src/bin/yaml_renderer.rs:410-441 of the reference: parse the YAML,
render, write ./<stem>.png): run by tests/conftest.py at the start of the session, checked here against Scene.render
of the same YAML through the library."""
import os

import numpy as np
import pytest
from PIL import Image

import conftest

pytestmark = pytest.mark.gpu


def _run(key):
    r = conftest.CLI_RUNS.get(key)
    assert r is not None, "the session did not run bin/yaml_renderer (no /dev/kfd, no binary, or not started with -m gpu)"
    assert not isinstance(r, Exception), r
    assert r.returncode == 0, r.stderr
    return r


def test_cli_writes_the_png_of_the_frame_the_library_renders(rt):
    r = _run("default")
    out = r.stdout.splitlines()
    assert out[0].startswith("Parsed in: ") and out[1] == "Rendering" and out[2].startswith("Rendered in: ")  # yaml_renderer.rs:406-415
    png = os.path.join(conftest.CLI_RUNS["dir"], "own_scene.png")  # ./<stem>.png, yaml_renderer.rs:419-429
    im = np.array(Image.open(png).convert("RGB"))
    rgb, u8, st = rt.Scene.from_yaml(conftest.CLI_RUNS["yaml"]).render(rt.MODE_FAST, want_u8=True)
    assert im.shape == (100, 200, 3) and np.array_equal(im, u8) and np.array_equal(u8, rt.quantize(rgb))
    assert len(np.unique(im.reshape(-1, 3), axis=0)) > 100  # a picture, not a blank canvas
    assert "%d rays" % st.rays in out[2]


def test_cli_flags_override_size_mode_and_format(rt):
    _run("flags")
    text = open(os.path.join(conftest.CLI_RUNS["dir"], "small.ppm")).read()
    rgb, _ = rt.Scene.from_yaml(conftest.CLI_RUNS["yaml"]).set_resolution(96, 64).render(rt.MODE_STRICT)
    assert text == rt.canvas_to_ppm(rgb)
    assert text.startswith("P3\n96 64\n255\n") and max(len(ln) for ln in text.splitlines()) <= 70  # src/tuple.rs:343-371


def test_host_in_plain_c_renders_on_one_gpu_and_on_a_list_of_gpus():
    """examples/render_c.c: a scene laid out by hand in the structs of include/rtamd.h, rtamd_render and rtamd_multi_render
    called from C99 -- the frames must agree to the bit (the program's exit status says so) and show the scene."""
    r = _run("render_c")
    out = r.stdout.splitlines()
    assert out[0].startswith("one GPU: 400x225, ") and " rays, kernel " in out[0]
    rays = int(out[0].split(", ")[1].split()[0])
    assert rays > 3 * 400 * 225  # every pixel sees the floor or a ball: a shadow ray each, reflections on top
    centre = [float(v) for v in out[0].split("centre pixel ")[1].split()]
    assert all(0.0 < c < 1.5 for c in centre)
    assert out[1].startswith("3 ranks: largest difference from the one-GPU frame 0 ")


def test_cli_at_3840x2160_moves_three_bytes_per_pixel_unless_asked_for_more(rt):
    """An image file needs the 8-bit channels only: the program fetches those alone (25 MB instead of 224 MB over the PCIe
    link) into page-locked memory.  Same PNG either way; the repeated frames say what a frame costs once the buffers exist."""
    import re
    pngs, times = {}, {}
    for key in ("4k_u8", "4k_f64"):
        r = _run(key)
        m = re.search(r"Rendered again in: ([0-9.]+) ms \(median of 5, (\d+) bytes per pixel into (\S+) memory; kernel ([0-9.]+) ms\)", r.stdout)
        assert m, r.stdout
        times[key] = (float(m.group(1)), int(m.group(2)), m.group(3), float(m.group(4)))
        pngs[key] = np.array(Image.open(os.path.join(conftest.CLI_RUNS["dir"], key + ".png")).convert("RGB"))
        print(key, r.stdout.splitlines()[2:4])
    assert pngs["4k_u8"].shape == (2160, 3840, 3) and np.array_equal(pngs["4k_u8"], pngs["4k_f64"])
    u8, _ = rt.Scene.from_yaml(conftest.CLI_RUNS["yaml"]).set_resolution(3840, 2160).render(rt.MODE_FAST, want_f64=False)
    assert np.array_equal(pngs["4k_u8"], u8)
    assert times["4k_u8"][1:3] == (3, "page-locked") and times["4k_f64"][1:3] == (24, "page-locked")
    # generous bounds (shared test machines): flatten + upload + kernel + copy of one 4K frame
    assert times["4k_u8"][0] < 10.0 and times["4k_f64"][0] < 25.0 and times["4k_u8"][0] < times["4k_f64"][0]


def test_host_in_plain_c_builds_its_scene_call_by_call(oracle):
    """examples/build_world_c.c: the CSG scene handed over through the scene builder of include/rtamd_host.h, flattened by
    the library, rendered in strict arithmetic: the frame the oracle renders of that scene."""
    r = _run("build_world_c")
    assert "flat scene equals the library's own csg scene: yes" in r.stdout and "2 csg nodes in 2 objects" in r.stdout
    got = np.fromfile(os.path.join(conftest.CLI_RUNS["dir"], "csg.f64"), dtype=np.float64).reshape(50, 100, 3)
    want = np.load(os.path.join(conftest.GOLDEN, "oracle", "csg_100x50.npy"))
    assert np.abs(got - want).max() <= 1e-10  # the bar of strict arithmetic (tests/test_gpu_parity.py)
