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
