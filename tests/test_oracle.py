"""The CPU oracle against what pins it: the known answers transcribed from the reference's unit tests,
the reference author's committed renders, and its own committed float64 frames."""
import os
import re
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import ALL_SCENES, GOLDEN, ROOT, golden_frame, sg_path


def test_known_answer_checks_pass():
    """oracle/kat_tests.cpp holds the inputs and expected values of the reference's #[test] functions
    for this path (src/scene.rs:287-708, src/intersection.rs:155-320, src/shapes/*.rs, src/material.rs:159-285,
    src/pattern.rs:112-232, src/matrix.rs:333-790, src/tuple.rs:398-635)."""
    r = subprocess.run([os.path.join(ROOT, "oracle", "build", "kat_tests")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    m = re.search(r"kat: (\d+) checks, (\d+) failures", r.stdout)
    assert m and int(m.group(1)) >= 464 and int(m.group(2)) == 0


@pytest.mark.parametrize("name", ALL_SCENES)
def test_oracle_reproduces_committed_frames(oracle, manifest, name):
    """Regression: the committed float64 frames are what the oracle computes today, bit for bit."""
    w, h = manifest[name]["small"]
    rgb, _, st = oracle.OracleScene(sg_path(name)).render(w, h, threads=4)
    assert np.array_equal(rgb, golden_frame(name, (w, h)))
    assert st["rays"] == manifest[name]["small_stats"]["rays"]
    assert st["shade_points"] == manifest[name]["small_stats"]["shade_points"]


# Renders committed by the reference's author (images/*.png) that correspond to a scene at HEAD.
REF_IMAGES = [("cover", "032.png"), ("cylinders", "033.png"), ("tables", "030.png"), ("glass_spheres", "034.png"),
              ("groups", "035.png"), ("puppets", "031.png"), ("rr_code", "020.png"), ("teapot_room", "025.png")]


@pytest.mark.parametrize("name,image", REF_IMAGES)
def test_oracle_matches_reference_render(oracle, name, image):
    """8-bit output of the oracle == the reference's own render of the same scene.  Whole frames were
    compared when the fixtures were made (manifest.json: all equal); here a band of rows (a lattice of
    pixels for the two scenes that are slow on a CPU) keeps the suite at seconds."""
    im = np.array(Image.open(os.path.join(GOLDEN, "ref_images", image)).convert("RGB"))
    h, w = im.shape[:2]
    if name in ("teapot_room", "groups"):
        ys, xs = np.meshgrid(np.arange(3, h, 9), np.arange(2, w, 11), indexing="ij")
        pix = (ys * w + xs).ravel()
    else:
        rows = np.arange(h // 3, h // 3 + max(8, h // 10))
        rows = np.concatenate([rows, np.arange(h - 12, h - 4)])
        pix = (rows[:, None] * w + np.arange(w)[None, :]).ravel()
    _, u8, _ = oracle.OracleScene(sg_path(name)).render_pixels(pix, w, h, threads=8)
    assert np.array_equal(u8, im.reshape(-1, 3)[pix])


def test_faithful_and_cached_modes_agree(oracle):
    """`faithful` (per-call inverses, children re-evaluated once per light: the reference's work) and
    `cached` produce the same bits; only the ray count differs (two lights on the cover scene)."""
    o = oracle.OracleScene(sg_path("cover"))
    pix = np.arange(0, 64 * 64, 7)
    a, _, sa = o.render_pixels(pix, 64, 64, faithful=False)
    b, _, sb = o.render_pixels(pix, 64, 64, faithful=True)
    assert np.array_equal(a, b)
    assert sb["rays"] > sa["rays"] and sb["inversions"] > 100 * sa["inversions"]


def test_color_at_known_answers(oracle):
    """World::color_at on the default world (src/scene.rs:324-366 and :458-468)."""
    o = oracle.OracleScene(sg_path("default_world"))
    rays = np.array([[0, 0, -5, 0, 1, 0], [0, 0, -5, 0, 0, 1]], dtype=float)
    c = o.color_at(rays, 4)
    assert np.allclose(c[0], 0, atol=1e-5)
    assert np.allclose(c[1], [0.38066, 0.47583, 0.2855], atol=1e-5)
    frame, _, _ = o.render(11, 11)
    assert np.allclose(frame[5, 5], [0.38066, 0.47583, 0.2855], atol=1e-5)
    assert np.array_equal(o.color_at(rays, 0)[1], c[1])  # an opaque, matt hit does not depend on the budget
