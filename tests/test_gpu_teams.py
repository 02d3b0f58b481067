"""Teams of lanes per pixel (small frames, fast arithmetic): a shade point's shadow rays for all its lights are traced
side by side by the lanes next to the pixel's own instead of one after the other.  The rays, their arithmetic and the
order of the sums (World::shade_hit, src/scene.rs:92-121) are the same, so the frames must be the same bits, every
pixel must be rendered exactly once, and every ray count must agree -- with one light (teams of 2), two and three
lights (teams of 4), in whole frames and in parts of frames whose groups of pixels mix both kinds of work."""
import numpy as np
import pytest

import sgtext as sg
from conftest import sg_path

pytestmark = pytest.mark.gpu

COUNTS = ("rays", "primary_rays", "shadow_rays", "reflect_rays", "refract_rays", "container_passes", "shade_points", "light_samples",
          "csg_overflows")


def _frames(rt, ctx, monkeypatch, w, h, n_parts=1):
    """{setting: (frame, stats of the whole frame)} for teams never / whenever allowed / by the library's own rule."""
    import importlib

    import torch
    rows_mod = importlib.import_module("ray-tracer_amd.rows")
    out = {}
    for setting in ("0", "1", None):
        if setting is None:
            monkeypatch.delenv("RTAMD_TEAMS", raising=False)
        else:
            monkeypatch.setenv("RTAMD_TEAMS", setting)
        frame = np.zeros((h, w, 3))
        ctx.reset_stats()
        for part in range(n_parts):
            rows = ctx.part_rows(8, part, n_parts)
            buf = torch.full((rows, w, 3), float("nan"), dtype=torch.float64, device="cuda")
            ctx.render_rows(8, part, n_parts, rt.MODE_FAST, d_rgb=buf.data_ptr())
            torch.cuda.synchronize()
            frame[rows_mod.part_row_indices(h, 8, part, n_parts)] = buf.cpu().numpy()
        out[setting] = (frame, ctx.stats().as_dict())
    return out


def _check(out, w, h, expect_teams):
    base, base_st = out["0"]
    assert np.isfinite(base).all() and base_st["team_shade_points"] == 0 and base_st["primary_rays"] == w * h
    for setting in ("1", None):
        frame, st = out[setting]
        assert np.array_equal(frame, base), setting
        assert [st[c] for c in COUNTS] == [base_st[c] for c in COUNTS], (setting, st, base_st)
    assert (out["1"][1]["team_shade_points"] > 0) == expect_teams


@pytest.mark.parametrize("name,w,h,teams", [("reflect_refract", 400, 200, True), ("cover", 480, 270, True), ("csg", 400, 200, True),
                                            ("teapot_lowres_room", 240, 120, True), ("groups", 240, 120, False)])
def test_committed_scenes_same_bits(rt, gpu_ctx, monkeypatch, name, w, h, teams):
    """One light (reflect_refract, csg, teapot), two (cover); four lights are more than a team holds (groups)."""
    gpu_ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(w, h).flatten())
    _check(_frames(rt, gpu_ctx, monkeypatch, w, h), w, h, teams)


@pytest.mark.parametrize("n_parts", [5, 25])
def test_parts_of_a_frame_render_every_pixel_once(rt, gpu_ctx, monkeypatch, n_parts):
    """Parts of 40 and of 8 rows: waves go from pieces of marked groups to whole groups and back."""
    w, h = 400, 200
    gpu_ctx.upload(rt.Scene.from_sg(sg_path("cylinders")).set_resolution(w, h).flatten())
    _check(_frames(rt, gpu_ctx, monkeypatch, w, h, n_parts), w, h, True)


def test_three_lights_glass_and_mirror(rt, oracle, gpu_ctx, monkeypatch, tmp_path):
    """Three lights (a team's last lane gets the third), a glass ball inside a glass ball (container passes run beside the
    team's shadow rays), a mirror, a ball that casts no shadow; checked against the oracle as well."""
    s = sg.SG()
    s.shape("plane", s.mat(color=(0.8, 0.8, 0.8), specular=0.2, reflective=0.3, pattern=("checkers", (0.9, 0.9, 0.9), (0.2, 0.3, 0.4), sg.translation(0.13, 0.01, 0.27))),
            sg.translation(0, -1, 0))
    glass = s.mat(color=(0.1, 0.1, 0.1), diffuse=0.2, transparency=0.9, reflective=0.9, ri=1.5, shininess=300)
    s.shape("sphere", glass, sg.translation(-0.6, 0.1, 0.2))
    s.shape("sphere", s.mat(color=(0.2, 0.1, 0.1), diffuse=0.3, transparency=0.8, ri=1.0003), sg.translation(-0.6, 0.1, 0.2) @ sg.scaling(0.5, 0.5, 0.5))
    s.shape("sphere", s.mat(color=(0.9, 0.9, 1.0), diffuse=0.3, reflective=0.8, shininess=40.5), sg.translation(1.5, 0.0, 1.0))
    s.shape("cube", s.mat(color=(0.9, 0.4, 0.2)), sg.translation(0.4, -0.6, -1.3) @ sg.scaling(0.4, 0.4, 0.4), shadow=False)
    s.light((-4.9, 4.9, -1.0), (0.5, 0.5, 0.5))
    s.light((3.0, 6.0, -4.0), (0.4, 0.3, 0.3))
    s.light((0.5, 2.5, -6.0), (0.2, 0.25, 0.3))
    w, h = 320, 160
    s.set_camera(w, h, 1.0, (0.3, 1.2, -5.2), (0, 0, 0))
    path = s.write(tmp_path / "three_lights.sg")
    gpu_ctx.upload(rt.Scene.from_sg(path).set_resolution(w, h).flatten())
    out = _frames(rt, gpu_ctx, monkeypatch, w, h)
    _check(out, w, h, True)
    ref, _, _ = oracle.OracleScene(path).render(w, h, threads=8)
    assert float(np.abs(out["1"][0] - ref).max()) <= 1e-8
