"""The hierarchy of bounds records (and the re-cut pools of triangles, the boxes on their runs, the one-triangle-per-lane
form) only skips work: with all of it switched off at upload (RTAMD_BRUTE_FORCE: every leaf and every triangle of every
run for every ray, which is the reference's World::intersect, src/scene.rs:19-26) a frame must come out the same to the
bit in both arithmetic modes, with the same rays -- and with more tests executed."""
import numpy as np
import pytest

from conftest import sg_path

pytestmark = pytest.mark.gpu

RAYS = ("rays", "primary_rays", "shadow_rays", "reflect_rays", "refract_rays", "container_passes", "shade_points", "light_samples")


@pytest.mark.parametrize("name,w,h", [("cover", 192, 108), ("tables", 160, 90), ("groups", 120, 80), ("cylinders", 160, 90), ("csg", 128, 72),
                                      ("teapot_lowres_room", 96, 54), ("glass_spheres", 128, 128), ("puppets", 120, 80)])
def test_frames_do_not_depend_on_the_bounds_records(rt, gpu_ctx, monkeypatch, name, w, h):
    flat = rt.Scene.from_sg(sg_path(name)).set_resolution(w, h).flatten()
    out = {}
    for brute in (False, True):
        if brute:
            monkeypatch.setenv("RTAMD_BRUTE_FORCE", "1")
        else:
            monkeypatch.delenv("RTAMD_BRUTE_FORCE", raising=False)
        gpu_ctx.upload(flat)  # (the switch is read when a scene is uploaded)
        gpu_ctx.count_work(True)
        frames = []
        for mode in (rt.MODE_FAST, rt.MODE_STRICT):
            gpu_ctx.reset_stats()
            rgb, u8 = gpu_ctx.render(w, h, mode, want_u8=True)
            frames.append((rgb, u8, gpu_ctx.stats().as_dict()))
        gpu_ctx.count_work(False)
        out[brute] = frames
    monkeypatch.delenv("RTAMD_BRUTE_FORCE", raising=False)
    gpu_ctx.upload(flat)
    for (rgb, u8, st), (rgb_b, u8_b, st_b) in zip(out[False], out[True]):
        assert np.array_equal(rgb, rgb_b) and np.array_equal(u8, u8_b)
        assert [st[k] for k in RAYS] == [st_b[k] for k in RAYS]
        executed = lambda s: sum(v for k, v in s.items() if k.startswith("tests_") and k != "tests_box")
        assert executed(st_b) >= executed(st)
