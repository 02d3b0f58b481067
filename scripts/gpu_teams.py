#!/usr/bin/env python3
"""Development aid: frames with and without teams of lanes per pixel (RTAMD_TEAMS) -- equal bits, kernel times."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import torch
ctx = rt.Context(0)
cases = [("reflect_refract", 400, 200), ("cover", 480, 270), ("cover", 960, 540), ("cover", 1920, 270), ("cover", 3840, 2160), ("glass_spheres", 600, 600),
         ("tables", 400, 200), ("teapot_room", 400, 200), ("csg", 400, 200), ("cylinders", 400, 200)]
for name, W, H in cases:
    ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    out = {}
    for teams in ("0", "1", "auto"):
        os.environ["RTAMD_TEAMS"] = teams
        if teams == "auto":
            del os.environ["RTAMD_TEAMS"]
        ts = []
        for it in range(6):
            ctx.reset_stats()
            ctx.render_rows(8, 0, 1, rt.MODE_FAST, d_rgb=buf.data_ptr())
            st = ctx.stats()
            ts.append(st.kernel_ms)
        out[teams] = (buf.clone(), np.median(ts[1:]), st.primary_rays, st.shadow_rays, st.reflect_rays, st.refract_rays, st.container_passes, st.shade_points, st.light_samples)
    same = torch.equal(out["0"][0], out["1"][0]) and torch.equal(out["0"][0], out["auto"][0])
    print(f"{name} {W}x{H}: teams off {out['0'][1]:.3f} ms, on {out['1'][1]:.3f} ms, by rule {out['auto'][1]:.3f} ms, same bits {same}, counts equal {out['0'][2:] == out['1'][2:] == out['auto'][2:]}", flush=True)
    if out['0'][2:] != out['1'][2:]:
        print("   off", out['0'][2:], "\n   on ", out['1'][2:])
# parts of a frame: every pixel once
name, W, H = "cylinders", 400, 200
ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
for n_parts in (1, 5, 25):
    for teams in ("0", "1"):
        os.environ["RTAMD_TEAMS"] = teams
        bad = []
        for part in range(n_parts):
            rows = ctx.part_rows(8, part, n_parts)
            buf = torch.empty((rows, W, 3), dtype=torch.float64, device="cuda")
            ctx.reset_stats()
            ctx.render_rows(8, part, n_parts, rt.MODE_FAST, d_rgb=buf.data_ptr())
            st = ctx.stats()
            if st.primary_rays != rows * W:
                bad.append((part, st.primary_rays, rows * W))
        print("parts", n_parts, "teams", teams, "wrong primary counts:", bad, flush=True)
