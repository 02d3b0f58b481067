#!/usr/bin/env python3
"""Development aid: kernel time of one scene at growing frame sizes (fixed costs show as a non-zero intercept)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import torch
name = sys.argv[1] if len(sys.argv) > 1 else "cover"
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = rt.Context(0)
for W, H in ((960, 540), (1920, 1080), (3840, 2160), (7680, 4320), (15360, 4320)):
    ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten(budget))
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    ts = []
    for it in range(4):
        ctx.reset_stats()
        ctx.render_rows(8, 0, 1, rt.MODE_FAST, d_rgb=buf.data_ptr())
        st = ctx.stats()
        ts.append(st.kernel_ms)
    rays = st.primary_rays + st.shadow_rays + st.reflect_rays + st.refract_rays
    print(f"{name} budget {budget} {W}x{H}: kernel {min(ts):.3f} ms, {rays / min(ts) / 1e3:.0f} Mrays/s, {min(ts) * 1e6 / (W * H):.3f} ns/pixel", flush=True)
    del buf
