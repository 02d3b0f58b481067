#!/usr/bin/env python3
"""Development aid: kernel time per ray as a function of the number of leaves (synthetic cube grids)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")

def scene(n, kind="cube", rot=False, refl=0.0):
    y = ["- add: camera\n  width: 1920\n  height: 1080\n  field-of-view: 0.9\n  from: [0, 6, -14]\n  to: [0, 0, 0]\n  up: [0, 1, 0]\n",
         "- add: light\n  at: [-10, 20, -10]\n  intensity: [1, 1, 1]\n"]
    side = int(np.ceil(np.sqrt(max(n, 1))))
    for i in range(n):
        gx, gz = i % side, i // side
        x, z = (gx - side / 2) * 2.5, (gz - side / 2) * 2.5
        tr = f"    - [scale, 0.8, 0.8, 0.8]\n" + (f"    - [rotate-y, 0.3]\n" if rot else "") + f"    - [translate, {x}, 0, {z}]\n"
        extra = "  min: -1\n  max: 1\n  closed: true\n" if kind in ("cylinder", "cone") else ""
        y.append(f"- add: {kind}\n{extra}  transform:\n{tr}  material:\n    color: [0.8, 0.3, 0.3]\n    specular: 0\n    reflective: {refl}\n")
    return "".join(y)

ctx = rt.Context(0)
mode = rt.MODE_FAST if (len(sys.argv) < 2 or sys.argv[1] == "fast") else rt.MODE_STRICT
import torch
for kind, rot in (("cube", False), ("cube", True), ("sphere", False), ("cylinder", False)):
    for n in (0, 1, 8, 16, 32, 64):
        sc = rt.Scene.from_yaml_str(scene(n, kind, rot))
        ctx.upload(sc.flatten())
        buf = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
        ts = []
        for it in range(4):
            ctx.reset_stats()
            ctx.render_rows(8, 0, 1, mode, d_rgb=buf.data_ptr())
            st = ctx.stats()
            ts.append(st.kernel_ms)
        rays = st.primary_rays + st.shadow_rays + st.reflect_rays + st.refract_rays
        t = min(ts)
        # SIMD cycles per ray-wave: t * 2.4e9 * 1024 SIMDs / (rays / 64)
        cyc = t * 1e-3 * 2.4e9 * 1024 / (rays / 64)
        print(f"{kind:9s} rot={int(rot)} n={n:3d} rays={rays:9d} kernel={t:7.3f} ms  SIMD-cycles/ray-wave={cyc:8.0f}  per-leaf={cyc / max(n, 1):7.0f}", flush=True)
