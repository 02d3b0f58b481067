#!/usr/bin/env python3
"""Development aid: per-phase cycle stamps of the main loop (needs the -DRTAMD_PROFILE build:
RTAMD_LIB_DIR=ray-tracer_amd/lib_prof RTAMD_PROFILE_DUMP=1 python3 scripts/gpu_phase.py [scene] [W H])."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import torch
name = sys.argv[1] if len(sys.argv) > 1 else "cover"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
for mode, mname in ((rt.MODE_FAST, "fast"), (rt.MODE_STRICT, "strict")):
    for it in range(3):
        ctx.reset_stats()
        ctx.render_rows(8, 0, 1, mode, d_rgb=buf.data_ptr())
        st = ctx.stats()
    print(f"{name} {W}x{H} {mname}: kernel {st.kernel_ms:.3f} ms, rays {st.rays}, containers {st.container_passes}, shade points {st.shade_points}", flush=True)
