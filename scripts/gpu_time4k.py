#!/usr/bin/env python3
"""Development aid: best-of-5 kernel time (FAST) of the named scenes at one frame size, with the kernel variant used.
python3 scripts/gpu_time4k.py W H scene [scene ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import torch
W, H = int(sys.argv[1]), int(sys.argv[2])
ctx = rt.Context(0)
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
for name in sys.argv[3:]:
    ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
    ts = []
    for it in range(5):
        ctx.reset_stats()
        ctx.render_rows(8, 0, 1, rt.MODE_FAST, d_rgb=buf.data_ptr())
        ts.append(ctx.stats().kernel_ms)
    print(f"{name} {W}x{H} fast: kernel {min(ts):.3f} ms  sum {float(buf.sum()):.6f}", flush=True)
