#!/usr/bin/env python3
"""Development aid: kernel time of each rank's part of a frame (what one GPU of an N-GPU job renders),
measured one part after the other on a single GPU.  usage: gpu_parts.py [scene] [W H] [tile_rows]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import torch
name = sys.argv[1] if len(sys.argv) > 1 else "cover"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
R = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
for n in (1, 2, 4, 8):
    times = []
    for p in range(n):
        best = 1e9
        for it in range(4):
            ctx.reset_stats()
            ctx.render_rows(R, p, n, rt.MODE_FAST, d_rgb=buf.data_ptr())
            best = min(best, ctx.stats().kernel_ms)
        times.append(best)
    print(f"{name} {W}x{H} tile_rows {R} parts {n}: kernel ms per part " + " ".join(f"{t:.3f}" for t in times) +
          f"  -> slowest {max(times):.3f}, speed-up over one part {0 if n == 1 else base / max(times):.2f}", flush=True)
    if n == 1:
        base = times[0]
