#!/usr/bin/env python3
"""Development aid: wall time of rtamd_render (host buffers: device allocation, kernel, copy back) against the
kernel alone.  usage: gpu_host_render.py [scene] [W H]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cover"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
import ctypes as C
import numpy as np
gpu = rt._gpu
for pinned in (False, True):
    new = rt.host_empty if pinned else np.empty
    rgb = new((H, W, 3), dtype=np.float64)
    rgb[:] = 0  # touch the pages: a canvas that exists already
    for it in range(4):
        ctx.reset_stats()
        t0 = time.perf_counter()
        rc = gpu.rtamd_render(ctx._h, rt.MODE_FAST, rgb.ctypes.data, None)
        dt = (time.perf_counter() - t0) * 1e3
        assert rc == 0
        print(f"{name} {W}x{H} rtamd_render into {'page-locked' if pinned else 'ordinary'} memory: {dt:.2f} ms wall, kernel {ctx.stats().kernel_ms:.2f} ms", flush=True)
