#!/usr/bin/env python3
"""Development aid: rtamd_multi_render with the device list naming GPU 0 one to four times, against one context (the
overhead of the split itself: part buffers, strided copies, several kernels sharing the card)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
name, W, H = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("cover", 3840, 2160)
flat = rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten()
for n in (1, 2, 4):
    m = rt.MultiContext([0] * n).upload(flat)
    for k0 in (1, 2):
        ts = []
        for it in range(6):
            t0 = time.perf_counter()
            m.render_on_device(rt.MODE_FAST, 8, k0, 1)
            ts.append((time.perf_counter() - t0) * 1e3)
        st, r, p = m.stats()
        print(f"{name} {W}x{H} contexts {n} shares {k0}:1  wall {np.median(ts[1:]):.3f} ms  render/rank {np.round(r, 3)}  place/rank {np.round(p, 3)}", flush=True)
    m.close()
