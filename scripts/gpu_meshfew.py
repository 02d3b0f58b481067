#!/usr/bin/env python3
"""Development aid: the two forms of the triangle test (loop over the triangles of a run / one triangle per
lane) must give the same bits; prints kernel times for a few settings of RTAMD_MESH_FEW."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "teapot_room"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
ctx = rt.Context(0)
ctx.upload(rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten())
for mode, mname in ((rt.MODE_STRICT, "strict"), (rt.MODE_FAST, "fast")):
    ref = None
    for few in (0, 1, 2, 4, 8, 16, 64):
        os.environ["RTAMD_MESH_FEW"] = str(few)
        best = 1e9
        for it in range(3):
            ctx.reset_stats()
            rgb = ctx.render(W, H, mode)
            best = min(best, ctx.stats().kernel_ms)
        if ref is None:
            ref = rgb
        same = np.array_equal(rgb, ref)
        print(f"{name} {W}x{H} {mname} few={few}: kernel {best:.3f} ms, bits equal to few=0: {same}"
              + ("" if same else f" (max |diff| {np.abs(rgb - ref).max():.3e}, {int((rgb != ref).any(axis=2).sum())} pixels)"), flush=True)
