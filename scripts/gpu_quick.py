#!/usr/bin/env python3
"""Quick GPU-vs-oracle comparison over the golden scenes (development aid; prints one line per scene/mode)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
import oracle_lib

scenes = sys.argv[1:] or ["default_world", "reflect_refract", "cover", "cylinders", "tables", "glass_spheres", "groups",
                          "puppets", "rr_code", "csg", "teapot_lowres_room", "teapot_room"]
sizes = {"default_world": (11, 11), "teapot_room": (60, 30), "glass_spheres": (96, 96), "cover": (96, 96)}
ctx = rt.Context(0)
for name in scenes:
    sg = os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")
    w, h = sizes.get(name, (120, 60))
    sc = rt.Scene.from_sg(sg).set_resolution(w, h)
    flat = sc.flatten()
    ctx.upload(flat)
    t0 = time.time()
    ref, ref8, ost = oracle_lib.OracleScene(sg).render(w, h, threads=8)
    t_or = time.time() - t0
    for mode, mname in ((rt.MODE_STRICT, "strict"), (rt.MODE_FAST, "fast")):
        ctx.reset_stats()
        rgb, u8 = ctx.render(w, h, mode, want_u8=True)
        st = ctx.stats()
        d = np.abs(rgb - ref)
        bad = ~np.isfinite(rgb).all()
        d8 = np.abs(u8.astype(int) - ref8.astype(int)).max(axis=2)
        exact = (rgb == ref).all(axis=2).mean()
        print(f"{name:20s} {mname:6s} {w}x{h} maxabs {np.nanmax(d):.3e} bitexact {100*exact:7.3f}% u8>1 {100*(d8>1).mean():.4f}% "
              f"u8max {d8.max()} nonfinite {bad} rays {st.rays} (oracle {ost['rays']}) kernel {st.kernel_ms:.3f} ms oracle {t_or:.1f}s",
              flush=True)
