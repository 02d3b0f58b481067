#!/usr/bin/env python3
"""Development aid for the GPU box: one tool, several sub-commands (each prints one line per scene and mode).

    scripts/gpu_dev.py time   [--modes fast,strict] [--frames 6] scene W H [scene W H ...]
        kernel time of a scene at a size (median and min over the frames after the first), rays, Mrays/s
    scripts/gpu_dev.py hash   [--modes fast,strict] [--out file.json] [--against file.json] scene W H ...
        SHA-256 of the f64 frame and of the 8-bit frame plus the ray counts: two builds that render the same bits print
        the same lines; with --against the run fails when a frame differs from the recorded one
    scripts/gpu_dev.py check  [scene ...]
        small frames against the CPU oracle, both modes (max |diff|, share of bit-identical pixels, 8-bit distance)
    scripts/gpu_dev.py suite  [--out file.json] [--against file.json]
        `time` + `hash` over the benchmark configurations at their full sizes (what a kernel change is judged by)
    scripts/gpu_dev.py work   scene W H ...
        tests the traversals execute per ray, by kind (the counting twins of the kernels), and the mix of rays
    scripts/gpu_dev.py host   scene W H
        per-call time of Scene.render into pageable / page-locked host memory, f64 and 8-bit only
    scripts/gpu_dev.py parts  scene W H
        kernel time of each rank's part of a frame (what one GPU of 1, 2, 4, 8 renders), one part after the other on one GPU
    scripts/gpu_dev.py multi  scene W H
        rtamd_multi_render with the device list naming GPU 0 one to four times: what the split itself costs
    scripts/gpu_dev.py worst  [--stride N] scene W H ...
        the frame at its full size against the oracle on every N-th pixel (default: about 14 000 pixels), both modes:
        the distribution of the differences and the pixels with the largest ones
    scripts/gpu_dev.py phase  scene W H
        three frames per mode; with the -DRTAMD_PROFILE build (RTAMD_LIB_DIR=ray-tracer_amd/lib_prof RTAMD_PROFILE_DUMP=1)
        the library prints cycles per phase of the main loop for each
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")

SUITE = [("reflect_refract", 400, 200), ("cover", 480, 270), ("cover", 3840, 2160), ("cylinders", 3840, 2160), ("tables", 3840, 2160),
         ("teapot_room", 400, 200), ("teapot_room", 3840, 2160), ("glass_spheres", 7680, 4320), ("groups", 7680, 4320), ("csg", 7680, 4320)]
MODES = {"fast": rt.MODE_FAST, "strict": rt.MODE_STRICT}


def sg_path(name):
    return os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")


def triples(args):
    if len(args) % 3:
        raise SystemExit("scenes come as: name W H")
    return [(args[i], int(args[i + 1]), int(args[i + 2])) for i in range(0, len(args), 3)]


def time_and_hash(ctx, torch, name, W, H, modes, frames, want_hash):
    out = {}
    ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(W, H).flatten())
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    buf8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    for mname in modes:
        ts = []
        for _ in range(frames):
            ctx.reset_stats()
            ctx.render_rows(8, 0, 1, MODES[mname], d_rgb=buf.data_ptr())
            st = ctx.stats()
            ts.append(st.kernel_ms)
        rays = st.primary_rays + st.shadow_rays + st.reflect_rays + st.refract_rays
        rec = {"ms": float(np.median(ts[1:])), "ms_min": float(min(ts[1:])), "rays": int(rays), "containers": int(st.container_passes)}
        line = f"{name} {W}x{H} {mname}: kernel {rec['ms']:.3f} ms (min {rec['ms_min']:.3f}), {rays / 1e6:.1f} Mrays, {rays / rec['ms'] / 1e3:.0f} Mrays/s"
        if want_hash:
            ctx.render_rows(8, 0, 1, MODES[mname], d_rgb=buf.data_ptr(), d_rgb8=buf8.data_ptr())
            torch.cuda.synchronize()
            rec["f64"] = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16]
            rec["u8"] = hashlib.sha256(buf8.cpu().numpy().tobytes()).hexdigest()[:16]
            line += f" f64 {rec['f64']} u8 {rec['u8']}"
        print(line, flush=True)
        out[mname] = rec
    return out


def cmd_time_hash(a, want_hash, todo=None):
    import torch
    ctx = rt.Context(0)
    modes = a.modes.split(",")
    todo = todo or triples(a.scenes)
    res = {}
    for name, W, H in todo:
        res[f"{name} {W}x{H}"] = time_and_hash(ctx, torch, name, W, H, modes, a.frames, want_hash)
    if getattr(a, "out", None):
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    rc = 0
    if getattr(a, "against", None):
        old = json.load(open(a.against))
        for key, per_mode in res.items():
            for m, rec in per_mode.items():
                o = old.get(key, {}).get(m)
                if not o:
                    continue
                same = all(rec.get(k) == o.get(k) for k in ("f64", "u8", "rays", "containers") if k in o and k in rec)
                print(f"{key} {m}: {'same bits' if same else 'DIFFERENT'}; {o['ms']:.3f} -> {rec['ms']:.3f} ms ({100 * (rec['ms'] / o['ms'] - 1):+.1f} %)", flush=True)
                rc |= 0 if same else 1
    return rc


def cmd_check(a):
    import oracle_lib
    scenes = a.scenes or ["default_world", "reflect_refract", "cover", "cylinders", "tables", "glass_spheres", "groups", "puppets", "rr_code",
                          "csg", "teapot_lowres_room", "teapot_room"]
    sizes = {"default_world": (11, 11), "teapot_room": (60, 30), "glass_spheres": (96, 96), "cover": (96, 96)}
    ctx = rt.Context(0)
    rc = 0
    for name in scenes:
        w, h = sizes.get(name, (120, 60))
        ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(w, h).flatten())
        t0 = time.time()
        ref, ref8, ost = oracle_lib.OracleScene(sg_path(name)).render(w, h, threads=8)
        t_or = time.time() - t0
        for mname, tol in (("strict", 1e-10), ("fast", 1e-8)):
            ctx.reset_stats()
            rgb, u8 = ctx.render(w, h, MODES[mname], want_u8=True)
            st = ctx.stats()
            d = np.abs(rgb - ref)
            d8 = np.abs(u8.astype(int) - ref8.astype(int)).max(axis=2)
            exact = (rgb == ref).all(axis=2).mean()
            traced = st.primary_rays + st.shadow_rays + st.reflect_rays + st.refract_rays  # (st.rays also counts container passes)
            ok = np.isfinite(rgb).all() and np.nanmax(d) <= tol and (d8 <= 1).mean() >= 0.999 and traced == ost["rays"]
            rc |= 0 if ok else 1
            print(f"{name:20s} {mname:6s} {w}x{h} {'ok ' if ok else 'BAD'} maxabs {np.nanmax(d):.3e} bitexact {100 * exact:7.3f}% u8>1 {100 * (d8 > 1).mean():.4f}% "
                  f"rays {traced} (oracle {ost['rays']}) kernel {st.kernel_ms:.3f} ms oracle {t_or:.1f}s", flush=True)
    return rc


def cmd_work(a):
    """Tests the traversals execute, per ray (the counting twins of the kernels), and the ray mix."""
    import torch
    ctx = rt.Context(0)
    for name, W, H in triples(a.scenes):
        ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(W, H).flatten())
        buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        ctx.reset_stats()
        ctx.count_work(True)
        ctx.render_rows(8, 0, 1, rt.MODE_FAST, d_rgb=buf.data_ptr())
        st = ctx.stats()
        ctx.count_work(False)
        rays = st.rays
        per = {n[6:]: getattr(st, n) / rays for n, _ in st._fields_ if n.startswith("tests_") and getattr(st, n)}
        print(f"{name} {W}x{H}: {rays / 1e6:.1f} M traversals (primary {st.primary_rays / rays:.2f}, shadow {st.shadow_rays / rays:.2f}, reflect "
              f"{st.reflect_rays / rays:.2f}, refract {st.refract_rays / rays:.2f}, containers {st.container_passes / rays:.2f}); lane-tests per traversal: "
              + ", ".join(f"{k} {v:.2f}" for k, v in per.items()), flush=True)
    return 0


def cmd_host(a):
    name, W, H = triples(a.scenes)[0]
    sc = rt.Scene.from_sg(sg_path(name)).set_resolution(W, H)
    ctx = rt.Context(0)
    ctx.upload(sc.flatten())
    for label, kw in (("f64 + u8, pageable", dict(want_u8=True)), ("f64 + u8, page-locked", dict(want_u8=True, pinned=True)),
                      ("u8 only, page-locked", dict(want_u8=True, want_f64=False, pinned=True))):
        try:
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                ctx.render(W, H, rt.MODE_FAST, **kw)
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{name} {W}x{H} rtamd_render {label}: {np.median(ts[1:]):.2f} ms per call", flush=True)
        except TypeError as e:
            print(f"{label}: not offered by this build ({e})")
    return 0


def cmd_parts(a):
    import torch
    name, W, H = triples(a.scenes)[0]
    ctx = rt.Context(0)
    ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(W, H).flatten())
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    base = None
    for n in (1, 2, 4, 8):
        times = []
        for p in range(n):
            best = 1e9
            for _ in range(4):
                ctx.reset_stats()
                ctx.render_rows(8, p, n, rt.MODE_FAST, d_rgb=buf.data_ptr())
                best = min(best, ctx.stats().kernel_ms)
            times.append(best)
        base = base or times[0]
        print(f"{name} {W}x{H} parts {n}: kernel ms per part " + " ".join(f"{t:.3f}" for t in times) +
              f"  -> slowest {max(times):.3f}, speed-up over one part {base / max(times):.2f}", flush=True)
    return 0


def cmd_multi(a):
    name, W, H = triples(a.scenes)[0]
    flat = rt.Scene.from_sg(sg_path(name)).set_resolution(W, H).flatten()
    for n in (1, 2, 4):
        m = rt.MultiContext([0] * n).upload(flat)
        for k0 in (1, 2):
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                m.render_on_device(rt.MODE_FAST, 8, k0, 1)
                ts.append((time.perf_counter() - t0) * 1e3)
            _, r, p = m.stats()
            print(f"{name} {W}x{H} contexts {n} shares {k0}:1  wall {np.median(ts[1:]):.3f} ms  render/rank {np.round(r, 3)}  place/rank {np.round(p, 3)}", flush=True)
        m.close()
    return 0


def cmd_worst(a):
    import oracle_lib
    ctx = rt.Context(0)
    for name, W, H in triples(a.scenes):
        stride = a.stride or max(1, W * H // 14000)
        pix = np.arange(stride // 2, W * H, stride, dtype=np.int64)
        ref, ref8, _ = oracle_lib.OracleScene(sg_path(name)).render_pixels(pix, W, H, threads=16)
        ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(W, H).flatten())
        for mname, mode in MODES.items():
            rgb, u8 = ctx.render(W, H, mode, want_u8=True)
            d = np.abs(rgb.reshape(-1, 3)[pix] - ref).max(axis=1)
            d8 = np.abs(u8.reshape(-1, 3)[pix].astype(int) - ref8.astype(int)).max(axis=1)
            top = np.argsort(d)[::-1][:6]
            counts = " ".join(f">{t:g}: {int((d > t).sum())}" for t in (1e-14, 1e-12, 1e-10, 1e-9, 1e-8))
            print(f"{name} {W}x{H} {mname}: {pix.size} pixels, max {d.max():.3e}, {counts}, 8-bit off {int((d8 > 0).sum())}; worst "
                  + " ".join(f"({int(pix[i] % W)},{int(pix[i] // W)}: {d[i]:.2e})" for i in top), flush=True)
    return 0


def cmd_phase(a):
    import torch
    ctx = rt.Context(0)
    for name, W, H in triples(a.scenes):
        ctx.upload(rt.Scene.from_sg(sg_path(name)).set_resolution(W, H).flatten())
        buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        for mname, mode in MODES.items():
            for _ in range(3):
                ctx.reset_stats()
                ctx.render_rows(8, 0, 1, mode, d_rgb=buf.data_ptr())
                st = ctx.stats()  # (the profile build prints its phase timers here)
            print(f"{name} {W}x{H} {mname}: kernel {st.kernel_ms:.3f} ms, rays {st.rays}, containers {st.container_passes}, shade points {st.shade_points}", flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    for c in ("time", "hash", "suite"):
        p = sub.add_parser(c)
        p.add_argument("--modes", default="fast,strict" if c != "suite" else "fast")
        p.add_argument("--frames", type=int, default=6)
        p.add_argument("--out")
        p.add_argument("--against")
        p.add_argument("scenes", nargs="*")
    p = sub.add_parser("check")
    p.add_argument("scenes", nargs="*")
    p = sub.add_parser("host")
    p.add_argument("scenes", nargs="*")
    for c in ("work", "parts", "multi", "phase", "worst"):
        p = sub.add_parser(c)
        p.add_argument("scenes", nargs="*")
        if c == "worst":
            p.add_argument("--stride", type=int, default=0)
    a = ap.parse_args()
    if a.cmd == "time":
        return cmd_time_hash(a, False)
    if a.cmd == "hash":
        return cmd_time_hash(a, True)
    if a.cmd == "suite":
        return cmd_time_hash(a, True, SUITE)
    if a.cmd == "check":
        return cmd_check(a)
    if a.cmd in ("work", "parts", "multi", "phase", "worst"):
        return {"work": cmd_work, "parts": cmd_parts, "multi": cmd_multi, "phase": cmd_phase, "worst": cmd_worst}[a.cmd](a)
    return cmd_host(a)


if __name__ == "__main__":
    sys.exit(main())
