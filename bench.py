#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) and frame time of the GPU render loop.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one frame of the workload (default: scenes/cover.yaml of the reference at 3840x2160, recursion
budget 4), rendered with the scene already resident in HBM: every rank renders its interleaved row tiles
(tile i -> part i % n_parts; a rank holds one part, rank 0 possibly more: --rank0-share), the float64 frame
(the reference's Canvas) is gathered on rank 0 over RCCL and put back in row order.  Rank 0 prints ONE JSON line.  `value` = rays traced by all ranks per second
(primary, reflection, refraction and shadow rays; the n1/n2 container passes are not counted).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X: 256 CU x 64 FP64 FMA lanes/clk x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0

# algorithmic flops of one leaf test, ray transform included (33 flops; a plane needs only the y row)
LEAF_FLOPS = {"sphere": 33 + 29, "plane": 14, "cube": 33 + 23, "cylinder_open": 33 + 34, "cylinder_closed": 33 + 55,
              "cone_open": 33 + 40, "cone_closed": 33 + 63, "tri": 51}
SHADE_POINT_FLOPS = 100
LIGHT_SAMPLE_FLOPS = 60
# flops of the tests the kernel executes, per lane-test counted by rtamd_stats (fma = 2): what the formulation on the
# device costs, not what the reference's costs -- an axis-aligned cube is two world-space slabs (6 fma, min, max, min,
# 2 compares) and, for the rays those do not reject, a third (3 fma, min, max, 4 compares); a triangle is the first half
# of Moeller-Trumbore (cross, dot, reciprocal, sub, dot, compares) and for the survivors the second
EXECUTED_FLOPS = {"tests_sphere": 33 + 29, "tests_plane": 14, "tests_cube": 33 + 23, "tests_slab2": 17, "tests_slab3": 12,
                  "tests_cylinder": 33 + 34, "tests_cone": 33 + 40, "tests_caps": 21, "tests_tri_first": 30, "tests_tri_second": 21,
                  "tests_box": 22, "tests_run_transform": 33}


def source_stamp():
    """Hash of the sources the render kernels are built from: a measured-traffic file is only quoted for the kernel
    it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rtamd_kernel.inc", "rtamd_device.h", "rtamd_api.hip"):
        with open(os.path.join(ROOT, "ray-tracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def flops_per_ray(sg_path):
    """Sum over the leaves of the scene graph; a run of triangles pays the ray transform once."""
    total, in_run = 0, False
    with open(sg_path) as f:
        for line in f:
            t = line.split()
            if not t or t[0] != "node":
                continue
            kind = t[1]
            if kind in ("tri", "stri"):
                total += LEAF_FLOPS["tri"] + (0 if in_run else 33)
                in_run = True
                continue
            in_run = False
            if kind in ("sphere", "plane", "cube"):
                total += LEAF_FLOPS[kind]
            elif kind in ("cylinder", "cone"):
                total += LEAF_FLOPS[f"{kind}_{'closed' if t[10] == '1' else 'open'}"]
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cover")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2],
                    help="frames rendering at once on a GPU (0: 1 on one GPU, 2 on several -- the tail of a frame, a few "
                         "long chains of dependent rays, then overlaps the start of the next; kernel durations, and with "
                         "them roofline.achieved, then include the time a kernel shares the GPU)")
    ap.add_argument("--rank0-share", default="auto",
                    help="with several GPUs: 'equal', 'K0:K' (rank 0 holds K0 interleaved parts of the frame, every other rank K), "
                         "or 'auto': a few such splits are tried during warm-up and the fastest is timed -- rank 0's rows do not "
                         "travel, every other rank's cross one xGMI link")
    ap.add_argument("--gather", default="f64", choices=["f64", "u8"],
                    help="what the ranks render and rank 0 assembles: f64 = Canvas.pixels, 24 bytes per pixel (the headline); u8 = the 8-bit "
                         "channels an image file holds, 3 bytes per pixel -- an eighth of the bytes over every xGMI link")
    ap.add_argument("--brute-force", action="store_true",
                    help="diagnostic: no bounds records, no re-cut of triangle pools, every run of triangles tested for every ray "
                         "(what the reference's World::intersect does), so that the algorithmic roofline fraction is a utilisation")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="with one GPU: still go through the process group, the gather and the two frames in flight of the "
                         "multi-GPU path (a rehearsal of that code on a one-GPU box; the figures are not comparable)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.rehearse_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # a plain `python bench.py --rehearse-collective`
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.brute_force:
        os.environ["RTAMD_BRUTE_FORCE"] = "1"  # read by the library at upload and at launch
    rt = importlib.import_module("ray-tracer_amd")
    rows_mod = importlib.import_module("ray-tracer_amd.rows")
    mode = rt.MODE_FAST if a.mode == "fast" else rt.MODE_STRICT
    sg = os.path.join(ROOT, "tests", "golden", "scenes", a.scene + ".sg")
    scene = rt.Scene.from_sg(sg).set_resolution(a.width, a.height)
    in_flight = a.in_flight or (2 if use_dist else 1)
    flat = scene.flatten()
    ctxs = [rt.Context(local_rank) for _ in range(in_flight)]  # a context owns the scratch of one render under way
    for c in ctxs:
        c.upload(flat)
    ctx = ctxs[0]
    W, H, R = a.width, a.height, a.tile_rows

    frame_dtype = torch.float64 if a.gather == "f64" else torch.uint8

    def render_into(c, buf, stream, m=mode):
        if a.gather == "f64":
            c.render_rows(R, fg.part, fg.n_parts, m, d_rgb=buf.data_ptr(), stream=stream, span=fg.span)
        else:
            c.render_rows(R, fg.part, fg.n_parts, m, d_rgb8=buf.data_ptr(), stream=stream, span=fg.span)

    def make_gather(k0, k):
        g = rows_mod.FrameGather(W, H, R, rank, world, dev, collective=use_dist, shares=rows_mod.shares_for(world, k0, k), dtype=frame_dtype)
        if ctx.part_rows(R, g.part, g.n_parts, g.span) != g.n_rows:
            raise SystemExit("row partition of the library and of rows.py disagree")
        return g

    streams = [torch.cuda.current_stream()] if in_flight == 1 else [torch.cuda.Stream() for _ in range(in_flight)]
    fg = None

    def step(k, events=None):  # frame k's rows travel to rank 0 while the next frames render (rows.FrameGather)
        j = k % in_flight
        with torch.cuda.stream(streams[j]):
            buf = fg.begin()
            if events:
                events[0].record()  # events bracket the render kernel itself, on the stream it is launched on
            render_into(ctxs[j], buf, streams[j].cuda_stream)
            if events:
                events[1].record()
            fg.submit()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # How the frame is shared out (untimed, part of the warm-up): every split is run for a few frames, pipelined as
    # the timed steps are, and the fastest is kept.  All ranks see the same (max-reduced) times, so they agree.
    trials = {}
    if not use_dist or a.rank0_share == "equal":
        share = (1, 1)
    elif a.rank0_share != "auto":
        share = tuple(int(x) for x in a.rank0_share.split(":"))
    else:
        for cand in ((1, 1), (1, 1), (3, 2), (2, 1), (5, 2), (3, 1)):  # (the first pass warms the GPU up; its time is replaced)
            fg = make_gather(*cand)
            for k in range(2 * in_flight):
                step(k)
            fg.finish()
            sync()
            t0 = time.perf_counter()
            for k in range(8):
                step(k)
            fg.finish()
            sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trials["%d:%d" % cand] = float(t[0]) / 8 * 1e3
            fg = None
        share = rows_mod.pick_share(trials)
    fg = make_gather(*share)
    for k in range(max(a.warmup, in_flight if a.warmup else 0)):
        step(k)
    fg.finish()
    sync()
    for c in ctxs:
        c.reset_stats()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k, ev[k])
    frame = fg.finish()  # every frame is complete on rank 0 inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    sts = [c.stats() for c in ctxs]
    step_kernel_ms = [s.elapsed_time(e) for s, e in ev]  # this rank's render kernel, step by step
    kernel_ms = float(np.mean(step_kernel_ms))
    my_kernel_ms = kernel_ms
    counts = torch.tensor([sum(st.primary_rays + st.shadow_rays + st.reflect_rays + st.refract_rays for st in sts),
                           sum(st.shade_points for st in sts), sum(st.light_samples for st in sts),
                           sum(st.container_passes for st in sts)], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(counts)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(tmax[0]), float(tmax[1])

    # Outside the timed region: a frame's two phases one after the other (nothing pipelined), slowest rank of each --
    # the render kernel alone on its GPU, then the gather of the rows to rank 0 and their de-interleaving there.
    cur = torch.cuda.current_stream()
    phase = []
    for _ in range(3):
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        buf = fg.begin()
        e0.record()
        render_into(ctx, buf, cur.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fg.submit()
        fg.finish()
        torch.cuda.synchronize()
        phase.append((e0.elapsed_time(e1), (time.perf_counter() - t1) * 1e3))
    phase = torch.tensor(np.median(np.array(phase), axis=0), dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(phase, op=dist.ReduceOp.MAX)
    render_alone_ms, gather_ms = float(phase[0]), float(phase[1])
    # per rank, for a run on several GPUs to be readable afterwards: its kernel in the timed steps, its kernel alone, its rows
    per_rank = torch.tensor([my_kernel_ms, float(np.median(np.array([p[0] for p in [phase.tolist()]]))), float(fg.n_rows)], dtype=torch.float64, device=dev)
    per_rank_all = [torch.zeros_like(per_rank) for _ in range(world)] if use_dist else [per_rank]
    if use_dist:
        dist.all_gather(per_rank_all, per_rank)
    per_rank_all = [t.tolist() for t in per_rank_all]
    rays, shade_points, light_samples, containers = [float(x) / a.steps for x in counts]
    # One more frame (untimed) with the kernels that count the tests their traversals execute: the timed frames use
    # the kernels without the counters, which cost 5-15 %.  Same scene, same rows, so the counts are those of a timed frame.
    spare = torch.empty((max(fg.n_rows, 1), W, 3), dtype=frame_dtype, device=dev)
    ctx.reset_stats()
    ctx.count_work(True)
    render_into(ctx, spare, cur.cuda_stream)
    torch.cuda.synchronize()
    st = ctx.stats()
    ctx.count_work(False)
    counting_kernel_ms = st.kernel_ms
    tests = torch.tensor([float(getattr(st, n)) for n in EXECUTED_FLOPS], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tests)
    lane_tests = {n: float(x) for n, x in zip(EXECUTED_FLOPS, tests)}
    # the reference's own arithmetic order, once, for the record (not the headline)
    strict_ms = None
    if world == 1 and not use_dist and a.mode == "fast":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = []
        for _ in range(3):
            e0.record()
            render_into(ctx, spare, cur.cuda_stream, rt.MODE_STRICT)
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
        strict_ms = float(np.median(t))
    del spare

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        fpr = flops_per_ray(sg)
        shading_flops = shade_points * SHADE_POINT_FLOPS + light_samples * LIGHT_SAMPLE_FLOPS
        flops = rays * fpr + shading_flops  # (the n1/n2 container passes are this design's own: not algorithmic rays)
        achieved = flops / (kernel_ms * 1e-3) / 1e12  # per rank-kernel: flops of all ranks / slowest kernel
        executed_flops = sum(lane_tests[n] * f for n, f in EXECUTED_FLOPS.items()) + shading_flops
        executed = executed_flops / (kernel_ms * 1e-3) / 1e12
        # HBM bytes per launch from the PMC counters: measured by scripts/collect_profiles.sh in rocprofv3 runs of their
        # own; quoted only if that file was measured on the kernel sources of this run
        traffic = None
        pmc = os.path.join(ROOT, "profiles", f"pmc_hbm_{a.scene}_{W}x{H}_{a.mode}.json")
        if world == 1 and not use_dist and not a.brute_force and os.path.exists(pmc):
            rec = json.load(open(pmc))
            if rec.get("source_stamp") == source_stamp():
                traffic = rec.get("hbm_bytes_per_launch")
        out = {
            "metric": "Mrays/sec (primary+secondary) & frame time, cover.yaml 4K, 1/2/4/8 MI355X",
            "value": rays / (ms_per_step * 1e-3) / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": f"the reference's own scene ({a.scene}, committed as a scene graph under tests/golden/scenes/); no dataset, no random input",
            "config": {"workload": f"{a.scene} scene graph at {W}x{H}, recursion budget 4, {a.mode} arithmetic",
                       "rays_per_frame": rays, "pixels": W * H, "parallelism": f"rows{world}",
                       "row_tiles": f"cyclic, {R} rows" + ("" if world == 1 else f"; rank 0 holds {share[0]} of {fg.n_parts} interleaved parts, "
                                                           f"every other rank {share[1]}"),
                       **({"share_trials_ms_per_frame": trials} if trials else {}),
                       "gather": {"what": "f64 pixels (Canvas.pixels), 24 bytes per pixel" if a.gather == "f64" else "8-bit channels, 3 bytes per pixel",
                                  "bytes_sent_per_rank_per_frame": fg.sent_bytes, "bytes_per_frame": W * H * 3 * (8 if a.gather == "f64" else 1)},
                       "per_rank": [{"rank": r, "rows": int(v[2]), "kernel_ms_timed_steps": v[0], "kernel_ms_alone": v[1]} for r, v in enumerate(per_rank_all)],
                       "frames_in_flight": in_flight, **({"rehearsal": "collective path on one rank"} if use_dist and world == 1 else {}),
                       **({"brute_force": "no bounds records, every triangle of every run tested for every ray"} if a.brute_force else {}),
                       **({"strict_arithmetic_kernel_ms": strict_ms} if strict_ms is not None else {})},
            "phases": {"render_ms": render_alone_ms, "gather_ms": gather_ms,
                       "note": "one frame at a time outside the timed region, slowest rank: render kernel, then gather to rank 0 + de-interleave"},
            "roofline": {"bound": "fp64_valu", "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                         "frac": achieved / (FP64_VALU_PEAK_TFLOPS * world), "traffic": traffic,
                         "kernel_ms": kernel_ms,
                         "kernel_ms_steps": {"min": float(np.min(step_kernel_ms)), "median": float(np.median(step_kernel_ms)),
                                             "max": float(np.max(step_kernel_ms)), "n": len(step_kernel_ms), "of": "rank 0"},
                         "flops_per_ray": fpr, "algorithmic_flops_per_frame": flops,
                         "executed": executed, "executed_frac": executed / (FP64_VALU_PEAK_TFLOPS * world),
                         "executed_flops_per_frame": executed_flops, "executed_lane_tests_per_frame": lane_tests,
                         "counting_kernel_ms": counting_kernel_ms,
                         "hbm_algorithmic_bytes_per_frame": W * H * (24 if a.gather == "f64" else 3),
                         "hbm_frac": W * H * (24 if a.gather == "f64" else 3) / (kernel_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)},
        }
        if fpr > 100000 or achieved > FP64_VALU_PEAK_TFLOPS * world:
            # the brute-force constant counts leaf tests that the bounds records (triangles, groups of bounded leaves) skip
            out["roofline"]["note"] = ("algorithmic flops count every leaf for every ray; bounds records skip most of those tests here, so frac "
                                       "is not a utilisation: executed_frac (flops of the tests the kernel counted) is, and --brute-force gives the "
                                       "algorithmic figure with nothing skipped")
        if world == 1 and a.cpu_seconds > 0 and a.gather == "f64":
            out["cpu_baseline"], out["parity"] = cpu_baseline(a, sg, frame, rt)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_model():
    """Model name of the host CPU the baseline ran on (/proc/cpuinfo)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(a, sg, frame, rt):
    """The oracle in `faithful` mode (per-call inverses, per-light duplicated recursion: the work the
    reference does), one thread, on a regular subsample of the frame's pixels sized to the time budget."""
    import oracle_lib
    W, H = a.width, a.height
    o = oracle_lib.OracleScene(sg)
    npx = W * H
    probe = np.linspace(0, npx - 1, 96).astype(np.int64)
    t0 = time.perf_counter()
    o.render_pixels(probe, W, H, faithful=True)
    per_px = (time.perf_counter() - t0) / probe.size
    n = int(max(256, min(npx, a.cpu_seconds / max(per_px, 1e-9))))
    stride = max(1, npx // n)
    pix = np.arange(stride // 2, npx, stride, dtype=np.int64)
    t0 = time.perf_counter()
    rgb, u8, st = o.render_pixels(pix, W, H, faithful=True)
    dt = time.perf_counter() - t0
    # the same oracle with cached inverses and children evaluated once per shade point, on the same sample: how much
    # of the speed-up is owed to the reference's redundant work rather than to the GPU
    sub = pix
    t0 = time.perf_counter()
    _, _, st_c = o.render_pixels(sub, W, H, faithful=False)
    dt_c = time.perf_counter() - t0
    import torch
    gpu = frame.reshape(-1, 3)[torch.from_numpy(pix).to(frame.device)].cpu().numpy()
    g8 = rt.quantize(gpu)
    d8 = np.abs(g8.astype(int) - u8.astype(int)).max(axis=1)
    base = {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"{pix.size} pixels (every {stride}-th of the {W}x{H} frame), oracle in faithful mode, 1 thread, {dt:.1f} s",
            "rays_reference_equivalent_per_frame": st["rays"] / pix.size * npx,
            "est_frame_seconds": dt / pix.size * npx, "host_cpus": os.cpu_count(), "host_cpu_model": cpu_model(),
            "oracle_build": "g++ -O3 -ffp-contract=off (oracle/Makefile)",
            "cached_mode": {"value": st_c["rays"] / dt_c / 1e6, "unit": "Mrays/s", "est_frame_seconds": dt_c / sub.size * npx,
                            "sample": f"{sub.size} pixels, {dt_c:.1f} s"}}
    parity = {"pixels_checked": int(pix.size), "max_abs_f64": float(np.abs(gpu - rgb).max()),
              "u8_within_1_lsb": float((d8 <= 1).mean()), "u8_equal": float((d8 == 0).mean())}
    return base, parity


if __name__ == "__main__":
    main()
