#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by scripts/collect_profiles.sh) into profiles/:
  <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats table, unchanged
  <tag>_pmc_summary.json     per-dispatch averages of every collected counter for the render kernel
  pmc_hbm_<scene>_<W>x<H>_<mode>.json   HBM bytes per launch, read by bench.py for roofline.traffic
usage: scripts/summarize_profiles.py <tag> <scene> <W> <H> <mode>
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, scene, W, H, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def timed_kernel(name):
    """The kernel the timed frames of bench.py run: the arithmetic mode asked for, not its counting twin (bench.py also
    renders one frame with the counting kernel and, in fast mode, three with the strict one)."""
    base = name.split("(")[0]
    prefixes = (f"rtamd_render_{mode}_",) + (("rtamd_render_teams_", "rtamd_render_teams3_") if mode == "fast" else ())  # (small frames: the kernels with teams)
    return base.startswith(prefixes) and not base.endswith("_count")


stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
kernel_avg_ns = None
if stats:
    shutil.copyfile(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for row in csv.DictReader(open(stats[0])):
        if timed_kernel(row["Name"]):
            kernel_avg_ns = float(row["AverageNs"])
            kernel_name = row["Name"]

counters = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for row in csv.DictReader(open(f)):
        if not timed_kernel(row["Kernel_Name"]):
            continue
        per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, by_dispatch in per.items():
        vals = list(by_dispatch.values())
        counters[name] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}

sys.path.insert(0, ROOT)
import bench  # source_stamp(): which kernel sources these counters belong to

summary = {"workload": f"{scene} {W}x{H} {mode}", "kernel": kernel_name if kernel_avg_ns else None, "kernel_avg_ns": kernel_avg_ns, "counters": counters, "source_stamp": bench.source_stamp()}
f64 = [counters.get(f"SQ_INSTS_VALU_{k}_F64", {}).get("mean_per_dispatch") for k in ("FMA", "MUL", "ADD")]
if None not in f64 and kernel_avg_ns:
    # wave-instructions x 64 lanes, fma = 2 flops: what the vector units issued in FP64 arithmetic (an upper bound on useful
    # work: idle lanes of a wave and lanes a test was not meant for are counted too)
    issued = (2 * f64[0] + f64[1] + f64[2]) * 64
    summary["fp64_issued"] = {"flops_per_launch": issued, "tflops": issued / kernel_avg_ns / 1e3,
                              "frac_of_78.6_tflops": issued / kernel_avg_ns / 1e3 / 78.6,
                              "share_of_valu_instructions": sum(f64) / counters["SQ_INSTS_VALU"]["mean_per_dispatch"]
                              if "SQ_INSTS_VALU" in counters else None}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch_kb, write_kb = counters["FETCH_SIZE"]["mean_per_dispatch"], counters["WRITE_SIZE"]["mean_per_dispatch"]
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes,
    # so the read side is doubled (an upper bound for narrow reads).
    hbm = {"fetch_size_kib": fetch_kb, "write_size_kib": write_kb, "read_bytes_corrected": 2 * fetch_kb * 1024,
           "write_bytes": write_kb * 1024, "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
           "algorithmic_bytes_per_launch": W * H * 24, "source": f"profiles/{tag}_pmc_summary.json",
           "source_stamp": bench.source_stamp()}
    summary["hbm"] = hbm
    json.dump(hbm, open(os.path.join(dst, f"pmc_hbm_{scene}_{W}x{H}_{mode}.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
