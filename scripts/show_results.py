#!/usr/bin/env python3
"""Development aid: the table rows of profiles/r02_configs.jsonl and the headline figures of the profile summaries."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
names = ['1', '2', '3a', '3b', '4', '5a', '5b', '5c', '4 brute force']
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", f"{tag}_configs.jsonl"))]
for n, d in zip(names, rows):
    r = d['roofline']; c = d.get('cpu_baseline') or {}; p = d.get('parity') or {}
    wl = d['config']['workload'].split(' scene graph at ')
    print(f"| {n} | {wl[0]} {wl[1].split(',')[0]} | {d['ms_per_step']:.2f} | {d['config'].get('strict_arithmetic_kernel_ms', 0):.2f} | {d['value']:.0f} | {r['frac']:.2f} | {r['executed_frac']:.2f} | {c.get('est_frame_seconds', '-')} | {p.get('pixels_checked')} {p.get('max_abs_f64')} {p.get('u8_equal')} {p.get('u8_within_1_lsb')} | traffic {r.get('traffic')}")
for t in (tag, tag + '_teapot', tag + '_csg', tag + '_small'):
    d = json.load(open(os.path.join(ROOT, "profiles", f"{t}_pmc_summary.json")))
    c = d['counters']
    print(t, d['kernel'], round(d['kernel_avg_ns'] / 1e6, 3), 'ms; VALU', round(c['SQ_INSTS_VALU']['mean_per_dispatch'] / 1e6), 'M SALU', round(c['SQ_INSTS_SALU']['mean_per_dispatch'] / 1e6),
          'M; fp64 share', round(d['fp64_issued']['share_of_valu_instructions'], 3), 'frac', round(d['fp64_issued']['frac_of_78.6_tflops'], 3), 'hbm', round(d['hbm']['hbm_bytes_per_launch'] / 1e9, 3),
          'GB w', round(d['hbm']['write_bytes'] / 1e9, 3), 'r', round(d['hbm']['read_bytes_corrected'] / 1e9, 3), d['source_stamp'])
import bench
print("tree:", bench.source_stamp())
