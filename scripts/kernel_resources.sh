#!/bin/bash
# Registers, spills and scratch of every render kernel (no GPU needed): scripts/kernel_resources.sh [fast|strict|levels|teams|teams3] [extra hipcc flags]
mode=${1:-fast}; shift || true
cd "$(dirname "$0")/../ray-tracer_amd"
contract=$( { [ "$mode" = strict ] || [ "$mode" = levels ]; } && echo off || echo fast)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=$contract "$@" -Rpass-analysis=kernel-resource-usage \
    -c csrc/rtamd_kernels_$mode.hip -o /tmp/rtamd_res_$mode.o 2>&1 |
python3 -c '
import re, sys
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark:\s+([\w \[\]/]+): (\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": re.sub(r"^_Z\d+|6DScene5DWork$", "", v)}; rows.append(cur)
    else:
        cur[k] = v
print("%-44s %5s %5s %6s %6s %8s %4s" % ("kernel", "SGPR", "VGPR", "sSpill", "vSpill", "scratch", "occ"))
for r in rows:
    print("%-44s %5s %5s %6s %6s %8s %4s" % (r["name"], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("SGPRs Spill", r.get("Spilled SGPRs")), r.get("VGPRs Spill", r.get("Spilled VGPRs")), r.get("ScratchSize [bytes/lane]", r.get("ScratchSize")), r.get("Occupancy [waves/SIMD]", r.get("Occupancy"))))
'
