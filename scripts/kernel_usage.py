#!/usr/bin/env python3
"""Registers, spills, scratch, occupancy and LDS per kernel, from the compiler's own report:
    scripts/kernel_usage.py [fast|strict|teams]   (compiles the kernels of that arithmetic mode once more, about a minute)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
contract = "off" if mode == "strict" else "fast"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-ffp-contract={contract}", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.join(ROOT, "ray-tracer_amd", "csrc", f"rtamd_kernels_{mode}.hip"), "-o", "/dev/null"]
text = subprocess.run(cmd, capture_output=True, text=True).stderr
keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"SGPRs"), ("scratch B/lane", r"ScratchSize \[bytes/lane\]"), ("SGPR spills", r"SGPRs Spill"),
        ("VGPR spills", r"VGPRs Spill"), ("waves/SIMD", r"Occupancy \[waves/SIMD\]"), ("LDS B/block", r"LDS Size \[bytes/block\]")]
print("kernel".ljust(56) + "".join(k.rjust(16) for k, _ in keys))
for block in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
    name = block.split()[0]
    vals = []
    for _, pat in keys:
        m = re.search(pat + r": (\d+)", block)
        vals.append(m.group(1) if m else "?")
    print(name[:55].ljust(56) + "".join(v.rjust(16) for v in vals))
