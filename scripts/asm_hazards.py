#!/usr/bin/env python3
"""Inline-assembly instructions that read the result of a transcendental instruction too early (no GPU needed).

    scripts/asm_hazards.py [fast|strict|teams ...]

On gfx950 a vector instruction that reads what v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos wrote needs a
wait state in between.  The compiler inserts it for the instructions it selects, but it does not look inside inline
assembly: `v_max_f64` written as asm right behind a `v_rcp_f64` of its operand reads a stale register.  This compiles
the kernels to assembly and reports every inline-asm instruction within two instructions of a transcendental whose
destination it reads.  Exit status 1 when any is found."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f64|f32|f16)")


def regs(tok):
    m = re.match(r"[|-]*v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[|-]*v(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def main():
    modes = sys.argv[1:] or ["fast", "strict", "teams"]
    bad = 0
    for mode in modes:
        asm = f"/tmp/rtamd_hazard_{mode}.s"
        contract = "off" if mode == "strict" else "fast"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-ffp-contract={contract}", "--offload-device-only",
                        "-S", os.path.join(ROOT, "ray-tracer_amd", "csrc", f"rtamd_kernels_{mode}.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
        recent, in_asm, kernel = [], False, "?"  # recent: (age, destination registers) of transcendental instructions
        for ln in open(asm):
            s = ln.strip()
            m = re.match(r"(_Z\w+):", s)
            if m:
                kernel = m.group(1)
            if "#ASMSTART" in s:
                in_asm = True
                continue
            if "#ASMEND" in s:
                in_asm = False
                continue
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            ops = s.replace(",", " ").split()
            if in_asm and len(ops) > 2:
                src = set().union(*[regs(t) for t in ops[2:]])
                for age, dst in recent:
                    if age < 2 and src & dst:
                        print(f"{mode}: {kernel}: `{s}` reads v{sorted(src & dst)} {age + 1} instruction(s) after a transcendental wrote it")
                        bad += 1
            recent = [(a + 1, d) for a, d in recent if a + 1 < 3]
            if TRANS.match(ops[0]) and len(ops) > 1:
                recent.append((0, regs(ops[1])))
    print(f"{bad} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
