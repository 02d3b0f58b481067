#!/usr/bin/env python3
"""Static instruction table of one render kernel, by section of csrc/rtamd_kernel.inc and by instruction class.

    scripts/isa_audit.py [--mode fast|strict|levels|teams|teams3] [--kernel rtamd_render_fast_plain_bounds_slab] [--asm file.s] [--loops]

Compiles the kernel source for gfx950 with line tables (no GPU needed), walks the assembly of the one kernel and
books every instruction under the source line its `.loc` names (the innermost inlined line) and under a class:
FP64 arithmetic (add/mul/fma), FP64 min/max, other FP64 (rcp, rsq, conversions, ...), compares, selects, moves,
lane traffic (v_readlane / v_writelane / v_readfirstlane / DPP: what scalar-register spills turn into), other vector
ALU, LDS, vector memory (scratch = spills of vector registers), scalar memory, scalar ALU, waits and branches.
The counts are static (instructions in the code object, not executed ones); `--loops` adds the table per innermost
loop label so that the loops of the traversal can be read one by one.  The sections are line ranges of the kernel
source found by their marker comments, so the table follows the file as it changes.
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ray-tracer_amd", "csrc", "rtamd_kernel.inc")

CLASSES = ["f64_arith", "f64_minmax", "f64_other", "cmp", "select", "mov", "lane", "valu_other", "lds", "vmem", "scratch",
           "smem", "salu", "wait", "branch"]


def classify(op):
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64", "v_pk_fma_f64")):
        return "f64_arith"
    if op.startswith(("v_max_f64", "v_min_f64")):
        return "f64_minmax"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "select"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mov_b32_dpp", "ds_swizzle", "ds_bpermute", "ds_permute")) or "dpp" in op:
        return "lane"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "mov"
    if op.startswith("v_") and "f64" in op:
        return "f64_other"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_atomic", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_getpc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "valu_other"


def sections():
    """(first line, name) of every section of render_body, from marker text in the source."""
    marks = [
        ("template <int VARIANT, bool SLAB_", "prologue"),
        ("refill idle lanes", "refill"),
        ("one traversal for every busy lane", "traversal set-up"),
        ("auto traverse = [&](auto simple_tag)", "sinks"),
        ("for (int vi = 0; vi < nv; vi++)", "record load"),
        ("if (BOUNDS) {", "bounds loop"),
        ("if (SLAB && SIMPLE && all_slab", "cube run (simple)"),
        ("if (SLAB && !SIMPLE && all_slab", "cube run (class-major)"),
        ("auto f64 = [&](int i) {  // i-th double", "general leaf"),
        ("if (cont_mask != 0)", "after traversal: n1/n2"),
        ("what the traversal was for", "hit / prepare_computations"),
        ("PROF_MARK(5);  // colour / container results", "shade set-up"),
        ("const int light_rounds", "lighting"),
        ("PROF_MARK(7);  // lighting", "return / next ray"),
        ("counters: wave -> block", "epilogue"),
    ]
    lines = open(SRC).read().split("\n")
    out, at = [], 0
    for text, name in marks:
        for i in range(at, len(lines)):
            if text in lines[i]:
                out.append((i + 1, name))
                at = i + 1
                break
        else:
            raise SystemExit(f"marker not found in rtamd_kernel.inc: {text!r}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="fast", choices=["fast", "strict", "levels", "teams", "teams3"])
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--asm", default=None, help="assembly made with -gline-tables-only (skips the compile)")
    ap.add_argument("--loops", action="store_true")
    ap.add_argument("--flags", default="", help="extra hipcc flags")
    a = ap.parse_args()
    kernel = a.kernel or f"rtamd_render_{a.mode}_plain_bounds_slab"
    asm = a.asm
    if asm is None:
        asm = f"/tmp/rtamd_audit_{a.mode}.s"
        contract = "off" if a.mode in ("strict", "levels") else "fast"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-ffp-contract={contract}", "-gline-tables-only",
               "--offload-device-only", "-S", os.path.join(ROOT, "ray-tracer_amd", "csrc", f"rtamd_kernels_{a.mode}.hip"), "-o", asm] + a.flags.split()
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    secs = sections()
    inc_file = None  # file number of rtamd_kernel.inc in this function's .loc/.file table
    files = {}
    text = open(asm).read().split("\n")
    for ln in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
    inc_ids = {k for k, v in files.items() if v.endswith("rtamd_kernel.inc")}
    start = None
    for i, ln in enumerate(text):
        if re.match(rf"_Z\d+{kernel}6DScene5DWork:", ln):
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel {kernel} not in {asm}")
    table = collections.defaultdict(collections.Counter)
    loops = collections.defaultdict(collections.Counter)
    cur_line, cur_label = 0, "entry"
    total = collections.Counter()
    for ln in text[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            # a line of the helpers above render_body (vector algebra, reciprocals, the leaf tests) is an inlined call:
            # it stays booked under the section of the last line of render_body seen before it
            if int(m.group(1)) in inc_ids and int(m.group(2)) >= secs[0][0]:
                cur_line = int(m.group(2))
            continue
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            cur_label = m.group(1)
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if not re.match(r"[a-z]", op):
            continue
        c = classify(op)
        name = "other headers"
        for first, nm in secs:
            if cur_line >= first:
                name = nm
        table[name][c] += 1
        total[c] += 1
        loops[cur_label][c] += 1
    order = ["other headers"] + [n for _, n in secs]
    hdr = "%-28s" % "section" + "".join("%11s" % c for c in CLASSES) + "%8s" % "all"
    print(f"kernel {kernel}: static instruction counts (code object, not executed); inlined helpers of lines 1-{secs[0][0] - 1} are booked")
    print("under the section of the last render_body line before them")
    print(hdr)
    for n in order:
        if n in table:
            print("%-28s" % n + "".join("%11d" % table[n][c] for c in CLASSES) + "%8d" % sum(table[n].values()))
    print("%-28s" % "total" + "".join("%11d" % total[c] for c in CLASSES) + "%8d" % sum(total.values()))
    if a.loops:
        print("\nper basic block (label, classes):")
        for lb, cn in loops.items():
            if sum(cn.values()) >= 12:
                print("%-12s" % lb + " ".join(f"{c}={cn[c]}" for c in CLASSES if cn[c]) + f"  all={sum(cn.values())}")


if __name__ == "__main__":
    main()
