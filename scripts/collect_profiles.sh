#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes over the default bench workload.
# usage: scripts/collect_profiles.sh <tag> [bench args...]      output: gpurun_out/prof_<tag>/
# One kernel-trace pass, then one --pmc pass per counter group (a pass holds <= 8 SQ, 4 TCC, 2 GRBM).
set -euo pipefail
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=("$@")
run() {  # name, rocprofv3 options...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 "${args[@]}" \
        > "$out/$name.log" 2>&1
    echo "pass $name done"
}
run kt --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
run pmc_sq3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
