#!/bin/bash
# A/B of two builds on one box: scripts/ab.sh <lib dir A> <lib dir B> scene W H [scene W H ...]
# (three alternating rounds of scripts/gpu_dev.py time, fast arithmetic; prints the medians side by side)
a=$1; b=$2; shift 2
for r in 1 2 3; do
  for d in "$a" "$b"; do
    RTAMD_LIB_DIR=$d timeout -k 10 300 python -u scripts/gpu_dev.py time --modes ${AB_MODES:-fast} --frames 8 "$@" | sed "s|^|$d r$r: |" || exit 1
  done
done
