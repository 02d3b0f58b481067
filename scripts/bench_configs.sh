#!/bin/bash
# Runs on the GPU box: bench.py over every benchmark configuration, one JSON line each -> gpurun_out/configs_<tag>.jsonl
set -euo pipefail
tag=$1
out=gpurun_out/configs_$tag.jsonl
: > "$out"
run() { timeout -k 10 280 python3 bench.py --steps 5 --warmup 2 --cpu-seconds 6 "$@" >> "$out" 2>> gpurun_out/configs_$tag.err; echo "done $*"; }
run --scene reflect_refract --width 400 --height 200
run --scene cover
run --scene cylinders
run --scene tables
run --scene teapot_room
run --scene glass_spheres --width 7680 --height 4320
run --scene groups --width 7680 --height 4320
run --scene csg --width 7680 --height 4320
