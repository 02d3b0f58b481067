set -e
bash scripts/bench_configs.sh r02 > gpurun_out/cfg.log 2>&1
timeout -k 10 900 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --scene teapot_room --brute-force >> gpurun_out/configs_r02.jsonl 2>> gpurun_out/configs_r02.err
bash scripts/collect_profiles.sh r02 > gpurun_out/collect.log 2>&1
bash scripts/collect_profiles.sh r02_teapot --scene teapot_room > gpurun_out/collect_teapot.log 2>&1
bash scripts/collect_profiles.sh r02_csg --scene csg --width 7680 --height 4320 > gpurun_out/collect_csg.log 2>&1
bash scripts/collect_profiles.sh r02_small --scene reflect_refract --width 400 --height 200 > gpurun_out/collect_small.log 2>&1
tail -n 2 gpurun_out/collect_small.log
