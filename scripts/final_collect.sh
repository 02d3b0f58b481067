set -e
bash scripts/bench_configs.sh r03 > gpurun_out/cfg.log 2>&1
timeout -k 10 900 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --scene teapot_room --brute-force >> gpurun_out/configs_r03.jsonl 2>> gpurun_out/configs_r03.err
bash scripts/collect_profiles.sh r03 > gpurun_out/collect.log 2>&1
bash scripts/collect_profiles.sh r03_teapot --scene teapot_room > gpurun_out/collect_teapot.log 2>&1
bash scripts/collect_profiles.sh r03_csg --scene csg --width 7680 --height 4320 > gpurun_out/collect_csg.log 2>&1
bash scripts/collect_profiles.sh r03_small --scene reflect_refract --width 400 --height 200 > gpurun_out/collect_small.log 2>&1
bash scripts/collect_profiles.sh r03_groups --scene groups --width 7680 --height 4320 > gpurun_out/collect_groups.log 2>&1
tail -n 2 gpurun_out/collect_small.log
