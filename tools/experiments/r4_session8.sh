#!/bin/bash
# round 4, GPU session 8: the round's profile collection (tools/collect_profiles.sh r4f) + small-launch table + host API rates + regret
cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh r4f 2>&1 | tail -8
mkdir -p gpurun_out/r4s8
timeout -k 10 300 python tools/small_launch_table.py > gpurun_out/r4s8/small_launches.txt 2>&1; tail -6 gpurun_out/r4s8/small_launches.txt
timeout -k 10 200 python tools/host_api_rate.py dense > gpurun_out/r4s8/host_api_rates.txt 2>&1; timeout -k 10 200 python tools/host_api_rate.py paleo >> gpurun_out/r4s8/host_api_rates.txt 2>&1; grep -v amdgpu gpurun_out/r4s8/host_api_rates.txt | tail -12
REGRET_ONLY_BASELINE_SIZES=1 timeout -k 10 400 python tools/auto_regret.py > gpurun_out/r4s8/auto_regret_baseline_sizes.txt 2>&1; tail -4 gpurun_out/r4s8/auto_regret_baseline_sizes.txt
