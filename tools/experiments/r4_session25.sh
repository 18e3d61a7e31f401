#!/bin/bash
# round 4, GPU session 25: AUTO's rules for runs to convergence after the read-ahead rings: GPU suite, the regret sweep again
out=gpurun_out/r4s25; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
timeout -k 10 900 python tools/auto_regret.py > $out/auto_regret.txt 2>&1; tail -8 $out/auto_regret.txt | tee -a $out/summary.txt
REGRET_ONLY_BASELINE_SIZES=1 timeout -k 10 400 python tools/auto_regret.py > $out/auto_regret_baseline_sizes.txt 2>&1; tail -3 $out/auto_regret_baseline_sizes.txt | tee -a $out/summary.txt
