#!/bin/bash
# round 4, GPU session 24: the final build's regret table, host-entry rates, full regret sweep and fuzzers
out=gpurun_out/r4s24; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "from ldsr_amd import _lib; print(_lib.lib().ldsr_version().decode())" 2>/dev/null | tee -a $out/summary.txt
REGRET_ONLY_BASELINE_SIZES=1 timeout -k 10 400 python tools/auto_regret.py > $out/auto_regret_baseline_sizes.txt 2>&1; tail -4 $out/auto_regret_baseline_sizes.txt | tee -a $out/summary.txt
timeout -k 10 200 python tools/host_api_rate.py dense > $out/host_api_rates.txt 2>&1; timeout -k 10 200 python tools/host_api_rate.py paleo >> $out/host_api_rates.txt 2>&1; grep -v amdgpu $out/host_api_rates.txt | tail -12 | tee -a $out/summary.txt
timeout -k 10 900 python tools/auto_regret.py > $out/auto_regret.txt 2>&1; tail -6 $out/auto_regret.txt | tee -a $out/summary.txt
echo "== fuzz" | tee -a $out/summary.txt
timeout -k 10 600 python tools/fuzz_steady.py 800 601 2>&1 | tail -3 | tee -a $out/fuzz.txt
LDSR_FORCE_FILL=1 timeout -k 10 900 python tools/fuzz_parity.py 1500 602 2>&1 | tail -6 | tee -a $out/fuzz.txt
timeout -k 10 700 python tools/fuzz_parity.py 1000 603 2>&1 | tail -6 | tee -a $out/fuzz.txt
LDSR_FORCE_FILL=1 timeout -k 10 800 python tools/fuzz_lead.py 1000 604 2>&1 | tail -3 | tee -a $out/fuzz.txt
