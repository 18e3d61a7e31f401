#!/bin/bash
# round 4, GPU session 11: the round's final build -- GPU suite, every profile pass, fuzzers, host-entry rates
out=gpurun_out/r4s11; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
bash tools/collect_profiles.sh r4f > $out/collect.log 2>&1; tail -3 $out/collect.log | tee -a $out/summary.txt
echo "== fuzzers" | tee -a $out/summary.txt
timeout -k 10 250 python tools/fuzz_steady.py 250 411 2>&1 | tail -3 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 250 python tools/fuzz_parity.py 250 412 2>&1 | tail -3 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 200 python tools/fuzz_lead.py 150 413 2>&1 | tail -3 | tee -a $out/summary.txt
echo "== host api rates" | tee -a $out/summary.txt
timeout -k 10 300 python tools/host_api_rate.py > $out/host_api_rates.txt 2>&1; tail -12 $out/host_api_rates.txt | tee -a $out/summary.txt
echo "== smoke" | tee -a $out/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $out/summary.txt
