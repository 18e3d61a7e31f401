#!/bin/bash
# round 4, GPU session 7: wave-parallel series_prep (tests, whole-step times against round 3), fuzzers
out=gpurun_out/r4s7; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
for w in cfg2 cfg3 cfg4 cfg5; do
 for r in 1 2; do for so in ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so; do
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-entry --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $so step %.4f ms kernel %.4f ms  overhead %.1f us  %.4g units/s' % (d['ms_per_step'], d['roofline']['kernel_ms'], 1e3*(d['ms_per_step']-d['roofline']['kernel_ms']), d['value']))" | tee -a $out/summary.txt
 done; done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg3 -- python3 bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry > /dev/null 2> $out/trace.err
find $out/trace_cfg3 -name "*kernel_stats.csv" | head -1 | xargs cat | head -5 | tee -a $out/summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg2 -- python3 bench.py --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry > /dev/null 2>> $out/trace.err
find $out/trace_cfg2 -name "*kernel_stats.csv" | head -1 | xargs cat | head -5 | tee -a $out/summary.txt
echo "== fuzzers" | tee -a $out/summary.txt
timeout -k 10 250 python tools/fuzz_steady.py 250 401 2>&1 | tail -3 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 250 python tools/fuzz_parity.py 250 402 2>&1 | tail -3 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 200 python tools/fuzz_lead.py 150 403 2>&1 | tail -3 | tee -a $out/summary.txt
