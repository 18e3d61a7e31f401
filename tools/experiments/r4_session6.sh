#!/bin/bash
# round 4, GPU session 6: the pruned library (tests), the full bench line with host_entry, small-launch table
out=gpurun_out/r4s6; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest.log | tee -a $out/summary.txt
python bench.py --steps 20 --warmup 3 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; python -c "
import json; d=json.loads(open('$out/bench_cfg2.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline'].get('frac_at_sustained_clock'), d.get('host_entry'), d['cpu_baseline']['value'])" | tee -a $out/summary.txt
tail -3 $out/bench_cfg2.err | tee -a $out/summary.txt
for m in dense paleo; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --niter 1000 --tol 1e-5 --mask $m > $out/bench_cfg2_conv_$m.json 2>> $out/bench_cfg2.err; python -c "
import json; d=json.loads(open('$out/bench_cfg2_conv_$m.json').read().strip().splitlines()[-1]); print('conv $m', {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'], d['roofline']['kernel'], d.get('host_entry',{}).get('ms_per_call'))" | tee -a $out/summary.txt
done
timeout -k 10 300 python tools/small_launch_table.py 2>&1 | tail -30 | tee -a $out/summary.txt
timeout -k 10 200 python tools/host_api_rate.py dense 2>&1 | tail -8 | tee -a $out/summary.txt
timeout -k 10 200 python tools/host_api_rate.py paleo 2>&1 | tail -8 | tee -a $out/summary.txt
