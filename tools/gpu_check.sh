#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/gpu_check.sh [tag]
# Runs the GPU parity suite, then the bench on the three single-GPU workloads, printing kernel ms.
tag=${1:-x}
python -m pytest tests -m gpu -q > gpurun_out/test_$tag.log 2>&1; tail -3 gpurun_out/test_$tag.log
for w in "cfg2 dense" "cfg2 paleo" "cfg3 dense" "cfg4 dense" "cfg5 dense"; do
  set -- $w
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $1 --mask $2 > gpurun_out/bench_${tag}_$1_$2.json 2>> gpurun_out/bench_$tag.err
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_${tag}_$1_$2.json')); print('$1 $2: %.4g units/s  step %.3f ms  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
