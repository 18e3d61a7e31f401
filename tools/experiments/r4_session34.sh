#!/bin/bash
# round 4, GPU session 34: series_prep with the raw series staged in LDS (LDSR_PREP_STAGE=0: as before, from global memory)
out=gpurun_out/r4s34; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do for st in 0 1; do for w in cfg2 cfg3 cfg5; do
  LDSR_PREP_STAGE=$st python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-entry --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w stage=$st  %.4f ms/step  kernel %.4f  outside %.1f us  value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], 1e3*(d['ms_per_step']-d['roofline']['kernel_ms']), d['value']))" | tee -a $out/summary.txt
done; done; done
for st in 0 1; do echo "-- small launches, LDSR_PREP_STAGE=$st" | tee -a $out/summary.txt; LDSR_PREP_STAGE=$st python tools/small_launch_table.py 0 2>/dev/null | grep -v amdgpu.ids | tee -a $out/summary.txt; done
