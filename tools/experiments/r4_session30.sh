#!/bin/bash
# round 4, GPU session 30: the EM launch's block table written by series_prep (calls of <= 32 series) instead of a staged copy
out=gpurun_out/r4s30; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do for st in 1 0; do for w in cfg2 cfg5 cfg4; do
  LDSR_STAGE_TABLE=$st python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-entry --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w staged=$st  %.4f ms/step  kernel %.4f  value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" | tee -a $out/summary.txt
done; done; done
