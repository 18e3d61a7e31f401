#!/bin/bash
# round 4, GPU session 21: persistent workgroups that move from series to series (work-queue launches); base = commit b7d9c0a
out=gpurun_out/r4s21; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
run() {
  lbl=$1; shift
  for r in 1 2; do for v in _base ""; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v' or 'new', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "cfg5 conv" --workload cfg5 --niter 1000 --tol 1e-5
run "cfg4 conv" --workload cfg4 --niter 1000 --tol 1e-5
run "cfg3 conv" --workload cfg3 --niter 1000 --tol 1e-5
run "cfg2 conv" --workload cfg2 --niter 1000 --tol 1e-5
run "cfg2 paleo conv" --workload cfg2 --mask paleo --niter 1000 --tol 1e-5
run "cfg3 paleo conv" --workload cfg3 --mask paleo --niter 1000 --tol 1e-5
run "cfg5 conv scan" --workload cfg5 --niter 1000 --tol 1e-5 --algo 2
run "1000,1,2 x20000 conv" --workload custom --shape 1000,1,2,20000 --niter 1000 --tol 1e-5
run "813,3,3 x8192 paleo conv" --workload custom --shape 813,3,3,8192 --mask paleo --niter 1000 --tol 1e-5
run "3000,1,2 x2048 conv" --workload custom --shape 3000,1,2,2048 --niter 1000 --tol 1e-5
echo "-- LDSR_QUEUE_SLOTS=0 (one slot per cell, new build)" | tee -a $out/summary.txt
LDSR_QUEUE_SLOTS=0 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload cfg5 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 conv slots=cells %.4f ms' % d['roofline']['kernel_ms'])" | tee -a $out/summary.txt
python tools/small_launch_table.py 0 2>/dev/null | grep -v amdgpu.ids | tee -a $out/summary.txt
