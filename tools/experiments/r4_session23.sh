#!/bin/bash
# round 4, GPU session 23: persistent work-queue workgroups as small as the LDS allows (pair family), hopping series; base = commit b7d9c0a
out=gpurun_out/r4s23; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
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
run "1000,1,2 x20000 conv" --workload custom --shape 1000,1,2,20000 --niter 1000 --tol 1e-5
run "813,3,3 x8192 paleo conv" --workload custom --shape 813,3,3,8192 --mask paleo --niter 1000 --tol 1e-5
run "400,1,2 x8192 dense conv" --workload custom --shape 400,1,2,8192 --niter 1000 --tol 1e-5
for w in 1 2 4 8; do LDSR_QUEUE_WPB=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload cfg5 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 conv LDSR_QUEUE_WPB=$w %.4f ms' % d['roofline']['kernel_ms'])" | tee -a $out/summary.txt; done
for w in 2 4 8; do LDSR_QUEUE_WPB=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload cfg4 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 conv LDSR_QUEUE_WPB=$w %.4f ms' % d['roofline']['kernel_ms'])" | tee -a $out/summary.txt; done
