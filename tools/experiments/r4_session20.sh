#!/bin/bash
# round 4, GPU session 20: issue priority by the age of the wave's oldest cell (prio_by_age); base = commit b7d9c0a
out=gpurun_out/r4s20; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  lbl=$1; shift
  for r in 1 2 3; do for v in base prio; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "cfg5 conv" --workload cfg5 --niter 1000 --tol 1e-5
run "cfg4 conv" --workload cfg4 --niter 1000 --tol 1e-5
run "cfg3 conv" --workload cfg3 --niter 1000 --tol 1e-5
run "cfg2 conv" --workload cfg2 --niter 1000 --tol 1e-5
run "cfg2 paleo conv" --workload cfg2 --mask paleo --niter 1000 --tol 1e-5
run "cfg5 conv scan" --workload cfg5 --niter 1000 --tol 1e-5 --algo 2
