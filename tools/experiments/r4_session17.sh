#!/bin/bash
# round 4, GPU session 17: base (commit 69fd272) / v1 (lead walk for p <= 2, rings in the pair sweeps) / v2 (lead walk only)
out=gpurun_out/r4s17; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base v1 v2; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "813,3,3 x8192 paleo fixed" --workload custom --shape 813,3,3,8192 --mask paleo
run "813,3,3 x8192 paleo conv" --workload custom --shape 813,3,3,8192 --mask paleo --niter 1000 --tol 1e-5
run "cfg3 paleo fixed" --workload cfg3 --mask paleo
run "cfg5 conv" --workload cfg5 --niter 1000 --tol 1e-5
run "cfg5 fixed" --workload cfg5
run "cfg4 conv" --workload cfg4 --niter 1000 --tol 1e-5
run "cfg4 fixed" --workload cfg4
run "cfg2 paleo fixed" --workload cfg2 --mask paleo
run "cfg2 paleo conv" --workload cfg2 --mask paleo --niter 1000 --tol 1e-5
run "813,2,2 x8192 paleo conv" --workload custom --shape 813,2,2,8192 --mask paleo --niter 1000 --tol 1e-5
