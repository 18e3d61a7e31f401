#!/bin/bash
# round 4, GPU session 28: pair-family rings for chunks of 17 .. 23 steps (members without the steady form); base lib = the committed build
out=gpurun_out/r4s28; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  lbl=$1; shift
  for r in 1 2; do for v in "" _p23; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v' or 'base', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "600,1,2 x20000 dense" --workload custom --shape 600,1,2,20000
run "600,1,2 x20000 dense conv" --workload custom --shape 600,1,2,20000 --niter 1000 --tol 1e-5
run "600,2,4 x20000 dense" --workload custom --shape 600,2,4,20000
run "730,1,2 x8192 dense" --workload custom --shape 730,1,2,8192
run "320,1,2 x20000 dense" --workload custom --shape 320,1,2,20000
run "360,2,2 x20000 dense conv" --workload custom --shape 360,2,2,20000 --niter 1000 --tol 1e-5
run "600,1,2 x20000 paleo(force no lead)" --workload custom --shape 600,1,2,20000 --mask paleo --algo 3
