#!/bin/bash
# round 4, GPU session 26: the long-chunk scan kernel's one-step read pipeline (LDSR_SCAN_PREFETCH) for lone waves
out=gpurun_out/r4s26; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base pf; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "1200,1,2 x64 dense 200 it" --workload custom --shape 1200,1,2,64 --niter 200
run "1500,3,3 x64 dense 200 it" --workload custom --shape 1500,3,3,64 --niter 200
run "2000,1,2 x64 dense 200 it" --workload custom --shape 2000,1,2,64 --niter 200
LDSR_LEAD=0 run "1200,1,2 x64 paleo 200 it" --workload custom --shape 1200,1,2,64 --mask paleo --niter 200
run "1500,1,2 x8192 dense" --workload custom --shape 1500,1,2,8192
run "2000,3,3 x4096 dense" --workload custom --shape 2000,3,3,4096
