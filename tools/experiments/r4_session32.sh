#!/bin/bash
# round 4, GPU session 32: depth of the global-image ring (steps ahead: up to two pairs per step / three / more): 3,2,2 (base) against 4,3,2 and 6,4,2
out=gpurun_out/r4s32; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base d4 d6; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "3000,1,2 x2048 dense 30 it" --workload custom --shape 3000,1,2,2048 --niter 30
run "3000,1,2 x64 dense 100 it" --workload custom --shape 3000,1,2,64 --niter 100
run "5000,1,2 x2048 dense 30 it" --workload custom --shape 5000,1,2,2048 --niter 30
run "8000,2,2 x1024 dense 30 it" --workload custom --shape 8000,2,2,1024 --niter 30
run "4000,3,3 x1024 dense 30 it" --workload custom --shape 4000,3,3,1024 --niter 30
