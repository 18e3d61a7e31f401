#!/bin/bash
# round 4, GPU session 19: LLVM scheduling strategies (max-ilp, iterative-ilp) for lone waves; v1 = default strategy, same sources
out=gpurun_out/r4s19; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  lbl=$1; shift
  for r in 1 2; do for v in v1 ilp iilp; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
export LDSR_FORCE_FILL=1
run "813,1,3 x64 paleo 200 it (lone LEAD waves)" --workload custom --shape 813,1,3,64 --mask paleo --niter 200
run "2000,1,4 x64 paleo 200 it (lone LEAD waves)" --workload custom --shape 2000,1,4,64 --mask paleo --niter 200
unset LDSR_FORCE_FILL
run "cfg5 conv" --workload cfg5 --niter 1000 --tol 1e-5
run "cfg5 fixed" --workload cfg5
run "cfg4 conv" --workload cfg4 --niter 1000 --tol 1e-5
run "cfg4 fixed" --workload cfg4
run "800,4,4 x64 dense scan 200 it (lone scan waves)" --workload custom --shape 800,4,4,64 --algo 2 --niter 200
run "800,4,4 x8192 dense scan" --workload custom --shape 800,4,4,8192 --algo 2
for v in v1 ilp iilp; do echo "-- small launches $v" | tee -a $out/summary.txt; LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so timeout -k 10 300 python tools/small_launch_table.py 0 2>/dev/null | grep -v amdgpu.ids | tee -a $out/summary.txt; done
