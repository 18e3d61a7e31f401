#!/bin/bash
# round 4, GPU session 27: read-ahead rings in the long-chunk (17 .. 32 steps) scan kernels; base = commit 47dffbd
out=gpurun_out/r4s27; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_spfl.so timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest(spfl) rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base spfl; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "1200,1,2 x64 dense 200 it" --workload custom --shape 1200,1,2,64 --niter 200
run "1500,3,3 x64 dense 200 it" --workload custom --shape 1500,3,3,64 --niter 200
run "2000,1,2 x64 dense 200 it" --workload custom --shape 2000,1,2,64 --niter 200
run "1200,1,2 x64 paleo conv" --workload custom --shape 1200,1,2,64 --mask paleo --niter 1000 --tol 1e-5
run "1500,1,2 x8192 dense" --workload custom --shape 1500,1,2,8192
run "1500,1,2 x8192 dense conv" --workload custom --shape 1500,1,2,8192 --niter 1000 --tol 1e-5
run "2000,3,3 x4096 dense" --workload custom --shape 2000,3,3,4096
run "1200,2,4 x8192 paleo conv" --workload custom --shape 1200,2,4,8192 --mask paleo --niter 1000 --tol 1e-5
