#!/bin/bash
# round 4, GPU session 29: read-ahead ring for the global image of the multi-wave cells (T = 2049 .. 8192); base = commit 9e78e09
out=gpurun_out/r4s29; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_gimg.so timeout -k 10 600 python -m pytest tests/test_gpu_long_series.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest(gimg) rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base gimg; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "3000,1,2 x2048 dense 30 it" --workload custom --shape 3000,1,2,2048 --niter 30
run "3000,1,2 x64 dense 100 it" --workload custom --shape 3000,1,2,64 --niter 100
run "5000,1,2 x2048 dense 30 it" --workload custom --shape 5000,1,2,2048 --niter 30
run "8000,1,2 x1024 dense 30 it" --workload custom --shape 8000,1,2,1024 --niter 30
run "4000,3,3 x1024 dense 30 it" --workload custom --shape 4000,3,3,1024 --niter 30
run "3000,1,2 x2048 dense conv" --workload custom --shape 3000,1,2,2048 --niter 1000 --tol 1e-5
run "8000,3,3 x512 dense 30 it" --workload custom --shape 8000,3,3,512 --niter 30
