#!/bin/bash
# round 4, GPU session 31: the ring for wide inputs in the half-stored long chunks (L = 20, 24) of the one-wave-per-cell kernel
out=gpurun_out/r4s31; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_wl.so timeout -k 10 600 python -m pytest tests/test_gpu_long_series.py tests/test_gpu_parity.py tests/test_gpu_read_ahead.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest(wl) rc=$rc" | tee -a $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base wl; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "1200,1,8 x4096 dense 50 it" --workload custom --shape 1200,1,8,4096 --niter 50
run "1200,4,8 x4096 dense 50 it" --workload custom --shape 1200,4,8,4096 --niter 50
run "1500,2,8 x4096 dense 50 it" --workload custom --shape 1500,2,8,4096 --niter 50
run "1500,7,2 x4096 dense 50 it" --workload custom --shape 1500,7,2,4096 --niter 50
run "1200,1,8 x64 dense conv" --workload custom --shape 1200,1,8,64 --niter 1000 --tol 1e-5
run "1200,7,4 x4096 paleo(no lead) 50 it" --workload custom --shape 1200,7,4,4096 --mask paleo --niter 50 --algo 2
