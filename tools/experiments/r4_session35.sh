#!/bin/bash
# round 4, GPU session 35: rings in the long chunks (24 .. 32 steps) of the four-cells-per-wave kernels, padded p + q <= 4
out=gpurun_out/r4s35; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_ql.so timeout -k 10 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_read_ahead.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest(ql) rc=$rc" | tee -a $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
run() {
  lbl=$1; shift
  for r in 1 2; do for v in base ql; do
    LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  %s' % ('$v', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "400,1,2 x8192 dense" --workload custom --shape 400,1,2,8192
run "400,1,2 x8192 dense conv" --workload custom --shape 400,1,2,8192 --niter 1000 --tol 1e-5
run "512,2,2 x8192 dense" --workload custom --shape 512,2,2,8192
run "450,1,1 x20000 holes(paleo no lead)" --workload custom --shape 450,1,1,20000 --mask paleo --algo 4
