#!/bin/bash
# round 4, GPU session 16: read-ahead rings in the pair family's generic sweeps (L <= 16) and the pipelined walk of the
# closed-form lead's passes.  libldsr_hip_base.so = commit 69fd272 (scan kernel's ring only), libldsr_hip.so = this build
out=gpurun_out/r4s16; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$PWD/ldsr_amd/libldsr_hip_base.so; S=$PWD/ldsr_amd/libldsr_hip.so
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
run() {  # label, bench args...
  lbl=$1; shift
  for r in 1 2; do for so in $B $S; do
    LDSR_HIP_SO=$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-5s %.4f ms  %s  verified %s' % (os.path.basename('$so')[11:-3] or 'new', d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "cfg5 conv" --workload cfg5 --niter 1000 --tol 1e-5
run "cfg4 conv" --workload cfg4 --niter 1000 --tol 1e-5
run "cfg5 fixed" --workload cfg5
run "cfg4 fixed" --workload cfg4
run "cfg2 paleo fixed" --workload cfg2 --mask paleo
run "cfg2 paleo conv" --workload cfg2 --mask paleo --niter 1000 --tol 1e-5
run "cfg3 paleo fixed" --workload cfg3 --mask paleo
run "400,1,2 x8192 dense fixed" --workload custom --shape 400,1,2,8192
run "400,1,2 x8192 dense conv" --workload custom --shape 400,1,2,8192 --niter 1000 --tol 1e-5
run "300,2,2 x8192 dense fixed" --workload custom --shape 300,2,2,8192
run "813,3,3 x8192 paleo fixed" --workload custom --shape 813,3,3,8192 --mask paleo
run "813,3,3 x8192 paleo conv" --workload custom --shape 813,3,3,8192 --mask paleo --niter 1000 --tol 1e-5
run "cfg2 fixed" --workload cfg2
run "cfg2 conv" --workload cfg2 --niter 1000 --tol 1e-5
