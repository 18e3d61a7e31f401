#!/bin/bash
# round 4, GPU session 14: read-ahead of the LDS image in the scan kernel's short-chunk sweeps (SPF, em_scan_impl.h)
# libldsr_hip_base.so = HEAD before the change, libldsr_hip_spf.so = with the read-ahead ring
out=gpurun_out/r4s14; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$PWD/ldsr_amd/libldsr_hip_base.so; S=$PWD/ldsr_amd/libldsr_hip_spf.so
LDSR_HIP_SO=$S timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest_spf.log 2>&1; rc=$?; echo "pytest(spf) rc=$rc" | tee -a $out/summary.txt
tail -4 $out/pytest_spf.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
echo "== small launches (AUTO), base then spf" | tee -a $out/summary.txt
for so in $B $S $B $S; do
  echo "-- $so" | tee -a $out/summary.txt
  LDSR_HIP_SO=$so timeout -k 10 300 python tools/small_launch_table.py 0 2>/dev/null | grep -v amdgpu.ids | tee -a $out/summary.txt
done
echo "== large launches through the scan kernel (algo 2): kernel ms, base / spf interleaved" | tee -a $out/summary.txt
run() {  # label, bench args...
  lbl=$1; shift
  for r in 1 2; do for so in $B $S; do
    LDSR_HIP_SO=$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry "$@" 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('$lbl %-8s %.4f ms  %s  verified %s' % (os.path.basename('$so')[11:-3], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
  done; done
}
run "1000,1,2 x4096 dense scan" --workload cfg2 --algo 2
run "1000,1,2 x4096 paleo scan" --workload cfg2 --mask paleo --algo 2
run "1000,1,2 x4096 dense scan conv" --workload cfg2 --algo 2 --niter 1000 --tol 1e-5
run "800,4,4 x8192 dense scan" --workload custom --shape 800,4,4,8192 --algo 2
run "500,2,2 x8192 dense scan" --workload custom --shape 500,2,2,8192 --algo 2
run "1000,2,4 x8192 dense scan" --workload custom --shape 1000,2,4,8192 --algo 2
run "cfg5 conv scan" --workload cfg5 --niter 1000 --tol 1e-5 --algo 2
run "cfg5 conv AUTO" --workload cfg5 --niter 1000 --tol 1e-5
run "cfg3 (unchanged kernel)" --workload cfg3
