#!/bin/bash
# round 4, GPU session 12: closed-form variances of the steady form's transient block (pair_var_block) against the 2x2 DPP scan
out=gpurun_out/r4s12; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_full_configs.py -m gpu -q -x > $out/pytest_a.log 2>&1; rc=$?; echo "pytest(a) rc=$rc" | tee -a $out/summary.txt
tail -6 $out/pytest_a.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
echo "== cfg2 fixed" | tee -a $out/summary.txt
bash tools/ab.sh cfg2 dense ldsr_amd/libldsr_hip_scan2x2.so ldsr_amd/libldsr_hip.so | tee -a $out/summary.txt
echo "== cfg2 converged" | tee -a $out/summary.txt
for r in 1 2 3; do for so in ldsr_amd/libldsr_hip_scan2x2.so ldsr_amd/libldsr_hip.so; do
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload cfg2 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so conv %.4f ms/step kernel %.4f  %.4g units/s verified %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['verified']['ok']))" | tee -a $out/summary.txt
done; done
echo "== 20000 cells conv / fixed, 800,1,1" | tee -a $out/summary.txt
for r in 1 2; do for so in ldsr_amd/libldsr_hip_scan2x2.so ldsr_amd/libldsr_hip.so; do
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload custom --shape 1000,1,2,20000 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so 20000conv %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel']))" | tee -a $out/summary.txt
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload custom --shape 1000,1,2,20000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so 20000fixed %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel']))" | tee -a $out/summary.txt
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload custom --shape 800,1,1,4096 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so 800,1,1 fixed %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel']))" | tee -a $out/summary.txt
done; done
echo "== fuzz steady" | tee -a $out/summary.txt
timeout -k 10 400 python tools/fuzz_steady.py 400 421 2>&1 | tail -3 | tee -a $out/summary.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
