#!/bin/bash
# round 4, GPU session 2: two-loop steady form + cell ordering (libldsr_hip.so) against round 3's build
out=gpurun_out/r4s2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -15 $out/pytest.log | tee -a $out/summary.txt
echo "== cfg2 fixed" | tee -a $out/summary.txt
timeout -k 10 300 tools/ab.sh cfg2 dense ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so 2>&1 | tee -a $out/summary.txt
echo "== cfg2 fixed, no ordering" | tee -a $out/summary.txt
for r in 1 2; do LDSR_STEADY_ORDER=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload cfg2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noorder %.4f ms kernel  %.4g units/s' % (d['roofline']['kernel_ms'], d['value']))" | tee -a $out/summary.txt; done
echo "== cfg2 converged (niter 1000 tol 1e-5)" | tee -a $out/summary.txt
for r in 1 2 3; do for so in ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so; do
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload cfg2 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so conv %.4f ms/step kernel %.4f  %.4g units/s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" | tee -a $out/summary.txt
done; done
echo "== custom 20000 cells conv / fixed" | tee -a $out/summary.txt
for so in ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so; do
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload custom --shape 1000,1,2,20000 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so 20000conv %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel','')))" | tee -a $out/summary.txt
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload custom --shape 1000,1,2,20000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so 20000fixed %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel','')))" | tee -a $out/summary.txt
  LDSR_HIP_SO=$PWD/$so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload custom --shape 800,1,1,4096 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so 800,1,1 fixed %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel','')))" | tee -a $out/summary.txt
done
