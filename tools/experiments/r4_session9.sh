#!/bin/bash
# round 4, GPU session 9: steady form of the one-wave-per-cell kernel (config 3): tests, A/B against LDSR_SCAN_STEADY=0
out=gpurun_out/r4s9; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_a.log 2>&1; echo "pytest(a) rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest_a.log | tee -a $out/summary.txt
for r in 1 2 3; do for st in 1 0; do
  LDSR_SCAN_STEADY=$st python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload cfg3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 steady=$st kernel %.4f ms  %.4g units/s verified %s' % (d['roofline']['kernel_ms'], d['value'], d['verified']['ok']))" | tee -a $out/summary.txt
done; done
for shp in 1000,1,4,8192 1000,2,2,8192 1000,1,8,8192 800,4,4,8192 1000,1,2,1024 1000,8,8,4096; do for st in 1 0; do
  LDSR_SCAN_STEADY=$st python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload custom --shape $shp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shp steady=$st kernel %.4f ms %s verified %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
done; done
for st in 1 0; do
  LDSR_SCAN_STEADY=$st python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload cfg3 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 conv steady=$st kernel %.4f ms  %.4g units/s verified %s' % (d['roofline']['kernel_ms'], d['value'], d['verified']['ok']))" | tee -a $out/summary.txt
done
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
