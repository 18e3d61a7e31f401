#!/bin/bash
# round 4, GPU session 10: steady form of the one-wave-per-cell kernel as three launches (config 3): tests, A/B against LDSR_SCAN_STEADY=0
out=gpurun_out/r4s10; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_scan_steady.py tests/test_gpu_full_configs.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_a.log 2>&1; rc=$?; echo "pytest(a) rc=$rc" | tee -a $out/summary.txt
tail -15 $out/pytest_a.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do for st in 1 0; do
  LDSR_SCAN_STEADY=$st python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload cfg3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 steady=$st kernel %.4f ms  %.4g units/s verified %s %s' % (d['roofline']['kernel_ms'], d['value'], d['verified']['ok'], d['roofline']['kernel']))" | tee -a $out/summary.txt
done; done
for shp in 1000,1,4,8192 1000,2,2,8192 1000,1,8,8192 700,4,8,8192 1000,1,2,1024 1000,8,8,4096 900,2,4,200; do for st in 1 0; do
  LDSR_SCAN_STEADY=$st python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload custom --shape $shp --algo 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$shp steady=$st kernel %.4f ms %s verified %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel'], d['verified']['ok']))" | tee -a $out/summary.txt
done; done
for st in 1 0; do
  LDSR_SCAN_STEADY=$st python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload cfg3 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 conv steady=$st kernel %.4f ms  %.4g units/s verified %s' % (d['roofline']['kernel_ms'], d['value'], d['verified']['ok']))" | tee -a $out/summary.txt
  LDSR_SCAN_STEADY=$st python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload custom --shape 1000,1,2,8192 --algo 2 --niter 1000 --tol 1e-5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1000,1,2,8192 conv scan steady=$st kernel %.4f ms  %.4g units/s verified %s' % (d['roofline']['kernel_ms'], d['value'], d['verified']['ok']))" | tee -a $out/summary.txt
done
rocprofv3 --kernel-trace --stats -d $out/trace -o cfg3 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry --workload cfg3 > $out/cfg3_under_rocprof.json 2> $out/rocprof.log
python - <<'PY' | tee -a $out/summary.txt
import csv, glob
for f in glob.glob("gpurun_out/r4s10/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f max %.1f" % (float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
for f in glob.glob("gpurun_out/r4s10/trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[-8:]:
        print("   dispatch", r["Kernel_Name"][:60], "%.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -4 $out/pytest.log | tee -a $out/summary.txt
