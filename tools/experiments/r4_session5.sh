#!/bin/bash
# round 4, GPU session 5: full tests on the build with the function-call G phase + uniform cell index; cfg3/4/5 A/B; regret rows
out=gpurun_out/r4s5; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest.log | tee -a $out/summary.txt
for w in cfg3 cfg4 cfg5; do
  echo "== $w (s4 = before the uniform cell index)" | tee -a $out/summary.txt
  timeout -k 10 300 tools/ab.sh $w dense ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip_s4.so ldsr_amd/libldsr_hip.so 2>&1 | tee -a $out/summary.txt
done
echo "== scan shapes" | tee -a $out/summary.txt
for so in ldsr_amd/libldsr_hip_s4.so ldsr_amd/libldsr_hip.so ldsr_amd/libldsr_hip_s4.so ldsr_amd/libldsr_hip.so; do
 for shp in 1000,2,2,8192 1000,1,8,8192 2000,1,2,4096 213,3,3,200; do
  LDSR_HIP_SO=$PWD/$so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload custom --shape $shp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so $shp %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel','')))" | tee -a $out/summary.txt
 done
done
echo "== regret rows at the BASELINE launch sizes" | tee -a $out/summary.txt
REGRET_ONLY_BASELINE_SIZES=1 timeout -k 10 400 python tools/auto_regret.py 2>&1 | tee $out/auto_regret_baseline_sizes.txt | tail -12 | tee -a $out/summary.txt
echo "== full bench line" | tee -a $out/summary.txt
python bench.py --steps 20 --warmup 3 > $out/bench_cfg2.json 2> $out/bench_cfg2.err; python -c "
import json; d=json.loads(open('$out/bench_cfg2.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('host_entry'), d['cpu_baseline']['value'])" | tee -a $out/summary.txt
