#!/bin/bash
# round 4, GPU session 1: the odd-value image layout (libldsr_hip.so) against round 3's build (libldsr_hip_base.so)
out=gpurun_out/r4s1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest.log | tee -a $out/summary.txt
for w in cfg3 cfg2 cfg4 cfg5; do
  echo "== $w" | tee -a $out/summary.txt
  timeout -k 10 300 tools/ab.sh $w dense ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so 2>&1 | tee -a $out/summary.txt
done
echo "== custom 1024,4,4 / 1000,2,2 / 900,4,4" | tee -a $out/summary.txt
for so in ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so ldsr_amd/libldsr_hip_base.so ldsr_amd/libldsr_hip.so; do
 for shp in 1000,2,2,8192 813,3,3,8192 1000,1,1,8192; do
  LDSR_HIP_SO=$PWD/$so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload custom --shape $shp 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so $shp %.4f ms kernel %s' % (d['roofline']['kernel_ms'], d['roofline'].get('kernel','')))" | tee -a $out/summary.txt
 done
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq2 -- python3 bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/sq2.err
python3 tools/summarize_profiles.py $out/sq2 2>/dev/null | tail -5 | tee -a $out/summary.txt
find $out/sq2 -name "*counter_collection.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys,collections
t=collections.defaultdict(float)
for r in csv.DictReader(open('{}')):
    if 'em_scan' in r['Kernel_Name']: t[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(t)); print('conflict frac', t['SQ_LDS_BANK_CONFLICT']/max(t['SQ_LDS_IDX_ACTIVE'],1))
" | tee -a $out/summary.txt
