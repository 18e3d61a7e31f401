#!/bin/bash
# Usage (GPU box): tools/pmc_study.sh TAG "T,p,q,restarts" MASK ["extra bench args"]   -- SQ counter passes of one custom shape
tag=$1; shape=$2; mask=${3:-dense}; extra=${4:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/a -- python3 bench.py --workload custom --shape $shape --mask $mask $extra --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/a.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $out/b -- python3 bench.py --workload custom --shape $shape --mask $mask $extra --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/b.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d $out/c -- python3 bench.py --workload custom --shape $shape --mask $mask $extra --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/c.err
python3 - <<PY
import csv, glob, collections, json
agg=collections.defaultdict(list)
for f in glob.glob("$out/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "em_scan" in r["Kernel_Name"] or "em_pair" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
units=d["roofline"]["units_per_launch"]
v={k:sum(x)/len(x) for k,x in agg.items()}
wc=v["SQ_WAVE_CYCLES"]
print("$tag $shape $mask", d["roofline"]["kernel"], "kernel_ms(unprofiled n/a)")
print("  per unit: VALU %.0f (FMA64 %.0f MUL64 %.0f ADD64 %.0f TRANS64 %.0f INT32 %.0f) LDS %.0f SALU %.0f SMEM %.1f VMEM %.2f" % tuple(v[k]/units for k in ["SQ_INSTS_VALU","SQ_INSTS_VALU_FMA_F64","SQ_INSTS_VALU_MUL_F64","SQ_INSTS_VALU_ADD_F64","SQ_INSTS_VALU_TRANS_F64","SQ_INSTS_VALU_INT32","SQ_INSTS_LDS","SQ_INSTS_SALU","SQ_INSTS_SMEM","SQ_INSTS_VMEM"]))
print("  per wave-cycle: VALU active %.3f  wait_inst_any %.3f (lds %.3f)  wait_any %.3f  active_any %.3f  LDS active %.3f  SCA active %.3f  lds level %.2f  smem level %.2f" % (v["SQ_ACTIVE_INST_VALU"]/wc, v["SQ_WAIT_INST_ANY"]/wc, v["SQ_WAIT_INST_LDS"]/wc, v["SQ_WAIT_ANY"]/wc, v["SQ_ACTIVE_INST_ANY"]/wc, v["SQ_ACTIVE_INST_LDS"]/wc, v["SQ_ACTIVE_INST_SCA"]/wc, v["SQ_INST_LEVEL_LDS"]/wc, v["SQ_INST_LEVEL_SMEM"]/wc))
PY
