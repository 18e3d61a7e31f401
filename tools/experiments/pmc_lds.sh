cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_x; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/a -- python3 bench.py --workload custom --shape 1000,1,2,4096 --algo 3 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/a.err
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$out/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "em_pair" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, sum(v)/len(v))
PY
tail -3 $out/a.err
