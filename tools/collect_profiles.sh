#!/bin/bash
# Usage (GPU box, repo root): tools/collect_profiles.sh TAG   -> gpurun_out/prof_TAG/<workload>/{trace,fetch,write,sq}
# rocprofv3 passes of `python3 bench.py --workload W`: kernel trace + stats, then the PMC passes
# each in its own run (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc).
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in cfg2 cfg3 cfg4 cfg5; do
  out=gpurun_out/prof_$tag/$w
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry > $out/bench_under_rocprof.json 2> $out/trace.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry > /dev/null 2> $out/fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry > /dev/null 2> $out/write.err
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/sq -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry > /dev/null 2> $out/sq.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq2 -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry > /dev/null 2> $out/sq2.err
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/sq3 -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-host-entry > /dev/null 2> $out/sq3.err
  python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
  echo "$w done: $(ls $out)"
done
# "report both masks", and the reference's normal mode of use (tol = 1e-5, src/EM.cpp:272)
x=gpurun_out/prof_$tag/extra; mkdir -p $x
python3 bench.py --workload cfg2 --mask paleo --steps 20 --warmup 3 --no-cpu-baseline > $x/cfg2_paleo_bench.json 2> $x/err
python3 bench.py --workload cfg2 --niter 1000 --tol 1e-5 --steps 20 --warmup 3 --no-cpu-baseline > $x/cfg2_conv_bench.json 2>> $x/err
python3 bench.py --workload cfg2 --mask paleo --niter 1000 --tol 1e-5 --steps 20 --warmup 3 --no-cpu-baseline > $x/cfg2_conv_paleo_bench.json 2>> $x/err
python3 bench.py --workload cfg3 --mask paleo --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry > $x/cfg3_paleo_bench.json 2>> $x/err
python3 bench.py --workload cfg3 --niter 1000 --tol 1e-5 --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry > $x/cfg3_conv_bench.json 2>> $x/err
python3 bench.py --workload cfg4 --niter 1000 --tol 1e-5 --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry > $x/cfg4_conv_bench.json 2>> $x/err
python3 bench.py --workload cfg5 --niter 1000 --tol 1e-5 --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry > $x/cfg5_conv_bench.json 2>> $x/err
# the line the round driver records (`python bench.py`, defaults: cpu_baseline and host_entry included)
python3 bench.py > $x/cfg2_full_bench.json 2>> $x/err
echo "extra done: $(ls $x)"
