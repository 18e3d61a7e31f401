run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$TAG\", d[\"roofline\"][\"kernel\"], \"%.4f ms kernel\" % (d[\"roofline\"][\"kernel_ms\"]))"; }
export LDSR_LEAD=0
for m in dense paleo; do for n in 1024 2048 4096; do for a in 0 2; do TAG="400,1,2 $m n=$n converged algo=$a"; run --workload custom --shape 400,1,2,$n --mask $m --niter 1000 --tol 1e-5 --algo $a; done; done; done
