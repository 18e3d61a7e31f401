# converged runs (tol = 1e-5) on masked series without the closed-form lead: scan kernel against the pair / quad kernels (four-wave workgroups)
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$TAG\", d[\"roofline\"][\"kernel\"], \"%.4f ms kernel  %.4g units/s\" % (d[\"roofline\"][\"kernel_ms\"], d[\"value\"]))"; }
export LDSR_LEAD=0
for shp in 400,1,2,8192 400,1,2,16384 300,2,2,8192 500,1,4,8192 813,1,3,8192 1000,1,2,8192; do
  for a in 2 3 4; do TAG="$shp paleo converged algo=$a"; run --workload custom --shape $shp --mask paleo --niter 1000 --tol 1e-5 --algo $a; done
done
TAG="cfg5 converged LEAD=0 algo=2"; run --workload cfg5 --niter 1000 --tol 1e-5 --algo 2
TAG="cfg5 converged LEAD=0 algo=3"; run --workload cfg5 --niter 1000 --tol 1e-5 --algo 3
