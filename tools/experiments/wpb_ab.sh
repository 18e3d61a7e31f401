# same-box A/B of the pair family's waves per workgroup (LDSR_PAIR_WPB: 8 = one workgroup per CU as before, unset = two of four where the LDS allows, 2 = four of two)
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$TAG\", d[\"roofline\"][\"kernel\"], \"%.4f ms kernel  %.4g units/s\" % (d[\"roofline\"][\"kernel_ms\"], d[\"value\"]))"; }
for w in 8 4 2; do
  export LDSR_PAIR_WPB=$w
  for wl in cfg4 cfg5 cfg2; do TAG="$wl wpb=$w"; run --workload $wl; done
  TAG="cfg2 paleo wpb=$w"; run --workload cfg2 --mask paleo
  TAG="cfg5 converged wpb=$w"; run --workload cfg5 --niter 1000 --tol 1e-5
  TAG="cfg2 paleo converged wpb=$w"; run --workload cfg2 --mask paleo --niter 1000 --tol 1e-5
  for shp in 300,1,2,8192 300,1,2,10000 500,1,2,8192 500,2,4,6000 813,3,3,8192 200,4,4,8192 700,1,1,5000; do
    TAG="$shp dense wpb=$w"; run --workload custom --shape $shp
    TAG="$shp dense converged wpb=$w"; run --workload custom --shape $shp --niter 1000 --tol 1e-5 --algo 3
  done
  TAG="813,3,3,8192 paleo wpb=$w"; run --workload custom --shape 813,3,3,8192 --mask paleo
  TAG="2000,1,4,12000 paleo wpb=$w"; run --workload custom --shape 2000,1,4,12000 --mask paleo
done
