# where does the pair family start to pay?  launch sizes below the device-filling rule, scan (2) against pair (3) / quad (4)
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$TAG\", d[\"roofline\"][\"kernel\"], \"%.4f ms kernel\" % (d[\"roofline\"][\"kernel_ms\"]))"; }
for shp in 400,1,2 200,2,2 300,1,4; do for n in 1024 2048 3072 4096 6144; do for a in 2 3 4; do TAG="$shp n=$n algo=$a"; run --workload custom --shape $shp,$n --algo $a; done; done; done
