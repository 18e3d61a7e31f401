#!/bin/bash
# Usage (GPU box): tools/ab.sh WORKLOAD MASK lib1.so lib2.so [...]  -- interleaved rounds, same box
W=$1; M=$2; shift 2
for r in 1 2 3; do
  for so in "$@"; do
    LDSR_HIP_SO=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry --workload $W --mask $M 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so %.4f ms kernel  %.4g units/s' % (d['roofline']['kernel_ms'], d['value']))"
  done
done
