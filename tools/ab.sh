#!/bin/bash
# Usage (GPU box): tools/ab.sh A.so B.so [workload] [mask]   -- interleaved rounds, same process order
A=$1; B=$2; W=${3:-cfg2}; M=${4:-dense}
for r in 1 2 3 4; do
  for so in $A $B; do
    LDSR_HIP_SO=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $W --mask $M 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so %.4f ms kernel  %.4g units/s' % (d['roofline']['kernel_ms'], d['value']))"
  done
done
