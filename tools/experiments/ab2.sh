#!/bin/bash
# Usage (GPU box): tools/ab2.sh "<bench args>" lib1.so lib2.so [...]  -- interleaved rounds, same box
A=$1; shift
for r in 1 2 3; do
  for so in "$@"; do
    LDSR_HIP_SO=$PWD/$so python bench.py --steps 20 --warmup 3 --no-cpu-baseline $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$so', d['roofline']['kernel'], '%.4f ms kernel  %.4g units/s' % (d['roofline']['kernel_ms'], d['value']))"
  done
done
