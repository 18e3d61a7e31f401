#!/bin/bash
# round 4, GPU session 33: long fuzz runs on the round's final build (rings in every chunk length, global image, wide long chunks)
out=gpurun_out/r4s33; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "from ldsr_amd import _lib; print(_lib.lib().ldsr_version().decode())" 2>/dev/null | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 1000 python tools/fuzz_parity.py 3000 801 2>&1 | tail -5 | tee -a $out/summary.txt
timeout -k 10 1000 python tools/fuzz_parity.py 3000 802 2>&1 | tail -5 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 800 python tools/fuzz_lead.py 2000 803 2>&1 | tail -3 | tee -a $out/summary.txt
timeout -k 10 600 python tools/fuzz_steady.py 1000 804 2>&1 | tail -3 | tee -a $out/summary.txt
