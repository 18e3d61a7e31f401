#!/bin/bash
# round 4, GPU session 13: long fuzz runs on the round's final build (random shapes, masks, leads, launch sizes against the oracle)
out=gpurun_out/r4s13; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "from ldsr_amd import _lib; print(_lib.lib().ldsr_version().decode())" 2>/dev/null | tee -a $out/summary.txt
timeout -k 10 1000 python tools/fuzz_steady.py 1500 501 2>&1 | tail -3 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 1000 python tools/fuzz_parity.py 1500 502 2>&1 | tail -6 | tee -a $out/summary.txt
timeout -k 10 700 python tools/fuzz_parity.py 800 503 2>&1 | tail -6 | tee -a $out/summary.txt
LDSR_FORCE_FILL=1 timeout -k 10 800 python tools/fuzz_lead.py 800 504 2>&1 | tail -3 | tee -a $out/summary.txt
