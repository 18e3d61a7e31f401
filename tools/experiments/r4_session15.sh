#!/bin/bash
# round 4, GPU session 15/18: where a LONE wave of the closed-form-lead kernel (four cells per wave) spends its iteration
out=gpurun_out/r4s18; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LDSR_HIP_SO=$PWD/ldsr_amd/libldsr_hip_timing.so LDSR_FORCE_FILL=1
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $out/summary.txt
import sys, os
sys.path.insert(0, "tools")
os.environ.pop("SECTIONS_PAIR", None)
import scan_sections as S
for cells in (4, 64, 24576):
    S.run_lead(813, 1, 3, cells, 717)
for cells in (4, 16, 10240):
    S.run_lead(2000, 1, 4, cells, 1800)
PY
