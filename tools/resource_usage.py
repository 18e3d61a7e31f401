#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in a HIP translation unit (gfx950):
    python tools/resource_usage.py ldsr_amd/csrc/em_scan_L16.hip [name-filter]
Parses hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed)."""
import re
import subprocess
import sys


def table(src, flt=""):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
           "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        mangled = b.split()[0]
        name = subprocess.run(["c++filt", mangled], capture_output=True,
                              text=True).stdout.strip()
        if flt and flt not in name:
            continue

        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((name, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("ScratchSize [bytes/lane]"),
                     g("Occupancy [waves/SIMD]"), g("TotalSGPRs"), g("SGPRs Spill")))
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print("%-64s %5s %5s %6s %8s %4s %5s %6s" % ("kernel", "VGPR", "AGPR", "vspill", "scratchB", "occ", "SGPR", "sspill"))
    for r in rows:
        print("%-64s %5d %5d %6d %8d %4d %5d %6d" % r)
