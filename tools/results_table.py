#!/usr/bin/env python3
"""Write the round's measured table into README.md and BASELINE.md (between the `<!-- results:begin -->` /
`<!-- results:end -->` markers) from the committed evidence, so that every number in those tables has a
file behind it:

    python tools/results_table.py r04

reads profiles/ROUND_cfg{2..5}_bench.json (the unprofiled bench line of tools/collect_profiles.sh),
profiles/pmc_summary.json (rocprofv3 averages, counters, sustained clock of the same round) and the extra
lines profiles/ROUND_*_{paleo,conv,conv_paleo}_bench.json."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABEL = {
    "cfg2": "cfg2 T=1000 p=1 q=2, 4096 restarts — **the metric's config**",
    "cfg3": "cfg3 T=1000 p=4 q=8, 8192 restarts",
    "cfg4": "cfg4 cvLDS grid 10 × 1024, T=2000 p=1 q=4",
    "cfg5": "cfg5 48 series × 512, T=813 p=1 q=3",
}


def short(kernel):
    return kernel.replace("em_pair_kernel", "pair").replace("em_scan_kernel", "scan")


def line_of(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def table(rnd):
    prof = os.path.join(ROOT, "profiles")
    pmc = {e["workload"]: e for e in json.load(open(os.path.join(prof, "pmc_summary.json")))["entries"] if e["round"] == rnd}
    out = ["Round %s, one MI355X, `python bench.py --workload W` (niter = 100, tol = 0, fully observed series for cfg2 / cfg3, the "
           "configs' own masks for cfg4 / cfg5; boxes differ by ±4 %%).  Sources: `profiles/%s_cfgN_bench.json` (value, HIP-event kernel time, "
           "fractions), `profiles/%s_cfgN_kernel_stats.csv` (rocprofv3 average), `profiles/pmc_summary.json` (counters, clock).\n" % (rnd, rnd, rnd),
           "| workload | kernel | restart×EM-iter/s (whole job) | EM kernel ms: HIP events / rocprofv3 | `roofline.frac` | at sustained clock (GHz) | executed fp64 ÷ peak | VALU instr. per unit | HBM MB per launch (PMC, upper bound) |",
           "|---|---|---|---|---|---|---|---|---|"]
    host = cpu = None
    for w in ("cfg2", "cfg3", "cfg4", "cfg5"):
        f = os.path.join(prof, "%s_%s_bench.json" % (rnd, w))
        if not os.path.exists(f):
            continue
        d, r, e = line_of(f), line_of(f)["roofline"], pmc.get(w, {})
        out.append("| %s | `%s` | %.3g | %.3f / %.3f | %.3f | %s | %s | %.0f | %.1f |" % (
            LABEL[w], short(r["kernel"]), d["value"], r["kernel_ms"], e.get("rocprof_kernel_ms") or float("nan"), r["frac"],
            ("%.3f (%.2f)" % (r["frac_at_sustained_clock"], r["sustained_clock_ghz"])) if r.get("frac_at_sustained_clock") else "—",
            ("%.3f" % r["fp64_executed_frac"]) if r.get("fp64_executed_frac") else "—",
            e.get("valu_insts_per_unit", float("nan")), e.get("hbm_bytes_per_launch", float("nan")) / 1e6))
        if w == "cfg2":
            host, cpu = d.get("host_entry"), d.get("cpu_baseline")
            full = os.path.join(prof, "%s_cfg2_full_bench.json" % rnd)       # the driver's default invocation
            if os.path.exists(full):
                cpu = line_of(full).get("cpu_baseline") or cpu
    out.append("")
    extras = sorted(glob.glob(os.path.join(prof, "%s_cfg*_*_bench.json" % rnd)))
    extras = [f for f in extras if not f.endswith("_bench_under_rocprof.json")]
    if extras:
        out.append("The other mask and the runs to convergence (`--niter 1000 --tol 1e-5`: the reference's normal mode of use; a unit is an "
                   "iteration actually executed), one unprofiled bench line each:\n")
        out.append("| file | kernel | restart×EM-iter/s | ms per step | EM kernel ms |")
        out.append("|---|---|---|---|---|")
        for f in extras:
            d = line_of(f)
            out.append("| `profiles/%s` | `%s` | %.3g | %.3f | %.3f |" % (os.path.basename(f), short(d["roofline"]["kernel"]), d["value"],
                                                                      d["ms_per_step"], d["roofline"]["kernel_ms"]))
        out.append("")
    if host:
        out.append("cfg2 through the host-pointer entry the R shim calls (`host_entry`: PCIe in and out, the lead found from `y` by the library): "
                   "%.3g units/s, %.3f ms per call = %.2f of the device-resident rate.%s" % (
                       host["value"], host["ms_per_call"], host["vs_device_resident"],
                       ("  CPU baseline (`cpu_baseline`, the oracle on %d host cores, kind \"%s\"): %.3g units/s." % (
                           cpu["cores"], cpu["kind"], cpu["value"])) if cpu else ""))
        out.append("")
    return "\n".join(out)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    block = table(rnd)
    for name in ("README.md", "BASELINE.md"):
        p = os.path.join(ROOT, name)
        s = open(p).read()
        a, b = "<!-- results:begin -->", "<!-- results:end -->"
        if a not in s or b not in s:
            print(name, "has no results markers")
            continue
        s = s[:s.index(a) + len(a)] + "\n" + block + s[s.index(b):]
        open(p, "w").write(s)
        print("wrote", name)


if __name__ == "__main__":
    main()
