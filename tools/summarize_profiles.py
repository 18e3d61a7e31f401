#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_TAG/<workload>/...) into
the committed summaries:  profiles/ROUND_<workload>_{kernel_stats,pmc_fetch,pmc_write,pmc_sq,pmc_sq2}.csv
(EM-kernel and series_prep rows only), profiles/ROUND_<workload>_bench*.json and the entries of
profiles/pmc_summary.json that bench.py reads for roofline.traffic / issue_frac.

    python tools/summarize_profiles.py gpurun_out/prof_r2a r02

HBM bytes: (FETCH_SIZE + WRITE_SIZE) x 1024 per the CDNA guide; on gfx950 FETCH_SIZE reports half
the bytes of WIDE (16 B/lane) streaming reads.  This kernel's global reads are 8 B/lane, an
uncalibrated width, so both the raw figure and the doubled-read upper bound are recorded and
bench.py reports the upper bound."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(files):
    """gpurun merges a call's output into the local gpurun_out/: a tag used twice leaves two runs' files side
    by side (rocprofv3 prefixes them with the pid).  Only the newest run counts."""
    if not files:
        return files
    t = max(os.path.getmtime(f) for f in files)
    return [f for f in files if t - os.path.getmtime(f) < 30]


def counter_rows(d):
    rows = []
    for f in newest(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, name_part):
    agg = collections.defaultdict(list)
    disp = collections.defaultdict(set)
    for r in rows:
        if name_part in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    return {k: sum(v) / max(len(disp[k]), 1) for k, v in agg.items()}, rows


def write_filtered(rows, dst):
    keep = [r for r in rows if "em_scan" in r["Kernel_Name"] or "em_pair" in r["Kernel_Name"] or "em_serial" in r["Kernel_Name"]
            or "series_prep" in r["Kernel_Name"]]
    if not keep:
        return
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
        w.writeheader()
        w.writerows(keep)


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    summ_path = os.path.join(prof, "pmc_summary.json")
    try:
        summ = json.load(open(summ_path))
    except (OSError, ValueError):
        summ = {}
    entries = [e for e in summ.get("entries", []) if e.get("round") != rnd]
    for wdir in sorted(glob.glob(os.path.join(src, "cfg*"))):
        w = os.path.basename(wdir)
        bench = json.loads(open(os.path.join(wdir, "bench.json")).read().strip().splitlines()[-1])
        under = json.loads(open(os.path.join(wdir, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        kernel = bench["roofline"]["kernel"]
        stats = newest(glob.glob(os.path.join(wdir, "trace", "*", "*_kernel_stats.csv")))
        if stats:
            shutil.copy(stats[0], os.path.join(prof, "%s_%s_kernel_stats.csv" % (rnd, w)))
        json.dump(bench, open(os.path.join(prof, "%s_%s_bench.json" % (rnd, w)), "w"))
        json.dump(under, open(os.path.join(prof, "%s_%s_bench_under_rocprof.json" % (rnd, w)), "w"))
        vals = {}
        for sub in ("fetch", "write", "sq", "sq2", "sq3"):
            rows = counter_rows(os.path.join(wdir, sub))
            v, _ = per_kernel(rows, kernel.split("<")[0])
            vals.update(v)
            write_filtered(rows, os.path.join(prof, "%s_%s_pmc_%s.csv" % (rnd, w, sub)))
        units = bench["roofline"]["units_per_launch"]
        rocprof_ms = None
        if stats:
            for r in csv.DictReader(open(stats[0])):
                if kernel.split("<")[0] in r["Name"]:
                    rocprof_ms = float(r["AverageNs"]) / 1e6
        fetch, write = vals.get("FETCH_SIZE", 0.0), vals.get("WRITE_SIZE", 0.0)
        # sustained clock while the EM kernel runs: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 /
        # dispatch duration, of the timed dispatches of the sq2 pass (the first one is the warm-up)
        clocks = []
        for r in counter_rows(os.path.join(wdir, "sq2")):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and kernel.split("<")[0] in r["Kernel_Name"]:
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                if dur > 0:
                    clocks.append(float(r["Counter_Value"]) / 8.0 / dur)
        clocks = clocks[1:] if len(clocks) > 1 else clocks
        e = {
            "round": rnd, "workload": w, "mask": "dense", "kernel": kernel,
            "niter": bench["config"]["niter"], "tol": bench["config"]["tol"],
            "units_per_launch": units,
            "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
            "hbm_bytes_per_launch_raw": (fetch + write) * 1024,
            "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
            "valu_insts_per_unit": vals.get("SQ_INSTS_VALU", 0.0) / units,
            "lds_insts_per_unit": vals.get("SQ_INSTS_LDS", 0.0) / units,
            # executed fp64 arithmetic: wave instructions x 64 lanes, an FMA counts two flops
            "fp64_insts_per_unit": {k: vals.get("SQ_INSTS_VALU_%s_F64" % k, 0.0) / units
                                    for k in ("FMA", "MUL", "ADD", "TRANS")},
            "fp64_flops_executed_per_unit": 64.0 * (2 * vals.get("SQ_INSTS_VALU_FMA_F64", 0.0)
                                                    + vals.get("SQ_INSTS_VALU_MUL_F64", 0.0)
                                                    + vals.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                                                    + vals.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) / units,
            "lds_issue_stall_per_wave_cycle": (vals["SQ_WAIT_INST_LDS"] / vals["SQ_WAVE_CYCLES"])
            if vals.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_LDS" in vals else None,
            "waves": vals.get("SQ_WAVES"),
            "valu_active_per_wave_cycle": (vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"])
            if vals.get("SQ_WAVE_CYCLES") else None,
            "wait_inst_any_per_wave_cycle": (vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"])
            if vals.get("SQ_WAVE_CYCLES") else None,
            "wait_any_per_wave_cycle": (vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"])
            if vals.get("SQ_WAVE_CYCLES") else None,
            "lds_bank_conflict_cycles": vals.get("SQ_LDS_BANK_CONFLICT"),
            "lds_idx_active_cycles": vals.get("SQ_LDS_IDX_ACTIVE"),
            "sustained_clock_ghz": (sum(clocks) / len(clocks)) if clocks else None,
            "rocprof_kernel_ms": rocprof_ms,
            "bench_kernel_ms_same_command": under["roofline"]["kernel_ms"],
            "bench_kernel_ms_unprofiled": bench["roofline"]["kernel_ms"],
            "source": ["profiles/%s_%s_%s" % (rnd, w, s) for s in
                       ("kernel_stats.csv", "pmc_fetch.csv", "pmc_write.csv", "pmc_sq.csv", "pmc_sq2.csv",
                        "pmc_sq3.csv")],
        }
        entries.append(e)
        print(w, kernel, "rocprof %.3f ms, bench (same command) %.3f ms, unprofiled %.3f ms; VALU/unit %.0f, "
              "LDS/unit %.0f, HBM %.0f..%.0f KB/launch" % (rocprof_ms or -1, e["bench_kernel_ms_same_command"],
                                                          e["bench_kernel_ms_unprofiled"], e["valu_insts_per_unit"],
                                                          e["lds_insts_per_unit"], e["hbm_bytes_per_launch_raw"] / 1e3,
                                                          e["hbm_bytes_per_launch"] / 1e3))
    # the other mask and the converged runs (tools/collect_profiles.sh "extra"): one unprofiled bench line each
    for f in sorted(glob.glob(os.path.join(src, "extra", "*_bench.json"))):
        try:
            line = json.loads(open(f).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            print("skipped", f)
            continue
        json.dump(line, open(os.path.join(prof, "%s_%s" % (rnd, os.path.basename(f))), "w"))
        print(os.path.basename(f), "%.4g units/s, %.3f ms/step, kernel %.3f ms %s"
              % (line["value"], line["ms_per_step"], line["roofline"]["kernel_ms"], line["roofline"]["kernel"]))
    summ = {"_about": "rocprofv3 --pmc passes of `python3 bench.py --workload W --steps 3 --warmup 1 "
                      "--no-cpu-baseline` on MI355X, summarised by tools/summarize_profiles.py; per-launch "
                      "averages of the EM kernel.  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                      "(upper bound: gfx950 halves FETCH_SIZE for wide reads; _raw is the undoubled value).",
            "entries": entries}
    json.dump(summ, open(summ_path, "w"), indent=1)


if __name__ == "__main__":
    main()
