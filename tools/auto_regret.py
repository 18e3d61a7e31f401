#!/usr/bin/env python3
"""GPU box: how far is LDSR_ALGO_AUTO from the best explicit algorithm?  One process, device-resident
operands (ldsr_em_batch_device_lead, the bench's entry), shapes ON and OFF the BASELINE configs:
    python tools/auto_regret.py > gpurun_out/auto_regret.txt
For every (T, p, q, cells, mask, tol) it times AUTO and LDSR_ALGO_SCAN / PAIR / QUAD (where the
shape is supported), ms per call = series preparation + EM kernel, median of 5 after 2 warm-ups, and
prints the regret of AUTO = t(AUTO) / min over explicit algorithms.  Rows above 1.10 are flagged."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ldsr_amd import _lib, synth  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
NAMES = {0: "auto", 2: "scan", 3: "pair", 4: "quad"}


def problem(T, p, q, n, mask):
    y, u, v = synth.make_series(T, p, q, series_id=7, mask=mask)
    th0 = synth.make_init_packed(p, q, n, seed=3)
    return (y[None], np.ascontiguousarray(u.T[None]), np.ascontiguousarray(v.T[None]), 0,
            np.array([0, n], np.int32), th0, n)


def time_algo(prob, T, p, q, niter, tol, algo):
    if L.ldsr_em_workspace_bytes(1, T, p, q, prob[6], algo) == 0:
        return None, ""
    try:
        job = bench.Job(L, torch, dev, 0, prob, T, p, q, niter, tol, algo)
        for _ in range(2):
            job.step()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            job.step()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        buf = C.create_string_buffer(160)
        L.ldsr_last_em_kernel(0, buf, 160)
        return 1e3 * float(np.median(ts)), buf.value.decode()
    except Exception as e:      # noqa: BLE001 -- unsupported (shape, algo) combinations are expected
        return None, "error: %s" % str(e)[:60]


def main():
    shapes = [(150, 1, 2), (400, 1, 2), (600, 1, 2), (813, 3, 3), (1000, 1, 2), (1000, 4, 8), (1500, 1, 2),
              (3000, 1, 2), (600, 2, 4), (260, 4, 4)]
    worst = []
    print("%-22s %6s %-6s %-6s | %9s %9s %9s %9s | regret  AUTO's kernel" % ("T,p,q", "cells", "mask", "tol", "auto", "scan", "pair", "quad"))
    # BASELINE config 2's own launch size -- ONE device-filling round of 4096 cells -- was missing from the
    # round-3 table, and it is where the steady form's fallback tail sat
    cells_of = {(1000, 1, 2): (200, 2000, 4096, 20000), (1000, 4, 8): (200, 2000, 8192, 20000)}
    if os.environ.get("REGRET_ONLY_BASELINE_SIZES"):
        shapes = [(1000, 1, 2), (1000, 4, 8)]
        cells_of = {(1000, 1, 2): (4096,), (1000, 4, 8): (8192,)}
    for (T, p, q) in shapes:
        for n in cells_of.get((T, p, q), (200, 2000, 20000)):
            for mask in ("dense", "paleo"):
                for tol, niter in ((0.0, 50), (1e-5, 300)) if n not in (4096, 8192) else ((0.0, 100), (1e-5, 1000)):
                    prob = problem(T, p, q, n, mask)
                    # two interleaved passes, the faster one counts (the first job after an idle spell
                    # runs at lower clocks: a single pass charged that to whatever came first -- AUTO)
                    res = {}
                    for _pass in range(2):
                        for a in (2, 3, 4, 0):
                            t, nm = time_algo(prob, T, p, q, niter, tol, a)
                            if a not in res or (t is not None and (res[a][0] is None or t < res[a][0])):
                                res[a] = (t, nm)
                    best = min(t for a, (t, _) in res.items() if a != 0 and t is not None)
                    auto = res[0][0]
                    regret = auto / best
                    row = "%-22s %6d %-6s %-6g | %9s %9s %9s %9s | %5.2f%s  %s" % (
                        "%d,%d,%d" % (T, p, q), n, mask, tol,
                        *["%.3f" % res[a][0] if res[a][0] is not None else "-" for a in (0, 2, 3, 4)],
                        regret, " <<<" if regret > 1.10 else "", res[0][1])
                    print(row, flush=True)
                    if regret > 1.10:
                        worst.append(row)
    print("\nrows with AUTO more than 10 %% off the best explicit algorithm: %d" % len(worst))
    for r in worst:
        print(r)


if __name__ == "__main__":
    main()
