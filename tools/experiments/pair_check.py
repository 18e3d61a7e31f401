#!/usr/bin/env python3
"""GPU: the two- and four-cells-per-wave kernels (algo 3, 4) against the CPU oracle and the scan kernel (algo 2)
on config-2 / config-5 shaped problems: python tools/pair_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (HIP runtime initialised by torch first)
import ldsr_amd  # noqa: E402
from ldsr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run(T, p, q, n, niter, tol, mask, seed=5):
    y, u, v = synth.make_series(T, p, q, series_id=seed)
    if mask == "paleo":
        y = y.copy(); y[: T - T // 10] = np.nan
    elif mask == "holes":
        y = y.copy(); y[::7] = np.nan; y[3] = np.nan
    th0 = synth.make_init_packed(p, q, n, seed=seed)
    ref_th, ref_lik, ref_it, _ = O.em_batch(y[None], u.T[None].copy(), v.T[None].copy(),
                                            np.zeros(n, np.int32), th0, niter, tol, n_threads=8)
    ok = True
    import ctypes
    algos = [2]
    for a in (3, 4):
        if ldsr_amd._lib.lib().ldsr_em_plan(T, p, q, max(niter, 2), tol, a, ctypes.create_string_buffer(8), 8) == a:
            algos.append(a)
    for algo in algos:
        t0 = time.time()
        r = ldsr_amd.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=algo)
        dt = time.time() - t0
        same_it = np.array_equal(r["n_iter"], ref_it)
        fin = np.isfinite(ref_lik)
        dth = np.abs(r["theta"] - ref_th)[fin]
        bar = (1e-6 * np.abs(ref_th) + 1e-9)[fin]
        dl = np.abs(r["lik"] - ref_lik)[fin]
        good = same_it and np.all(dth <= bar) and np.all(dl <= 1e-6 * np.abs(ref_lik[fin]) + 1e-9)
        ok &= bool(good)
        print("T=%d p=%d q=%d n=%d niter=%d tol=%g %-6s algo %d: n_iter %s  max|dtheta| %.2e  max|dlik| %.2e  status %s  %.3fs  %s"
              % (T, p, q, n, niter, tol, mask, algo, "same" if same_it else "DIFF %d" % np.sum(r["n_iter"] != ref_it),
                 dth.max() if dth.size else 0, dl.max() if dl.size else 0, np.bincount(r["status"]).tolist(), dt,
                 "ok" if good else "MISMATCH"), flush=True)
    return ok


if __name__ == "__main__":
    allok = True
    for (T, p, q) in ((1000, 1, 2), (813, 1, 3), (864, 2, 4), (992, 1, 1), (650, 2, 2), (832, 1, 2), (800, 3, 3), (257, 1, 2), (300, 2, 2), (400, 1, 4), (512, 4, 4), (65, 1, 1), (96, 1, 2), (100, 2, 4), (200, 3, 3), (256, 1, 2)):
        for mask in ("dense", "paleo", "holes"):
            allok &= run(T, p, q, 37, 40, 0.0, mask)
            allok &= run(T, p, q, 101, 300, 1e-5, mask)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
