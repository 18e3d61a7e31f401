#!/usr/bin/env python3
"""GPU: the closed-form lead of the pair family (AUTO on paleo-type series through the host entry)
against the CPU oracle and the scan kernel:  python tools/lead_check.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import synth, _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402


def close(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(np.all((np.abs(a - b) <= 1e-6 * np.abs(b) + 1e-9) | (np.isnan(a) & np.isnan(b))))


def run(T, p, q, n, lead, niter, tol, holes=False, S=1):
    Y = np.empty((S, T)); U = np.empty((S, T, p)); V = np.empty((S, T, q))
    for s in range(S):
        y, u, v = synth.make_series(T, p, q, series_id=40 + s)
        y = y.copy(); y[: lead + 3 * s] = np.nan
        if holes:
            y[lead + 20: lead + 30] = np.nan
        Y[s] = y; U[s] = u.T; V[s] = v.T
    counts = np.full(S, n // S)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    th0 = synth.make_init_packed(p, q, int(off[-1]), seed=T + lead)
    soc = np.repeat(np.arange(S), counts).astype(np.int32)
    buf = ctypes.create_string_buffer(160)
    _lib.lib().ldsr_em_plan_lead(T, p, q, max(niter, 2), tol, 0, lead, buf, 160)
    ref = O.em_batch(Y, U, V, soc, th0, niter, tol, n_threads=16)
    ok = np.isfinite(ref[1])
    good_all = True
    for algo in (0, 2):
        t0 = time.time()
        r = ldsr_amd.em_batch(Y if S > 1 else Y[0], np.transpose(U, (0, 2, 1)).copy() if S > 1 else U[0].T.copy(),
                              np.transpose(V, (0, 2, 1)).copy() if S > 1 else V[0].T.copy(), th0,
                              cell_offsets=off, niter=niter, tol=tol, algo=algo)
        dt = time.time() - t0
        same = np.array_equal(r["n_iter"][ok], ref[2][ok])
        dth = np.abs(r["theta"] - ref[0])[ok]
        good = same and close(r["lik"][ok], ref[1][ok]) and close(r["theta"][ok], ref[0][ok])
        good_all &= bool(good)
        print("T=%d p=%d q=%d S=%d n=%d lead=%d niter=%d tol=%g holes=%d algo %d [%s]: n_iter %s max|dtheta| %.2e %.3fs %s"
              % (T, p, q, S, off[-1], lead, niter, tol, holes, algo, buf.value.decode() if algo == 0 else "scan",
                 "same" if same else "DIFF %d" % np.sum(r["n_iter"][ok] != ref[2][ok]),
                 dth.max() if dth.size else 0, dt, "ok" if good else "MISMATCH"), flush=True)
    return good_all


if __name__ == "__main__":
    allok = True
    allok &= run(1000, 1, 2, 8192, 900, 30, 0.0)
    allok &= run(1000, 1, 2, 8192, 900, 300, 1e-5)
    allok &= run(2000, 1, 4, 8192, 1800, 20, 0.0, holes=True)
    allok &= run(813, 1, 3, 8192, 730, 30, 0.0, S=4)
    allok &= run(813, 2, 2, 8192, 700, 200, 1e-5, S=2)
    allok &= run(600, 1, 1, 8192, 400, 25, 0.0)
    allok &= run(813, 3, 3, 8192, 760, 40, 0.0)           # the bundled Nakhon Phanom shape (46 of 813 observed)
    allok &= run(813, 3, 3, 8192, 760, 300, 1e-5)
    allok &= run(1200, 4, 1, 8192, 1000, 30, 0.0, holes=True)
    allok &= run(1000, 1, 2, 8192, 520, 300, 1e-5, holes=True)   # tails of 257..512 steps
    allok &= run(2000, 1, 4, 8192, 1500, 30, 0.0, S=3)
    allok &= run(4000, 2, 2, 8192, 3500, 200, 1e-5)
    allok &= run(8000, 1, 2, 8192, 7800, 8, 0.0)          # a very long lead (scan kernel: four waves per cell)
    allok &= run(4000, 2, 4, 8192, 3790, 60, 1e-5, holes=True)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
