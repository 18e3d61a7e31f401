#!/usr/bin/env python3
"""GPU: the pair kernel's steady-state sweeps (fully observed series, L >= 24) against the CPU oracle,
entry by entry of theta, after 1, 2, 5, 40 iterations: python tools/steady_check.py [T ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import ldsr_amd  # noqa: E402
from ldsr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

names = lambda p, q: ["A"] + ["B%d" % i for i in range(p)] + ["C"] + ["D%d" % i for i in range(q)] + ["Q", "R", "mu1", "V1"]


def run(T, p, q, n, niter, algo=3):
    y, u, v = synth.make_series(T, p, q, series_id=100 + T)
    th0 = synth.make_init_packed(p, q, n, seed=T)
    ref_th, ref_lik, ref_it, _ = O.em_batch(y[None], u.T[None].copy(), v.T[None].copy(),
                                            np.zeros(n, np.int32), th0, niter, 0.0, n_threads=8)
    r = ldsr_amd.em_batch(y, u, v, th0, niter=niter, tol=0.0, algo=algo)
    rel = np.abs(r["theta"] - ref_th) / (np.abs(ref_th) + 1e-9)
    dl = np.abs(r["lik"] - ref_lik)
    worst = rel.max(axis=0)
    print("T=%d niter=%d cells=%d: max|dlik| %.2e  " % (T, niter, n, dl.max()) +
          " ".join("%s %.1e" % (nm, w) for nm, w in zip(names(p, q), worst)), flush=True)
    return rel.max() < 1e-6 and dl.max() < 1e-8


if __name__ == "__main__":
    Ts = [int(a) for a in sys.argv[1:]] or [768, 1000, 737, 800, 1024]
    ok = True
    for T in Ts:
        for niter in (2, 3, 5, 40):
            ok &= run(T, 1, 2, 21, niter)
    print("ALL OK" if ok else "FAILURES")
