#!/usr/bin/env python3
"""GPU: many-series ragged grids (0..70 cells per series, own inputs and masks per series) on the
pair / quad / scan kernels against the CPU oracle:  python tools/grid_stress.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def close(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(np.all((np.abs(a - b) <= 1e-6 * np.abs(b) + 1e-9) | (np.isnan(a) & np.isnan(b))))


def main():
    rng = np.random.default_rng(5)
    bad = 0
    for (T, p, q, algos) in ((700, 1, 2, (2, 3)), (300, 2, 3, (2, 3, 4)), (97, 1, 1, (2, 3, 4)), (1024, 1, 2, (2, 3))):
        S = 64
        Y = np.empty((S, T)); U = np.empty((S, T, p)); V = np.empty((S, T, q))
        for s in range(S):
            y, u, v = synth.make_series(T, p, q, series_id=1000 + s)
            kind = s % 4
            if kind == 1:
                y[: int(T * 0.8)] = np.nan
            elif kind == 2:
                y[rng.random(T) < 0.3] = np.nan
            elif kind == 3:
                y[T // 2: T // 2 + T // 5] = np.nan
            Y[s] = y; U[s] = u.T; V[s] = v.T
        counts = rng.integers(0, 71, size=S)
        counts[3] = 0; counts[10] = 1
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        n = int(off[-1])
        th0 = synth.make_init_packed(p, q, n, seed=T)
        soc = np.repeat(np.arange(S), counts).astype(np.int32)
        for niter, tol in ((15, 0.0), (400, 1e-5)):
            ref = O.em_batch(Y, U, V, soc, th0, niter, tol, n_threads=16)
            ok = np.isfinite(ref[1])
            for algo in algos:
                r = ldsr_amd.em_batch(Y, np.transpose(U, (0, 2, 1)).copy(), np.transpose(V, (0, 2, 1)).copy(), th0,
                                      cell_offsets=off, niter=niter, tol=tol, algo=algo)
                good = (np.array_equal(r["n_iter"][ok], ref[2][ok]) and close(r["lik"][ok], ref[1][ok])
                        and close(r["theta"][ok], ref[0][ok]))
                print("T=%d p=%d q=%d %d cells in %d series niter=%d tol=%g algo %d: %s" %
                      (T, p, q, n, S, niter, tol, algo, "ok" if good else "MISMATCH"), flush=True)
                bad += 0 if good else 1
    print("grid stress: %d failures" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
