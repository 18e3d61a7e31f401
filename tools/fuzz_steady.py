"""Randomized GPU-vs-oracle parity fuzz of the pair kernel's steady-state sweeps (run on the GPU box):
    python tools/fuzz_steady.py [n] [seed]
Fully observed series of 737..1024 steps (sometimes next to masked ones in the same call), p, q in
1..4 (sometimes absent), y scaled by 1e-6..1e6, ordinary restarts mixed with cells built to need the
fallback for a few or for many iterations (A near 1, tiny C), ragged cell counts, fixed iteration
counts and early stopping, the library's multi-device cut on one GPU.  Identical n_iter, theta and
lik within the parity bar; exit status 1 if anything failed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import ldsr_amd  # noqa: E402
from ldsr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def close(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(np.all((np.abs(a - b) <= 1e-6 * np.abs(b) + 1e-9) | (np.isnan(a) & np.isnan(b))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = ran = 0
    for case in range(n):
        T = int(rng.integers(737, 1025))
        p, q = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        S = int(rng.integers(1, 4))
        has_u, has_v = rng.random() > 0.15, rng.random() > 0.15
        scale = 10.0 ** rng.uniform(-6, 6) if rng.random() < 0.4 else 1.0
        Ys, Us, Vs = [], [], []
        for s in range(S):
            y, u, v = synth.make_series(T, p, q, series_id=int(rng.integers(0, 10 ** 6)))
            if s > 0 and rng.random() < 0.4:             # a masked series next to the dense ones
                y = y.copy()
                y[rng.random(T) < 0.2] = np.nan
            Ys.append(y * scale); Us.append(u); Vs.append(v)
        Y = np.stack(Ys)
        U = np.stack(Us) if has_u else None
        V = np.stack(Vs) if has_v else None
        pe, qe = (p if has_u else 1), (q if has_v else 1)
        counts = rng.integers(1, 40, size=S)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        nc = int(off[-1])
        th0 = synth.make_init_packed(pe, qe, nc, seed=int(rng.integers(0, 10 ** 6)))
        slow = rng.random(nc) < 0.3
        th0[slow, 0] = rng.uniform(0.9, 0.999, slow.sum())          # A
        th0[slow, 1 + pe] = rng.uniform(0.005, 0.2, slow.sum())     # C
        niter, tol = ((int(rng.integers(2, 60)), 0.0) if rng.random() < 0.5 else (400, 1e-5))
        Uo = np.ascontiguousarray(np.transpose(U, (0, 2, 1))) if has_u else None
        Vo = np.ascontiguousarray(np.transpose(V, (0, 2, 1))) if has_v else None
        soc = np.repeat(np.arange(S), counts).astype(np.int32)
        import ctypes
        if ldsr_amd._lib.lib().ldsr_em_plan(T, pe, qe, max(niter, 2), float(tol), 3, ctypes.create_string_buffer(8), 8) != 3:
            continue                          # wide inputs: the pair kernel ends earlier (8 waves per CU must fit)
        ref_th, ref_lik, ref_it, ref_st = O.em_batch(Y, Uo, Vo, soc, th0, niter, tol, n_threads=16)
        devs = [0, 0] if rng.random() < 0.3 else None
        yy = Y if S > 1 else Y[0]
        uu = (U if S > 1 else U[0]) if has_u else None
        vv = (V if S > 1 else V[0]) if has_v else None
        r = ldsr_amd.em_batch(yy, uu, vv, th0, cell_offsets=off, niter=niter, tol=tol, algo=3, devices=devs)
        ran += 1
        ok = np.array_equal(r["n_iter"], ref_it) and np.array_equal(r["status"], ref_st) and close(r["lik"], ref_lik) and close(r["theta"], ref_th)
        if not ok:
            bad += 1
            print("FAIL case %d: T=%d p=%d q=%d S=%d u=%d v=%d scale=%.1e niter=%d tol=%g cells=%d devs=%s  n_iter differ: %d"
                  % (case, T, pe, qe, S, has_u, has_v, scale, niter, tol, nc, devs, int(np.sum(r["n_iter"] != ref_it))), flush=True)
    print("%d cases drawn, %d ran on the pair kernel, %d failed" % (n, ran, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
