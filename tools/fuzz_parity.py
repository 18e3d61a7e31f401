"""Randomized GPU-vs-oracle parity fuzz (run on the GPU box):  python tools/fuzz_parity.py [n] [seed]
Random T (2..8192: one, two and four waves per cell), p, q (1..8, sometimes absent), NA patterns, 1..5 series with own or shared
inputs, ragged cell counts, niter / tol; serial kernel, AUTO, the scan kernel and (where it applies) the pair kernel.  Prints one line per failing case and a
summary; exit status 1 if anything failed."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n):
        T = int(rng.choice([rng.integers(2, 40), rng.integers(40, 300), rng.integers(300, 1100), rng.integers(65, 513),
                            rng.integers(513, 1025), rng.integers(1100, 2049), rng.integers(2049, 4097), rng.integers(4097, 8193)]))
        p, q = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        if 65 <= T <= 1024 and rng.random() < 0.7:       # the pair / quad kernels' shapes
            p, q = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        S = int(rng.integers(1, 6))
        shared = bool(rng.integers(0, 2)) and S > 1
        has_u, has_v = rng.random() > 0.15, rng.random() > 0.15
        kind = rng.choice(["dense", "paleo", "scatter", "blocks"])
        Ys, Us, Vs = [], [], []
        for s in range(S):
            y, u, v = synth.make_series(max(T, 12), p, q, series_id=int(rng.integers(0, 10 ** 6)))
            y, u, v = y[:T].copy(), u[:, :T].copy(), v[:, :T].copy()
            if kind == "paleo":
                y[:int(T * rng.uniform(0.3, 0.9))] = np.nan
            elif kind == "scatter":
                y[rng.random(T) < rng.uniform(0.05, 0.6)] = np.nan
            elif kind == "blocks":
                for _ in range(3):
                    a = int(rng.integers(0, T))
                    y[a:a + int(rng.integers(1, max(2, T // 5)))] = np.nan
            Ys.append(y); Us.append(u); Vs.append(v)
        Y = np.stack(Ys)
        n_obs = np.sum(np.isfinite(Y), axis=1)
        if np.any(n_obs <= 2 * (q + 2)) or T <= 2 * (max(p, q) + 2):
            continue                                   # over-parameterised / singular: skip
        U = (Us[0] if shared else np.stack(Us)) if has_u else None
        V = (Vs[0] if shared else np.stack(Vs)) if has_v else None
        if (U is None) != (V is None) and not shared and S > 1:
            pass
        pe, qe = (p if has_u else 1), (q if has_v else 1)
        counts = rng.integers(0, 9, size=S)
        counts[int(rng.integers(0, S))] += 1
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        th0 = synth.make_init_packed(pe, qe, int(off[-1]), seed=int(rng.integers(0, 10 ** 6)))
        niter = int(rng.choice([2, 3, 10, 60, 200] if T <= 2048 else [2, 3, 10, 40]))
        tol = float(rng.choice([0.0, 1e-5, 1e-3]))
        soc = np.repeat(np.arange(S), counts).astype(np.int32)

        def tm(a):
            if a is None:
                return None
            a = np.asarray(a)
            if a.ndim == 2:
                a = np.repeat(a[None], S, axis=0)
            return np.ascontiguousarray(np.transpose(a, (0, 2, 1)))
        ref = O.em_batch(Y, tm(U), tm(V), soc, th0, niter, tol, n_threads=8)
        ok = np.isfinite(ref[1])
        import ctypes
        algos = [1, 0, 2] if T <= 8192 and max(pe, qe) <= 8 else [1, 0]
        for a in (3, 4):         # two / four cells per wave where they apply
            if ldsr_amd._lib.lib().ldsr_em_plan(T, pe, qe, max(niter, 2), tol, a, ctypes.create_string_buffer(8), 8) == a:
                algos.append(a)
        for algo in algos:     # serial, AUTO, scan kernel, pair kernel where it applies
            try:
                r = ldsr_amd.em_batch(Y, U, V, th0, cell_offsets=off, niter=niter, tol=tol, algo=algo)
            except Exception as e:   # noqa: BLE001
                print("case %d algo %d EXC %s  T=%d p=%d q=%d S=%d %s" % (case, algo, e, T, pe, qe, S, kind))
                bad += 1
                continue
            good = (np.array_equal(r["n_iter"][ok], ref[2][ok]) and close(r["lik"][ok], ref[1][ok])
                    and close(r["theta"][ok], ref[0][ok]))
            if not good:
                bad += 1
                d = np.nonzero(r["n_iter"][ok] != ref[2][ok])[0]
                print("case %d algo %d MISMATCH T=%d p=%d q=%d S=%d shared=%s %s niter=%d tol=%g cells=%s n_iter_diff=%s"
                      % (case, algo, T, pe, qe, S, shared, kind, niter, tol, counts.tolist(), d[:5].tolist()))
    print("fuzz: %d cases, %d failures" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
