"""Randomized GPU-vs-oracle parity of the closed-form lead (run on the GPU box with LDSR_FORCE_FILL=1 so
that AUTO takes the LEAD forms for small launches):  LDSR_FORCE_FILL=1 python tools/fuzz_lead.py [n] [seed]
Random T (300..6000), p, q (1..8, absent sometimes), a common lead of >= 192 steps with per-series
extras, tails of 80..512 steps with holes, 1..4 series, ragged cell counts, niter / tol.  Prints one
line per failing case and how many cases ran on which LEAD member; exit status 1 on any failure."""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import synth, _lib  # noqa: E402
from oracle import oracle as O  # noqa: E402


def close(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return bool(np.all((np.abs(a - b) <= 1e-6 * np.abs(b) + 1e-9) | (np.isnan(a) & np.isnan(b))))


def last_kernel():
    buf = C.create_string_buffer(160)
    _lib.lib().ldsr_last_em_kernel(0, buf, 160)
    return buf.value.decode()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    ran = collections.Counter()
    for case in range(n):
        tail = int(rng.integers(40, 500))
        T = tail + int(rng.choice([rng.integers(200, 800), rng.integers(800, 2500), rng.integers(2500, 5500)]))
        p, q = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        S = int(rng.integers(1, 5))
        has_u, has_v = rng.random() > 0.1, rng.random() > 0.1
        Ys, Us, Vs = [], [], []
        for s in range(S):
            y, u, v = synth.make_series(T, p, q, series_id=int(rng.integers(0, 10 ** 6)))
            y = y.copy()
            y[:T - tail + int(rng.integers(0, 12)) * (s > 0)] = np.nan
            if rng.random() < 0.5:
                a = T - tail + int(rng.integers(5, max(6, tail - 20)))
                y[a:a + int(rng.integers(1, 15))] = np.nan
            if rng.random() < 0.3:
                y[T - 1] = np.nan
            Ys.append(y); Us.append(u); Vs.append(v)
        Y = np.stack(Ys)
        if np.any(np.sum(np.isfinite(Y), axis=1) <= 2 * (q + 3)):
            continue
        U = np.stack(Us) if has_u else None
        V = np.stack(Vs) if has_v else None
        pe, qe = (p if has_u else 1), (q if has_v else 1)
        counts = rng.integers(1, 7, size=S)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        th0 = synth.make_init_packed(pe, qe, int(off[-1]), seed=int(rng.integers(0, 10 ** 6)))
        niter = int(rng.choice([3, 10, 60, 200]))
        tol = float(rng.choice([0.0, 1e-5, 1e-3]))
        soc = np.repeat(np.arange(S), counts).astype(np.int32)
        tm = lambda a: None if a is None else np.ascontiguousarray(np.transpose(a, (0, 2, 1)))   # noqa: E731
        ref = O.em_batch(Y, tm(U), tm(V), soc, th0, niter, tol, n_threads=8)
        ok = np.isfinite(ref[1])
        try:
            r = ldsr_amd.em_batch(Y, U, V, th0, cell_offsets=off, niter=niter, tol=tol)
        except Exception as e:   # noqa: BLE001
            print("case %d EXC %s  T=%d p=%d q=%d S=%d tail=%d" % (case, e, T, pe, qe, S, tail))
            bad += 1
            continue
        name = last_kernel()
        ran[name.split("<")[0] + ("+LEAD " if name.endswith(", true>") else " ") +
            ("wide" if max(pe, qe) > 4 else "narrow")] += 1
        good = (np.array_equal(r["n_iter"][ok], ref[2][ok]) and close(r["lik"][ok], ref[1][ok])
                and close(r["theta"][ok], ref[0][ok]))
        if not good:
            bad += 1
            print("case %d MISMATCH %s T=%d p=%d q=%d S=%d tail=%d niter=%d tol=%g cells=%s"
                  % (case, name, T, pe, qe, S, tail, niter, tol, counts.tolist()))
    for k, v in sorted(ran.items()):
        print("  %-40s %d" % (k, v))
    print("fuzz_lead: %d cases, %d failures" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
