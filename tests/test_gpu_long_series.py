"""GPU parity of the multi-wave scan kernel (2 or 4 wavefronts per cell, T in 2049..8192) and of
the FIT form of the scan kernel (Kalman_smoother / penalized_likelihood on the scan path),
against the oracle.  The reference handles any T with one sequential sweep (src/EM.cpp:70,99)."""
import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 1e-9
SCAN = 2


def _oracle_batch(y, u, v, th0, niter, tol):
    from oracle import oracle as O
    U = None if u is None else np.ascontiguousarray(u.T[None])
    V = None if v is None else np.ascontiguousarray(v.T[None])
    return O.em_batch(y[None], U, V, np.zeros(th0.shape[0], np.int32), th0, niter, tol, n_threads=8)


@pytest.mark.parametrize("T,p,q,mask", [
    (2049, 1, 2, "dense"),     # W=2, L=20: second wave only partly active
    (2600, 2, 3, "paleo"),     # W=2, L=24
    (4096, 1, 2, "paleo"),     # W=2, L=32, every lane full
    (4097, 1, 1, "dense"),     # W=4, L=20
    (6000, 1, 4, "paleo"),     # W=4, L=24
    (8000, 3, 2, "dense"),     # W=4, L=32
    (8192, 1, 2, "paleo"),     # the largest supported length
])
def test_long_series_on_the_multi_wave_scan_kernel(T, p, q, mask):
    import ldsr_amd
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, p, q, series_id=77, mask=mask)
    th0 = synth.make_init_packed(p, q, 10, seed=9)
    ref = _oracle_batch(y, u, v, th0, 40, 1e-5)
    r = ldsr_amd.em_batch(y, u, v, th0, niter=40, tol=1e-5, algo=SCAN)
    assert np.array_equal(r["n_iter"], ref[2]), (r["n_iter"], ref[2])
    assert parity_close(r["lik"], ref[1], RTOL, ATOL)
    assert parity_close(r["theta"], ref[0], RTOL, ATOL)
    # AUTO picks the scan kernel for fully observed series of these lengths and gives the very same
    # bits; a long all-missing lead it handles in closed form (the LEAD form of the pair family)
    a = ldsr_amd.em_batch(y, u, v, th0, niter=40, tol=1e-5)
    assert np.array_equal(a["n_iter"], r["n_iter"])
    if mask == "dense":
        assert np.array_equal(a["theta"], r["theta"])
    else:
        assert parity_close(a["lik"], ref[1], RTOL, ATOL) and parity_close(a["theta"], ref[0], RTOL, ATOL)
    # fixed iteration count (tol = 0) and the likelihood trace
    r0 = ldsr_amd.em_batch(y, u, v, th0[:3], niter=7, tol=0.0, algo=SCAN, return_liks=True)
    ref0 = _oracle_batch(y, u, v, th0[:3], 7, 0.0)
    assert np.all(r0["n_iter"] == 7) and parity_close(r0["theta"], ref0[0], RTOL, ATOL)
    assert parity_close(r0["liks"][:, -1], ref0[1], RTOL, ATOL)


def test_long_series_multi_series_grid_and_winner_fit():
    """Several long series with own inputs through the one-call restart entry: the multi-wave
    kernel under the work queue, then the FIT form for the winners."""
    import ldsr_amd
    from ldsr_amd import synth
    from oracle import oracle as O
    T, p, q, S, R = 2500, 1, 2, 3, 5
    ser = [synth.make_series(T, p, q, series_id=90 + s, mask="paleo", n_tail=300 + 50 * s) for s in range(S)]
    Y = np.stack([a[0] for a in ser])
    U = np.stack([a[1] for a in ser])
    V = np.stack([a[2] for a in ser])
    off = (np.arange(S + 1) * R).astype(np.int32)
    th0 = synth.make_init_packed(p, q, S * R, seed=12)
    r = ldsr_amd.em_restart_grid(Y, U, V, th0, cell_offsets=off, niter=30, tol=1e-5)
    for s in range(S):
        refs = [O.lds_em(Y[s], U[s], V[s], th0[c], 30, 1e-5) for c in range(off[s], off[s + 1])]
        k = O.select(np.array([m["lik"] for m in refs]), np.array([m["theta"][1 + p] for m in refs]))
        assert r["winner"][s] == off[s] + k
        assert r["n_iter"][s] == len(refs[k]["liks"])
        assert parity_close(r["theta"][s], refs[k]["theta"], RTOL, ATOL)
        for name in "XYVJ":
            assert parity_close(r[name][s], refs[k]["fit"][name], RTOL, ATOL), name
        assert parity_close(r["liks"][s, :r["n_iter"][s]], refs[k]["liks"], RTOL, ATOL)


@pytest.mark.parametrize("T,p,q,mask", [(85, 7, 7, "p1"), (813, 3, 3, "paleo"), (1000, 4, 8, "dense"),
                                        (1500, 1, 4, "paleo"), (3000, 2, 2, "dense"), (7000, 1, 2, "paleo")])
def test_fit_form_of_the_scan_kernel_matches_oracle_smoother(T, p, q, mask, p1case):
    """Kalman_smoother on the scan path: X, Y, V, J (including J[T-1] of src/EM.cpp:98), lik
    with and without stdlik, and penalized_likelihood (R/LDS_GA.R:28-44)."""
    import ldsr_amd
    from ldsr_amd import synth
    from oracle import oracle as O
    if mask == "p1":
        y, u, v = p1case["y"].copy(), p1case["u"], p1case["v"]
        y[[0, 40, 84]] = np.nan
    else:
        y, u, v = synth.make_series(T, p, q, series_id=55, mask=mask)
    th = synth.make_init_packed(p, q, 5, seed=14)
    th[:, 2 + p + q] = 0.3 + th[:, 0]           # Q
    th[:, 3 + p + q] = 0.05 + 0.5 * th[:, 1 + p]  # R
    th[:, 4 + p + q] = 0.2                       # mu1
    lam = 0.4
    g = ldsr_amd.smooth_batch(y, u, v, th)
    g0 = ldsr_amd.smooth_batch(y, u, v, th, stdlik=False)
    pl = ldsr_amd.penalized_likelihood(y, u, v, th, lam)
    for i in range(th.shape[0]):
        r = O.kalman_smoother(y, u, v, th[i])
        r0 = O.kalman_smoother(y, u, v, th[i], stdlik=False)
        for name in "XYVJ":
            assert parity_close(g[name][i], r[name], RTOL, ATOL), (name, i)
        assert parity_close(g["lik"][i], r["lik"], RTOL, ATOL)
        assert parity_close(g0["lik"][i], r0["lik"], RTOL, ATOL)
        X = r["X"]
        ssq = np.sum((X[1:] - th[i, 0] * X[:-1] - th[i, 1:1 + p] @ u[:, :-1]) ** 2)
        assert parity_close(pl[i], r0["lik"] - lam * ssq, RTOL, ATOL)


def test_likelihood_product_does_not_overflow_on_badly_scaled_series():
    """The log-determinant is accumulated as a folded (mantissa, exponent) product: series with
    Sigma_t ~ 1e+-16 per step (unstandardised y; the whole problem is rescaled, which keeps
    the reference's own (1 - K C) Vp update well conditioned) must give the oracle's finite likelihoods and
    stop at the oracle's iteration (the reference sums log(Sigma_t) per step, src/EM.cpp:122)."""
    import ldsr_amd
    from ldsr_amd import synth
    for scale in (1e8, 1e-8):
        y, u, v = synth.make_series(2000, 1, 2, series_id=66)
        y = y * scale
        th0 = synth.make_init_packed(1, 2, 8, seed=15)
        # the hidden state at the data's scale: B, D, mu1 ~ scale; Q, R, V1 ~ scale^2; A, C unchanged
        th0 *= np.array([1, scale, 1, scale, scale, scale ** 2, scale ** 2, scale, scale ** 2])
        ref = _oracle_batch(y, u, v, th0, 30, 1e-5)
        assert np.all(np.isfinite(ref[1]))
        for algo in (1, 2):
            r = ldsr_amd.em_batch(y, u, v, th0, niter=30, tol=1e-5, algo=algo)
            assert np.array_equal(r["n_iter"], ref[2]), (scale, algo)
            assert parity_close(r["lik"], ref[1], RTOL, ATOL), (scale, algo)
            assert parity_close(r["theta"], ref[0], RTOL, 1e-9 * min(1.0, scale * scale)), (scale, algo)
