"""GPU parity tests (run with -m gpu on the MI355X box).  Everything goes through the C ABI
of include/ldsr_hip.h; the CPU oracle is only the checker.

Bar (BASELINE.json north_star): identical EM iteration counts, then
|d| <= 1e-6*|ref| + 1e-9 on every theta entry and on the log-likelihood."""
import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 1e-9


@pytest.fixture(scope="module")
def eng():
    import ldsr_amd
    from ldsr_amd import _lib
    assert _lib.lib().ldsr_device_count() >= 1, "no GPU visible"
    return ldsr_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


import os

ALGOS = [int(a) for a in os.environ.get("LDSR_TEST_ALGOS", "1,2").split(",")]   # 1 = LDSR_ALGO_SERIAL, 2 = LDSR_ALGO_SCAN


def _oracle_batch(O, y, u, v, th0, niter, tol, soc=None, threads=8):
    Y = np.atleast_2d(y)
    U = None if u is None else (u.T[None] if u.ndim == 2 else np.transpose(u, (0, 2, 1)))
    V = None if v is None else (v.T[None] if v.ndim == 2 else np.transpose(v, (0, 2, 1)))
    S = Y.shape[0]
    if U is not None and U.shape[0] != S:
        U = np.repeat(U, S, axis=0)
    if V is not None and V.shape[0] != S:
        V = np.repeat(V, S, axis=0)
    if soc is None:
        soc = np.zeros(th0.shape[0], np.int32)
    return O.em_batch(Y, None if U is None else np.ascontiguousarray(U),
                      None if V is None else np.ascontiguousarray(V), soc, th0, niter, tol,
                      n_threads=threads)


def _assert_batch_parity(r, ref, what=""):
    ref_th, ref_lik, ref_it, ref_st = ref
    assert np.array_equal(r["n_iter"], ref_it), "%s: iteration counts differ at cells %s" % (
        what, np.nonzero(r["n_iter"] != ref_it)[0][:10])
    assert parity_close(r["lik"], ref_lik, RTOL, ATOL), what
    assert parity_close(r["theta"], ref_th, RTOL, ATOL), what


# ---- the reference's own known-answer test, through the GPU ---------------------------------
def test_known_answer_first_two_iterations(eng, p1case):
    """tests/testthat/test-LDS-EM.R:21-35 with the same call sequence and tolerance."""
    c = p1case
    smooth1 = eng.Kalman_smoother(c["y"], c["u"], c["v"], c["theta0"])
    theta1 = eng.Mstep(c["y"], c["u"], c["v"], smooth1)
    smooth2 = eng.Kalman_smoother(c["y"], c["u"], c["v"], theta1)
    theta2 = eng.Mstep(c["y"], c["u"], c["v"], smooth2)
    tol = 1e-6
    assert smooth1["lik"] == pytest.approx(-11.678657, abs=tol)
    assert smooth1["X"][0, [0, 84]] == pytest.approx([1.293356, -0.987671], abs=tol)
    assert theta1["A"][0, 0] == pytest.approx(0.606066, abs=tol)
    assert theta1["C"][0, 0] == pytest.approx(-0.005995, abs=tol)
    assert theta1["Q"][0, 0] == pytest.approx(3.640236, abs=tol)
    assert smooth2["lik"] == pytest.approx(-0.114224, abs=tol)
    assert theta2["A"][0, 0] == pytest.approx(0.603945, abs=tol)
    assert theta2["C"][0, 0] == pytest.approx(-0.012004, abs=tol)
    assert theta2["Q"][0, 0] == pytest.approx(3.644322, abs=tol)


@pytest.mark.parametrize("algo", ALGOS)
def test_known_answer_convergence(eng, p1case, algo):
    """tests/testthat/test-LDS-EM.R:37-41: 68 iterations, lik = -0.039093."""
    c = p1case
    fit = eng.LDS_EM(c["y"], c["u"], c["v"], c["theta0"], 100, 1e-5, algo=algo)
    assert len(fit["liks"]) == 68
    assert fit["lik"] == pytest.approx(-0.039093, abs=1e-6)
    assert fit["fit"]["lik"] == pytest.approx(fit["lik"], abs=1e-12)


def test_smoother_full_fit_matches_oracle(eng, O, p1case):
    c = p1case
    g = eng.Kalman_smoother(c["y"], c["u"], c["v"], c["theta0"])
    r = O.kalman_smoother(c["y"], c["u"], c["v"], c["theta0"])
    for k in "XYVJ":
        assert parity_close(g[k][0], r[k], RTOL, ATOL), k
    assert parity_close(g["lik"], r["lik"], RTOL, ATOL)
    g = eng.Kalman_smoother(c["y"], c["u"], c["v"], c["theta0"], stdlik=False)
    r = O.kalman_smoother(c["y"], c["u"], c["v"], c["theta0"], stdlik=False)
    assert parity_close(g["lik"], r["lik"], RTOL, ATOL)


def test_penalized_likelihood_population(eng, O, p1case):
    """GA fitness (R/LDS_GA.R:28-44) for a population of thetas in one call."""
    from ldsr_amd import synth
    c = p1case
    pop = synth.make_init_packed(7, 7, 40, seed=21)
    pop[:, 15] = 0.3 + pop[:, 0]            # Q
    pop[:, 16] = 0.05 + 0.5 * pop[:, 8]     # R
    lam = 0.7
    got = eng.penalized_likelihood(c["y"], c["u"], c["v"], pop, lam)
    for i in range(0, 40, 5):
        ks = O.kalman_smoother(c["y"], c["u"], c["v"], pop[i], stdlik=False)
        X = ks["X"]
        ssq = np.sum((X[1:] - pop[i, 0] * X[:-1] - pop[i, 1:8] @ c["u"][:, :-1]) ** 2)
        assert parity_close(got[i], ks["lik"] - lam * ssq, RTOL, ATOL)


def test_propagate_matches_oracle(eng, O, p1case):
    c = p1case
    y = c["y"].copy()
    y[[3, 4, 50]] = np.nan
    g = eng.propagate(c["theta0"], c["u"], c["v"], y)
    r = O.propagate(c["theta0"], c["u"], c["v"], y)
    for k in "XYV":
        assert parity_close(g[k][0], r[k], RTOL, ATOL), k
    assert parity_close(g["lik"], r["lik"], RTOL, ATOL)


# ---- branches the reference only smoke-tests --------------------------------------------------
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("case", ["u_absent", "v_absent", "nan_mask", "p2_q7", "both_p3_q5"])
def test_branches_on_p1_data(eng, O, p1case, case, algo):
    c = p1case
    y, u, v = c["y"].copy(), c["u"], c["v"]
    if case == "u_absent":
        u = None
    if case == "v_absent":
        v = None
    if case == "nan_mask":
        y[[0, 10, 11, 84]] = np.nan
    if case == "p2_q7":
        u = u[:2]
    if case == "both_p3_q5":
        u, v = u[:3], v[2:7]
    p = 1 if u is None else u.shape[0]
    q = 1 if v is None else v.shape[0]
    from ldsr_amd import synth
    th0 = synth.make_init_packed(p, q, 40, seed=11)
    th0[0] = np.concatenate([[0.5], np.full(p, 0.5), [0.5], np.full(q, 0.5), [1, 1, 1, 1.0]])
    r = eng.em_batch(y, u, v, th0, niter=100, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, y, u, v, th0, 100, 1e-5)
    _assert_batch_parity(r, ref, case)
    if u is None:
        assert np.all(r["theta"][:, 1] == 0.0)          # B.zeros(1,p), src/EM.cpp:186
    if v is None:
        assert np.all(r["theta"][:, 2 + p] == 0.0)      # D.zeros(1,q), src/EM.cpp:154


@pytest.mark.parametrize("algo", ALGOS)
def test_np_bundled_data_restarts(eng, O, npcase, refdata, algo):
    """Config 1: bundled Nakhon Phanom data, the reference test's slice t(NPpc[601:813])
    (start.year = 1800, T = 213, 46 observations, p = q = 3), 50 restarts, niter=1000."""
    c = npcase(1800)
    from ldsr_amd import synth
    th0 = synth.make_init_packed(3, 3, 50, seed=5)
    r = eng.em_batch(c["y"], c["u"], c["v"], th0, niter=1000, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, c["y"], c["u"], c["v"], th0, 1000, 1e-5)
    _assert_batch_parity(r, ref, "NP T=213")
    k = eng.select_restart(r["lik"], r["theta"], 3, 3)
    assert k == O.select(ref[1], ref[0][:, 4])


@pytest.mark.parametrize("algo", ALGOS)
def test_np_full_length_paleo_mask(eng, O, npcase, algo):
    """NP data from 1200 (T = 813, 767 leading NA): mu1 / V1 stay at their initial values."""
    c = npcase(1200)
    from ldsr_amd import synth
    th0 = synth.make_init_packed(3, 3, 16, seed=6)
    r = eng.em_batch(c["y"], c["u"], c["v"], th0, niter=300, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, c["y"], c["u"], c["v"], th0, 300, 1e-5)
    _assert_batch_parity(r, ref, "NP T=813")


# ---- synthetic configs of BASELINE.json at oracle-sized cell counts ---------------------------
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("T,p,q,mask,n", [
    (1000, 1, 2, "dense", 192),     # config 2 shape
    (1000, 1, 2, "paleo", 64),
    (1000, 4, 8, "dense", 64),      # config 3 shape (5x5 and 9x9 systems)
    (2000, 1, 4, "dense", 32),      # config 4 shape
    (813, 1, 3, "paleo", 32),       # config 5 shape
    (64, 1, 1, "dense", 8),
    (65, 2, 1, "dense", 8),
    (256, 3, 5, "dense", 8),
    (257, 8, 8, "paleo", 8),
    (1024, 2, 2, "dense", 8),
    (1025, 1, 1, "dense", 8),
    (2048, 2, 4, "paleo", 8),
    (2049, 1, 2, "dense", 8),       # beyond the scan kernel: AUTO / SCAN must refuse or fall back
])
def test_synthetic_converged_runs(eng, O, algo, T, p, q, mask, n):
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, p, q, series_id=T + p + q, mask=mask)
    th0 = synth.make_init_packed(p, q, n, seed=T)
    if algo == 2 and T > 8192:
        with pytest.raises(Exception):
            eng.em_batch(y, u, v, th0, niter=1000, tol=1e-5, algo=algo)
        algo = 0                     # LDSR_ALGO_AUTO falls back to the serial kernel
    r = eng.em_batch(y, u, v, th0, niter=1000, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, y, u, v, th0, 1000, 1e-5)
    assert np.all(np.isfinite(ref[1]))
    _assert_batch_parity(r, ref, "T=%d p=%d q=%d %s" % (T, p, q, mask))


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("T,p,q", [(2, 1, 1), (3, 1, 1), (3, 1, 2), (5, 2, 2), (7, 2, 2), (9, 8, 8)])
def test_tiny_series_fixed_iterations(eng, O, algo, T, p, q):
    """Smallest lengths the reference can run (T >= 2).  These problems have more parameters
    than data, EM drives R -> 0, so only a few iterations are compared."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(max(T, 12), p, q, series_id=50 + T)
    y, u, v = y[:T], u[:, :T], v[:, :T]
    th0 = synth.make_init_packed(p, q, 6, seed=T)
    if T <= max(p, q):               # Svv / Tuu singular: flagged, as arma::inv would throw
        r = eng.em_batch(y, u, v, th0, niter=3, tol=0.0, algo=algo)
        assert np.all(r["status"] == 2)
        return
    r = eng.em_batch(y, u, v, th0, niter=3, tol=0.0, algo=algo)
    ref = _oracle_batch(O, y, u, v, th0, 3, 0.0)
    _assert_batch_parity(r, ref, "tiny T=%d" % T)


@pytest.mark.parametrize("T", [128, 129, 192, 193, 256, 384, 385, 512, 513, 640, 641, 768, 769, 813, 832, 833,
                               896, 897, 960, 961, 1024, 1025, 1280, 1281, 1536, 1537, 1792, 1793, 2048])
def test_scan_chunk_length_boundaries(eng, O, T):
    """The scan kernel picks its chunk length L (2, 3, 4, 6, 8, 10, 12..16, 20, 24, 28, 32 steps
    per lane) from T; exercise both sides of every switch point, with a ragged NA mask so the
    general (non-dense) path runs."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, 1, 2, series_id=T)
    y[T // 3:T // 3 + 5] = np.nan
    y[0] = np.nan
    y[T - 1] = np.nan
    th0 = synth.make_init_packed(1, 2, 6, seed=T)
    r = eng.em_batch(y, u, v, th0, niter=60, tol=1e-5, algo=2)
    ref = _oracle_batch(O, y, u, v, th0, 60, 1e-5)
    _assert_batch_parity(r, ref, "T=%d" % T)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("p", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("q", [1, 2, 4, 6, 8])
def test_every_padded_input_size(eng, O, algo, p, q):
    """All 16 (PP, QQ) kernel instantiations (rows padded to 1, 2, 4, 8), dense and masked."""
    from ldsr_amd import synth
    T = 150
    y, u, v = synth.make_series(T, p, q, series_id=10 * p + q)
    th0 = synth.make_init_packed(p, q, 5, seed=p * 8 + q)
    for mask in ("dense", "scattered"):
        ym = y.copy()
        if mask == "scattered":
            ym[np.random.default_rng(p + q).choice(T, 30, replace=False)] = np.nan
        r = eng.em_batch(ym, u, v, th0, niter=40, tol=1e-5, algo=algo)
        ref = _oracle_batch(O, ym, u, v, th0, 40, 1e-5)
        _assert_batch_parity(r, ref, "p=%d q=%d %s" % (p, q, mask))


@pytest.mark.parametrize("p,q", [(9, 2), (3, 12), (16, 16), (11, 13)])
def test_wide_inputs_run_on_the_serial_kernel(eng, O, p, q):
    """More than 8 rows of u or v: LDSR_ALGO_AUTO falls back to the serial kernel, the scan
    kernel refuses; the standalone smoother / M-step take the same widths."""
    from ldsr_amd import synth
    T = 120
    y, u, v = synth.make_series(T, p, q, series_id=p * 17 + q, mask="paleo", n_tail=80)
    th0 = synth.make_init_packed(p, q, 6, seed=p + q)
    with pytest.raises(Exception):
        eng.em_batch(y, u, v, th0, niter=30, tol=1e-5, algo=2)
    r = eng.em_batch(y, u, v, th0, niter=30, tol=1e-5)
    ref = _oracle_batch(O, y, u, v, th0, 30, 1e-5)
    _assert_batch_parity(r, ref, "wide p=%d q=%d" % (p, q))
    s = eng.Kalman_smoother(y, u, v, th0[0])
    so = O.kalman_smoother(y, u, v, th0[0])
    assert parity_close(s["X"][0], so["X"], RTOL, ATOL) and parity_close(s["lik"], so["lik"], RTOL, ATOL)
    th1 = eng.pack_theta(eng.Mstep(y, u, v, s), p, q)
    assert parity_close(th1, O.mstep(y, u, v, so), RTOL, ATOL)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("frac", [0.1, 0.5, 0.9])
def test_scattered_missing_values(eng, O, algo, frac):
    """NA scattered at random through the series (the reference branches per step, src/EM.cpp:82)."""
    from ldsr_amd import synth
    T = 1000
    y, u, v = synth.make_series(T, 1, 2, series_id=31)
    y[np.random.default_rng(int(frac * 10)).random(T) < frac] = np.nan
    th0 = synth.make_init_packed(1, 2, 24, seed=12)
    r = eng.em_batch(y, u, v, th0, niter=150, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, y, u, v, th0, 150, 1e-5)
    _assert_batch_parity(r, ref, "scattered %.1f" % frac)


@pytest.mark.parametrize("algo", ALGOS)
def test_fixed_niter_and_liks_trace(eng, O, algo):
    from ldsr_amd import synth
    y, u, v = synth.make_series(500, 1, 2, series_id=3)
    th0 = synth.make_init_packed(1, 2, 16, seed=2)
    r = eng.em_batch(y, u, v, th0, niter=25, tol=0.0, algo=algo, return_liks=True)
    assert np.all(r["n_iter"] == 25)
    for c in range(4):
        ref = O.lds_em(y, u, v, th0[c], 25, 0.0)
        assert parity_close(r["liks"][c], ref["liks"], RTOL, ATOL)
    # EM never decreases the (incomplete-data) likelihood
    assert np.all(np.diff(r["liks"], axis=1) > -1e-9)


@pytest.mark.parametrize("algo", ALGOS)
def test_cv_fold_grid_shared_inputs(eng, O, algo):
    """cvLDS shape (R/LDS_reconstruction.R:270-285): same u, v; one NA mask of y per fold;
    fresh restarts per fold; ragged cell counts per fold, one fold with no restarts."""
    from ldsr_amd import synth
    T, p, q = 400, 1, 4
    y, u, v = synth.make_series(T, p, q, series_id=9, mask="paleo", n_tail=100)
    folds = []
    for k in range(4):
        yk = y.copy()
        yk[300 + 25 * k:300 + 25 * k + 11] = np.nan     # contiguous block of the instrumental period
        folds.append(yk)
    Y = np.stack(folds)
    counts = [5, 0, 9, 3]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    th0 = synth.make_init_packed(p, q, off[-1], seed=8)
    r = eng.em_batch(Y, u, v, th0, cell_offsets=off, niter=200, tol=1e-5, algo=algo)
    soc = np.repeat(np.arange(4), counts).astype(np.int32)
    ref = _oracle_batch(O, Y, u, v, th0, 200, 1e-5, soc=soc)
    _assert_batch_parity(r, ref, "cv grid")


@pytest.mark.parametrize("algo", ALGOS)
def test_multi_series_own_inputs(eng, O, algo):
    """48-station shape (config 5): independent series with their own u, v and NA pattern."""
    from ldsr_amd import synth
    T, p, q, S, n = 300, 1, 3, 5, 7
    ys, us, vs = zip(*[synth.make_series(T, p, q, series_id=100 + s, mask="paleo",
                                         n_tail=30 + 12 * s) for s in range(S)])
    Y, U, V = np.stack(ys), np.stack(us), np.stack(vs)
    off = (np.arange(S + 1) * n).astype(np.int32)
    th0 = synth.make_init_packed(p, q, S * n, seed=4)
    r = eng.em_batch(Y, U, V, th0, cell_offsets=off, niter=150, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, Y, U, V, th0, 150, 1e-5, soc=np.repeat(np.arange(S), n).astype(np.int32))
    _assert_batch_parity(r, ref, "multi series")


def test_np_reconstruction_reaches_published_optimum(eng, O, npcase, refdata):
    """Config 1 end to end: the vignette's run (bundled NPannual / NPpc from 1200, T = 813,
    p = q = 3, niter = 1000, tol = 1e-5) with 512 random restarts in one launch.  The package
    ships its own result NPlds (lik 0.8249222558, from a few dozen unseeded restarts): the best
    of 512 restarts with C > 0 must be at least that good (it is better: ~0.8395), and must be
    exactly what the oracle computes from the same initial theta."""
    c = npcase(1200)
    init = eng.make_init(3, 3, 512, r_seed=2020)
    win = eng.LDS_EM_restart(c["y"], c["u"], c["v"], init, niter=1000, tol=1e-5)
    lik_pub = refdata["NPlds"]["lik"][0]
    assert win["theta"]["C"][0, 0] > 0
    assert win["lik"] > lik_pub - 1e-3, (win["lik"], lik_pub)
    assert np.sum(win["all"]["lik"] > lik_pub - 1e-3) >= 1
    assert win["fit"]["X"].shape == (1, 813) and np.all(np.isfinite(win["fit"]["V"]))
    # the winner is exactly what the oracle gets from the same init
    k = win["all"]["selected"]
    ref = O.lds_em(c["y"], c["u"], c["v"], eng.pack_theta(init[k], 3, 3), 1000, 1e-5)
    assert len(ref["liks"]) == len(win["liks"])
    assert parity_close(eng.pack_theta(win["theta"], 3, 3), ref["theta"], RTOL, ATOL)
    assert parity_close(win["lik"], ref["lik"], RTOL, ATOL)


def test_cv_grid_matches_oracle_engine(eng, O, npcase):
    """cvLDS grid (section 8 f-2) on the bundled NP data: folds x restarts in one launch; the
    per-fold winners and their fitted Y must match the same host logic run on the oracle."""
    from ldsr_amd import cv
    c = npcase(1800)
    inst = np.nonzero(~np.isnan(c["y"]))[0]
    Z = cv.make_Z(c["y"][inst], nRuns=6, frac=0.25, rng=np.random.default_rng(2))

    def em_batch(Y, u_, v_, th0, cell_offsets=None, niter=1000, tol=1e-5):
        soc = np.repeat(np.arange(Y.shape[0]), np.diff(cell_offsets)).astype(np.int32)
        th, lik, nit, st = _oracle_batch(O, Y, u_, v_, th0, niter, tol, soc=soc)
        return {"theta": th, "lik": lik, "n_iter": nit, "status": st}

    def smooth_batch(Y, u_, v_, th, cell_offsets=None):
        fits = [O.kalman_smoother(Y[f], u_, v_, th[f]) for f in range(Y.shape[0])]
        return {k: np.stack([f_[k] for f_ in fits]) for k in "XYVJ"}

    ref_eng = {"em_batch": em_batch, "smooth_batch": smooth_batch,
               "select": lambda l, t, p_, q_: O.select(l, t[:, 1 + p_])}
    kw = dict(num_restarts=8, niter=300, tol=1e-5, r_seed=7, mu=c["mu"])
    g = cv.cv_grid(c["y"], c["u"], c["v"], inst, Z, **kw)
    r = cv.cv_grid(c["y"], c["u"], c["v"], inst, Z, engine=ref_eng, **kw)
    assert np.array_equal(g["winner"], r["winner"])
    assert np.array_equal(g["all"]["n_iter"], r["all"]["n_iter"])
    assert parity_close(g["Ycv"], r["Ycv"], RTOL, ATOL)
    for f, z in enumerate(Z):
        mg = cv.calculate_metrics(g["Ycv"][f], c["y"][inst] + c["mu"], z)
        mr = cv.calculate_metrics(r["Ycv"][f], c["y"][inst] + c["mu"], z)
        assert parity_close(list(mg.values()), list(mr.values()), 1e-5, 1e-8)


def test_restart_selection_and_winner_fit(eng, O, p1case):
    c = p1case
    init = eng.make_init(7, 7, 24, seed=1)
    win = eng.LDS_EM_restart(c["y"], c["u"], c["v"], init, niter=100, tol=1e-5)
    th0 = np.stack([eng.pack_theta(t, 7, 7) for t in init])
    ref = _oracle_batch(O, c["y"], c["u"], c["v"], th0, 100, 1e-5)
    k = O.select(ref[1], ref[0][:, 8])
    assert win["all"]["selected"] == k
    assert win["theta"]["C"][0, 0] > 0 or not np.any(ref[0][:, 8] > 0)
    full = O.lds_em(c["y"], c["u"], c["v"], th0[k], 100, 1e-5)
    for key in "XYVJ":
        assert parity_close(win["fit"][key][0], full["fit"][key], RTOL, ATOL), key
    assert parity_close(win["liks"], full["liks"], RTOL, ATOL)
    assert win["init"] is init[k]


def test_singular_inputs_are_flagged_not_trapped(eng):
    """Duplicate rows of v make Svv singular: arma::inv throws in the reference; the device
    path marks every cell LDSR_CELL_SINGULAR and the host API raises."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(100, 1, 2, series_id=1)
    v[1] = v[0]
    th0 = synth.make_init_packed(1, 2, 4, seed=1)
    r = eng.em_batch(y, u, v, th0, niter=10)
    assert np.all(r["status"] == 2) and np.all(np.isnan(r["lik"]))
    with pytest.raises(Exception):
        eng.LDS_EM(y, u, v, th0[0], 10, 1e-5)


def test_argument_errors(eng):
    from ldsr_amd import synth
    y, u, v = synth.make_series(50, 1, 2)
    th0 = synth.make_init_packed(1, 2, 2)
    with pytest.raises(Exception):
        eng.em_batch(y, u, v, th0, niter=1)              # reference reads lik[1]
    with pytest.raises(Exception):
        eng.em_batch(y, np.zeros((17, 50)), v, synth.make_init_packed(17, 2, 2))   # p > 16


# ---- full BASELINE sizes: size-independent properties + sampled oracle parity -----------------
@pytest.mark.parametrize("algo", ALGOS)
def test_config2_full_size(eng, O, algo):
    """Config 2 as benchmarked: T=1000, p=1, q=2, 4096 restarts, niter=100, tol=0."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(1000, 1, 2, series_id=0)
    th0 = synth.make_init_packed(1, 2, 4096, seed=1)
    r = eng.em_batch(y, u, v, th0, niter=100, tol=0.0, algo=algo, return_liks=True)
    assert np.all(r["n_iter"] == 100)
    assert np.all(r["status"] == 0)
    assert np.all(np.diff(r["liks"], axis=1) > -1e-9)            # EM monotonicity, every cell
    # sharding independence: a sub-batch gives bit-identical results
    sub = eng.em_batch(y, u, v, th0[1000:1100], niter=100, tol=0.0, algo=algo)
    assert np.array_equal(sub["theta"], r["theta"][1000:1100])
    # sign symmetry of the model (x,B,C,mu1 -> -x,-B,-C,-mu1 leaves the likelihood unchanged)
    flip = th0[:64].copy()
    flip[:, 1] *= -1
    flip[:, 2] *= -1
    f = eng.em_batch(y, u, v, flip, niter=100, tol=0.0, algo=algo)
    assert parity_close(f["lik"], r["lik"][:64], 1e-9, 1e-12)
    assert parity_close(f["theta"][:, 2], -r["theta"][:64, 2], 1e-9, 1e-12)
    # oracle on every 16th cell
    idx = np.arange(0, 4096, 16)
    ref = _oracle_batch(O, y, u, v, th0[idx], 100, 0.0, threads=16)
    _assert_batch_parity({k: r[k][idx] for k in ("theta", "lik", "n_iter")}, ref, "cfg2 full")


@pytest.mark.parametrize("algo", [2])
@pytest.mark.parametrize("mask", ["dense", "paleo"])
def test_config2_full_size_converged_all_cells(eng, O, algo, mask):
    """Config 2 as the survey specifies its parity run (niter=1000, tol=1e-5): EVERY one of the
    4096 cells must stop at the oracle's iteration and agree on theta and lik."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(1000, 1, 2, series_id=0, mask=mask)
    th0 = synth.make_init_packed(1, 2, 4096, seed=1)
    r = eng.em_batch(y, u, v, th0, niter=1000, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, y, u, v, th0, 1000, 1e-5, threads=16)
    assert np.all(np.isfinite(ref[1]))
    _assert_batch_parity(r, ref, "cfg2 converged %s" % mask)
    assert eng.select_restart(r["lik"], r["theta"], 1, 2) == O.select(ref[1], ref[0][:, 2])


@pytest.mark.parametrize("n_dev", [1, 2, 3, 5])
def test_multi_device_entry_is_bit_identical(eng, n_dev):
    """ldsr_em_batch_multi: host threads, one contiguous cell slice per listed device (here the
    one GPU of the box listed n_dev times); results must be bit-identical to the single call,
    whatever the cut points (ragged series, shared and own inputs, an empty series)."""
    from ldsr_amd import synth
    T, p, q, S = 200, 1, 3, 4
    ys, us, vs = zip(*[synth.make_series(T, p, q, series_id=300 + s, mask="paleo", n_tail=50 + 20 * s)
                       for s in range(S)])
    Y, U, V = np.stack(ys), np.stack(us), np.stack(vs)
    counts = [7, 0, 11, 5]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    th0 = synth.make_init_packed(p, q, int(off[-1]), seed=6)
    for uu, vv in ((U, V), (us[0], vs[0])):            # own inputs, then shared inputs
        one = eng.em_batch(Y, uu, vv, th0, cell_offsets=off, niter=60, tol=1e-5, return_liks=True)
        multi = eng.em_batch(Y, uu, vv, th0, cell_offsets=off, niter=60, tol=1e-5, return_liks=True,
                             devices=[0] * n_dev)
        for k in ("theta", "lik", "n_iter", "status", "liks"):
            assert np.array_equal(one[k], multi[k], equal_nan=True), (n_dev, k)


def test_multi_device_entry_reports_bad_device(eng):
    from ldsr_amd import synth
    y, u, v = synth.make_series(50, 1, 2)
    with pytest.raises(Exception) as e:
        eng.em_batch(y, u, v, synth.make_init_packed(1, 2, 8), niter=5, devices=[0, 99])
    assert "device 99" in str(e.value)


@pytest.mark.parametrize("algo", ALGOS)
def test_degenerate_initial_thetas_do_not_trap(eng, O, algo):
    """Nothing clamps Q or R in the reference (src/EM.cpp:177,210) and NaN likelihoods are
    tolerated by the selection (na.rm, R/LDS_reconstruction.R:54).  Hostile inits must come back
    with a status word, never hang or fault, and must not disturb their healthy neighbours."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(300, 1, 2, series_id=8)
    th0 = synth.make_init_packed(1, 2, 16, seed=5)
    P = th0.shape[1]
    iQ, iR, iV1, iC, iA = 5, 6, 8, 2, 0
    th0[1, :] = np.nan                    # all NaN
    th0[3, iR] = np.nan
    th0[5, iR] = -1.0                     # negative R: log(Sigma) of a negative number
    th0[7, iQ] = 0.0; th0[7, iV1] = 0.0   # zero variances: 0/0 in the smoother gain
    th0[9, iA] = np.inf
    th0[11, iC] = 0.0                     # y carries no information about x at iteration 0
    th0[13, iQ] = 1e300; th0[13, iR] = 1e-300
    r = eng.em_batch(y, u, v, th0, niter=40, tol=1e-5, algo=algo)
    ref = _oracle_batch(O, y, u, v, th0, 40, 1e-5)
    assert r["theta"].shape == (16, P)
    healthy = [0, 2, 4, 6, 8, 10, 12, 14, 15]
    assert np.all(np.isfinite(ref[1][healthy]))
    _assert_batch_parity({k: r[k][healthy] for k in ("theta", "lik", "n_iter")},
                         tuple(a[healthy] for a in ref), "healthy neighbours")
    for c in (1, 3, 9):                   # NaN / Inf inputs can only give NaN
        assert np.isnan(r["lik"][c]) and r["status"][c] == 1 and np.isnan(ref[1][c])
    assert np.all((r["status"] == 0) == np.isfinite(r["lik"]))
    # wherever the oracle stays finite the GPU result must be finite and equal (cell 13 is
    # excluded: with Q = 1e300 the factor 1 - K C sits in the last bit and FMA contraction
    # decides its sign; it only has to come back with a status)
    for c in (5, 7, 11):
        if np.isfinite(ref[1][c]) and np.all(np.isfinite(ref[0][c])):
            assert r["n_iter"][c] == ref[2][c], c
            assert parity_close(r["lik"][c], ref[1][c], 1e-6, 1e-9), c


def test_plain_c_client_links_the_abi(eng, O, p1case, tmp_path):
    """The drop-in boundary is a C ABI: a gcc-built client with no HIP / Python dependency links
    libldsr_hip.so, runs the reference's known-answer case plus random restarts through
    ldsr_em_batch_multi + ldsr_select_restart, and must print what the oracle computes."""
    import subprocess
    from ldsr_amd import _lib, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "client")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_client", "client.c"), "-o", exe,
                           "-L", os.path.dirname(_lib.SO_PATH), "-lldsr_hip", "-lm",
                           "-Wl,-rpath," + os.path.dirname(_lib.SO_PATH)])
    c = p1case
    th0 = synth.make_init_packed(7, 7, 12, seed=3)
    th0[0] = c["theta0"]
    case = tmp_path / "case.txt"
    with open(case, "w") as f:
        f.write("85 7 7 12 100 1e-5\n")
        for arr in (c["y"], c["u"].T.ravel(), c["v"].T.ravel(), th0.ravel()):
            f.write(" ".join(repr(float(x)) for x in arr) + "\n")
    out = subprocess.check_output([exe, str(case)], text=True).split("\n")
    k, nit, lik = out[0].split()
    theta = np.array([float(x) for x in out[1].split()])
    ref = _oracle_batch(O, c["y"], c["u"], c["v"], th0, 100, 1e-5)
    kref = O.select(ref[1], ref[0][:, 8])
    assert int(k) == kref and int(nit) == ref[2][kref]
    assert parity_close(float(lik), ref[1][kref], RTOL, ATOL)
    assert parity_close(theta, ref[0][kref], RTOL, ATOL)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("niter", [2, 3, 4])
def test_smallest_iteration_caps(eng, O, p1case, algo, niter):
    """niter = 2 is the least the reference can run (E, M, E; the loop of src/EM.cpp:259 never
    executes): liks has niter entries and theta is the one that produced the last fit."""
    c = p1case
    fit = eng.LDS_EM(c["y"], c["u"], c["v"], c["theta0"], niter, 1e-5, algo=algo)
    ref = O.lds_em(c["y"], c["u"], c["v"], c["theta0"], niter, 1e-5)
    assert len(fit["liks"]) == len(ref["liks"]) == niter
    assert parity_close(fit["liks"], ref["liks"], RTOL, ATOL)
    assert parity_close(eng.pack_theta(fit["theta"], 7, 7), ref["theta"], RTOL, ATOL)
    for k in "XYVJ":
        assert parity_close(fit["fit"][k][0], ref["fit"][k], RTOL, ATOL), k


@pytest.mark.parametrize("algo", [2, 3])     # scan kernel, pair kernel
def test_large_batch_is_consistent_with_small_batches(eng, algo):
    """65 536 restarts in one call (16 x the benchmark grid): every cell finishes, and any
    slice of the big batch is bit-identical to the same cells run as a small batch, for both
    schedules (tol = 0 static mapping, tol > 0 work queue) -- per kernel: LDSR_ALGO_AUTO chooses
    by launch size (pair kernel for launches that fill the device, scan kernel below), and the
    two differ at the 1e-13 level."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(1000, 1, 2, series_id=0, mask="paleo")
    th0 = synth.make_init_packed(1, 2, 65536, seed=1)
    for niter, tol in ((20, 0.0), (60, 1e-4)):
        big = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=algo)
        assert np.all(big["status"] == 0) and np.all(big["n_iter"] >= 3)
        for lo in (0, 30000, 65536 - 257):
            small = eng.em_batch(y, u, v, th0[lo:lo + 257], niter=niter, tol=tol, algo=algo)
            for k in ("theta", "lik", "n_iter"):
                assert np.array_equal(small[k], big[k][lo:lo + 257]), (tol, lo, k)


@pytest.mark.parametrize("T,p,q,mask", [(1700, 7, 5, "paleo"), (2048, 8, 8, "dense"), (1281, 4, 5, "dense"),
                                        (1900, 2, 8, "paleo")])
def test_long_wide_series_use_the_global_image_scan(eng, O, T, p, q, mask):
    """T > 1024 with wide inputs: the chunk-transposed series image exceeds the 160 KiB LDS, so
    the scan kernel reads the prepared arrays from global memory (GIMG variant).  Same parity
    bar; LDSR_ALGO_SCAN must accept these shapes."""
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, p, q, series_id=T + p, mask=mask, n_tail=T // 3)
    y[5:9] = np.nan
    th0 = synth.make_init_packed(p, q, 10, seed=T)
    for niter, tol in ((40, 1e-5), (12, 0.0)):
        r = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=2)
        ref = _oracle_batch(O, y, u, v, th0, niter, tol)
        _assert_batch_parity(r, ref, "GIMG T=%d p=%d q=%d" % (T, p, q))
