"""GPU parity of the two-cells-per-wave kernel (LDSR_ALGO_PAIR, em_pair_impl.h) against the CPU
oracle: the bar of SURVEY.md Appendix B -- identical n_iter per cell, then theta and lik within
|d| <= 1e-6 |ref| + 1e-9 -- on every chunk length it is compiled for (T = 65 .. 1024), dense and
masked series, both schedules (static pairs at tol = 0, per-half work queue at tol > 0), ragged
grids (odd cell counts, one-cell series, empty series), plus the plan (AUTO picks it where it
applies) and its error behaviour.  Follows /root/reference/tests/testthat/test-LDS-EM.R's way of
testing (fixed inputs, fixed initial values, compare numbers)."""
import ctypes as C

import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 1e-9
PAIR, SCAN, QUAD = 3, 2, 4      # LDSR_ALGO_PAIR (two cells per wave), _SCAN, _QUAD (four cells per wave)


@pytest.fixture(scope="module")
def eng():
    import ldsr_amd
    return ldsr_amd


def _series(T, p, q, sid, mask):
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, p, q, series_id=sid)
    y = y.copy()
    if mask == "paleo":
        y[: T - T // 10] = np.nan
    elif mask == "holes":
        y[::7] = np.nan
        y[3] = np.nan
        y[T - 1] = np.nan
    elif mask == "head":            # observed start, missing end (the reverse of the paleo case)
        y[T // 3:] = np.nan
    return y, u, v


def _oracle(Y, U, V, soc, th0, niter, tol):
    from oracle import oracle as O
    return O.em_batch(np.atleast_2d(Y), np.ascontiguousarray(U), np.ascontiguousarray(V),
                      np.asarray(soc, np.int32), th0, niter, tol, n_threads=16)


def _check(r, ref, what):
    ref_th, ref_lik, ref_it, _ = ref
    bad = np.nonzero(r["n_iter"] != ref_it)[0]
    assert bad.size == 0, "%s: iteration counts differ at cells %s" % (what, bad[:10])
    assert parity_close(r["lik"], ref_lik, RTOL, ATOL), what
    assert parity_close(r["theta"], ref_th, RTOL, ATOL), what


def _plan_name(T, p, q, tol, algo=0):
    from ldsr_amd import _lib
    buf = C.create_string_buffer(160)
    a = _lib.lib().ldsr_em_plan(T, p, q, 100, float(tol), algo, buf, 160)
    return a, buf.value.decode()


# T values: both ends of every third chunk length's range, the BASELINE shapes (1000, 813), and
# the limits of the kernel (65, 1024)
@pytest.mark.parametrize("T", [65, 66, 96, 97, 128, 150, 160, 200, 224, 256, 257, 300, 352, 400, 448, 500, 512,
                               513, 544, 545, 600, 640, 641, 700, 768, 769, 813, 832, 833, 900,
                               960, 961, 992, 993, 1000, 1023, 1024])
@pytest.mark.parametrize("mask", ["dense", "paleo"])
def test_every_chunk_length_matches_oracle(eng, T, mask):
    from ldsr_amd import synth
    p, q = 1, 2
    y, u, v = _series(T, p, q, 100 + T, mask)
    th0 = synth.make_init_packed(p, q, 21, seed=T)          # odd count: one half-wave idles
    # AUTO's plan (a launch that fills the device assumed): four cells per wave up to T = 512, two above
    assert _plan_name(T, p, q, 0.0)[1] == "em_pair_kernel<1, 2, %d, %d, false, false>" % (
        (max(5, -(-T // 16)), 16) if T <= 512 else (-(-T // 32), 32))
    assert _plan_name(T, p, q, 0.0, PAIR)[1] == "em_pair_kernel<1, 2, %d, 32, false, false>" % max(3, -(-T // 32))
    for niter, tol in ((25, 0.0), (300, 1e-5)):
        ref = _oracle(y, u.T[None], v.T[None], np.zeros(21), th0, niter, tol)
        for algo in (PAIR, QUAD) if T <= 512 else (PAIR,):
            r = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=algo)
            _check(r, ref, "T=%d %s tol=%g algo=%d" % (T, mask, tol, algo))


@pytest.mark.parametrize("p,q", [(1, 1), (1, 2), (1, 3), (1, 4), (2, 1), (2, 2), (2, 3), (2, 4),
                                 (3, 1), (4, 2), (3, 3), (4, 4)])
@pytest.mark.parametrize("mask", ["dense", "holes", "head"])
def test_every_compiled_width(eng, p, q, mask):
    from ldsr_amd import synth
    pad = lambda n: 1 if n <= 1 else 2 if n <= 2 else 4
    KP = (1 + pad(p) + pad(q) + 1) // 2          # 16-byte value pairs per step of the series image
    T = {2: 1000, 3: 928, 4: 864, 5: 800}[KP]    # the longest series that leaves room for 8 waves per CU
    y, u, v = _series(T, p, q, 7 * p + q, mask)
    th0 = synth.make_init_packed(p, q, 48, seed=p * 10 + q)
    ref = _oracle(y, u.T[None], v.T[None], np.zeros(48), th0, 400, 1e-5)
    r = eng.em_batch(y, u, v, th0, niter=400, tol=1e-5, algo=PAIR)
    _check(r, ref, "p=%d q=%d %s" % (p, q, mask))
    # the one-cell-per-wave scan kernel agrees to rounding and stops at the same iterations
    r2 = eng.em_batch(y, u, v, th0, niter=400, tol=1e-5, algo=SCAN)
    assert np.array_equal(r["n_iter"], r2["n_iter"])
    assert parity_close(r["theta"], r2["theta"], 1e-9, 1e-12)
    # four cells per wave on the first 300 steps of the same series (odd cell count: idle rows)
    ys, us, vs = y[:300].copy(), u[:, :300].copy(), v[:, :300].copy()
    ys[0] = y[0] if np.isfinite(y[0]) else 0.1
    ref4 = _oracle(ys, us.T[None], vs.T[None], np.zeros(45), th0[:45], 200, 1e-5)
    r4 = eng.em_batch(ys, us, vs, th0[:45], niter=200, tol=1e-5, algo=QUAD)
    _check(r4, ref4, "quad p=%d q=%d %s" % (p, q, mask))


def test_absent_inputs(eng):
    """u or v absent (the reference's 1x1 sentinel, src/EM.cpp:71-75,172,212-213): B / D come
    back as exact zeros."""
    from ldsr_amd import synth
    T = 900
    y, u, v = _series(T, 1, 2, 31, "paleo")
    th0 = synth.make_init_packed(1, 2, 16, seed=4)
    th0_nou = th0.copy()
    r = eng.em_batch(y, None, v, th0_nou, niter=60, tol=1e-5, algo=PAIR)
    r2 = eng.em_batch(y, None, v, th0_nou, niter=60, tol=1e-5, algo=1)
    assert np.array_equal(r["n_iter"], r2["n_iter"])
    assert parity_close(r["theta"], r2["theta"], RTOL, ATOL)
    assert np.all(r["theta"][:, 1] == 0.0)


def test_ragged_grid_of_series(eng):
    """Several series with their own inputs and masks; 0, 1, 2, 15, 16, 17 and 33 cells per series
    (workgroups never straddle a series; a series without cells launches nothing)."""
    from ldsr_amd import synth
    T, p, q = 813, 1, 3
    counts = [0, 1, 2, 15, 16, 17, 33, 0, 5]
    S = len(counts)
    Y = np.empty((S, T)); U = np.empty((S, T, p)); V = np.empty((S, T, q))
    for s in range(S):
        y, u, v = _series(T, p, q, 500 + s, ["dense", "paleo", "holes"][s % 3])
        Y[s] = y; U[s] = u.T; V[s] = v.T
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    n = int(off[-1])
    th0 = synth.make_init_packed(p, q, n, seed=77)
    soc = np.repeat(np.arange(S), counts)
    for niter, tol in ((30, 0.0), (500, 1e-5)):
        ref = _oracle(Y, U, V, soc, th0, niter, tol)
        r = eng.em_batch(Y, np.transpose(U, (0, 2, 1)).copy(), np.transpose(V, (0, 2, 1)).copy(), th0,
                         cell_offsets=off, niter=niter, tol=tol, algo=PAIR, return_liks=True)
        _check(r, ref, "ragged tol=%g" % tol)
        # the trace: liks[c, :n_iter] finite, NaN padded behind (ABI contract of the batch entry)
        for c in (0, n // 2, n - 1):
            k = r["n_iter"][c]
            assert np.all(np.isfinite(r["liks"][c, :k])) and np.all(np.isnan(r["liks"][c, k:]))
            assert r["liks"][c, k - 1] == r["lik"][c]


def test_cells_do_not_depend_on_their_partner_or_on_the_cut(eng):
    """A cell's result is bit-identical whatever shares its wave: whole batch, reversed batch,
    single cells, and the library's multi-device cut (three slices on one GPU)."""
    from ldsr_amd import synth
    T, p, q = 1000, 1, 2
    y, u, v = _series(T, p, q, 9, "dense")
    th0 = synth.make_init_packed(p, q, 40, seed=12)
    for tol in (0.0, 1e-5):
        a = eng.em_batch(y, u, v, th0, niter=80, tol=tol, algo=PAIR)
        b = eng.em_batch(y, u, v, th0[::-1].copy(), niter=80, tol=tol, algo=PAIR)
        assert np.array_equal(a["theta"], b["theta"][::-1]) and np.array_equal(a["n_iter"], b["n_iter"][::-1])
        c = eng.em_batch(y, u, v, th0[7:8].copy(), niter=80, tol=tol, algo=PAIR)
        assert np.array_equal(a["theta"][7], c["theta"][0]) and a["lik"][7] == c["lik"][0]
        d = eng.em_batch(y, u, v, th0, niter=80, tol=tol, algo=PAIR, devices=[0, 0, 0])
        assert np.array_equal(a["theta"], d["theta"]) and np.array_equal(a["lik"], d["lik"])
        # the same for four cells per wave (first 400 steps)
        a = eng.em_batch(y[:400], u[:, :400], v[:, :400], th0, niter=80, tol=tol, algo=QUAD)
        b = eng.em_batch(y[:400], u[:, :400], v[:, :400], th0[::-1].copy(), niter=80, tol=tol, algo=QUAD)
        assert np.array_equal(a["theta"], b["theta"][::-1]) and np.array_equal(a["n_iter"], b["n_iter"][::-1])
        c = eng.em_batch(y[:400], u[:, :400], v[:, :400], th0[5:6].copy(), niter=80, tol=tol, algo=QUAD)
        assert np.array_equal(a["theta"][5], c["theta"][0]) and a["lik"][5] == c["lik"][0]


def test_steady_and_fallback_cells_share_waves(eng):
    """Fully observed series of 737..1024 steps take the steady-state sweeps when a cell's Riccati
    recursion has converged within the first L-1 steps, the generic sweeps otherwise -- decided per
    cell and per EM iteration.  Cells built to fail that test for many iterations (A near 1, tiny C:
    hundreds of steps to converge), for a few (moderate A, small C) and never, interleaved so
    that waves hold every combination: oracle parity with identical iteration counts, and results
    bit-identical whatever the partner is."""
    from ldsr_amd import synth
    p, q = 1, 2
    for T in (1000, 768, 1024, 737):
        y, u, v = _series(T, p, q, 77 + T, "dense")
        th0 = synth.make_init_packed(p, q, 48, seed=T)
        th0[0::3, 0], th0[0::3, 2] = 0.97, 0.03          # A, C: slow for tens of iterations
        th0[1::6, 0], th0[1::6, 2] = 0.80, 0.15          # a handful of iterations
        for niter, tol in ((60, 0.0), (400, 1e-5)):
            ref = _oracle(y, u.T[None], v.T[None], np.zeros(48), th0, niter, tol)
            r = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=PAIR)
            _check(r, ref, "steady/fallback T=%d tol=%g" % (T, tol))
            b = eng.em_batch(y, u, v, th0[::-1].copy(), niter=niter, tol=tol, algo=PAIR)
            assert np.array_equal(r["theta"], b["theta"][::-1]) and np.array_equal(r["lik"], b["lik"][::-1])
            c = eng.em_batch(y, u, v, th0[3:4].copy(), niter=niter, tol=tol, algo=PAIR)
            assert np.array_equal(r["theta"][3], c["theta"][0]) and r["lik"][3] == c["lik"][0]


def test_results_do_not_depend_on_the_workgroup_size(eng, tmp_path):
    """Two four-wave workgroups per CU (the default where the LDS allows) against one of eight
    (LDSR_PAIR_WPB=8, read once per process -> a child process): bit-identical, static and queue."""
    import os
    import subprocess
    import sys
    from ldsr_amd import synth
    T, p, q = 300, 2, 2
    y, u, v = _series(T, p, q, 21, "holes")
    th0 = synth.make_init_packed(p, q, 300, seed=5)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    np.savez(tmp_path / "in.npz", y=y, u=u, v=v, th0=th0)
    code = ("import sys, numpy as np; sys.path.insert(0, @ROOT@); import ldsr_amd; d = np.load(@IN@); out = {}\n"
            "for algo in (3, 4):\n"
            "    for tol in (0.0, 1e-5):\n"
            "        r = ldsr_amd.em_batch(d['y'], d['u'], d['v'], d['th0'], niter=60, tol=tol, algo=algo)\n"
            "        out['t%d_%g' % (algo, tol)] = r['theta']; out['n%d_%g' % (algo, tol)] = r['n_iter']\n"
            "np.savez(@OUT@, **out)\n")
    code = code.replace("@ROOT@", repr(root)).replace("@IN@", repr(str(tmp_path / "in.npz"))).replace(
        "@OUT@", repr(str(tmp_path / "out8.npz")))
    env = dict(os.environ, LDSR_PAIR_WPB="8")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
    ref = np.load(tmp_path / "out8.npz")
    for algo in (PAIR, QUAD):
        for tol in (0.0, 1e-5):
            r = eng.em_batch(y, u, v, th0, niter=60, tol=tol, algo=algo)
            assert np.array_equal(r["theta"], ref["t%d_%g" % (algo, tol)]), (algo, tol)
            assert np.array_equal(r["n_iter"], ref["n%d_%g" % (algo, tol)]), (algo, tol)


def test_config2_whole_batch_converged(eng):
    """BASELINE config 2 whole (4096 restarts, T=1000, p=1, q=2, niter=1000, tol=1e-5) through the
    per-half work queue; AUTO must have picked the pair kernel."""
    import bench
    Y, U, V, shared, off, th0, n = bench.build_problem("cfg2", "dense", 1, 0)
    assert _plan_name(1000, 1, 2, 1e-5, PAIR)[1] == "em_pair_kernel<1, 2, 32, 32, true, false>"
    ref = _oracle(Y, U, V, np.zeros(n), th0, 1000, 1e-5)
    r = eng.em_batch(Y[0], U[0].T.copy(), V[0].T.copy(), th0, niter=1000, tol=1e-5)
    _check(r, ref, "cfg2")
    from oracle import oracle as O
    p = 1
    assert eng.select_restart(r["lik"], r["theta"], 1, 2) == O.select(ref[1], ref[0][:, 1 + p])


def test_auto_choice(eng):
    """AUTO takes the pair kernel only for launches whose sixteen-cell workgroups fill the device
    (T = 1000: below ~3600 cells the scan kernel's four-cell workgroups spread over more CUs; short
    series, whose four-wave workgroups sit two per CU, from 1024 cells), and with
    tol > 0 only when every series is fully observed (iteration counts of masked series spread too
    widely for two cells per wave; the device entry / ldsr_em_plan cannot look at y and stay on
    the scan kernel)."""
    from ldsr_amd import synth
    T, p, q = 1000, 1, 2
    assert _plan_name(T, p, q, 0.0)[1].startswith("em_pair_kernel")
    assert _plan_name(T, p, q, 1e-5)[1].startswith("em_scan_kernel")
    big = synth.make_init_packed(p, q, 4096, seed=2)
    small = big[:24].copy()

    def last_kernel():
        from ldsr_amd import _lib
        buf = C.create_string_buffer(160)
        assert _lib.lib().ldsr_last_em_kernel(0, buf, 160) == 0
        return buf.value.decode()

    # fully observed: the pair kernel with and without early stopping (bit-identical to algo = PAIR)
    y, u, v = _series(T, p, q, 4, "dense")
    for niter, tol in ((300, 1e-5), (12, 0.0)):
        a = eng.em_batch(y, u, v, big, niter=niter, tol=tol)
        assert last_kernel() == "em_pair_kernel<1, 2, 32, 32, %s, false>" % ("true" if tol > 0 else "false")
        b = eng.em_batch(y, u, v, big, niter=niter, tol=tol, algo=PAIR)
        assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["lik"], b["lik"]), tol
    # ... unless the launch is small
    a = eng.em_batch(y, u, v, small, niter=12, tol=0.0)
    assert last_kernel().startswith("em_scan_kernel<1, 2, 16, 1,")
    # a scattered mask with early stopping: the scan kernel (iteration counts spread too widely)
    y, u, v = _series(T, p, q, 4, "holes")
    a = eng.em_batch(y, u, v, big, niter=100, tol=1e-5)
    assert last_kernel() == "em_scan_kernel<1, 2, 16, 1, true, false, false>"
    # a paleo-type mask (900 unobserved steps, then the record): the closed-form lead -- four cells
    # per wave from ~1536 cells; with early stopping (and a lead below 1024 steps) only when the
    # launch fills the device: 4096 cells fill the two-cells-per-wave kernel's workgroups only
    y, u, v = _series(T, p, q, 4, "paleo")
    for niter, tol in ((300, 1e-5), (12, 0.0)):
        a = eng.em_batch(y, u, v, big, niter=niter, tol=tol)
        assert last_kernel() == ("em_pair_kernel<1, 2, 4, 32, true, true>" if tol > 0 else
                                 "em_pair_kernel<1, 2, 7, 16, false, true>")
        b = eng.em_batch(y, u, v, big, niter=niter, tol=tol, algo=SCAN)
        assert np.array_equal(a["n_iter"], b["n_iter"])
        assert parity_close(a["theta"], b["theta"], 1e-8, 1e-11) and parity_close(a["lik"], b["lik"], 1e-9, 1e-12)
    # short series (four-wave workgroups, two per CU): four cells per wave from 3072 cells, two
    # from 1024, the scan kernel for a few
    y, u, v = _series(400, p, q, 4, "dense")
    big8 = synth.make_init_packed(p, q, 8192, seed=3)
    for n, same_as in ((8192, QUAD), (4096, QUAD), (3000, PAIR), (1024, PAIR), (1000, SCAN), (24, SCAN)):
        a = eng.em_batch(y, u, v, big8[:n], niter=8, tol=0.0)
        b = eng.em_batch(y, u, v, big8[:n], niter=8, tol=0.0, algo=same_as)
        assert np.array_equal(a["theta"], b["theta"]), n


def test_restart_grid_on_the_pair_kernel(eng, monkeypatch):
    """ldsr_em_restart_grid with a launch large enough for AUTO to take the pair kernel: the
    winner's trace and fit are those of the batch run's kernel, also when the traces do not fit
    the device (cap forced to 0) and the winner is re-run alone -- a single cell, which AUTO by
    itself would hand to the scan kernel."""
    from ldsr_amd import synth
    from oracle import oracle as O
    T, p, q = 1000, 1, 2
    y, u, v = _series(T, p, q, 6, "dense")
    th0 = synth.make_init_packed(p, q, 4096, seed=9)
    a = eng.em_restart_grid(y, u, v, th0, niter=40, tol=0.0)
    ref = eng.em_batch(y, u, v, th0, niter=40, tol=0.0, algo=PAIR, return_liks=True)
    w = int(a["winner"][0])
    assert w == eng.select_restart(ref["lik"], ref["theta"], p, q)
    assert np.array_equal(a["theta"][0], ref["theta"][w]) and a["lik"][0] == ref["lik"][w]
    assert np.array_equal(a["liks"][0][:40], ref["liks"][w][:40])
    monkeypatch.setenv("LDSR_LIKS_TRACE_MAX_BYTES", "0")
    b = eng.em_restart_grid(y, u, v, th0, niter=40, tol=0.0)
    for k in ("winner", "theta", "lik", "n_iter", "X", "Y", "V", "J"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["liks"], b["liks"], equal_nan=True)
    fit = O.kalman_smoother(y, u, v, ref["theta"][w])     # the winner's fit (packed theta)
    assert parity_close(np.ravel(a["X"][0]), np.ravel(fit["X"]), RTOL, ATOL)
    assert parity_close(np.ravel(a["V"][0]), np.ravel(fit["V"]), RTOL, ATOL)


def test_large_scale_values(eng):
    """y scaled by 1e+-8 (R and Q collapse / explode): the scaled step matrix of the dense F1 and
    the folded log-determinant keep every product in range."""
    from ldsr_amd import synth
    T, p, q = 1000, 1, 2
    y, u, v = _series(T, p, q, 3, "dense")
    th0 = synth.make_init_packed(p, q, 16, seed=5)
    for sc in (1e8, 1e-8):
        ref = _oracle(y * sc, u.T[None], v.T[None], np.zeros(16), th0, 40, 0.0)
        r = eng.em_batch(y * sc, u, v, th0, niter=40, tol=0.0, algo=PAIR)
        _check(r, ref, "scale %g" % sc)


def test_singular_series_and_bad_shapes(eng):
    from ldsr_amd import synth, _lib
    T, p, q = 700, 1, 2
    y, u, v = _series(T, p, q, 1, "dense")
    v = v.copy(); v[1] = v[0]                       # Svv singular: status 2 (arma::inv would throw)
    th0 = synth.make_init_packed(p, q, 5, seed=1)
    for tol in (0.0, 1e-5):
        r = eng.em_batch(y, u, v, th0, niter=20, tol=tol, algo=PAIR, return_liks=True)
        assert np.all(r["status"] == 2) and np.all(np.isnan(r["theta"])) and np.all(r["n_iter"] == 0)
        assert np.all(np.isnan(r["liks"]))
    # shapes outside the kernel: explicit PAIR is an error, AUTO goes elsewhere
    assert _plan_name(600, 1, 2, 0.0, QUAD)[0] == -1 and _plan_name(513, 3, 3, 0.0, QUAD)[0] == -1
    # (an odd number of values per step no longer pads the image: nine values at L = 32 fit since round 4)
    assert _plan_name(512, 3, 3, 0.0, QUAD) == (QUAD, "em_pair_kernel<4, 4, 32, 16, false, false>")
    for (T2, p2, q2) in ((64, 1, 2), (1025, 1, 2), (1000, 5, 2), (1000, 1, 5), (1000, 1, 4), (900, 4, 4)):
        assert _plan_name(T2, p2, q2, 0.0, PAIR)[0] == -1
        a, name = _plan_name(T2, p2, q2, 0.0, 0)
        assert a in (1, 2) and "em_pair" not in name
        y2, u2, v2 = _series(T2, p2, q2, 2, "dense")
        with pytest.raises(_lib.LdsrError):
            eng.em_batch(y2, u2, v2, synth.make_init_packed(p2, q2, 4, seed=1), niter=5, tol=0.0, algo=PAIR)


def test_niter_caps_and_interrupt_free_path(eng):
    """niter = 2, 3, 4 (the reference's E0, M, E1 prologue src/EM.cpp:251-256 and the first loop
    passes) with tol > 0."""
    from ldsr_amd import synth
    T, p, q = 1000, 1, 2
    y, u, v = _series(T, p, q, 21, "paleo")
    th0 = synth.make_init_packed(p, q, 9, seed=8)
    for niter in (2, 3, 4):
        ref = _oracle(y, u.T[None], v.T[None], np.zeros(9), th0, niter, 1e-5)
        r = eng.em_batch(y, u, v, th0, niter=niter, tol=1e-5, algo=PAIR)
        _check(r, ref, "niter=%d" % niter)
        assert np.all(r["n_iter"] == niter)
