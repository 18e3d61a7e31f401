"""GPU parity of the steady form of the one-wave-per-cell kernel (em_scan_steady.h): fully observed
series of 641..1024 steps run as three launches -- generic iterations until a cell's variance
recursion settles within the transient block, em_scan_steady_kernel, generic iterations for the
cells that kernel gave back.  The bar is SURVEY.md Appendix B's (identical n_iter per cell, then
theta and lik within |d| <= 1e-6 |ref| + 1e-9) against the CPU oracle, on every chunk length the
form exists for, every padded width, both schedules, cells that are slow at theta0 (they take
their first iterations in the first launch), cells forced through the give-back path
(LDSR_SCAN_GIVEBACK, a test hook read per call), series mixed with masked ones in one call, and
the likelihood traces.  BASELINE config 3 whole is tests/test_gpu_full_configs.py."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 1e-9
SCAN = 2


@pytest.fixture(scope="module")
def eng():
    import ldsr_amd
    return ldsr_amd


def _oracle(Y, U, V, soc, th0, niter, tol):
    from oracle import oracle as O
    return O.em_batch(np.atleast_2d(Y), np.ascontiguousarray(U), np.ascontiguousarray(V),
                      np.asarray(soc, np.int32), th0, niter, tol, n_threads=16)


def _check(r, ref, what):
    ref_th, ref_lik, ref_it, _ = ref
    bad = np.nonzero(r["n_iter"] != ref_it)[0]
    assert bad.size == 0, "%s: iteration counts differ at cells %s" % (what, bad[:10])
    assert np.all(r["status"] == 0), what
    assert parity_close(r["lik"], ref_lik, RTOL, ATOL), what
    assert parity_close(r["theta"], ref_th, RTOL, ATOL), what


def _last_kernel():
    from ldsr_amd import _lib
    buf = C.create_string_buffer(160)
    assert _lib.lib().ldsr_last_em_kernel(0, buf, 160) == 0
    return buf.value.decode()


def _slow_thetas(th0, p, q):
    """Every fourth cell: A near 1 with a small gain -- the variance recursion needs hundreds of steps."""
    th = th0.copy()
    th[::4, 0] = 0.995                    # A
    th[::4, 1 + p] = 0.05                 # C
    th[::4, 2 + p + q] = 1e-3             # Q
    th[::4, 3 + p + q] = 1.0              # R
    return th


# the form's chunk lengths: L = 12 (T = 641..768), 13, 14, 15, 16 (961..1024)
@pytest.mark.parametrize("T", [641, 704, 705, 768, 769, 832, 833, 896, 897, 960, 961, 1000, 1023, 1024])
def test_every_chunk_length_matches_oracle(eng, T):
    from ldsr_amd import synth
    p, q = 2, 3
    y, u, v = synth.make_series(T, p, q, series_id=300 + T)
    th0 = _slow_thetas(synth.make_init_packed(p, q, 19, seed=T), p, q)
    for niter, tol in ((25, 0.0), (300, 1e-5)):
        ref = _oracle(y, u.T[None], v.T[None], np.zeros(19), th0, niter, tol)
        r = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=SCAN)
        L = max(12, -(-T // 64))               # (no member with chunks of 11 steps)
        assert _last_kernel() == "em_scan_steady_kernel<2, 4, %d, %s>" % (L, "true" if tol > 0 else "false")
        _check(r, ref, "T=%d tol=%g" % (T, tol))


@pytest.mark.parametrize("p,q", [(1, 1), (1, 2), (2, 1), (1, 4), (3, 3), (4, 2), (1, 8), (5, 1), (4, 8), (8, 8), (7, 3)])
def test_every_padded_width(eng, p, q):
    from ldsr_amd import synth
    T = 1000
    y, u, v = synth.make_series(T, p, q, series_id=11 * p + q)
    th0 = _slow_thetas(synth.make_init_packed(p, q, 24, seed=100 + 9 * p + q), p, q)
    ref = _oracle(y, u.T[None], v.T[None], np.zeros(24), th0, 400, 1e-5)
    r = eng.em_batch(y, u, v, th0, niter=400, tol=1e-5, algo=SCAN, return_liks=True)
    assert _last_kernel().startswith("em_scan_steady_kernel<")
    _check(r, ref, "p=%d q=%d" % (p, q))
    # the traces: n_iter values, NaN beyond
    for c in range(24):
        n = r["n_iter"][c]
        assert np.all(np.isfinite(r["liks"][c, :n])) and np.all(np.isnan(r["liks"][c, n:]))
        assert r["liks"][c, n - 1] == r["lik"][c]
    # the static schedule on the same cells, a fixed number of iterations
    ref0 = _oracle(y, u.T[None], v.T[None], np.zeros(24), th0, 12, 0.0)
    r0 = eng.em_batch(y, u, v, th0, niter=12, tol=0.0, algo=SCAN)
    _check(r0, ref0, "p=%d q=%d fixed" % (p, q))


@pytest.mark.parametrize("give", [0, 1, 7])
def test_cells_given_back_finish_in_the_third_launch(eng, give):
    """LDSR_SCAN_GIVEBACK=k: the steady kernel hands every cell back at iteration k (a cell whose
    variance recursion stops settling within the block does the same); the third launch finishes
    them with generic iterations, same results."""
    from ldsr_amd import synth
    T, p, q = 1000, 4, 8
    y, u, v = synth.make_series(T, p, q, series_id=77)
    th0 = synth.make_init_packed(p, q, 21, seed=5)
    os.environ["LDSR_SCAN_GIVEBACK"] = str(give)
    try:
        for niter, tol in ((20, 0.0), (300, 1e-5)):
            ref = _oracle(y, u.T[None], v.T[None], np.zeros(21), th0, niter, tol)
            r = eng.em_batch(y, u, v, th0, niter=niter, tol=tol, algo=SCAN)
            _check(r, ref, "give back at %d, tol=%g" % (give, tol))
    finally:
        del os.environ["LDSR_SCAN_GIVEBACK"]


def test_masked_and_fully_observed_series_in_one_call(eng):
    """The device entry cannot look at y: it issues the three launches and the kernels decide per
    series.  Three series -- fully observed, scattered NA, fully observed -- with ragged cell counts."""
    import torch
    from ldsr_amd import _lib, synth
    T, p, q = 900, 1, 2
    ys, us, vs = [], [], []
    for s in range(3):
        y, u, v = synth.make_series(T, p, q, series_id=40 + s)
        y = y.copy()
        if s == 1:
            y[::5] = np.nan
        ys.append(y), us.append(u.T), vs.append(v.T)
    Y, U, V = np.stack(ys), np.stack(us), np.stack(vs)
    off = np.array([0, 13, 20, 41], np.int32)
    th0 = _slow_thetas(synth.make_init_packed(p, q, 41, seed=3), p, q)
    soc = np.repeat(np.arange(3), np.diff(off))
    for niter, tol in ((15, 0.0), (300, 1e-5)):
        ref = _oracle(Y, U, V, soc, th0, niter, tol)
        L = _lib.lib()
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        dY, dU, dV, dT0 = t(Y), t(U), t(V), t(th0)
        dT = torch.empty_like(dT0)
        dl = torch.empty(41, dtype=torch.float64, device=dev)
        dn = torch.empty(41, dtype=torch.int32, device=dev)
        ds = torch.empty(41, dtype=torch.int32, device=dev)
        wsb = L.ldsr_em_workspace_bytes(3, T, p, q, 41, SCAN)
        ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
        wp = (ws.data_ptr() + 255) & ~255
        offc = (C.c_int * 4)(*[int(x) for x in off])
        st = torch.cuda.current_stream(dev)
        _lib.check(L.ldsr_em_batch_device(0, C.c_void_p(st.cuda_stream), 3, T, p, q, dY.data_ptr(), dU.data_ptr(),
                                          dV.data_ptr(), 0, offc, dT0.data_ptr(), niter, tol, SCAN, dT.data_ptr(),
                                          dl.data_ptr(), dn.data_ptr(), ds.data_ptr(), None, C.c_void_p(wp), wsb))
        torch.cuda.synchronize()
        r = {"theta": dT.cpu().numpy(), "lik": dl.cpu().numpy(), "n_iter": dn.cpu().numpy(), "status": ds.cpu().numpy()}
        _check(r, ref, "mixed series tol=%g" % tol)
        # the host-pointer entry sees the NA and takes the one launch: same numbers
        r2 = eng.em_batch(Y, U.transpose(0, 2, 1), V.transpose(0, 2, 1), th0, cell_offsets=off, niter=niter, tol=tol,
                          algo=SCAN)
        assert _last_kernel().startswith("em_scan_kernel<")
        assert np.array_equal(r2["n_iter"], r["n_iter"])
        assert parity_close(r2["theta"], r["theta"], 1e-9, 1e-12)


def test_results_do_not_depend_on_the_launch_size_or_schedule(eng):
    """A cell's arithmetic depends on its own theta only: the same cell alone, among 300 others, in the
    static schedule and in the work queue (tol tiny: the queue schedule without early stops)."""
    from ldsr_amd import synth
    T, p, q = 1000, 1, 2
    y, u, v = synth.make_series(T, p, q, series_id=9)
    th0 = _slow_thetas(synth.make_init_packed(p, q, 301, seed=8), p, q)
    big = eng.em_batch(y, u, v, th0, niter=30, tol=0.0, algo=SCAN)
    for c in (0, 1, 150, 300):
        one = eng.em_batch(y, u, v, th0[c:c + 1], niter=30, tol=0.0, algo=SCAN)
        assert np.array_equal(one["theta"][0], big["theta"][c]) and one["lik"][0] == big["lik"][c]
    q_ = eng.em_batch(y, u, v, th0, niter=30, tol=1e-300, algo=SCAN)
    assert np.array_equal(q_["theta"], big["theta"]) and np.array_equal(q_["lik"], big["lik"])
