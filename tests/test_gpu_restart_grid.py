"""GPU tests of the one-call restart entry points (ldsr_em_restart_grid / _groups) and of the
asynchronous device entry.  The per-cell numerics are covered by test_gpu_parity.py; here the
new host/device plumbing is checked against the batch entry + host logic (bit-identical) and
against the oracle where a number is compared."""
import ctypes as C

import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import ldsr_amd
    from ldsr_amd import _lib
    assert _lib.lib().ldsr_device_count() >= 1, "no GPU visible"
    return ldsr_amd


def _cv_like(T=300, p=2, q=3, F=5, R=12, seed=11):
    from ldsr_amd import synth
    y, u, v = synth.make_series(T, p, q, series_id=31, mask="paleo", n_tail=60)
    Y = np.repeat(y[None], F, axis=0)
    for f in range(F):
        Y[f, T - 60 + 9 * f:T - 60 + 9 * f + 7] = np.nan
    off = (np.arange(F + 1) * R).astype(np.int32)
    if R > 6:
        off[2] -= 5      # ragged: folds need not have equal restart counts
    th0 = synth.make_init_packed(p, q, int(off[-1]), seed=seed)
    return Y, u, v, off, th0, p, q


def _reference_by_parts(eng, Y, u, v, off, th0, p, q, niter, tol):
    """The same result assembled from the batch entry + host selection + smoother call."""
    b = eng.em_batch(Y, u, v, th0, cell_offsets=off, niter=niter, tol=tol, return_liks=True)
    S = len(off) - 1
    win = np.array([off[s] + eng.select_restart(b["lik"][off[s]:off[s + 1]],
                                                b["theta"][off[s]:off[s + 1]], p, q)
                    for s in range(S)])
    fit = eng.smooth_batch(Y, u, v, b["theta"][win], cell_offsets=np.arange(S + 1, dtype=np.int32))
    return b, win, fit


@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)])
@pytest.mark.parametrize("tol", [1e-5, 0.0])
def test_restart_grid_equals_batch_plus_host_logic(eng, devices, tol):
    Y, u, v, off, th0, p, q = _cv_like()
    niter = 60
    r = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=niter, tol=tol, devices=devices)
    b, win, fit = _reference_by_parts(eng, Y, u, v, off, th0, p, q, niter, tol)
    for k in ("theta", "lik", "n_iter", "status"):
        assert np.array_equal(r["all"][k], b[k], equal_nan=True), k
    assert np.array_equal(r["winner"], win)
    assert np.array_equal(r["theta"], b["theta"][win])
    assert np.array_equal(r["lik"], b["lik"][win])
    assert np.array_equal(r["n_iter"], b["n_iter"][win])
    assert np.array_equal(r["liks"], b["liks"][win], equal_nan=True)
    for s in range(len(win)):                     # trace: n_iter finite entries, then NaN
        assert np.all(np.isfinite(r["liks"][s, :r["n_iter"][s]]))
        assert np.all(np.isnan(r["liks"][s, r["n_iter"][s]:]))
        assert r["liks"][s, r["n_iter"][s] - 1] == r["lik"][s]
    for k in "XYVJ":
        assert np.array_equal(r[k], fit[k]), k
    assert parity_close(fit["lik"], r["lik"], 1e-9, 1e-12)   # the fit is the winner's last E-step


def test_restart_grid_without_per_cell_outputs_and_optional_rows(eng):
    Y, u, v, off, th0, p, q = _cv_like(F=3, R=6)
    full = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=40, tol=1e-5)
    slim = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=40, tol=1e-5, return_all=False)
    assert "all" not in slim
    for k in ("winner", "theta", "lik", "n_iter", "X", "Y", "V", "J"):
        assert np.array_equal(full[k], slim[k]), k
    assert np.array_equal(full["liks"], slim["liks"], equal_nan=True)


def test_restart_grid_winner_rerun_path_is_identical(eng, monkeypatch):
    """Traces too large to keep on the device (cap forced to 0): the winners are re-run alone."""
    Y, u, v, off, th0, p, q = _cv_like()
    a = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=80, tol=1e-5)
    monkeypatch.setenv("LDSR_LIKS_TRACE_MAX_BYTES", "0")
    b = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=80, tol=1e-5)
    for k in ("winner", "theta", "lik", "n_iter", "X", "Y", "V", "J"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["liks"], b["liks"], equal_nan=True)


def test_restart_grid_series_without_a_selectable_winner(eng):
    """A fold whose restarts all end with NaN likelihoods has no winner (R's which.max on
    all-NA gives integer(0)): winner -1 and NaN rows, the other folds are unaffected."""
    Y, u, v, off, th0, p, q = _cv_like(F=3, R=4)
    th0 = th0.copy()
    th0[off[1]:off[2], 0] = np.nan               # NaN A in every restart of fold 1: NaN likelihoods throughout
    r = eng.em_restart_grid(Y, u, v, th0, cell_offsets=off, niter=3, tol=0.0)
    assert np.all(np.isnan(r["all"]["lik"][off[1]:off[2]]))
    assert r["winner"][1] == -1 and r["winner"][0] >= 0 and r["winner"][2] >= 0
    assert np.all(np.isnan(r["theta"][1])) and np.all(np.isnan(r["X"][1])) and np.all(np.isnan(r["liks"][1]))
    assert np.all(np.isfinite(r["X"][[0, 2]]))


def test_lds_em_restart_matches_oracle_winner(eng, p1case):
    from oracle import oracle as O
    from ldsr_amd import synth
    c = p1case
    th0 = synth.make_init_packed(7, 7, 24, seed=5)
    init = [eng.unpack_theta(t, 7, 7) for t in th0]
    win = eng.LDS_EM_restart(c["y"], c["u"], c["v"], init, niter=200, tol=1e-5)
    ref = [O.lds_em(c["y"], c["u"], c["v"], t, 200, 1e-5) for t in th0]
    k = O.select(np.array([r["lik"] for r in ref]), np.array([r["theta"][8] for r in ref]))
    assert win["all"]["selected"] == k
    assert len(win["liks"]) == len(ref[k]["liks"])
    assert parity_close(win["liks"], ref[k]["liks"], 1e-6, 1e-9)
    assert parity_close(win["lik"], ref[k]["lik"], 1e-6, 1e-9)
    for name in "XYVJ":
        assert parity_close(win["fit"][name][0], ref[k]["fit"][name], 1e-6, 1e-9), name
    assert win["fit"]["X"].shape == (1, 85)
    assert win["init"] is init[k]


def test_heterogeneous_ensemble_groups(eng, p1case):
    """tests/testthat/test-ensemble.R:4-18: ensemble members with different numbers of rows in
    u and v; every member must equal its own single-member run."""
    from ldsr_amd import synth
    c = p1case
    members = [(c["u"][:3], c["v"][:2]), (c["u"][1:3], c["v"][:5]), (None, c["v"][:4]),
               (c["u"], c["v"])]
    inits = [synth.make_init_packed(1 if u is None else u.shape[0], v.shape[0], 10 + 3 * i, seed=40 + i)
             for i, (u, v) in enumerate(members)]
    res = eng.ensemble_restart(c["y"], members, inits, niter=50, tol=1e-5)
    assert len(res) == len(members)
    for (u, v), th0, r in zip(members, inits, res):
        solo = eng.em_restart_grid(c["y"], u, v, th0, niter=50, tol=1e-5)
        for k in ("winner", "theta", "lik", "n_iter", "X", "Y", "V", "J"):
            assert np.array_equal(r[k], solo[k]), k
        assert np.array_equal(r["all"]["theta"], solo["all"]["theta"])


def test_cv_ensemble_folds_times_members(eng, npcase):
    """cvLDS with lists of u, v (R/LDS_reconstruction.R:377-381; tests/testthat/test-ensemble.R:9-18,
    non-contiguous folds): folds x members in one call == the mean of per-member cv_grid runs."""
    from ldsr_amd import cv
    c = npcase(1800)
    inst = np.nonzero(~np.isnan(c["y"]))[0]
    Z = [np.array([1, 5, 9, 13, 40]), np.arange(20, 27), np.array([0, 2, 44, 45])]
    members = [(c["u"][:3], c["v"][:2]), (c["u"][:2], c["v"][:3])]         # 3 and 2 rows, as the reference's test
    r = cv.cv_grid_ensemble(c["y"], members, inst, Z, num_restarts=6, niter=60, tol=1e-5, seed=3, mu=c["mu"])
    assert r["Ycv"].shape == (3, inst.size) and np.all(np.isfinite(r["Ycv"]))
    solo = [cv.cv_grid(c["y"], u, v, inst, Z, num_restarts=6, niter=60, tol=1e-5, seed=3 + 1000 * g)
            for g, (u, v) in enumerate(members)]
    assert np.allclose(r["Ycv"], np.mean([s["Ycv"] for s in solo], axis=0) + c["mu"], rtol=0, atol=1e-12)
    for g in range(2):
        assert np.array_equal(r["members"][g]["theta"], solo[g]["theta"])


def _device_call(L, stream_ptr, S, T, p, q, d, off_c, niter, tol, ws_ptr, wsb):
    from ldsr_amd import _lib
    _lib.check(L.ldsr_em_batch_device(
        0, C.c_void_p(stream_ptr), S, T, p, q, d["y"].data_ptr(), d["u"].data_ptr(),
        d["v"].data_ptr(), 1, off_c, d["th0"].data_ptr(), niter, tol, 0, d["th"].data_ptr(),
        d["lik"].data_ptr(), d["nit"].data_ptr(), d["st"].data_ptr(), None, C.c_void_p(ws_ptr), wsb))


def test_device_entry_is_asynchronous_and_keeps_no_host_pointer(eng):
    """ldsr_em_batch_device on a non-blocking stream: the host block table / cell_offsets are
    clobbered and freed right after each call returns, calls are issued back to back without any
    synchronisation, and the results must equal the synchronous host-pointer entry."""
    import torch
    from ldsr_amd import _lib
    L = _lib.lib()
    Y, u, v, off, th0, p, q = _cv_like(T=400, F=6, R=20)
    S, T = Y.shape
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)          # non-blocking w.r.t. the null stream
    ref = eng.em_batch(Y, u, v, th0, cell_offsets=off, niter=50, tol=1e-5)
    d = {"y": torch.from_numpy(Y).to(dev), "u": torch.from_numpy(np.ascontiguousarray(u.T)).to(dev),
         "v": torch.from_numpy(np.ascontiguousarray(v.T)).to(dev), "th0": torch.from_numpy(th0).to(dev)}
    n = th0.shape[0]
    wsb = L.ldsr_em_workspace_bytes(S, T, p, q, n, 0)
    outs = []
    torch.cuda.synchronize(dev)
    for rep in range(20):                           # wraps the 16-slot staging ring
        o = {"th": torch.empty_like(d["th0"]), "lik": torch.empty(n, dtype=torch.float64, device=dev),
             "nit": torch.empty(n, dtype=torch.int32, device=dev),
             "st": torch.empty(n, dtype=torch.int32, device=dev),
             "ws": torch.empty(wsb + 256, dtype=torch.uint8, device=dev)}
        off_c = (C.c_int * (S + 1))(*[int(x) for x in off])
        _device_call(L, stream.cuda_stream, S, T, p, q, {**d, **o}, off_c, 50, 1e-5,
                     (o["ws"].data_ptr() + 255) & ~255, wsb)
        for i in range(S + 1):                      # clobber, then drop, the host table
            off_c[i] = -12345
        del off_c
        outs.append(o)
    stream.synchronize()
    for o in outs:
        assert np.array_equal(o["nit"].cpu().numpy(), ref["n_iter"])
        assert np.array_equal(o["th"].cpu().numpy(), ref["theta"])
        assert np.array_equal(o["lik"].cpu().numpy(), ref["lik"])


def test_queue_is_reset_between_launches_sharing_a_workspace(eng):
    """Regression for the work-queue schedule (tol > 0): the per-series queue heads live in the
    workspace and are reset by series_prep of every launch, so back-to-back launches on ONE
    workspace -- with different grids -- each process every cell exactly once."""
    import torch
    from ldsr_amd import _lib
    L = _lib.lib()
    Y, u, v, off, th0, p, q = _cv_like(T=400, F=6, R=20)
    S, T = Y.shape
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    n = th0.shape[0]
    d = {"y": torch.from_numpy(Y).to(dev), "u": torch.from_numpy(np.ascontiguousarray(u.T)).to(dev),
         "v": torch.from_numpy(np.ascontiguousarray(v.T)).to(dev), "th0": torch.from_numpy(th0).to(dev)}
    wsb = L.ldsr_em_workspace_bytes(S, T, p, q, n, 0)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) & ~255
    off2 = off.copy()
    off2[1:] -= off[1] // 2                          # a smaller first series: another grid
    th0b = th0[off[1] // 2:]
    for offs, t0 in ((off, th0), (off2, th0b), (off, th0)):
        m = t0.shape[0]
        o = {"th0": torch.from_numpy(np.ascontiguousarray(t0)).to(dev),
             "th": torch.full((m, t0.shape[1]), np.nan, dtype=torch.float64, device=dev),
             "lik": torch.full((m,), np.nan, dtype=torch.float64, device=dev),
             "nit": torch.full((m,), -1, dtype=torch.int32, device=dev),
             "st": torch.full((m,), -1, dtype=torch.int32, device=dev)}
        off_c = (C.c_int * (S + 1))(*[int(x) for x in offs])
        _device_call(L, stream.cuda_stream, S, T, p, q, {**d, **o}, off_c, 40, 1e-5, ws_ptr, wsb)
        torch.cuda.synchronize(dev)
        ref = eng.em_batch(Y, u, v, t0, cell_offsets=offs, niter=40, tol=1e-5)
        assert np.array_equal(o["nit"].cpu().numpy(), ref["n_iter"])      # no cell skipped (-1) or redone
        assert np.array_equal(o["th"].cpu().numpy(), ref["theta"])


def test_interrupt_callback_stops_cells_and_reports_it(eng):
    """C ABI level: a registered callback is polled by the calling thread while the device
    works; once it returns non-zero every cell stops at a multiple of 64 iterations with status
    LDSR_CELL_INTERRUPTED and the call returns LDSR_EINTERRUPTED.  Unregistered: no polling."""
    from ldsr_amd import _lib, synth
    L = _lib.lib()
    y, u, v = synth.make_series(1000, 1, 2, series_id=6)
    th0 = synth.make_init_packed(1, 2, 1024, seed=2)
    calls = {"n": 0}

    @C.CFUNCTYPE(C.c_int, C.c_void_p)
    def cb(_):
        calls["n"] += 1
        return 1 if calls["n"] >= 5 else 0

    assert L.ldsr_set_interrupt_callback(C.cast(cb, C.c_void_p), None) == 0
    try:
        for devices in (None, [0, 0]):              # the caller's thread polls in both forms
            calls["n"] = 0
            with pytest.raises(_lib.LdsrError, match="error 4"):
                eng.em_batch(y, u, v, th0, niter=40000, tol=0.0, devices=devices)
            assert calls["n"] >= 5
    finally:
        assert L.ldsr_set_interrupt_callback(None, None) == 0
    r = eng.em_batch(y, u, v, th0[:64], niter=130, tol=0.0)       # no callback: runs to the cap
    assert np.all(r["n_iter"] == 130) and np.all(r["status"] == 0)


@pytest.mark.parametrize("shape", ["masked_queue", "long_series_queue", "dense_pair_queue"])
def test_interrupt_with_early_stopping_leaves_no_stale_cells(eng, shape):
    """tol > 0 runs use the work queue: cells converge after a few dozen iterations and waves keep
    pulling new ones.  Round 2 polled the flag on the CELL's iteration counter (a cell that stops
    before its 64th iteration never polled) and kept pulling after an abort, so an interrupted
    converging grid ran to completion; and cells never pulled kept whatever the arena held.  Now:
    the poll counts the wave's iterations, an aborted wave only marks what the queue still hands
    it.  The call must return LDSR_EINTERRUPTED well before the grid is done and every cell must
    end as finished (status 0 / 1 with its own n_iter) or as interrupted (status 3) -- no cell may
    keep the sentinel the outputs were pre-filled with."""
    import time
    from ldsr_amd import _lib, synth
    L = _lib.lib()
    T, p, q, n, mask = {"masked_queue": (1000, 1, 2, 200000, "scatter"),
                        "long_series_queue": (2500, 1, 2, 40000, "dense"),
                        "dense_pair_queue": (1000, 1, 2, 200000, "dense")}[shape]
    y, u, v = synth.make_series(T, p, q, series_id=6)
    if mask == "scatter":
        y = y.copy()
        y[::7] = np.nan
    th0 = synth.make_init_packed(p, q, n, seed=2)
    P = 6 + p + q
    theta = np.full((n, P), -7.0)
    lik = np.full(n, -7.0)
    nit = np.full(n, -7, np.int32)
    st = np.full(n, -7, np.int32)
    off = np.array([0, n], np.int32)
    U = np.ascontiguousarray(u.T)
    V = np.ascontiguousarray(v.T)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    calls = {"n": 0}

    @C.CFUNCTYPE(C.c_int, C.c_void_p)
    def cb(_):
        calls["n"] += 1
        return 1 if calls["n"] >= 4 else 0

    def run():
        return L.ldsr_em_batch(0, 1, T, p, q, y.ctypes.data_as(dp), U.ctypes.data_as(dp), V.ctypes.data_as(dp), 0,
                               off.ctypes.data_as(ip), th0.ctypes.data_as(dp), 1000, 1e-5, 0,
                               theta.ctypes.data_as(dp), lik.ctypes.data_as(dp), nit.ctypes.data_as(ip),
                               st.ctypes.data_as(ip), None)

    t0 = time.perf_counter()
    assert run() == 0                                         # uninterrupted: the whole grid
    t_full = time.perf_counter() - t0
    assert np.all(st == 0) and nit.max() < 1000 and nit.min() >= 3
    done_full = nit.copy()
    for a in (theta, lik):
        a[...] = -7.0
    nit[...] = -7
    st[...] = -7
    assert L.ldsr_set_interrupt_callback(C.cast(cb, C.c_void_p), None) == 0
    try:
        t0 = time.perf_counter()
        rc = run()
        t_int = time.perf_counter() - t0
    finally:
        assert L.ldsr_set_interrupt_callback(None, None) == 0
    assert rc == 4 and calls["n"] >= 4                         # LDSR_EINTERRUPTED
    # (the device -> host copy of the per-cell arrays is skipped on an interrupted call: what came
    # back is what the interrupted call wrote, or nothing)
    untouched = np.all(st == -7)
    if not untouched:
        assert not np.any(st == -7), "cells with stale outputs"
        assert np.all((st == 0) | (st == 3))
        fin = st == 0
        assert np.array_equal(nit[fin], done_full[fin])       # finished cells are complete results
        assert np.all(nit[st == 3] < 1000)
        assert np.any(st == 3)
    assert t_int < 0.7 * t_full, (t_int, t_full)             # it stopped, it did not run the grid out
