"""The R .Call shim (ldsr_amd/r_shim/ldsrhip_call.c) EXECUTED against tests/r_mock/rmock.c, a
miniature of the R C API entry points it uses (R itself is not installed in this image).

CPU part: the registration table (names / arity of src/RcppExports.cpp:132-143 for the entries
it replaces), argument checking, error unwinding and PROTECT balance.  GPU part: the reference's
own known-answer test (tests/testthat/test-LDS-EM.R:21-41) driven through the .Call entry points,
and the batch / grid entries against the Python binding of the same C ABI (bit-identical)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 10, 13, 14, 16, 19


class RMock:
    def __init__(self, so):
        L = self.L = C.CDLL(so)
        vp, ip = C.c_void_p, C.c_int
        for name, res, args in [
                ("rmock_init", ip, []), ("rmock_unload", None, []), ("rmock_n_routines", ip, []),
                ("rmock_routine_name", C.c_char_p, [ip]), ("rmock_routine_nargs", ip, [ip]),
                ("rmock_last_error", C.c_char_p, []), ("rmock_protect_depth", ip, []),
                ("rmock_interrupt_polls", ip, []), ("rmock_reset", None, []),
                ("rmock_interrupt_after", None, [ip]),
                ("rmock_call", ip, [C.c_char_p, ip, C.POINTER(vp), C.POINTER(vp)]),
                ("rmock_real_matrix", vp, [ip, ip, C.POINTER(C.c_double)]),
                ("rmock_int_matrix", vp, [ip, ip, C.POINTER(C.c_int)]),
                ("rmock_scalar", vp, [ip, C.c_double]), ("rmock_list", vp, [ip, ip]),
                ("rmock_list_set", None, [vp, ip, C.c_char_p, vp]), ("rmock_type", ip, [vp]),
                ("rmock_len", C.c_long, [vp]), ("rmock_nrow", ip, [vp]), ("rmock_ncol", ip, [vp]),
                ("rmock_data", vp, [vp]), ("rmock_elt", vp, [vp, ip]), ("rmock_name", C.c_char_p, [vp, ip])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        assert L.rmock_init() == 1

    def to_sexp(self, x):
        L = self.L
        if isinstance(x, np.ndarray) and x.dtype.kind in "iub":   # an R integer / logical matrix
            a = np.asfortranarray(np.atleast_2d(x), dtype=np.int32)
            return L.rmock_int_matrix(a.shape[0], a.shape[1], a.ctypes.data_as(C.POINTER(C.c_int)))
        if isinstance(x, np.ndarray):                       # R matrices are column-major
            a = np.asfortranarray(np.atleast_2d(x), dtype=np.float64)
            return L.rmock_real_matrix(a.shape[0], a.shape[1], a.ctypes.data_as(C.POINTER(C.c_double)))
        if isinstance(x, bool):
            return L.rmock_scalar(LGLSXP, float(x))
        if isinstance(x, int):
            return L.rmock_scalar(INTSXP, float(x))
        if isinstance(x, float):
            return L.rmock_scalar(REALSXP, x)
        if isinstance(x, dict):
            lst = L.rmock_list(len(x), 1)
            for i, (k, v) in enumerate(x.items()):
                L.rmock_list_set(lst, i, k.encode(), self.to_sexp(v))
            return lst
        if isinstance(x, (list, tuple)):
            lst = L.rmock_list(len(x), 0)
            for i, v in enumerate(x):
                L.rmock_list_set(lst, i, None, self.to_sexp(v))
            return lst
        raise TypeError(type(x))

    def from_sexp(self, s):
        L = self.L
        t, n = L.rmock_type(s), L.rmock_len(s)
        if t == REALSXP:
            a = np.ctypeslib.as_array(C.cast(L.rmock_data(s), C.POINTER(C.c_double)), (max(n, 1),))[:n].copy()
            nr, nc = L.rmock_nrow(s), L.rmock_ncol(s)
            return a.reshape((nr, nc), order="F") if (nr or nc) else a
        if t in (INTSXP, LGLSXP):
            return np.ctypeslib.as_array(C.cast(L.rmock_data(s), C.POINTER(C.c_int)), (max(n, 1),))[:n].copy()
        if t == VECSXP:
            names = [L.rmock_name(s, i).decode() for i in range(n)]
            vals = [self.from_sexp(L.rmock_elt(s, i)) for i in range(n)]
            return dict(zip(names, vals)) if any(names) else vals
        raise TypeError("SEXP type %d" % t)

    def call(self, name, *args):
        """.Call(name, ...): returns the converted result; raises RuntimeError on Rf_error.  The
        PROTECT stack must be balanced either way."""
        L = self.L
        a = (C.c_void_p * len(args))(*[self.to_sexp(x) for x in args])
        res = C.c_void_p()
        rc = L.rmock_call(name.encode(), len(args), a, C.byref(res))
        depth = L.rmock_protect_depth()
        try:
            assert rc != 2, "no .Call routine %s with %d arguments" % (name, len(args))
            if rc == 1:
                raise RuntimeError(L.rmock_last_error().decode())
            assert depth == 0, "PROTECT stack unbalanced after %s: depth %d" % (name, depth)
            return self.from_sexp(res.value)
        finally:
            L.rmock_reset()


@pytest.fixture(scope="module")
def R(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("rmock") / "libldsrhip_mock.so")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-shared", "-fPIC",
                           "-I", os.path.join(ROOT, "tests", "r_api_stub"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "ldsr_amd", "r_shim", "ldsrhip_call.c"),
                           os.path.join(ROOT, "tests", "r_mock", "rmock.c"),
                           "-L", os.path.join(ROOT, "ldsr_amd"), "-lldsr_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "ldsr_amd"), "-o", so])
    return RMock(so)


def _theta_list(th, p, q):
    """Packed theta -> the reference's named list of matrices (src/EM.cpp:221-228)."""
    th = np.asarray(th, float)
    return {"A": th[0:1].reshape(1, 1), "B": th[1:1 + p].reshape(1, p), "C": th[1 + p:2 + p].reshape(1, 1),
            "D": th[2 + p:2 + p + q].reshape(1, q), "Q": th[2 + p + q:3 + p + q].reshape(1, 1),
            "R": th[3 + p + q:4 + p + q].reshape(1, 1), "mu1": th[4 + p + q:5 + p + q].reshape(1, 1),
            "V1": th[5 + p + q:].reshape(1, 1)}


def _has_gpu():
    from ldsr_amd import _lib
    return _lib.lib().ldsr_device_count() > 0


# ---- CPU: registration, argument checks, error unwinding ------------------------------------
def test_registration_table_mirrors_the_reference(R):
    got = {R.L.rmock_routine_name(i).decode(): R.L.rmock_routine_nargs(i) for i in range(R.L.rmock_n_routines())}
    # arity of the entries they replace: src/RcppExports.cpp:132-143
    assert got == {"ldsrhip_LDS_EM_batch": 6, "ldsrhip_LDS_EM_grid": 6, "ldsrhip_LDS_EM": 6,
                   "ldsrhip_LDS_EM_batch_raw": 6, "ldsrhip_LDS_EM_grid_raw": 6, "ldsrhip_LDS_EM_groups": 7,
                   "ldsrhip_Kalman_smoother": 5, "ldsrhip_propagate": 5, "ldsrhip_Mstep": 4}


def test_loaded_as_ldsr_it_registers_the_reference_table(R, refdata):
    """Deployment shape (ii): the same shim object built as ldsr.so answers R_init_ldsr with the
    nine routines of src/RcppExports.cpp:132-143 -- names AND arities -- plus the two batched
    entries; the five metric routines (host code) are called here through .Call."""
    L = R.L
    L.rmock_init_as_ldsr.restype = C.c_int
    assert L.rmock_init_as_ldsr() == 1
    try:
        got = {L.rmock_routine_name(i).decode(): L.rmock_routine_nargs(i) for i in range(L.rmock_n_routines())}
        reference = {"_ldsr_Kalman_smoother": 5, "_ldsr_Mstep": 4, "_ldsr_LDS_EM": 6, "_ldsr_propagate": 5,
                     "_ldsr_NSE": 2, "_ldsr_nRMSE": 3, "_ldsr_corr": 2, "_ldsr_KGE": 2, "_ldsr_RE": 3}
        assert {k: v for k, v in got.items() if k.startswith("_ldsr_")} == reference
        assert {k: v for k, v in got.items() if not k.startswith("_ldsr_")} == {
            "ldsrhip_LDS_EM_batch": 6, "ldsrhip_LDS_EM_grid": 6, "ldsrhip_LDS_EM_batch_raw": 6,
            "ldsrhip_LDS_EM_grid_raw": 6, "ldsrhip_LDS_EM_groups": 7}
        # the metric entries on fold 0 of the reference-held NPcv object
        from ldsr_amd import cv
        c = refdata["NPcv"]
        z = np.asarray(c["Z"][0]) - 1
        sim, obs = np.asarray(c["Ycv"])[0][z], np.asarray(c["target"])[z]
        assert R.call("_ldsr_NSE", sim, obs)[0] == pytest.approx(c["metrics_dist"]["CE"][0], rel=1e-10)
        assert R.call("_ldsr_KGE", sim, obs)[0] == pytest.approx(c["metrics_dist"]["KGE"][0], rel=1e-10)
        assert R.call("_ldsr_corr", sim, obs)[0] == pytest.approx(np.corrcoef(sim, obs)[0, 1], rel=1e-12)
        assert R.call("_ldsr_nRMSE", sim, obs, float(np.mean(c["target"])))[0] == pytest.approx(
            c["metrics_dist"]["nRMSE"][0], rel=1e-10)
        tr = np.delete(np.asarray(c["target"]), z)
        assert R.call("_ldsr_RE", sim, obs, float(tr.mean()))[0] == pytest.approx(c["metrics_dist"]["RE"][0], rel=1e-10)
        with pytest.raises(RuntimeError, match="one \\(positive\\) length"):
            R.call("_ldsr_NSE", sim, obs[:5])
        # integer / logical vectors are coerced like Rcpp's NumericVector parameters: NSE(1:10, obs)
        ints = np.arange(1, sim.size + 1)
        assert R.call("_ldsr_NSE", ints, obs)[0] == R.call("_ldsr_NSE", ints.astype(float), obs)[0]
        assert R.call("_ldsr_corr", obs, ints % 2 == 0)[0] == pytest.approx(
            np.corrcoef(obs, (ints % 2 == 0).astype(float))[0, 1], rel=1e-12)
        with pytest.raises(RuntimeError, match="must be numeric"):
            R.call("_ldsr_NSE", [1.0, 2.0], obs)
        assert L.rmock_protect_depth() == 0
    finally:
        assert L.rmock_init() == 1            # back to the side-car table for the other tests


def test_argument_errors_unwind_cleanly(R, p1case):
    c = p1case
    y = c["y"][None, :]
    th = _theta_list(c["theta0"], 7, 7)
    bad = dict(th)
    del bad["Q"]
    with pytest.raises(RuntimeError, match="element 'Q' not found"):       # lists are read BY NAME
        R.call("ldsrhip_LDS_EM_batch", y, c["u"], c["v"], [bad], 10, 1e-5)
    short = dict(th, B=np.ones((1, 3)))
    with pytest.raises(RuntimeError, match="theta\\$B has the wrong length"):
        R.call("ldsrhip_Kalman_smoother", y, c["u"], c["v"], short, True)
    with pytest.raises(RuntimeError, match="ncol\\(y\\) columns"):
        R.call("ldsrhip_LDS_EM_batch", y, c["u"][:, :50], c["v"], [th], 10, 1e-5)
    with pytest.raises(RuntimeError, match="niter must be >= 2"):           # src/EM.cpp:256 reads lik[1]
        R.call("ldsrhip_LDS_EM_batch", y, c["u"], c["v"], [th], 1, 1e-5)
    with pytest.raises(RuntimeError, match="one init list per column"):
        R.call("ldsrhip_LDS_EM_grid", np.stack([c["y"], c["y"]], axis=1), c["u"], c["v"], [[th]], 10, 1e-5)
    with pytest.raises(RuntimeError, match="one element per ensemble member"):
        R.call("ldsrhip_LDS_EM_groups", c["y"][:, None], [c["u"], c["u"]], [c["v"]], [[[th]]], 10, 1e-5, False)
    with pytest.raises(RuntimeError, match="one init list per column of Y"):
        R.call("ldsrhip_LDS_EM_groups", c["y"][:, None], [c["u"]], [c["v"]], [[[th], [th]]], 10, 1e-5, False)
    with pytest.raises(RuntimeError, match="u must be numeric"):
        R.call("ldsrhip_LDS_EM_batch", y, [1.0], c["v"], [th], 10, 1e-5)
    assert R.L.rmock_protect_depth() == 0


def test_without_a_gpu_the_shim_raises_an_r_error(R, p1case):
    if _has_gpu():
        pytest.skip("GPU present")
    c = p1case
    th = _theta_list(c["theta0"], 7, 7)
    with pytest.raises(RuntimeError, match="no ROCm device"):
        R.call("ldsrhip_LDS_EM_batch", c["y"][None, :], c["u"], c["v"], [th], 10, 1e-5)
    with pytest.raises(RuntimeError, match="ldsr_smooth_batch"):
        R.call("ldsrhip_Kalman_smoother", c["y"][None, :], c["u"], c["v"], th, True)


# ---- GPU: the reference's known-answer test through .Call -------------------------------------
@pytest.mark.gpu
def test_known_answer_through_the_call_entries(R, p1case):
    """tests/testthat/test-LDS-EM.R:21-41, same call sequence and tolerance, via .Call."""
    c = p1case
    y, u, v = c["y"][None, :], c["u"], c["v"]
    theta0 = _theta_list(c["theta0"], 7, 7)
    smooth1 = R.call("ldsrhip_Kalman_smoother", y, u, v, theta0, True)
    theta1 = R.call("ldsrhip_Mstep", y, u, v, smooth1)
    smooth2 = R.call("ldsrhip_Kalman_smoother", y, u, v, theta1, True)
    theta2 = R.call("ldsrhip_Mstep", y, u, v, smooth2)
    tol = 1e-6
    assert smooth1["lik"][0] == pytest.approx(-11.678657, abs=tol)
    assert smooth1["X"].shape == (1, 85) and smooth1["J"].shape == (1, 85)
    assert smooth1["X"][0, [0, 84]] == pytest.approx([1.293356, -0.987671], abs=tol)
    assert (theta1["A"][0, 0], theta1["C"][0, 0], theta1["Q"][0, 0]) == pytest.approx(
        (0.606066, -0.005995, 3.640236), abs=tol)
    assert theta1["B"].shape == (1, 7) and theta1["D"].shape == (1, 7)
    assert smooth2["lik"][0] == pytest.approx(-0.114224, abs=tol)
    assert (theta2["A"][0, 0], theta2["C"][0, 0], theta2["Q"][0, 0]) == pytest.approx(
        (0.603945, -0.012004, 3.644322), abs=tol)
    fit = R.call("ldsrhip_LDS_EM", y, u, v, theta0, 100, 1e-5)
    assert list(fit) == ["theta", "fit", "liks", "lik"]                 # src/EM.cpp:276-279
    assert fit["liks"].shape == (68, 1)                                 # length(liks) == 68
    assert fit["lik"][0] == pytest.approx(-0.039093, abs=tol)
    prop = R.call("ldsrhip_propagate", theta0, u, v, y, True)
    assert list(prop) == ["X", "Y", "V", "lik"]                         # src/EM.cpp:352-355


@pytest.mark.gpu
def test_batch_entry_equals_the_python_binding(R, p1case):
    import ldsr_amd
    from ldsr_amd import synth
    c = p1case
    y = c["y"].copy()
    y[[2, 3, 60]] = np.nan                                   # NA_real_ is a NaN
    for u, v, p, q in ((c["u"][:3], c["v"][:2], 3, 2), (None, c["v"][:4], 1, 4)):
        th0 = synth.make_init_packed(p, q, 12, seed=31)
        init = [_theta_list(t, p, q) for t in th0]
        u_r = np.zeros((1, 1)) if u is None else u          # the matrix(0) sentinel (src/EM.cpp:71)
        m = R.call("ldsrhip_LDS_EM_batch", y[None, :], u_r, v, init, 60.0, 1e-5)   # niter as a double
        ref = ldsr_amd.em_restart_grid(y, u, v, th0, niter=60, tol=1e-5)
        assert list(m) == ["theta", "fit", "liks", "lik", "index", "all"]
        assert m["index"][0] == ref["winner"][0] + 1                       # 1-based
        packed = np.concatenate([m["theta"][k].ravel() for k in ("A", "B", "C", "D", "Q", "R", "mu1", "V1")])
        assert np.array_equal(packed, ref["theta"][0])
        assert m["theta"]["B"].shape == (1, p) and m["theta"]["D"].shape == (1, q)
        if u is None:
            assert m["theta"]["B"][0, 0] == 0.0                            # B.zeros(1, p), src/EM.cpp:186
        n_it = int(ref["n_iter"][0])
        assert m["liks"].shape == (n_it, 1) and np.array_equal(m["liks"][:, 0], ref["liks"][0, :n_it])
        for k in "XYVJ":
            assert m["fit"][k].shape == (1, 85) and np.array_equal(m["fit"][k][0], ref[k][0])
        assert m["fit"]["lik"][0] == m["lik"][0] == ref["lik"][0]
        assert np.array_equal(m["all"]["lik"], ref["all"]["lik"], equal_nan=True)
        assert np.array_equal(m["all"]["C"], ref["all"]["theta"][:, 1 + p])
        assert np.array_equal(m["all"]["n_iter"], ref["all"]["n_iter"])
    assert R.L.rmock_interrupt_polls() >= 2                 # R_CheckUserInterrupt before each launch


@pytest.mark.gpu
def test_grid_entry_equals_the_python_binding(R, npcase):
    """cvLDS's fold loop (R/LDS_reconstruction.R:373-375) in one .Call: Y is T x F."""
    import ldsr_amd
    from ldsr_amd import synth
    c = npcase(1800)
    T = c["y"].size
    inst = np.nonzero(~np.isnan(c["y"]))[0]
    folds = [inst[3:8], inst[20:26], inst[[1, 9, 30]]]
    Y = np.repeat(c["y"][None], 3, axis=0)
    for f, z in enumerate(folds):
        Y[f, z] = np.nan
    counts = [5, 8, 3]                                         # fresh restarts per fold, ragged
    th0 = synth.make_init_packed(3, 3, sum(counts), seed=33)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    inits = [[_theta_list(t, 3, 3) for t in th0[off[f]:off[f + 1]]] for f in range(3)]
    ms = R.call("ldsrhip_LDS_EM_grid", np.ascontiguousarray(Y.T), c["u"], c["v"], inits, 80, 1e-5)
    ref = ldsr_amd.em_restart_grid(Y, c["u"], c["v"], th0, cell_offsets=off, niter=80, tol=1e-5)
    assert len(ms) == 3
    for f, m in enumerate(ms):
        assert m["index"][0] == ref["winner"][f] - off[f] + 1
        assert m["fit"]["Y"].shape == (1, T) and np.array_equal(m["fit"]["Y"][0], ref["Y"][f])
        assert m["lik"][0] == ref["lik"][f]
        assert m["theta"]["C"][0, 0] == ref["theta"][f, 4]


@pytest.mark.gpu
def test_user_interrupt_stops_a_running_launch(R, p1case):
    """The reference polls Rcpp::checkUserInterrupt() every 100 iterations (src/EM.cpp:261-262).
    Here the .Call thread polls R_CheckUserInterrupt (inside R_ToplevelExec) about once per
    millisecond while the GPU works; a pending interrupt stops every cell within 64 iterations and
    the routine raises an R error with the PROTECT stack unwound."""
    import time
    from ldsr_amd import synth
    y, u, v = synth.make_series(1000, 1, 2, series_id=5)
    th0 = synth.make_init_packed(1, 2, 2048, seed=8)
    init = [_theta_list(t, 1, 2) for t in th0]
    args = (y[None, :], u, v, init, 40000, 0.0)               # ~0.3 s of EM if left alone
    R.call("ldsrhip_LDS_EM_batch", y[None, :], u, v, init[:8], 5, 0.0)     # warm up (allocations)
    R.L.rmock_interrupt_after(10)                              # the 10th poll from now is a Ctrl-C
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="interrupted by the user"):
        R.call("ldsrhip_LDS_EM_batch", *args)
    dt = time.perf_counter() - t0
    assert dt < 0.15, "the launch was not cut short: %.3f s" % dt
    assert R.L.rmock_protect_depth() == 0
    m = R.call("ldsrhip_LDS_EM_batch", y[None, :], u, v, init[:8], 5, 0.0)   # the next call is unaffected
    assert m["liks"].shape == (5, 1)


@pytest.mark.gpu
def test_raw_trajectories_of_the_winners(R, npcase):
    """use.raw / return.raw (R/LDS_reconstruction.R:279-281, :219-222): the batched entries also return
    propagate(theta, u, v, y) of every winner -- here against the oracle's propagate (src/EM.cpp:295-356)
    at the returned theta, fold by fold; integer-typed inputs are coerced on the way in."""
    from oracle import oracle as O
    from ldsr_amd import synth
    c = npcase(1800)
    T = c["y"].size
    inst = np.nonzero(~np.isnan(c["y"]))[0]
    folds = [inst[3:8], inst[20:26]]
    Y = np.repeat(c["y"][None], 2, axis=0)
    for f, z in enumerate(folds):
        Y[f, z] = np.nan
    th0 = synth.make_init_packed(3, 3, 12, seed=41)
    inits = [[_theta_list(t, 3, 3) for t in th0[6 * f:6 * f + 6]] for f in range(2)]
    plain = R.call("ldsrhip_LDS_EM_grid", np.ascontiguousarray(Y.T), c["u"], c["v"], inits, 60, 1e-5)
    ms = R.call("ldsrhip_LDS_EM_grid_raw", np.ascontiguousarray(Y.T), c["u"], c["v"], inits, 60, 1e-5)
    for f, (m, m0) in enumerate(zip(ms, plain)):
        assert list(m) == ["theta", "fit", "liks", "lik", "index", "raw"] and list(m["raw"]) == ["X", "Y", "V", "lik"]
        assert np.array_equal(m["fit"]["Y"], m0["fit"]["Y"]) and m["index"][0] == m0["index"][0]
        th = np.concatenate([m["theta"][k].ravel() for k in ("A", "B", "C", "D", "Q", "R", "mu1", "V1")])
        ref = O.propagate(th, c["u"], c["v"], Y[f])
        for k in "XYV":
            assert m["raw"][k].shape == (1, T)
            assert np.allclose(m["raw"][k][0], np.asarray(ref[k]).ravel(), rtol=1e-9, atol=1e-12), (f, k)
        assert m["raw"]["lik"][0] == pytest.approx(ref["lik"], rel=1e-9)
    one = R.call("ldsrhip_LDS_EM_batch_raw", Y[:1], c["u"], c["v"], inits[0], 60, 1e-5)
    assert list(one) == ["theta", "fit", "liks", "lik", "index", "raw", "all"]
    assert np.array_equal(one["raw"]["Y"], ms[0]["raw"]["Y"])


@pytest.mark.gpu
def test_ensemble_members_in_one_call(R, npcase):
    """LDS_reconstruction's ensemble loop (R/LDS_reconstruction.R:242-246) and cvLDS's folds x members
    loop (:377-381) through ONE .Call: members differ in p and q (tests/testthat/test-ensemble.R:4-5: three
    rows and two), one of them has no u at all; every (member, fold) model equals the grid entry run for
    that member alone, bit for bit."""
    from ldsr_amd import synth
    c = npcase(1800)
    inst = np.nonzero(~np.isnan(c["y"]))[0]
    Y = np.repeat(c["y"][None], 2, axis=0)
    Y[1, inst[5:12]] = np.nan
    members = [(c["u"], c["v"], 3, 3), (c["u"][:2], c["v"][:2], 2, 2), (np.zeros((1, 1)), c["v"][:1], 1, 1)]
    us, vs, inits = [], [], []
    for k, (u, v, p, q) in enumerate(members):
        th0 = synth.make_init_packed(p, q, 10, seed=50 + k)
        us.append(u); vs.append(v)
        inits.append([[_theta_list(t, p, q) for t in th0[5 * f:5 * f + 5]] for f in range(2)])
    res = R.call("ldsrhip_LDS_EM_groups", np.ascontiguousarray(Y.T), us, vs, inits, 50, 1e-5, True)
    assert len(res) == 3 and all(len(r) == 2 for r in res)
    for k, (u, v, p, q) in enumerate(members):
        alone = R.call("ldsrhip_LDS_EM_grid_raw", np.ascontiguousarray(Y.T), u, v, inits[k], 50, 1e-5)
        for f in range(2):
            a, b = res[k][f], alone[f]
            assert a["index"][0] == b["index"][0] and a["lik"][0] == b["lik"][0]
            for name in ("A", "B", "C", "D", "Q", "R", "mu1", "V1"):
                assert np.array_equal(a["theta"][name], b["theta"][name])
            assert a["theta"]["B"].shape == (1, p) and a["theta"]["D"].shape == (1, q)
            assert np.array_equal(a["fit"]["Y"], b["fit"]["Y"]) and np.array_equal(a["raw"]["Y"], b["raw"]["Y"])
    assert R.L.rmock_protect_depth() == 0
