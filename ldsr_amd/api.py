"""Host-side mirror of the reference's operator interface for the EM/Kalman path.

Same names, argument meaning and return shape as the R / Rcpp surface it replaces
(paths under /root/reference):

    make_init        R/LDS_reconstruction.R:14-30
    LDS_EM_restart   R/LDS_reconstruction.R:42-62   (the foreach fan-out + selection)
    LDS_EM           R/RcppExports.R:41-43  -> src/EM.cpp:245-280
    Kalman_smoother  R/RcppExports.R:15-17  -> src/EM.cpp:22-131
    Mstep            R/RcppExports.R:24-26  -> src/EM.cpp:139-229
    propagate        R/RcppExports.R:57-59  -> src/EM.cpp:295-356

Everything numeric runs on the GPU through the C ABI of include/ldsr_hip.h (ctypes); this
module only marshals.  A `theta` is a dict with the reference's list names
(A, B, C, D, Q, R, mu1, V1; B is 1 x p, D is 1 x q); packed arrays [A, B.., C, D.., Q, R,
mu1, V1] are accepted wherever a theta is.  u / v are p x T and q x T arrays as in R, or
None for an absent input (the reference's `matrix(0)` sentinel).  y is length T with NaN
for missing values.
"""
import ctypes as C

import numpy as np

from . import _lib
from .synth import make_init_packed

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

ALGO_AUTO, ALGO_SERIAL, ALGO_SCAN, ALGO_PAIR, ALGO_QUAD = 0, 1, 2, 3, 4


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def pack_theta(theta, p, q):
    if isinstance(theta, dict):
        out = np.concatenate([
            np.ravel(theta["A"]), np.ravel(theta["B"]), np.ravel(theta["C"]), np.ravel(theta["D"]),
            np.ravel(theta["Q"]), np.ravel(theta["R"]), np.ravel(theta["mu1"]),
            np.ravel(theta["V1"])]).astype(np.float64)
    else:
        out = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1)
    if out.size != 6 + p + q:
        raise ValueError("theta has %d entries, expected 6+p+q = %d" % (out.size, 6 + p + q))
    return out


def unpack_theta(th, p, q):
    th = np.asarray(th, dtype=np.float64)
    return {"A": th[0:1].reshape(1, 1).copy(), "B": th[1:1 + p].reshape(1, p).copy(),
            "C": th[1 + p:2 + p].reshape(1, 1).copy(),
            "D": th[2 + p:2 + p + q].reshape(1, q).copy(),
            "Q": th[2 + p + q:3 + p + q].reshape(1, 1).copy(),
            "R": th[3 + p + q:4 + p + q].reshape(1, 1).copy(),
            "mu1": th[4 + p + q:5 + p + q].reshape(1, 1).copy(),
            "V1": th[5 + p + q:6 + p + q].reshape(1, 1).copy()}


def make_init(p, q, num_restarts, seed=None, r_seed=None):
    """List of `num_restarts` random initial thetas with the reference's distribution.
    r_seed=k reproduces R's `set.seed(k); make_init(p, q, num_restarts)` draw for draw
    (ldsr_amd/rrng.py); otherwise the counter-based generator of synth.py is used."""
    if r_seed is not None:
        from .rrng import make_init_packed_r
        packed = make_init_packed_r(p, q, num_restarts, r_seed)
        return [unpack_theta(t, p, q) for t in packed]
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0])
    packed = make_init_packed(p, q, num_restarts, seed=seed)
    return [unpack_theta(t, p, q) for t in packed]


def _series(y, u, v):
    """Marshal one or several series.  y: [T] or [S,T]; u: [p,T] or [S,p,T] or None."""
    y = np.asarray(y, dtype=np.float64)
    if y.ndim == 2 and y.shape[0] == 1:
        y = y[0]
    multi = y.ndim == 2
    Y = np.ascontiguousarray(y if multi else y[None, :])
    S, T = Y.shape

    def prep(a, name):
        if a is None:
            return None, 1, True
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 2:
            if a.shape[1] != T:
                raise ValueError("%s must have T = %d columns" % (name, T))
            return np.ascontiguousarray(a.T)[None], a.shape[0], True     # [1,T,k] shared
        if a.ndim == 3:
            if a.shape[0] != S or a.shape[2] != T:
                raise ValueError("%s must be [S, k, T]" % name)
            return np.ascontiguousarray(np.transpose(a, (0, 2, 1))), a.shape[1], False
        raise ValueError("%s must be 2-D (k x T) or 3-D (S x k x T)" % name)

    U, p, us = prep(u, "u")
    V, q, vs = prep(v, "v")
    shared = 1 if (us and vs) else 0
    if not shared:      # mixed: replicate the shared one
        if U is not None and us:
            U = np.ascontiguousarray(np.repeat(U, S, axis=0))
        if V is not None and vs:
            V = np.ascontiguousarray(np.repeat(V, S, axis=0))
    if S == 1:
        shared = 0
    return Y, U, V, S, T, p, q, shared


def em_batch(y, u, v, theta0, cell_offsets=None, niter=1000, tol=1e-5, device=0, algo=ALGO_AUTO,
             return_liks=False, devices=None):
    """All cells in one launch.  theta0: packed [n_cells, 6+p+q].  cell_offsets: [S+1]
    (default: every cell belongs to series 0).  Returns dict of arrays theta, lik, n_iter,
    status (and liks [n_cells, niter], NaN padded).  devices=[0,1,...] shards the cells over
    several GPUs inside the library (host threads, no collective)."""
    Y, U, V, S, T, p, q, shared = _series(y, u, v)
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    if theta0.ndim != 2 or theta0.shape[1] != 6 + p + q:
        raise ValueError("theta0 must be [n_cells, %d]" % (6 + p + q))
    n = theta0.shape[0]
    if cell_offsets is None:
        if S != 1:
            raise ValueError("cell_offsets is required with several series")
        cell_offsets = [0, n]
    off = np.ascontiguousarray(cell_offsets, dtype=np.int32)
    if off.size != S + 1 or off[-1] != n:
        raise ValueError("cell_offsets must have S+1 entries ending at n_cells")
    theta = np.empty_like(theta0)
    lik = np.empty(n)
    n_iter = np.empty(n, dtype=np.int32)
    status = np.empty(n, dtype=np.int32)
    liks = np.empty((n, niter)) if return_liks else None
    L = _lib.lib()
    if devices is not None:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        _lib.check(L.ldsr_em_batch_multi(devs.size, _i(devs), S, T, p, q, _d(Y), _d(U), _d(V), shared,
                                         _i(off), _d(theta0), int(niter), float(tol), int(algo),
                                         _d(theta), _d(lik), _i(n_iter), _i(status), _d(liks)))
    else:
        _lib.check(L.ldsr_em_batch(device, S, T, p, q, _d(Y), _d(U), _d(V), shared, _i(off),
                                   _d(theta0), int(niter), float(tol), int(algo), _d(theta), _d(lik),
                                   _i(n_iter), _i(status), _d(liks)))
    out = {"theta": theta, "lik": lik, "n_iter": n_iter, "status": status}
    if return_liks:
        out["liks"] = liks
    return out


def smooth_batch(y, u, v, theta, cell_offsets=None, stdlik=True, device=0, mode="smooth"):
    Y, U, V, S, T, p, q, shared = _series(y, u, v)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    n = theta.shape[0]
    if cell_offsets is None:
        cell_offsets = [0, n]
    off = np.ascontiguousarray(cell_offsets, dtype=np.int32)
    X, Yh, Vv, J = (np.empty((n, T)) for _ in range(4))
    lik = np.empty(n)
    L = _lib.lib()
    if mode == "smooth":
        _lib.check(L.ldsr_smooth_batch(device, S, T, p, q, _d(Y), _d(U), _d(V), shared, _i(off),
                                       _d(theta), int(bool(stdlik)), _d(X), _d(Yh), _d(Vv), _d(J),
                                       _d(lik)))
        return {"X": X, "Y": Yh, "V": Vv, "J": J, "lik": lik}
    _lib.check(L.ldsr_propagate_batch(device, S, T, p, q, _d(Y), _d(U), _d(V), shared, _i(off),
                                      _d(theta), int(bool(stdlik)), _d(X), _d(Yh), _d(Vv), _d(lik)))
    return {"X": X, "Y": Yh, "V": Vv, "lik": lik}


def penalized_likelihood(y, u, v, theta_packed, lam, cell_offsets=None, device=0):
    """R/LDS_GA.R:28-44 for a batch of thetas [n, 6+p+q]: lik(stdlik=FALSE) - lam * ssq."""
    Y, U, V, S, T, p, q, shared = _series(y, u, v)
    theta = np.ascontiguousarray(np.atleast_2d(theta_packed), dtype=np.float64)
    n = theta.shape[0]
    off = np.ascontiguousarray([0, n] if cell_offsets is None else cell_offsets, dtype=np.int32)
    pl = np.empty(n)
    _lib.check(_lib.lib().ldsr_penalized_lik_batch(device, S, T, p, q, _d(Y), _d(U), _d(V), shared,
                                                   _i(off), _d(theta), float(lam), _d(pl)))
    return pl


def _dims(u, v):
    p = 1 if u is None else np.asarray(u).shape[-2]
    q = 1 if v is None else np.asarray(v).shape[-2]
    return p, q


def Kalman_smoother(y, u, v, theta, stdlik=True, device=0):
    """-> {"X","Y","V","J": 1 x T arrays, "lik": float}   (src/EM.cpp:126-130)"""
    p, q = _dims(u, v)
    r = smooth_batch(y, u, v, pack_theta(theta, p, q)[None, :], stdlik=stdlik, device=device)
    return {"X": r["X"], "Y": r["Y"], "V": r["V"], "J": r["J"], "lik": float(r["lik"][0])}


def propagate(theta, u, v, y, stdlik=True, device=0):
    """-> {"X","Y","V": 1 x T arrays, "lik": float}   (src/EM.cpp:352-355)"""
    p, q = _dims(u, v)
    r = smooth_batch(y, u, v, pack_theta(theta, p, q)[None, :], stdlik=stdlik, device=device,
                     mode="propagate")
    return {"X": r["X"], "Y": r["Y"], "V": r["V"], "lik": float(r["lik"][0])}


def Mstep(y, u, v, fit, device=0):
    """fit: result of Kalman_smoother -> theta dict   (src/EM.cpp:221-228)"""
    Y, U, V, S, T, p, q, shared = _series(y, u, v)
    X = np.ascontiguousarray(np.asarray(fit["X"], dtype=np.float64).reshape(1, T))
    Vv = np.ascontiguousarray(np.asarray(fit["V"], dtype=np.float64).reshape(1, T))
    J = np.ascontiguousarray(np.asarray(fit["J"], dtype=np.float64).reshape(1, T))
    th = np.empty((1, 6 + p + q))
    st = np.empty(1, dtype=np.int32)
    off = np.array([0, 1], dtype=np.int32)
    L = _lib.lib()
    _lib.check(L.ldsr_mstep_batch(device, 1, T, p, q, _d(Y), _d(U), _d(V), 0, _i(off), _d(X),
                                  _d(Vv), _d(J), _d(th), _i(st)))
    if st[0] == 2:
        raise _lib.LdsrError("Mstep: matrix is singular")   # arma::inv throws here
    return unpack_theta(th[0], p, q)


def _grid_outputs(S, T, P, n, niter, want_all):
    out = {"winner": np.empty(S, dtype=np.int32), "theta": np.empty((S, P)), "lik": np.empty(S),
           "n_iter": np.empty(S, dtype=np.int32), "liks": np.empty((S, niter)),
           "X": np.empty((S, T)), "Y": np.empty((S, T)), "V": np.empty((S, T)),
           "J": np.empty((S, T))}
    allr = None
    if want_all:
        allr = {"theta": np.empty((n, P)), "lik": np.empty(n), "n_iter": np.empty(n, dtype=np.int32),
                "status": np.empty(n, dtype=np.int32)}
    return out, allr


def em_restart_grid(y, u, v, theta0, cell_offsets=None, niter=1000, tol=1e-5, devices=(0,),
                    algo=ALGO_AUTO, return_all=True):
    """LDS_EM_restart for a whole grid of series / CV folds in ONE library call
    (ldsr_em_restart_grid): every (series, restart) cell runs on the GPU(s), each series'
    winner is picked by the reference's rule and only the winners' models cross PCIe.

    Returns dict: winner [S] (global cell index, -1 = none), theta [S, P], lik [S], n_iter [S],
    liks [S, niter] (NaN padded), X / Y / V / J [S, T]; plus "all" (per-cell theta, lik,
    n_iter, status) when return_all."""
    Y, U, V, S, T, p, q, shared = _series(y, u, v)
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    P = 6 + p + q
    if theta0.ndim != 2 or theta0.shape[1] != P:
        raise ValueError("theta0 must be [n_cells, %d]" % P)
    n = theta0.shape[0]
    if cell_offsets is None:
        if S != 1:
            raise ValueError("cell_offsets is required with several series")
        cell_offsets = [0, n]
    off = np.ascontiguousarray(cell_offsets, dtype=np.int32)
    if off.size != S + 1 or off[-1] != n:
        raise ValueError("cell_offsets must have S+1 entries ending at n_cells")
    devs = np.ascontiguousarray(devices, dtype=np.int32)
    out, allr = _grid_outputs(S, T, P, n, int(niter), return_all)
    a = allr or {}
    _lib.check(_lib.lib().ldsr_em_restart_grid(
        devs.size, _i(devs), S, T, p, q, _d(Y), _d(U), _d(V), shared, _i(off), _d(theta0),
        int(niter), float(tol), int(algo), _d(a.get("theta")), _d(a.get("lik")),
        _i(a.get("n_iter")), _i(a.get("status")), _i(out["winner"]), _d(out["theta"]),
        _d(out["lik"]), _i(out["n_iter"]), _d(out["liks"]), _d(out["X"]), _d(out["Y"]),
        _d(out["V"]), _d(out["J"])))
    if allr is not None:
        out["all"] = allr
    return out


def ensemble_restart(y, members, inits, niter=1000, tol=1e-5, devices=(0,), algo=ALGO_AUTO,
                     cell_offsets=None):
    """The ensemble loop of LDS_reconstruction (R/LDS_reconstruction.R:242-246): `members` is a
    list of (u, v) pairs that may differ in p and q (tests/testthat/test-ensemble.R:4-5),
    `inits` the matching list of packed theta0 arrays [n_restarts, 6+p+q].  All members run
    concurrently inside ONE library call (ldsr_em_restart_groups); returns one
    em_restart_grid-style dict per member.  y may hold several series / CV folds [S, T]
    (the nested fold x member loop of cvLDS, R/LDS_reconstruction.R:377-381): every member then
    runs the whole grid, with `cell_offsets` [S+1] (shared by the members; default: equal
    split of each member's restarts over the series)."""
    import ctypes as C
    if len(members) != len(inits):
        raise ValueError("members and inits must have the same length")
    G = len(members)
    groups = (_lib.Group * max(G, 1))()
    keep, outs = [], []
    for g, ((u, v), th0) in enumerate(zip(members, inits)):
        Y, U, V, S, T, p, q, shared = _series(y, u, v)
        th0 = np.ascontiguousarray(th0, dtype=np.float64)
        P = 6 + p + q
        if th0.ndim != 2 or th0.shape[1] != P:
            raise ValueError("inits[%d] must be [n, %d]" % (g, P))
        n = th0.shape[0]
        if cell_offsets is not None:
            off = np.ascontiguousarray(cell_offsets, dtype=np.int32)
        elif n % S == 0:
            off = (np.arange(S + 1) * (n // S)).astype(np.int32)
        else:
            raise ValueError("inits[%d]: %d restarts do not split evenly over %d series" % (g, n, S))
        if off.size != S + 1 or off[-1] != n:
            raise ValueError("cell_offsets must have S+1 entries ending at every member's restart count")
        out, allr = _grid_outputs(S, T, P, n, int(niter), True)
        keep.append((Y, U, V, th0, off))
        outs.append((out, allr))
        gr = groups[g]
        gr.n_series, gr.T, gr.p, gr.q, gr.shared_uv = S, T, p, q, shared
        gr.y, gr.u, gr.v, gr.cell_offsets, gr.theta0 = _d(Y), _d(U), _d(V), _i(off), _d(th0)
        gr.theta_all, gr.lik_all = _d(allr["theta"]), _d(allr["lik"])
        gr.n_iter_all, gr.status_all = _i(allr["n_iter"]), _i(allr["status"])
        gr.winner, gr.theta_w, gr.lik_w, gr.n_iter_w = (_i(out["winner"]), _d(out["theta"]),
                                                        _d(out["lik"]), _i(out["n_iter"]))
        gr.liks_w, gr.X, gr.Y, gr.V, gr.J = (_d(out["liks"]), _d(out["X"]), _d(out["Y"]),
                                             _d(out["V"]), _d(out["J"]))
    devs = np.ascontiguousarray(devices, dtype=np.int32)
    _lib.check(_lib.lib().ldsr_em_restart_groups(devs.size, _i(devs), G, C.byref(groups), int(niter),
                                                 float(tol), int(algo)))
    res = []
    for out, allr in outs:
        out["all"] = allr
        res.append(out)
    return res


def _model_from_grid(r, s, p, q):
    """Row s of an em_restart_grid result in LDS_EM's return shape (src/EM.cpp:276-279)."""
    n_it = int(r["n_iter"][s])
    return {"theta": unpack_theta(r["theta"][s], p, q),
            "fit": {"X": r["X"][s:s + 1].copy(), "Y": r["Y"][s:s + 1].copy(),
                    "V": r["V"][s:s + 1].copy(), "J": r["J"][s:s + 1].copy(),
                    "lik": float(r["lik"][s])},
            "liks": r["liks"][s, :n_it].copy(), "lik": float(r["lik"][s])}


def LDS_EM(y, u, v, theta0, niter=1000, tol=1e-5, device=0, algo=ALGO_AUTO):
    """-> {"theta", "fit", "liks", "lik"}   (src/EM.cpp:276-279)"""
    p, q = _dims(u, v)
    r = em_restart_grid(y, u, v, pack_theta(theta0, p, q)[None, :], niter=niter, tol=tol,
                        devices=(device,), algo=algo)
    if r["all"]["status"][0] == 2:
        raise _lib.LdsrError("LDS_EM: matrix is singular")
    if r["winner"][0] < 0:      # a lone restart with a NaN likelihood: the reference returns it as is
        b = em_batch(y, u, v, pack_theta(theta0, p, q)[None, :], niter=niter, tol=tol, device=device,
                     algo=algo, return_liks=True)
        th = b["theta"][0]
        return {"theta": unpack_theta(th, p, q), "fit": Kalman_smoother(y, u, v, th, device=device),
                "liks": b["liks"][0, :int(b["n_iter"][0])].copy(), "lik": float(b["lik"][0])}
    return _model_from_grid(r, 0, p, q)


def select_restart(lik, theta_packed, p, q):
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    th = np.ascontiguousarray(theta_packed, dtype=np.float64)
    return int(_lib.lib().ldsr_select_restart(lik.size, _d(lik), _d(th), p, q))


def LDS_EM_restart(y, u, v, init, niter=1000, tol=1e-5, return_init=True, device=0,
                   algo=ALGO_AUTO, devices=None):
    """One LDS_EM per element of `init`, all in one GPU launch, then the reference's selection
    (highest likelihood among models with C > 0 if any).  Returns the winning model in LDS_EM's
    shape (+ "init"), plus "all" = per-restart lik / theta / n_iter / status arrays.  Only the
    winner's fit and likelihood trace cross PCIe (one ldsr_em_restart_grid call)."""
    p, q = _dims(u, v)
    theta0 = np.stack([pack_theta(t, p, q) for t in init])
    r = em_restart_grid(y, u, v, theta0, niter=niter, tol=tol,
                        devices=(device,) if devices is None else devices, algo=algo)
    if np.any(r["all"]["status"] == 2):
        raise _lib.LdsrError("LDS_EM_restart: matrix is singular")
    k = int(r["winner"][0])
    if k < 0:
        raise _lib.LdsrError("LDS_EM_restart: no restart produced a finite likelihood")
    ans = _model_from_grid(r, 0, p, q)
    if return_init:
        ans["init"] = init[k]
    ans["all"] = dict(r["all"], selected=k)
    return ans
