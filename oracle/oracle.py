"""ctypes binding of the CPU oracle (oracle/libldsr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by the product package ldsr_amd.
Packed theta layout: [A, B(p), C, D(q), Q, R, mu1, V1]; u/v column-major (time-major
p-vectors, exactly the bytes of an R p x T matrix).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "libldsr_oracle.so")
    src = os.path.join(_HERE, "ldsr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libldsr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libldsr_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _prep(y, u, v):
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
    T = y.size
    has_u = u is not None
    has_v = v is not None
    if has_u:
        u = np.asarray(u, dtype=np.float64)
        assert u.ndim == 2 and u.shape[1] == T, "u must be p x T"
        p = u.shape[0]
        uf = np.ascontiguousarray(u.T).reshape(-1)      # time-major == R column-major
    else:
        p, uf = 1, np.zeros(T, dtype=np.float64)
    if has_v:
        v = np.asarray(v, dtype=np.float64)
        assert v.ndim == 2 and v.shape[1] == T, "v must be q x T"
        q = v.shape[0]
        vf = np.ascontiguousarray(v.T).reshape(-1)
    else:
        q, vf = 1, np.zeros(T, dtype=np.float64)
    return y, uf, vf, T, p, q, int(has_u), int(has_v)


def kalman_smoother(y, u, v, theta, stdlik=True):
    y, uf, vf, T, p, q, hu, hv = _prep(y, u, v)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    assert th.size == 6 + p + q
    X, Y, V, J = (np.empty(T) for _ in range(4))
    lik = C.c_double()
    rc = lib().oracle_kalman_smoother(T, p, q, _d(y), _d(uf), _d(vf), hu, hv, _d(th),
                                      int(stdlik), _d(X), _d(Y), _d(V), _d(J), C.byref(lik))
    if rc:
        raise RuntimeError("oracle_kalman_smoother rc=%d" % rc)
    return {"X": X, "Y": Y, "V": V, "J": J, "lik": lik.value}


def mstep(y, u, v, fit):
    y, uf, vf, T, p, q, hu, hv = _prep(y, u, v)
    X = np.ascontiguousarray(fit["X"], dtype=np.float64)
    V = np.ascontiguousarray(fit["V"], dtype=np.float64)
    J = np.ascontiguousarray(fit["J"], dtype=np.float64)
    th = np.empty(6 + p + q)
    rc = lib().oracle_mstep(T, p, q, _d(y), _d(uf), _d(vf), hu, hv, _d(X), _d(V), _d(J), _d(th))
    if rc:
        raise RuntimeError("oracle_mstep rc=%d" % rc)
    return th


def lds_em(y, u, v, theta0, niter=1000, tol=1e-5):
    y, uf, vf, T, p, q, hu, hv = _prep(y, u, v)
    th0 = np.ascontiguousarray(theta0, dtype=np.float64)
    assert th0.size == 6 + p + q
    th = np.empty(6 + p + q)
    X, Y, V, J = (np.empty(T) for _ in range(4))
    liks = np.empty(max(niter, 2))
    n_iter = C.c_int()
    lik = C.c_double()
    rc = lib().oracle_lds_em(T, p, q, _d(y), _d(uf), _d(vf), hu, hv, _d(th0), int(niter),
                             C.c_double(tol), _d(th), _d(X), _d(Y), _d(V), _d(J), _d(liks),
                             C.byref(n_iter), C.byref(lik))
    if rc:
        raise RuntimeError("oracle_lds_em rc=%d" % rc)
    return {"theta": th, "fit": {"X": X, "Y": Y, "V": V, "J": J, "lik": lik.value},
            "liks": liks[:n_iter.value].copy(), "lik": lik.value}


def propagate(theta, u, v, y, stdlik=True):
    y, uf, vf, T, p, q, hu, hv = _prep(y, u, v)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    X, Y, V = (np.empty(T) for _ in range(3))
    lik = C.c_double()
    rc = lib().oracle_propagate(T, p, q, _d(th), _d(uf), _d(vf), hu, hv, _d(y), int(stdlik),
                                _d(X), _d(Y), _d(V), C.byref(lik))
    if rc:
        raise RuntimeError("oracle_propagate rc=%d" % rc)
    return {"X": X, "Y": Y, "V": V, "lik": lik.value}


def em_batch(y_all, u_all, v_all, series_of_cell, theta0, niter, tol, n_threads=1):
    """y_all [S,T]; u_all [S,T,p] or None; v_all [S,T,q] or None (time-major);
    theta0 [n_cells, P].  Returns theta, lik, n_iter, status."""
    y_all = np.ascontiguousarray(y_all, dtype=np.float64)
    S, T = y_all.shape
    hu, hv = u_all is not None, v_all is not None
    u_all = np.ascontiguousarray(u_all, dtype=np.float64) if hu else np.zeros((S, T, 1))
    v_all = np.ascontiguousarray(v_all, dtype=np.float64) if hv else np.zeros((S, T, 1))
    p, q = u_all.shape[2], v_all.shape[2]
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    n_cells, P = theta0.shape
    assert P == 6 + p + q
    soc = np.ascontiguousarray(series_of_cell, dtype=np.int32)
    assert soc.size == n_cells
    theta = np.empty_like(theta0)
    lik = np.empty(n_cells)
    n_iter = np.empty(n_cells, dtype=np.int32)
    status = np.empty(n_cells, dtype=np.int32)
    rc = lib().oracle_em_batch(S, T, p, q, _d(y_all), _d(u_all), _d(v_all), int(hu), int(hv),
                               n_cells, soc.ctypes.data_as(_ip), _d(theta0), int(niter),
                               C.c_double(tol), int(n_threads), _d(theta), _d(lik),
                               n_iter.ctypes.data_as(_ip), status.ctypes.data_as(_ip))
    if rc:
        raise RuntimeError("oracle_em_batch rc=%d" % rc)
    return theta, lik, n_iter, status


def select(lik, Cs):
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    Cs = np.ascontiguousarray(Cs, dtype=np.float64)
    return int(lib().oracle_select(lik.size, _d(lik), _d(Cs)))


def pack_theta(A, B, Cc, D, Q, R, mu1, V1):
    return np.concatenate([[A], np.atleast_1d(B), [Cc], np.atleast_1d(D), [Q, R, mu1, V1]]).astype(
        np.float64)


def unpack_theta(th, p, q):
    th = np.asarray(th)
    return {"A": th[0], "B": th[1:1 + p], "C": th[1 + p], "D": th[2 + p:2 + p + q],
            "Q": th[2 + p + q], "R": th[3 + p + q], "mu1": th[4 + p + q], "V1": th[5 + p + q]}
