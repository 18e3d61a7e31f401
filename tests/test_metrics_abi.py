"""The five skill metrics of the C ABI (ldsr_metric_*: host code of libldsr_hip.so,
ldsr_amd/csrc/metrics.hip) -- what a DLL that replaces ldsr.so registers as _ldsr_NSE .. _ldsr_RE
(src/RcppExports.cpp:137-141).  Pinned on the reference-held NPcv object: its 150 per-fold numbers
(30 folds x R2, RE, CE, nRMSE, KGE) are a deterministic function of its stored Ycv, target and
folds through calculate_metrics (R/utils.R:56-70) and these five functions (src/utils.cpp:13-97).
No GPU needed."""
import ctypes as C

import numpy as np

from ldsr_amd import _lib, cv

_dp = C.POINTER(C.c_double)


def _p(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(_dp)


def _metrics_c(sim, obs, z):
    """calculate_metrics (R/utils.R:56-70) on the C ABI's metric entries."""
    L = _lib.lib()
    sim, obs = np.asarray(sim, float), np.asarray(obs, float)
    mask = np.ones(obs.size, bool)
    mask[z] = False
    tr_o, tr_s = obs[mask], sim[mask]
    ok = ~np.isnan(tr_o)
    tr_o, tr_s = np.ascontiguousarray(tr_o[ok]), np.ascontiguousarray(tr_s[ok])
    vs, vo = np.ascontiguousarray(sim[z]), np.ascontiguousarray(obs[z])
    return {"R2": L.ldsr_metric_nse(tr_o.size, _p(tr_s), _p(tr_o)),
            "RE": L.ldsr_metric_re(vo.size, _p(vs), _p(vo), float(tr_o.mean())),
            "CE": L.ldsr_metric_nse(vo.size, _p(vs), _p(vo)),
            "nRMSE": L.ldsr_metric_nrmse(vo.size, _p(vs), _p(vo), float(np.nanmean(obs))),
            "KGE": L.ldsr_metric_kge(vo.size, _p(vs), _p(vo))}


def test_metric_entries_reproduce_the_stored_npcv_numbers(refdata):
    c = refdata["NPcv"]
    Z = [np.asarray(z) - 1 for z in c["Z"]]
    Ycv, target = np.asarray(c["Ycv"]), np.asarray(c["target"])
    for f, z in enumerate(Z):
        m = _metrics_c(Ycv[f], target, z)
        for k in ("R2", "RE", "CE", "nRMSE", "KGE"):
            assert abs(m[k] - c["metrics_dist"][k][f]) <= 1e-10 * abs(c["metrics_dist"][k][f]) + 1e-12, (f, k)


def test_metric_entries_agree_with_the_python_mirror():
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for n in (2, 5, 12, 46, 1000):
        x, y = rng.normal(3.0, 2.0, n), rng.normal(3.0, 2.0, n)
        assert L.ldsr_metric_nse(n, _p(x), _p(y)) == np.float64(cv.NSE(x, y)) or abs(
            L.ldsr_metric_nse(n, _p(x), _p(y)) - cv.NSE(x, y)) < 1e-12
        assert abs(L.ldsr_metric_nrmse(n, _p(x), _p(y), 1.7) - cv.nRMSE(x, y, 1.7)) < 1e-12
        assert abs(L.ldsr_metric_corr(n, _p(x), _p(y)) - cv.corr(x, y)) < 1e-12
        assert abs(L.ldsr_metric_corr(n, _p(x), _p(y)) - np.corrcoef(x, y)[0, 1]) < 1e-12
        assert abs(L.ldsr_metric_kge(n, _p(x), _p(y)) - cv.KGE(x, y)) < 1e-12
        assert abs(L.ldsr_metric_re(n, _p(x), _p(y), 2.5) - cv.RE(x, y, 2.5)) < 1e-12
    # a perfect model: NSE = KGE = corr = 1, RE = 1, nRMSE = 0
    y = rng.normal(5.0, 1.0, 30)
    assert L.ldsr_metric_nse(30, _p(y), _p(y)) == 1.0 and L.ldsr_metric_re(30, _p(y), _p(y), 4.0) == 1.0
    assert L.ldsr_metric_nrmse(30, _p(y), _p(y), 1.0) == 0.0
    assert abs(L.ldsr_metric_kge(30, _p(y), _p(y)) - 1.0) < 1e-15 and abs(L.ldsr_metric_corr(30, _p(y), _p(y)) - 1.0) < 1e-15
