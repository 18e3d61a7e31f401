"""BASELINE.json configs 3, 4 and 5 run WHOLE on the GPU against the oracle (-m gpu).

The grids are exactly bench.py's (`build_problem`): cfg3 = 8192 restarts of one T=1000, p=4,
q=8 series; cfg4 = the cvLDS grid, 10 fold masks x 1024 restarts on shared u, v, T=2000, p=1,
q=4 (R/LDS_reconstruction.R:270-285,373-375); cfg5 = 48 own-input series x 512 restarts,
T=813, p=1, q=3, observed tails of 30..90 steps.  Each runs with the reference's defaults
niter=1000, tol=1e-5 (R/LDS_reconstruction.R:122-125) and must stop every cell at exactly the
oracle's iteration, then meet the Appendix-B bar on theta and lik, then pick the same winner
for every series / fold (R/LDS_reconstruction.R:50-58)."""
import numpy as np
import pytest

from conftest import parity_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-6, 1e-9


def _grid(name):
    import bench
    Y, U, V, shared, off, th0, n = bench.build_problem(name, "dense", 1, 0)
    assert n == th0.shape[0] == off[-1]
    return Y, U, V, shared, np.asarray(off, np.int32), th0


def _api_inputs(U, V, shared):
    """bench.build_problem hands time-major [S or 1, T, k]; the operator interface takes R's
    k x T matrices (shared) or [S, k, T]."""
    if shared:
        return np.ascontiguousarray(U[0].T), np.ascontiguousarray(V[0].T)
    return (np.ascontiguousarray(np.transpose(U, (0, 2, 1))),
            np.ascontiguousarray(np.transpose(V, (0, 2, 1))))


@pytest.mark.parametrize("name,cells,series", [("cfg3", 8192, 1), ("cfg4", 10240, 10),
                                               ("cfg5", 24576, 48)])
def test_baseline_config_whole_grid_converged(name, cells, series):
    import ldsr_amd
    from ldsr_amd import shard
    from oracle import oracle as O

    Y, U, V, shared, off, th0 = _grid(name)
    S = Y.shape[0]
    assert (S, th0.shape[0]) == (series, cells)
    u, v = _api_inputs(U, V, shared)
    y = Y if S > 1 else Y[0]
    if S == 1:
        u, v = (u[0], v[0]) if u.ndim == 3 else (u, v)
    r = ldsr_amd.em_batch(y, u, v, th0, cell_offsets=off, niter=1000, tol=1e-5)

    Uo = np.repeat(U, S, axis=0) if U.shape[0] != S else U
    Vo = np.repeat(V, S, axis=0) if V.shape[0] != S else V
    soc = np.repeat(np.arange(S), np.diff(off)).astype(np.int32)
    ref_th, ref_lik, ref_it, ref_st = O.em_batch(Y, np.ascontiguousarray(Uo),
                                                 np.ascontiguousarray(Vo), soc, th0, 1000, 1e-5,
                                                 n_threads=16)
    bad = np.nonzero(r["n_iter"] != ref_it)[0]
    assert bad.size == 0, "%s: %d cells stop at another iteration than the oracle, first %s" % (
        name, bad.size, bad[:8])
    assert np.array_equal(r["status"], ref_st)
    assert ref_it.max() > 100, "the grid should contain long-running cells"
    assert parity_close(r["lik"], ref_lik, RTOL, ATOL), name
    assert parity_close(r["theta"], ref_th, RTOL, ATOL), name
    # per-series / per-fold winner (R/LDS_reconstruction.R:50-58)
    p, q = U.shape[2], V.shape[2]
    win = shard.select_per_series(r["lik"], r["theta"], off, p, q)
    ref_win = np.array([off[s] + O.select(ref_lik[off[s]:off[s + 1]],
                                          ref_th[off[s]:off[s + 1], 1 + p]) for s in range(S)])
    assert np.array_equal(win, ref_win), name
