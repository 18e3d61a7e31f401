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


@pytest.mark.parametrize("mask", ["dense", "paleo"])
def test_bench_launch_matches_oracle(mask):
    """The EXACT launch bench.py times (BASELINE config 2 through ldsr_em_batch_device_lead: static
    schedule, niter=100, tol=0, operands resident in HBM, the lead / fully-observed hint the bench
    passes) against the CPU oracle on all 4096 cells -- round 2 checked this instantiation on 21..48
    cells only and the bench itself asserted nothing but iteration counts."""
    import ctypes as C

    import torch

    import bench
    from ldsr_amd import _lib
    from oracle import oracle as O

    w = bench.WORKLOADS["cfg2"]
    T, p, q, niter = w["T"], w["p"], w["q"], w["niter"]
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    prob = bench.build_problem("cfg2", mask, 1, 0)
    job = bench.Job(L, torch, dev, 0, prob, T, p, q, niter, 0.0, 0)
    job.step()
    torch.cuda.synchronize(dev)
    buf = C.create_string_buffer(160)
    assert L.ldsr_last_em_kernel(0, buf, 160) == 0
    kernel = buf.value.decode()
    assert kernel.startswith("em_pair_kernel<1, 2,") and ", false, " in kernel      # static schedule
    assert ("true>" in kernel) == (mask == "paleo")                                 # closed-form lead for the paleo mask
    assert job.units(niter, 0.0) == 4096 * niter

    Y, U, V, shared, off, th0, n = prob
    ref_th, ref_lik, ref_it, ref_st = O.em_batch(Y, U, V, np.zeros(n, np.int32), th0, niter, 0.0, n_threads=16)
    assert np.array_equal(job.d_nit.cpu().numpy(), ref_it)
    assert np.array_equal(job.d_st.cpu().numpy(), ref_st)
    assert parity_close(job.d_lik.cpu().numpy(), ref_lik, RTOL, ATOL)
    assert parity_close(job.d_th.cpu().numpy(), ref_th, RTOL, ATOL)
    # ... and the bench's own 64-cell check of that launch agrees
    v = bench.verify_sample(job, p, q, niter, 0.0)
    assert v["ok"] and v["cells"] == 64 and v["n_iter_equal"]
