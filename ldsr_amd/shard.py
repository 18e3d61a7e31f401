"""Sharding of the (series / fold, restart) cell grid over the GPUs of one node.

Cells never communicate during EM (the reference runs them as independent foreach tasks handed
to whichever worker is idle, R/LDS_reconstruction.R:46), so the grid is cut with no data-path
collective: every rank owns one of `world` contiguous parts of EVERY series' restarts -- of series
s the part (r + s) mod world, so the remainders of restart counts that `world` does not divide
(the reference's defaults: 50 restarts, cvLDS 20) rotate over the ranks instead of piling up on the
same ones (50 restarts x 48 series at 8 ranks: 300 cells each; without the rotation 288..336).  Series
differ a lot in iterations to converge (BASELINE config 5: 34 k to 130 k E-steps per series), while
the restarts of one series are statistically alike -- so equal shares of every series are equal
shares of the work (max / mean E-steps per rank 1.004 / 1.007 / 1.012 at 2 / 4 / 8 ranks on config 5,
against 1.18 / 1.22 / 1.32 for contiguous ranges of the flattened grid, round 2's cut).  Every rank
holds every series (<= 100 KB each).  The only cross-rank step is the gather of 8*(P+3) bytes per
cell before the per-series argmax (R/LDS_reconstruction.R:50-58); it uses whatever
torch.distributed backend the caller initialised (RCCL on GPUs, gloo in the CPU tests).  The
library cuts the same way for callers without torch.distributed (make_slices, ldsr_api.hip)."""
import numpy as np


def rank_slice(n_cells, world, rank, series=0):
    """Contiguous range [lo, hi): the rank's part of ONE series' n_cells restarts -- part
    (rank + series) mod world of the `world` contiguous parts (the same rule as make_slices in
    ldsr_api.hip)."""
    k = (rank + series) % world
    lo = n_cells * k // world
    hi = n_cells * (k + 1) // world
    return lo, hi


def rank_stripes(cell_offsets, world, rank):
    """The rank's cells of every series: (g_lo [S], local offsets [S+1]) -- of series s the rank
    owns the global cells [g_lo[s], g_lo[s] + loc[s+1] - loc[s]), stored locally from loc[s]."""
    off = np.asarray(cell_offsets, dtype=np.int64)
    S = off.size - 1
    g_lo = np.empty(S, dtype=np.int64)
    loc = np.zeros(S + 1, dtype=np.int32)
    for s in range(S):
        lo, hi = rank_slice(int(off[s + 1] - off[s]), world, rank, s)
        g_lo[s] = off[s] + lo
        loc[s + 1] = loc[s] + (hi - lo)
    return g_lo, loc


def stripe_index(cell_offsets, world, rank):
    """Global cell ids of the rank's cells, in local order."""
    g_lo, loc = rank_stripes(cell_offsets, world, rank)
    if loc[-1] == 0:
        return np.zeros(0, dtype=np.int64)
    return np.concatenate([np.arange(g_lo[s], g_lo[s] + loc[s + 1] - loc[s], dtype=np.int64)
                           for s in range(g_lo.size)])


def em_batch_sharded(y, u, v, theta0, cell_offsets=None, niter=1000, tol=1e-5, compute=None,
                     group=None, device=None, **kw):
    """Same contract as ldsr_amd.em_batch, computed cooperatively by every rank of the default
    (or given) process group; every rank returns the full result arrays.

    compute(y, u, v, theta0, cell_offsets=..., niter=..., tol=...) -> dict runs one rank's
    slice; the default is ldsr_amd.em_batch on this rank's GPU (LOCAL_RANK)."""
    import os

    import torch
    import torch.distributed as dist

    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    n, P = theta0.shape
    Y = np.asarray(y, dtype=np.float64)
    if cell_offsets is None:
        cell_offsets = [0, n]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    idx = stripe_index(cell_offsets, world, rank)          # this rank's cells, local order
    _, loc = rank_stripes(cell_offsets, world, rank)
    n_loc = idx.size

    if compute is None:
        from . import api
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device

        def compute(y_, u_, v_, th_, **k):
            return api.em_batch(y_, u_, v_, th_, device=dev, **k)

    res = {"theta": np.empty((n_loc, P)), "lik": np.empty(n_loc),
           "n_iter": np.empty(n_loc, np.int32), "status": np.empty(n_loc, np.int32)}
    if n_loc:
        r = compute(Y, u, v, theta0[idx], cell_offsets=loc, niter=niter, tol=tol, **kw)
        for k_ in res:
            res[k_][...] = r[k_]
    if world == 1:
        return res

    # gather: pack (theta | lik | n_iter | status) rows, pad to the largest share
    rows = np.concatenate([res["theta"], res["lik"][:, None],
                           res["n_iter"][:, None].astype(np.float64),
                           res["status"][:, None].astype(np.float64)], axis=1)
    sizes = [stripe_index(cell_offsets, world, r_).size for r_ in range(world)]
    max_rows = max(max(sizes), 1)
    backend = dist.get_backend(group)
    tdev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((max_rows, P + 3), dtype=torch.float64, device=tdev)
    buf[:rows.shape[0]] = torch.from_numpy(rows).to(tdev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = np.empty((n, P + 3))
    for r_ in range(world):
        out[stripe_index(cell_offsets, world, r_)] = parts[r_][:sizes[r_]].cpu().numpy()
    return {"theta": out[:, :P].copy(), "lik": out[:, P].copy(),
            "n_iter": out[:, P + 1].astype(np.int32), "status": out[:, P + 2].astype(np.int32)}


def select_per_series(lik, theta, cell_offsets, p, q, select=None):
    """Winner index (global cell id, or -1) for every series: the reference's selection rule
    applied to each series' restarts."""
    if select is None:
        from .api import select_restart as select
    off = np.asarray(cell_offsets)
    out = np.full(off.size - 1, -1, dtype=np.int64)
    for s in range(off.size - 1):
        a, b = int(off[s]), int(off[s + 1])
        if b > a:
            k = select(lik[a:b], theta[a:b], p, q)
            out[s] = a + k if k >= 0 else -1
    return out
