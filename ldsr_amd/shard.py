"""Sharding of the (series / fold, restart) cell grid over the GPUs of one node.

Cells never communicate during EM (the reference runs them as independent foreach tasks,
R/LDS_reconstruction.R:46), so the grid is cut into contiguous ranges, one per rank, and the
only cross-rank step is the gather of 8*(P+3) bytes per cell before the per-series argmax
(R/LDS_reconstruction.R:50-58).  No data-path collective; the gather uses whatever
torch.distributed backend the caller initialised (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def rank_slice(n_cells, world, rank):
    """Contiguous range [lo, hi) of the flattened cell grid owned by `rank`."""
    lo = n_cells * rank // world
    hi = n_cells * (rank + 1) // world
    return lo, hi


def local_offsets(cell_offsets, lo, hi):
    """Clip the per-series cell ranges to [lo, hi): returns (series ids present, local
    cell_offsets starting at 0)."""
    off = np.asarray(cell_offsets, dtype=np.int64)
    S = off.size - 1
    starts = np.clip(off[:-1], lo, hi)
    ends = np.clip(off[1:], lo, hi)
    keep = [s for s in range(S) if ends[s] > starts[s]]
    loc = np.zeros(len(keep) + 1, dtype=np.int32)
    for i, s in enumerate(keep):
        loc[i + 1] = loc[i] + (ends[s] - starts[s])
    return np.asarray(keep, dtype=np.int64), loc


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
    multi = Y.ndim == 2 and Y.shape[0] > 1
    if cell_offsets is None:
        cell_offsets = [0, n]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = rank_slice(n, world, rank)
    keep, loc = local_offsets(cell_offsets, lo, hi)

    if compute is None:
        from . import api
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device

        def compute(y_, u_, v_, th_, **k):
            return api.em_batch(y_, u_, v_, th_, device=dev, **k)

    def take(a, name):
        if a is None:
            return None
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 3:
            return a[keep]
        return a

    res = {"theta": np.empty((hi - lo, P)), "lik": np.empty(hi - lo),
           "n_iter": np.empty(hi - lo, np.int32), "status": np.empty(hi - lo, np.int32)}
    if hi > lo:
        y_loc = Y[keep] if multi else Y
        r = compute(y_loc, take(u, "u"), take(v, "v"), theta0[lo:hi], cell_offsets=loc,
                    niter=niter, tol=tol, **kw)
        for k_ in res:
            res[k_][...] = r[k_]
    if world == 1:
        return res

    # gather: pack (theta | lik | n_iter | status) rows, pad to the largest slice
    rows = np.concatenate([res["theta"], res["lik"][:, None],
                           res["n_iter"][:, None].astype(np.float64),
                           res["status"][:, None].astype(np.float64)], axis=1)
    max_rows = max(rank_slice(n, world, r_)[1] - rank_slice(n, world, r_)[0] for r_ in range(world))
    backend = dist.get_backend(group)
    tdev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((max_rows, P + 3), dtype=torch.float64, device=tdev)
    buf[:rows.shape[0]] = torch.from_numpy(rows).to(tdev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = np.empty((n, P + 3))
    for r_ in range(world):
        a, b = rank_slice(n, world, r_)
        out[a:b] = parts[r_][:b - a].cpu().numpy()
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
