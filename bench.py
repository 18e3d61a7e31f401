#!/usr/bin/env python3
"""bench.py -- restart x EM-iteration throughput of the EM/Kalman hot path on MI355X.

One "step" = one pass of the hot path over one batch: a single ldsr_em_batch_device call
(series preparation + the EM kernel) on BASELINE.json's config 2 -- synthetic T=1000, p=1,
q=2, 4096 restarts per GPU, niter=100, tol=0 (the stop rule of src/EM.cpp:272 can never fire,
so every cell runs exactly 100 E-steps) = 409 600 restart x EM-iteration units per GPU.
Inputs (y, u, v, theta0) are resident in HBM before the timed region.  --workload selects the
other BASELINE configs (cfg3 per GPU; cfg4 / cfg5 fixed grids) for DESIGN.md's table.

N > 1 (launched by torch.distributed.run): restarts shard embarrassingly -- every rank runs
its own 4096 restarts of the same series (restart index = global position in the counter-based
generator), no data-path collective; "scaling": "weak" (cfg4 / cfg5: "strong").

Prints ONE JSON line on rank 0; see the task contract for the fields.  Extra objects:
  roofline      algorithmic bytes (16*T*(3+p+q) per unit) / EM-kernel time
                measured with HIP events on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of src/EM.cpp, not RcppArmadillo) on the host cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# BASELINE.json configs.  cfg2 (the metric's config) and cfg3 are quoted per GPU -> weak scaling
# (every rank gets `restarts` restarts of the one series); cfg4 / cfg5 are fixed grids "sharded
# across the GPUs" -> strong scaling (the grid is cut into contiguous per-rank slices).
WORKLOADS = {
    "cfg2": dict(T=1000, p=1, q=2, series=1, restarts=4096, niter=100, scaling="weak"),
    "cfg3": dict(T=1000, p=4, q=8, series=1, restarts=8192, niter=100, scaling="weak"),
    "cfg4": dict(T=2000, p=1, q=4, series=10, restarts=1024, niter=100, scaling="strong",
                 kind="cvfolds"),      # 10 CV folds of one series: shared u, v; own NA mask
    "cfg5": dict(T=813, p=1, q=3, series=48, restarts=512, niter=100, scaling="strong",
                 kind="stations"),     # 48 independent series, observed tail of 30..90 steps
}


def build_problem(name, mask, world, rank):
    """Host arrays of this rank's slice of the workload's (series, restart) grid:
    Y [S,T], U [S or 1,T,p], V [S or 1,T,q] (time-major), shared_uv, cell_offsets [S+1],
    theta0 [cells, P], plus the global cell count."""
    from ldsr_amd import shard, synth
    w = WORKLOADS[name]
    T, p, q, S, R = w["T"], w["p"], w["q"], w["series"], w["restarts"]
    kind = w.get("kind", "single")
    if kind == "single":
        y, u, v = synth.make_series(T, p, q, series_id=0, mask=mask)
        n_global = world * R                       # weak: R restarts per rank
        lo, hi = shard.rank_slice(n_global, world, rank)
        th0 = synth.make_init_packed(p, q, hi - lo, seed=1, first=lo)
        return (y[None], np.ascontiguousarray(u.T[None]), np.ascontiguousarray(v.T[None]), 0,
                np.array([0, hi - lo], np.int32), th0, n_global)
    if kind == "cvfolds":
        # cvLDS (R/LDS_reconstruction.R:270-285): instrumental period = last 200 steps, fold k
        # hides a contiguous block of 21 instrumental points (make_Z blocks, R/utils.R:89-96)
        y, u, v = synth.make_series(T, p, q, series_id=4, mask="paleo", n_tail=200)
        Y = np.repeat(y[None], S, axis=0)
        for k in range(S):
            Y[k, T - 200 + 18 * k:T - 200 + 18 * k + 21] = np.nan
        U, V, shared = np.ascontiguousarray(u.T[None]), np.ascontiguousarray(v.T[None]), 1
    else:
        ys, us, vs = zip(*[synth.make_series(T, p, q, series_id=500 + s_, mask="paleo",
                                             n_tail=30 + (s_ * 60) // max(S - 1, 1))
                           for s_ in range(S)])
        Y = np.stack(ys)
        U = np.ascontiguousarray(np.stack([a.T for a in us]))
        V = np.ascontiguousarray(np.stack([a.T for a in vs]))
        shared = 0
    off = (np.arange(S + 1) * R).astype(np.int64)
    n_global = int(off[-1])
    lo, hi = shard.rank_slice(n_global, world, rank)
    keep, loc = shard.local_offsets(off, lo, hi)
    th0 = synth.make_init_packed(p, q, hi - lo, seed=1, first=lo)
    Yk = np.ascontiguousarray(Y[keep])
    if not shared:
        U, V = np.ascontiguousarray(U[keep]), np.ascontiguousarray(V[keep])
    return Yk, U, V, shared, loc, th0, n_global

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def bytes_per_unit(T, p, q):
    return 16 * T * (3 + p + q)


def load_pmc(workload, key):
    """A value of the committed rocprofv3 --pmc summary (profiles/pmc_summary.json), or None.
    PMC counters cannot be collected from inside the timed run; they come from separate
    rocprofv3 passes of this same command and are committed under profiles/."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get(key)
    except (OSError, ValueError):
        return None


def host_cores():
    """Usable host cores: affinity mask, capped by the cgroup CPU quota (the GPU box exposes
    256 CPUs but grants a 16-core share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def rscript_reference_probe(T, p, q, niter):
    """if (and only if) Rscript with the ldsr package is installed on this
    box, time the real RcppArmadillo path (LDS_EM on one restart) and report it separately.
    Never assumed: the build image and the GPU boxes seen so far have no R."""
    import shutil
    import subprocess
    rs = shutil.which("Rscript")
    if not rs:
        return {"available": False, "why": "Rscript not found on PATH"}
    code = ("suppressMessages(library(ldsr)); set.seed(1); T <- %d; p <- %d; q <- %d;"
            "u <- matrix(rnorm(p*T), p, T); v <- matrix(rnorm(q*T), q, T); y <- matrix(rnorm(T), 1, T);"
            "th <- make_init(p, q, 1)[[1]]; t0 <- proc.time()[3];"
            "for (i in 1:8) r <- ldsr:::LDS_EM(y, u, v, th, %d, 0); cat(8 * %d / (proc.time()[3] - t0))"
            % (T, p, q, niter, niter))
    try:
        out = subprocess.run([rs, "-e", code], capture_output=True, text=True, timeout=120)
        if out.returncode != 0:
            return {"available": False, "why": "Rscript present but ldsr not usable: "
                    + out.stderr.strip().splitlines()[-1][:120] if out.stderr.strip() else "error"}
        return {"available": True, "value": float(out.stdout.strip().split()[-1]),
                "unit": "restart*EM-iter/s", "cores": 1, "kind": "reference",
                "sample": "8 x LDS_EM(niter=%d, tol=0) on one R process" % niter}
    except Exception as e:   # noqa: BLE001 -- a probe must never break the benchmark
        return {"available": False, "why": "probe failed: %s" % type(e).__name__}


def cpu_baseline(p, q, niter, Y, U, V, seed):
    """The CPU oracle on the first series of the workload (Y [S,T], U [.,T,p], V [.,T,q])."""
    from oracle import oracle as O
    cores = host_cores()
    from ldsr_amd import synth
    cells = 512 * cores
    th0 = synth.make_init_packed(p, q, cells, seed=seed)
    Y = np.ascontiguousarray(Y[:1])
    U = np.ascontiguousarray(U[:1])
    V = np.ascontiguousarray(V[:1])
    soc = np.zeros(cells, np.int32)
    O.em_batch(Y, U, V, soc[:cores], th0[:cores], 3, 0.0, n_threads=cores)      # warm
    t0 = time.perf_counter()
    O.em_batch(Y, U, V, soc, th0, niter, 0.0, n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": cells * niter / dt, "unit": "restart*EM-iter/s", "cores": cores,
            "kind": "port",
            "sample": "%d restarts x %d EM iterations of the same series, %d host threads, %.1f s "
                      "(CPU oracle = scalar fp64 port of src/EM.cpp, not RcppArmadillo)"
                      % (cells, niter, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 serial, 2 scan")
    ap.add_argument("--mask", default="dense", choices=["dense", "paleo"])
    ap.add_argument("--niter", type=int, default=None, help="EM iteration cap (default: the workload's 100)")
    ap.add_argument("--tol", type=float, default=0.0,
                    help="stop tolerance; > 0 lets cells converge at their own pace and the "
                         "units are the iterations actually executed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
    # Rehearsal on a one-GPU box: LDSR_BENCH_BACKEND=gloo LDSR_BENCH_ONE_GPU=1 lets several ranks
    # share cuda:0 (the driver's real runs use one GPU per rank over RCCL).
    backend = os.environ.get("LDSR_BENCH_BACKEND", "nccl")
    if os.environ.get("LDSR_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ldsr_amd import _lib, shard, synth
    L = _lib.lib()

    w = WORKLOADS[args.workload]
    T, p, q, niter = w["T"], w["p"], w["q"], (args.niter or w["niter"])
    P = 6 + p + q
    Y, U, V, shared_uv, loc_off, th0, n_global = build_problem(args.workload, args.mask, world, rank)
    S_loc = Y.shape[0]
    cells = th0.shape[0]

    d_y = torch.from_numpy(Y).to(dev)          # [S][T]
    d_u = torch.from_numpy(U).to(dev)          # [S or 1][T][p]
    d_v = torch.from_numpy(V).to(dev)          # [S or 1][T][q]
    d_th0 = torch.from_numpy(th0).to(dev)
    d_th = torch.empty_like(d_th0)
    d_lik = torch.empty(cells, dtype=torch.float64, device=dev)
    d_nit = torch.empty(cells, dtype=torch.int32, device=dev)
    d_st = torch.empty(cells, dtype=torch.int32, device=dev)
    wsb = L.ldsr_em_workspace_bytes(S_loc, T, p, q, cells, args.algo)
    assert wsb > 0
    d_ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    off = (C.c_int * (S_loc + 1))(*[int(x) for x in loc_off])
    stream = torch.cuda.current_stream(dev)

    def step():
        _lib.check(L.ldsr_em_batch_device(
            local_rank, C.c_void_p(stream.cuda_stream), S_loc, T, p, q, d_y.data_ptr(),
            d_u.data_ptr(), d_v.data_ptr(), shared_uv, off, d_th0.data_ptr(), niter, args.tol, args.algo,
            d_th.data_ptr(), d_lik.data_ptr(), d_nit.data_ptr(), d_st.data_ptr(), None,
            C.c_void_p(ws_ptr), wsb))

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    L.ldsr_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tot_ms = C.c_double()
    n_l = C.c_int()
    _lib.check(L.ldsr_profile_collect(C.byref(tot_ms), C.byref(n_l)))
    L.ldsr_profile_enable(0)

    # the run must have done the work it claims
    nit = d_nit.cpu().numpy()
    st = d_st.cpu().numpy()
    if args.tol == 0.0:
        assert np.all(nit == niter), "cells stopped early"
    assert np.all(st == 0), "non-finite likelihoods in the bench batch"
    units_rank = int(nit.sum())            # E-steps actually executed by this rank per step

    units_all = units_rank
    if world > 1:
        tdev = dev if backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        usum = torch.tensor([units_rank], dtype=torch.int64, device=tdev)
        dist.all_reduce(usum, op=dist.ReduceOp.SUM)
        units_all = int(usum.item())

    if rank == 0:
        units_per_step = units_all        # = n_global * niter when tol == 0
        value = units_per_step * args.steps / dt
        kern_ms = tot_ms.value / max(n_l.value, 1)
        bpu = bytes_per_unit(T, p, q)
        alg_bytes = bpu * units_rank                          # per launch (this GPU)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "restart x EM-iteration / s (T=%d, p=%d, q=%d)" % (T, p, q),
            "value": value, "unit": "restart*EM-iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": w["scaling"], "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: synthetic T=%d p=%d q=%d, %d series x %d restarts%s, "
                                   "niter=%d, tol=0, %s mask"
                                   % (args.workload, T, p, q, w["series"], w["restarts"],
                                      "/GPU" if w["scaling"] == "weak" else " in total", niter,
                                      args.mask if w["series"] == 1 else "paleo-style"),
                       "cells_rank0": cells, "cells_total": n_global, "niter": niter, "tol": args.tol,
                       "algo": args.algo, "units_per_step": units_per_step,
                       "sharding": "contiguous cell ranges over %d rank(s), no collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_pmc(args.workload, "hbm_bytes_per_launch"),
                         "valu_busy_frac": load_pmc(args.workload, "valu_busy_frac"),
                         "kernel": "em_scan_kernel" if args.algo != 1 else "em_serial_kernel",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_unit": bpu,
                         "units_per_launch": units_rank,
                         "note": "algorithmic (logical) traffic; the series is served from LDS "
                                 "and the filtered states never leave registers, so measured "
                                 "HBM traffic (traffic, bytes per launch, PMC) is far below it; "
                                 "the binding resource is fp64 VALU issue (valu_busy_frac, PMC)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, q, niter, Y, U, V, seed=1)
            out["cpu_baseline"]["rscript_reference"] = rscript_reference_probe(T, p, q, niter)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
