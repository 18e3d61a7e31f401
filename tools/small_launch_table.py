#!/usr/bin/env python3
"""GPU box: wall time of ONE LDS_EM_restart / cvLDS grid as the reference actually calls them
(small launches: 20..600 cells, niter = 1000, tol = 1e-5 -- R/LDS_reconstruction.R:122-125,
vignettes/ldsr.Rmd:66-71,133-139) through ldsr_em_restart_grid, with the CPU oracle beside it:
    python tools/small_launch_table.py [algo]        (algo: 0 auto, 2 scan, 3 pair, 4 quad)
Rows: the bundled Nakhon Phanom series (T = 813, p = q = 3, 46 observations) with 50 and 20 restarts,
its test slice (T = 213), the Ping known-answer shape (T = 85, p = q = 7, fully observed) with 50
restarts, the vignette's cross-validation (30 folds x 20 restarts at T = 413)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import ldsr_amd  # noqa: E402
from ldsr_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def refdata():
    with open(os.path.join(ROOT, "tests", "golden", "reference_data.json")) as f:
        return json.load(f)


def np_case(rd, start_year):
    qa = np.array(rd["NPannual"]["Qa"]); years = np.array(rd["NPannual"]["year"])
    pcs = np.array(rd["NPpc"]["data"])
    obs = np.log(qa); mu = obs.mean()
    u = np.ascontiguousarray(pcs[:, start_year - 1200:])
    T = u.shape[1]
    y = np.full(T, np.nan)
    i0 = years[0] - start_year
    y[i0:i0 + len(obs)] = obs - mu
    return y, u, u


def p1_case(rd):
    obs = np.log(np.array(rd["P1annual"]["Qa"]))
    pc = np.array(rd["P1pc"]["data"])
    u = np.ascontiguousarray(pc[:, 321:406])
    return obs - obs.mean(), u, u


def run(name, Y, u, v, n_restarts, algo, cores):
    """Y: [T] or [F, T] (folds share u, v)."""
    Y2 = np.atleast_2d(Y)
    F, T = Y2.shape
    p, q = u.shape[0], v.shape[0]
    th0 = synth.make_init_packed(p, q, F * n_restarts, seed=11)
    off = (np.arange(F + 1) * n_restarts).astype(np.int32)
    yy = Y2 if F > 1 else Y2[0]
    for _ in range(2):                                         # warm-up (arena, code objects)
        api.em_restart_grid(yy, u, v, th0, cell_offsets=off, niter=1000, tol=1e-5, algo=algo)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = api.em_restart_grid(yy, u, v, th0, cell_offsets=off, niter=1000, tol=1e-5, algo=algo)
        ts.append(time.perf_counter() - t0)
    gpu_ms = 1e3 * float(np.median(ts))
    U = np.repeat(np.ascontiguousarray(u.T)[None], F, axis=0)
    V = np.repeat(np.ascontiguousarray(v.T)[None], F, axis=0)
    soc = np.repeat(np.arange(F), n_restarts).astype(np.int32)
    t0 = time.perf_counter()
    _, _, nit, _ = O.em_batch(Y2, U, V, soc, th0, 1000, 1e-5, n_threads=cores)
    cpu_ms = 1e3 * (time.perf_counter() - t0)
    esteps = int(nit.sum())
    same = bool(np.array_equal(nit, r["all"]["n_iter"]))
    print("%-44s cells %4d  E-steps %7d (max %4d)  GPU %8.2f ms   CPU oracle (%d threads) %9.1f ms   x%.0f  n_iter %s"
          % (name, F * n_restarts, esteps, int(nit.max()), gpu_ms, cores, cpu_ms, cpu_ms / gpu_ms,
             "same" if same else "DIFF"), flush=True)


if __name__ == "__main__":
    algo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cores = min(16, os.cpu_count() or 1)
    rd = refdata()
    y, u, v = np_case(rd, 1200)
    run("NP T=813 p=q=3, 50 restarts", y, u, v, 50, algo, cores)
    run("NP T=813 p=q=3, 20 restarts", y, u, v, 20, algo, cores)
    y2, u2, v2 = np_case(rd, 1800)
    run("NP test slice T=213, 50 restarts", y2, u2, v2, 50, algo, cores)
    yp, up, vp = p1_case(rd)
    run("P1 T=85 p=q=7 (fully observed), 50 restarts", yp, up, vp, 50, algo, cores)
    y4, u4, v4 = np_case(rd, 1600)
    inst = np.nonzero(~np.isnan(y4))[0]
    Z = [np.asarray(z) - 1 for z in rd["NPcv"]["Z"]]
    Y = np.repeat(y4[None], len(Z), axis=0)
    for f, z in enumerate(Z):
        Y[f, inst[z]] = np.nan
    run("vignette cvLDS: 30 folds x 20 restarts, T=413", Y, u4, v4, 20, algo, cores)
