"""Throughput of the long/wide corner (series image in global memory) vs the serial kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd
from ldsr_amd import synth
T, p, q, n, niter = 1700, 7, 5, 4096, 50
y, u, v = synth.make_series(T, p, q, series_id=3)
th0 = synth.make_init_packed(p, q, n, seed=1)
for algo, name in ((2, "scan (global image)"), (1, "serial")):
    m = n if algo == 2 else 1024
    ldsr_amd.em_batch(y, u, v, th0[:m], niter=5, tol=0.0, algo=algo)
    t0 = time.perf_counter()
    ldsr_amd.em_batch(y, u, v, th0[:m], niter=niter, tol=0.0, algo=algo)
    dt = time.perf_counter() - t0
    print("%-20s T=%d p=%d q=%d: %d cells x %d its in %.1f ms = %.3g units/s" % (name, T, p, q, m, niter, dt * 1e3, m * niter / dt))
