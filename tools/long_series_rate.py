"""Rates of the scan kernel across series lengths (T <= 8192): one wave per cell up to 2048 steps,
two up to 4096, four up to 8192, against the serial kernel.  Prints restart x EM-iteration / s and
time steps / s (units x T) so that different lengths compare.  DESIGN.md section 4.5."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import _lib, synth  # noqa: E402
import ctypes as C  # noqa: E402

p, q, n, niter = 1, 2, 4096, 30
L = _lib.lib()
for T in [int(a) for a in (sys.argv[1:] or "1000 2000 2048 2100 3000 4096 5000 8192".split())]:
    y, u, v = synth.make_series(T, p, q, series_id=3, mask="paleo")
    th0 = synth.make_init_packed(p, q, n, seed=1)
    for label, env, algo, cells in (("scan", {}, 2, n), ("serial", {}, 1, n)):
        for k, val in env.items():
            os.environ[k] = val
        buf = C.create_string_buffer(160)
        L.ldsr_em_plan(T, p, q, niter, 0.0, algo, buf, 160)
        ldsr_amd.em_batch(y, u, v, th0[:cells], niter=niter, tol=0.0, algo=algo)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r = ldsr_amd.em_batch(y, u, v, th0[:cells], niter=niter, tol=0.0, algo=algo)
        dt = (time.perf_counter() - t0) / reps
        for k in env:
            del os.environ[k]
        units = cells * niter
        print("T=%5d %-9s %-52s %8.2f ms  %.3g units/s  %.3g steps/s" % (T, label, buf.value.decode(), dt * 1e3,
                                                                        units / dt, units * T / dt), flush=True)
