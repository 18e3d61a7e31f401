#!/usr/bin/env python3
"""GPU: how long an observed tail still pays for the closed-form lead.  Run once per setting:
    LDSR_LEAD_MAX_TAIL=256 python tools/lead_tail_ab.py ; LDSR_LEAD_MAX_TAIL=512 python tools/lead_tail_ab.py
Prints the kernel AUTO took and the best wall time of the host-pointer entry (8192 cells, 100 iterations)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import synth, _lib  # noqa: E402

for T, p, q, tail in ((1000, 1, 2, 300), (1000, 1, 2, 400), (1000, 1, 2, 500), (2000, 1, 4, 300), (2000, 1, 4, 400),
                      (2000, 1, 4, 500), (1500, 3, 3, 350), (1500, 3, 3, 500), (813, 1, 3, 300), (4000, 2, 2, 500)):
    y, u, v = synth.make_series(T, p, q, series_id=11)
    y = y.copy(); y[:T - tail] = np.nan
    th0 = synth.make_init_packed(p, q, 8192, seed=T)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = ldsr_amd.em_batch(y, u, v, th0, niter=100, tol=0.0)
        best = min(best, time.perf_counter() - t0)
    buf = ctypes.create_string_buffer(160)
    _lib.lib().ldsr_last_em_kernel(0, buf, 160)
    print("T=%d p=%d q=%d tail=%d max_tail=%s: %-52s %.3f ms  lik[0]=%.10g" % (
        T, p, q, tail, os.environ.get("LDSR_LEAD_MAX_TAIL", "default"), buf.value.decode(), best * 1e3, r["lik"][0]), flush=True)
