"""Times the host-pointer entry ldsr_em_batch (H2D copies, allocation, launch, D2H) on config 2.
The number goes into DESIGN.md as the PCIe-inclusive rate; it is never bench.py's `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import synth  # noqa: E402

y, u, v = synth.make_series(1000, 1, 2)
th0 = synth.make_init_packed(1, 2, 4096, seed=1)
for _ in range(3):
    ldsr_amd.em_batch(y, u, v, th0, niter=100, tol=0.0)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    r = ldsr_amd.em_batch(y, u, v, th0, niter=100, tol=0.0)
dt = (time.perf_counter() - t0) / n
print("ldsr_em_batch (host pointers): %.3f ms per call = %.3g units/s" % (dt * 1e3, 409600 / dt))
