"""Per-wave time of BASELINE config 2's launch (GPU box; library built with -DLDSR_SCAN_TIMING:
tools/build_variant.sh timing "-DLDSR_SCAN_TIMING" em_pair_L32): the kernel's time is its slowest
wave's, so print the distribution over waves, what the slowest spent in the generic sweeps, and the
shader clock.  niter = 100, tol = 0 by default; --tol 1e-5 --niter 1000 for the converged run."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from ldsr_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--niter", type=int, default=100)
ap.add_argument("--tol", type=float, default=0.0)
ap.add_argument("--cells", type=int, default=4096)
a = ap.parse_args()
Y, U, V, shared, off, th0, n = bench.build_problem("cfg2", "dense", 1, 0)
th0 = th0[:a.cells]
r = api.em_batch(Y, np.transpose(U, (0, 2, 1))[0], np.transpose(V, (0, 2, 1))[0], th0, niter=max(a.niter, 12), tol=a.tol,
                 return_liks=True)
tk = r["liks"][:, :10]
cyc = tk[:, :9].sum(axis=1)                  # shader cycles the cell's wave spent (whole EM run of the cell)
clock = 100.0 * cyc / tk[:, 9]               # MHz
print("niter cap %d tol %g: n_iter min %d median %d max %d" % (a.niter, a.tol, r["n_iter"].min(), np.median(r["n_iter"]), r["n_iter"].max()))
print("shader clock %.0f MHz" % np.median(clock))
pc = np.percentile(cyc, [50, 90, 99, 100])
print("cycles per cell's wave (k): median %.0f  p90 %.0f  p99 %.0f  max %.0f   (max / median %.3f)"
      % (pc[0] / 1e3, pc[1] / 1e3, pc[2] / 1e3, pc[3] / 1e3, pc[3] / pc[0]))
print("wall (us, s_memrealtime) per cell: median %.1f max %.1f" % (np.median(tk[:, 9]) / 100, tk[:, 9].max() / 100))
names = ["consts+transient+verdict", "steady F1", "scan+steady F2", "rev composite+scan", "steady B2",
         "closed form+transient back", "generic sweeps", "loop top / M-step", "reduce+lik"]
mean = tk[:, :9].mean(axis=0)
for k in (7, 0, 1, 2, 3, 4, 5, 6, 8):
    print("    %-28s %9.0f  %5.1f %%" % (names[k], mean[k], 100 * mean[k] / mean.sum()))
worst = np.argsort(-cyc)[:8]
for c in worst:
    print("  cell %5d: %.0f k cycles, generic %.0f k (%.0f %%), wall %.1f us" % (c, cyc[c] / 1e3, tk[c, 6] / 1e3, 100 * tk[c, 6] / cyc[c], tk[c, 9] / 100))
