"""Where does ONE wave's EM iteration go?  (GPU box; library built with -DLDSR_SCAN_TIMING, see
em_scan_impl.h: tools/build_variant.sh timing "-DLDSR_SCAN_TIMING" em_scan_L2 em_scan_L4 em_scan_L13)

Prints shader-clock cycles per iteration of every section of the scan kernel for the small-launch
shapes of profiles/r03_small_launches.txt (lone waves: the launch's time is one wave's latency)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import api  # noqa: E402

NAMES = ["F1", "fwd scan", "F2", "B1+rev scan", "B2", "reduce+lik", "-", "M-step+loop"]


def run(T, p, q, cells, lead, niter=200):
    rng = np.random.default_rng(3)
    u = rng.standard_normal((p, T))
    v = rng.standard_normal((q, T))
    x = np.zeros(T)
    for t in range(1, T):
        x[t] = 0.8 * x[t - 1] + 0.3 * u[0, t - 1] + 0.3 * rng.standard_normal()
    y = x + 0.4 * v[0] + 0.3 * rng.standard_normal(T)
    y[:lead] = np.nan
    th0 = api.make_init_packed(p, q, cells, seed=5)
    r = api.em_batch(y[None, :], u, v, th0, niter=niter, tol=0.0, algo=2, return_liks=True)
    tk = r["liks"][:, :8] / niter
    med = np.median(tk, axis=0)
    tot = med.sum()
    print(f"T={T} p={p} q={q} cells={cells} lead={lead}: {tot:8.0f} cycles / iteration")
    for k, n in enumerate(NAMES):
        if n != "-":
            print(f"    {n:14s} {med[k]:8.0f}  {100 * med[k] / tot:5.1f} %")


if __name__ == "__main__":
    run(85, 7, 7, 50, 0)
    run(213, 3, 3, 50, 0)
    run(813, 3, 3, 50, 733)
    run(813, 3, 3, 50, 0)
