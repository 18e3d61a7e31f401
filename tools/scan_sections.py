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


if __name__ == "__main__" and not os.environ.get("SECTIONS_PAIR"):
    run(85, 7, 7, 50, 0)
    run(213, 3, 3, 50, 0)
    run(813, 3, 3, 50, 733)
    run(813, 3, 3, 50, 0)


PAIR_NAMES = ["consts+transient", "steady F1", "scan+steady F2", "rev composite+scan", "steady B2",
              "closed form+transient back", "generic fallback", "M-step+stop+loop", "reduce+lik"]


def run_pair(T=1024, p=1, q=2, cells=4096, niter=100):
    """BASELINE config 2 through AUTO (the two-cells-per-wave kernel with steady sweeps): cycles per
    iteration of a wave, two waves per SIMD -- shares, not latencies."""
    rng = np.random.default_rng(2)
    u = rng.standard_normal((p, T))
    v = rng.standard_normal((q, T))
    x = np.zeros(T)
    for t in range(1, T):
        x[t] = 0.8 * x[t - 1] + 0.3 * u[0, t - 1] + 0.3 * rng.standard_normal()
    y = x + 0.4 * v[0] + 0.3 * rng.standard_normal(T)
    th0 = api.make_init_packed(p, q, cells, seed=5)
    r = api.em_batch(y[None, :], u, v, th0, niter=niter, tol=0.0, return_liks=True)
    tk = r["liks"][:, :10] / niter
    med = np.mean(tk, axis=0)
    tot = med[:9].sum()
    print(f"shader clock while the kernel runs: {100.0 * tot / med[9]:.0f} MHz (s_memtime cycles per s_memrealtime tick of 10 ns)")
    print(f"pair kernel T={T} p={p} q={q} cells={cells}: {tot:8.0f} cycles / iteration (mean over cells)")
    tot_c = tk[:, :9].sum(axis=1)
    pc = np.percentile(tot_c, [50, 90, 99, 100])
    print("    cycles / iteration of a cell's wave: median %.0f  p90 %.0f  p99 %.0f  max %.0f" % tuple(pc))
    slow = tot_c >= np.percentile(tot_c, 99)
    print("    slowest 1 %% of the cells: generic fallback %.0f cycles / iteration (%.0f %% of their time), n_iter all %d"
          % (tk[slow, 6].mean(), 100 * tk[slow, 6].mean() / tot_c[slow].mean(), niter))
    order = [7, 0, 1, 2, 3, 4, 5, 6, 8]
    names = dict(zip([0, 1, 2, 3, 4, 5, 6, 7, 8], PAIR_NAMES))
    for k in order:
        print(f"    {names[k]:28s} {med[k]:8.0f}  {100 * med[k] / tot:5.1f} %")


def run_lead(T, p, q, cells, lead, niter=100):
    """A paleo-type launch through AUTO (closed-form lead + sweeps of the tail): slot 0 = iteration
    constants + the lead's first pass, slot 6 = the generic sweeps of the tail, slot 5 = the lead's
    second pass (behind the sweeps)."""
    rng = np.random.default_rng(4)
    u = rng.standard_normal((p, T))
    v = rng.standard_normal((q, T))
    x = np.zeros(T)
    for t in range(1, T):
        x[t] = 0.8 * x[t - 1] + 0.3 * u[0, t - 1] + 0.3 * rng.standard_normal()
    y = x + 0.4 * v[0] + 0.3 * rng.standard_normal(T)
    y[:lead] = np.nan
    th0 = api.make_init_packed(p, q, cells, seed=5)
    r = api.em_batch(y[None, :], u, v, th0, niter=niter, tol=0.0, return_liks=True)
    tk = r["liks"][:, :10] / niter
    med = np.mean(tk, axis=0)
    tot = med[:9].sum()
    print(f"LEAD launch T={T} p={p} q={q} cells={cells} lead={lead}: {tot:8.0f} cycles / iteration, clock {100.0 * tot / med[9]:.0f} MHz")
    for k, nme in ((7, "M-step+stop+loop"), (0, "constants + first lead pass"), (6, "sweeps of the tail"),
                   (5, "second lead pass"), (8, "reduce+lik")):
        print(f"    {nme:28s} {med[k]:8.0f}  {100 * med[k] / tot:5.1f} %")


if __name__ == "__main__" and os.environ.get("SECTIONS_PAIR"):
    run_pair()
    run_lead(2000, 1, 4, 10240, 1800)
    run_lead(813, 1, 3, 24576, 733)
