"""Prints the measured deviation of the GPU engine from the CPU oracle on config 2 (4096 cells,
converged runs).  Diagnostic for DESIGN.md section 5; run on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldsr_amd  # noqa: E402
from ldsr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

for mask in ("dense", "paleo"):
    y, u, v = synth.make_series(1000, 1, 2, series_id=0, mask=mask)
    th0 = synth.make_init_packed(1, 2, 4096, seed=1)
    ref = O.em_batch(y[None], u.T[None].copy(), v.T[None].copy(), np.zeros(4096, np.int32), th0,
                     1000, 1e-5, n_threads=16)
    for algo, name in ((1, "serial"), (2, "scan")):
        r = ldsr_amd.em_batch(y, u, v, th0, niter=1000, tol=1e-5, algo=algo)
        same = int(np.sum(r["n_iter"] == ref[2]))
        names = ["A", "B", "C", "D1", "D2", "Q", "R", "mu1", "V1"]
        d = np.abs(r["theta"] - ref[0])
        tolfrac = (d / (1e-6 * np.abs(ref[0]) + 1e-9)).max()       # 1.0 = at the parity bar
        dl = np.abs(r["lik"] - ref[1]) / np.abs(ref[1])
        per = "  ".join("%s %.1e" % (n, (d[:, i] / np.maximum(np.abs(ref[0][:, i]), 1e-300)).max())
                        for i, n in enumerate(names) if n != "mu1")
        print("%-5s %-6s n_iter equal %d/4096 (range %d..%d)  worst fraction of the parity bar %.1e  "
              "max rel dlik %.1e  max abs d(mu1) %.1e\n      max rel per parameter: %s"
              % (mask, name, same, ref[2].min(), ref[2].max(), tolfrac, dl.max(), d[:, 7].max(), per))
