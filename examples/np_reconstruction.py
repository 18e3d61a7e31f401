#!/usr/bin/env python3
"""The vignette's reconstruction of the Mekong at Nakhon Phanom (reference
vignettes/ldsr.Rmd:60-80; R/LDS_reconstruction.R:122-258) with the EM restarts on the GPU.

Only the hot path runs on the device (all restarts in one launch); the few lines of
pre/post-processing around it restate what LDS_reconstruction does in R: log-transform and
centre the flow (R/LDS_reconstruction.R:164-182), pad with NA outside the instrumental years
(:180-183), build 95 % intervals from the smoothed variance (:190-212).

    python examples/np_reconstruction.py [num_restarts]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ldsr_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_data.json")))
    years_obs = np.array(d["NPannual"]["year"])
    qa = np.array(d["NPannual"]["Qa"])
    pcs = np.array(d["NPpc"]["data"])               # 3 x 813, years 1200..2012
    start_year = 1200
    years = np.arange(start_year, start_year + pcs.shape[1])

    obs = np.log(qa)                                 # transform = 'log'
    mu = obs.mean()
    y = np.full(years.size, np.nan)
    y[years_obs[0] - start_year:years_obs[0] - start_year + obs.size] = obs - mu
    u = v = pcs

    init = ldsr_amd.make_init(u.shape[0], v.shape[0], n, r_seed=1)     # set.seed(1); make_init(3, 3, n)
    fit = ldsr_amd.LDS_EM_restart(y, u, v, init, niter=1000, tol=1e-5)

    X, V, Y = fit["fit"]["X"][0], fit["fit"]["V"][0], fit["fit"]["Y"][0] + mu
    C, R = fit["theta"]["C"][0, 0], fit["theta"]["R"][0, 0]
    sdY = np.sqrt(C * V * C + R)
    Q = np.exp(Y)                                    # exp_ci(): log-normal 5 % / 95 % quantiles
    Ql, Qu = np.exp(Y - 1.6448536269514722 * sdY), np.exp(Y + 1.6448536269514722 * sdY)
    a = fit["all"]
    print("restarts: %d   finite: %d   with C > 0: %d   iterations: %d..%d"
          % (n, np.isfinite(a["lik"]).sum(), (a["theta"][:, 4] > 0).sum(), a["n_iter"].min(), a["n_iter"].max()))
    print("best lik %.6f (package's NPlds: %.6f)   A %.4f  C %.5f  Q %.4f  R %.5f"
          % (fit["lik"], d["NPlds"]["lik"][0], fit["theta"]["A"][0, 0], C, fit["theta"]["Q"][0, 0], R))
    for yr in (1200, 1500, 1800, 1960, 2005):
        i = yr - start_year
        print("  %d  X %+.3f +- %.3f   Q %.0f  [%.0f, %.0f]" % (yr, X[i], 1.96 * np.sqrt(V[i]), Q[i], Ql[i], Qu[i]))


if __name__ == "__main__":
    main()
