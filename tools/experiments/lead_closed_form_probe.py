#!/usr/bin/env python3
"""CPU probe for the next-round plan: closed form of the smoother over the
all-missing lead [0, t1) of a paleo-type series.  Over missing steps K_t = 0, so
    Xs_t - Xp_t = J_t (Xs_{t+1} - Xp_{t+1}),   Vs_t - Vp_t = J_t^2 (Vs_{t+1} - Vp_{t+1}),
    prod_{k=t}^{t1-1} J_k = A^(t1-t) Vp_t / Vp_t1 =: c_t        (J_k = A Vp_k / Vp_{k+1} telescopes)
=>  Xs_t = Xp_t + c_t delta,  Vs_t = Vp_t + c_t^2 eps,  (delta, eps) = (Xs - Xp, Vs - Vp) at t1,
with Xp_{t+1} = A Xp_t + B u_t, Vp_{t+1} = A^2 Vp_t + Q.  Checks the formulas (and the M-step sums
built from them) against the oracle's smoother on synthetic paleo series:
    python tools/lead_closed_form_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldsr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    worst = 0.0
    for (T, p, q, t1, sid) in ((1000, 1, 2, 900, 3), (2000, 1, 4, 1800, 4), (813, 1, 3, 760, 5), (813, 3, 3, 700, 6)):
        y, u, v = synth.make_series(T, p, q, series_id=sid)
        y = y.copy(); y[:t1] = np.nan
        for seed in range(6):
            th = synth.make_init_packed(p, q, 1, seed=seed)[0]
            if seed >= 3:           # thetas after some EM iterations (small R, A near the fit)
                th = O.lds_em(y, u, v, th, 8 + seed, 0.0)["theta"]
            fit = O.kalman_smoother(y, u, v, th)
            A, B, C = th[0], th[1:1 + p], th[1 + p]
            Q, mu1, V1 = th[2 + p + q], th[4 + p + q], th[5 + p + q]
            Xp = np.empty(t1 + 1); Vp = np.empty(t1 + 1)
            Xp[0], Vp[0] = mu1, V1
            for t in range(t1):
                Xp[t + 1] = A * Xp[t] + B @ u[:, t]
                Vp[t + 1] = A * A * Vp[t] + Q
            delta = fit["X"][t1] - Xp[t1]
            eps = fit["V"][t1] - Vp[t1]
            c = A ** (t1 - np.arange(t1 + 1)) * Vp / Vp[t1]
            Xs = Xp + c * delta
            Vs = Vp + c * c * eps
            J = c[:-1] / c[1:]
            e = [np.max(np.abs(Xs[:t1] - fit["X"][:t1]) / (1e-300 + np.abs(fit["X"][:t1]).max())),
                 np.max(np.abs(Vs[:t1] - fit["V"][:t1]) / np.abs(fit["V"][:t1]).max()),
                 np.max(np.abs(J - fit["J"][:t1]) / np.abs(fit["J"][:t1]).max())]
            # the lead's share of the M-step sums (src/EM.cpp:180-193), closed form vs from the fit
            Pall = np.sum(Xp[:t1] ** 2 + Vp[:t1]) + 2 * delta * np.sum(c[:t1] * Xp[:t1]) + (delta ** 2 + eps) * np.sum(c[:t1] ** 2)
            Pall_ref = np.sum(fit["X"][:t1] ** 2 + fit["V"][:t1])
            Tx1x = (np.sum(Xp[1:] * Xp[:-1]) + delta * np.sum(c[1:] * Xp[:-1] + c[:-1] * Xp[1:])
                    + (delta ** 2 + eps) * np.sum(c[:-1] * c[1:]) + np.sum(A * Vp[:-1]))
            Tx1x_ref = np.sum(fit["X"][1:t1 + 1] * fit["X"][:t1] + fit["V"][1:t1 + 1] * fit["J"][:t1])
            e += [abs(Pall - Pall_ref) / abs(Pall_ref), abs(Tx1x - Tx1x_ref) / abs(Tx1x_ref)]
            worst = max(worst, max(e))
            print("T=%d p=%d q=%d lead=%d seed %d (A=%.3f): rel err Xs %.1e Vs %.1e J %.1e  sums Pall %.1e Tx1x %.1e"
                  % (T, p, q, t1, seed, A, *e))
    print("worst relative error %.2e" % worst)


if __name__ == "__main__":
    main()
