// kernels_serial.hip -- series preparation and the one-thread-per-cell kernels.
//
// em_serial_kernel is the general path (any T): each thread owns one (series, restart) cell
// and walks time sequentially with the reference's own expressions
// (/root/reference/src/EM.cpp:70-104), fusing every M-step sum (:151-161,:180-193) into the
// backward sweep.  The filtered pair (Xu_t, Vu_t) is the only thing carried from the forward
// to the backward sweep; it goes through an HBM strip laid out [t][cell] so that the 64 cells
// of a wavefront store and load it coalesced.  The fast path for T <= 2048 is the
// wave-per-cell scan kernel in kernels_scan.hip.
#include "ldsr_device.h"
#include "ldsr_kernels.h"

// ---------------------------------------------------------------------------------------
// series_prep_kernel: one block per series.  Builds the zero-padded time-major copies of
// u and v, and the theta-independent statistics (SeriesConst).
// ---------------------------------------------------------------------------------------
__device__ static bool invert_small(double *a, int n) {
    // Gauss-Jordan with partial pivoting on an n x n matrix stored with stride LDSR_MAXPQ;
    // entries outside n x n are left untouched (identity padding).  One thread.
    double inv[LDSR_MAXPQ * LDSR_MAXPQ];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) inv[i * LDSR_MAXPQ + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = fabs(a[c * LDSR_MAXPQ + c]);
        for (int r = c + 1; r < n; r++) {
            const double m = fabs(a[r * LDSR_MAXPQ + c]);
            if (m > best) { best = m; piv = r; }
        }
        if (!(best > 0.0) || !isfinite(best)) return false;
        if (piv != c)
            for (int j = 0; j < n; j++) {
                double t = a[c * LDSR_MAXPQ + j];
                a[c * LDSR_MAXPQ + j] = a[piv * LDSR_MAXPQ + j];
                a[piv * LDSR_MAXPQ + j] = t;
                t = inv[c * LDSR_MAXPQ + j];
                inv[c * LDSR_MAXPQ + j] = inv[piv * LDSR_MAXPQ + j];
                inv[piv * LDSR_MAXPQ + j] = t;
            }
        const double d = 1.0 / a[c * LDSR_MAXPQ + c];
        for (int j = 0; j < n; j++) {
            a[c * LDSR_MAXPQ + j] *= d;
            inv[c * LDSR_MAXPQ + j] *= d;
        }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double f = a[r * LDSR_MAXPQ + c];
            for (int j = 0; j < n; j++) {
                a[r * LDSR_MAXPQ + j] -= f * a[c * LDSR_MAXPQ + j];
                inv[r * LDSR_MAXPQ + j] -= f * inv[c * LDSR_MAXPQ + j];
            }
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[i * LDSR_MAXPQ + j] = inv[i * LDSR_MAXPQ + j];
    return true;
}

#define PREP_TILE 256
__global__ __launch_bounds__(256) void series_prep_kernel(PrepParams prm) {
    const int s = blockIdx.x;
    const int T = prm.T, p = prm.p, q = prm.q, PP = prm.PP, QQ = prm.QQ;
    const int tid = threadIdx.x;
    const double *y = prm.y + (long)s * T;
    const bool own_uv = (!prm.shared_uv) || s == 0;
    const double *u = prm.u ? prm.u + (prm.shared_uv ? 0 : (long)s * T * p) : nullptr;
    const double *v = prm.v ? prm.v + (prm.shared_uv ? 0 : (long)s * T * q) : nullptr;
    double *yp = prm.yp + (long)s * T;
    double *up = prm.up + (prm.shared_uv ? 0 : (long)s * T * PP);
    double *vp = prm.vp + (prm.shared_uv ? 0 : (long)s * T * QQ);

    // The series is streamed through LDS in tiles of PREP_TILE steps (coalesced loads); each of
    // the 138 statistics is owned by one thread and summed in ascending t like the reference.
    __shared__ double ty[PREP_TILE], tu[PREP_TILE * LDSR_MAXPQ], tv[PREP_TILE * LDSR_MAXPQ];
    __shared__ SeriesConst sc;
    // role of this thread
    const int role = tid < 64 ? 0 : tid < 128 ? 1 : tid < 136 ? 2 : tid == 136 ? 3 : 4;
    const int k = (role == 2) ? tid - 128 : ((tid & 63) >> 3), l = tid & 7;
    double acc = 0.0;
    int n = 0, first = -1, last = -1;
    const bool live = (role == 0 && v && k < q && l < q) || (role == 1 && u && k < p && l < p) ||
                      (role == 2 && v && k < q) || role == 3;
    for (int tb = 0; tb < T; tb += PREP_TILE) {
        const int nt = min(PREP_TILE, T - tb);
        for (int i = tid; i < nt; i += 256) {
            const double yv = y[tb + i];
            ty[i] = yv;
            yp[tb + i] = yv;
        }
        if (u)
            for (int i = tid; i < nt * p; i += 256) tu[i] = u[(long)tb * p + i];
        if (v)
            for (int i = tid; i < nt * q; i += 256) tv[i] = v[(long)tb * q + i];
        __syncthreads();
        if (own_uv) {
            for (int i = tid; i < nt * PP; i += 256) {
                const int tt = i / PP, kk = i - tt * PP;
                // u[:,T-1] is never read by the reference (src/EM.cpp:74,190-193): zero it
                up[(long)tb * PP + i] = (u && kk < p && tb + tt < T - 1) ? tu[tt * p + kk] : 0.0;
            }
            for (int i = tid; i < nt * QQ; i += 256) {
                const int tt = i / QQ, kk = i - tt * QQ;
                vp[(long)tb * QQ + i] = (v && kk < q) ? tv[tt * q + kk] : 0.0;
            }
        }
        if (live) {
            // branch-free bodies so the LDS reads of successive steps pipeline
            if (role == 0) {
#pragma unroll 8
                for (int tt = 0; tt < nt; tt++) {
                    const double pr = tv[tt * q + k] * tv[tt * q + l];
                    acc += isfinite(ty[tt]) ? pr : 0.0;
                }
            } else if (role == 1) {
                const int ne = min(nt, T - 1 - tb);
#pragma unroll 8
                for (int tt = 0; tt < ne; tt++) acc += tu[tt * p + k] * tu[tt * p + l];
            } else if (role == 2) {
#pragma unroll 8
                for (int tt = 0; tt < nt; tt++) {
                    const double yv = ty[tt];
                    acc += isfinite(yv) ? yv * tv[tt * q + k] : 0.0;
                }
            } else {
#pragma unroll 8
                for (int tt = 0; tt < nt; tt++) {
                    const double yv = ty[tt];
                    const bool o = isfinite(yv);
                    acc += o ? yv * yv : 0.0;
                    n += o ? 1 : 0;
                    first = (o && first < 0) ? tb + tt : first;
                    last = o ? tb + tt : last;
                }
            }
        }
        __syncthreads();
    }
    if (role == 0) sc.Svv_inv[k * LDSR_MAXPQ + l] = live ? acc : (k == l ? 1.0 : 0.0);
    if (role == 1) sc.Tuu_inv[k * LDSR_MAXPQ + l] = live ? acc : (k == l ? 1.0 : 0.0);
    if (role == 2) sc.Syv[k] = live ? acc : 0.0;
    if (role == 3) {
        sc.Syy = acc;
        sc.n_obs = n;
        sc.t_first_obs = first;
        sc.t_last_obs = last;
    }
    __syncthreads();
    if (tid == 0) {
        bool ok = sc.n_obs > 0;
        if (v) ok = invert_small(sc.Svv_inv, q) && ok;
        if (u) ok = invert_small(sc.Tuu_inv, p) && ok;
        sc.status = ok ? 0 : 2;
        for (int kk = 0; kk < LDSR_MAXPQ; kk++) {
            double a = 0.0;
            for (int ll = 0; ll < LDSR_MAXPQ; ll++) a += sc.Svv_inv[kk * LDSR_MAXPQ + ll] * sc.Syv[ll];
            sc.wv[kk] = a;
        }
        prm.sc[s] = sc;
    }
}

// ---------------------------------------------------------------------------------------
// em_serial_kernel: one thread per cell, whole EM loop on the device.
// ---------------------------------------------------------------------------------------
template <int PP, int QQ, bool STAGE>
__global__ __launch_bounds__(64) void em_serial_kernel(EmParams prm) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    const int s = prm.blk_series[b];
    const int c0 = prm.blk_cell0[b], nc = prm.blk_ncell[b];
    const int T = prm.T;
    const double *gy = prm.yp + (long)s * T;
    const double *gu = prm.up + (long)s * prm.u_stride;
    const double *gv = prm.vp + (long)s * prm.v_stride;
    const double *y, *u, *v;
    if constexpr (STAGE) {
        for (int i = threadIdx.x; i < T; i += 64) smem[i] = gy[i];
        for (int i = threadIdx.x; i < T * PP; i += 64) smem[T + i] = gu[i];
        for (int i = threadIdx.x; i < T * QQ; i += 64) smem[T + T * PP + i] = gv[i];
        __syncthreads();
        y = smem;
        u = smem + T;
        v = smem + T + T * PP;
    } else {
        y = gy;
        u = gu;
        v = gv;
    }
    if ((int)threadIdx.x >= nc) return;
    const int cell = c0 + threadIdx.x;
    const int P = 6 + prm.p + prm.q;
    const SeriesConst *sc = prm.sc + s;
    const int n_obs = sc->n_obs;
    double *sx = prm.scratch + cell;
    const long sst = prm.scratch_stride;

    Theta<PP, QQ> th;
    load_theta(th, prm.theta0 + (long)cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
    if (sc->status != 0) {
        for (int k = 0; k < P; k++) prm.theta[(long)cell * P + k] = NAN;
        prm.lik[cell] = NAN;
        prm.n_iter[cell] = 0;
        prm.status[cell] = 2;
        return;
    }

    double lik = NAN, lik1 = NAN, lik2 = NAN;
    int it = 0;
    for (;;) {
        // ---------------- E-step, forward filter (src/EM.cpp:48-90) + likelihood (:113-124)
        double Xp = th.mu1, Vp = th.V1, Xu = 0.0, Vu = 0.0, acc = 0.0;
        for (int t = 0; t < T; t++) {
            if (t > 0) {
                double bu = 0.0;
#pragma unroll
                for (int k = 0; k < PP; k++) bu = fma(th.B[k], u[(t - 1) * PP + k], bu);
                Xp = th.A * Xu + bu;
                Vp = th.A * Vu * th.A + th.Q;
            }
            double dv = 0.0;
#pragma unroll
            for (int k = 0; k < QQ; k++) dv = fma(th.D[k], v[t * QQ + k], dv);
            const double Yp = th.C * Xp + dv;
            const double yt = y[t];
            if (isfinite(yt)) {
                const double Sigma = th.C * Vp * th.C + th.R;
                const double K = Vp * th.C / Sigma;
                const double delta = yt - Yp;
                Xu = Xp + K * delta;
                Vu = (1.0 - K * th.C) * Vp;
                acc += delta / Sigma * delta + log(Sigma);
            } else {
                Xu = Xp;
                Vu = Vp;
            }
            sx[(2L * t) * sst] = Xu;
            sx[(2L * t + 1) * sst] = Vu;
        }
        lik2 = lik1;
        lik1 = lik;
        lik = (-0.5 * n_obs * LDSR_LOG_2PI - 0.5 * acc) / n_obs;
        if (prm.liks) prm.liks[(long)cell * prm.niter + it] = lik;
        it++;
        // stop rule of src/EM.cpp:272 (needs three likelihoods), or iteration cap
        if (it >= prm.niter) break;
        if (it >= 3 && fabs(lik - lik1) < prm.tol && fabs(lik1 - lik2) < prm.tol) break;

        // ---------------- backward smoother (:94-104) fused with the M-step sums (:151-193)
        Sums<PP, QQ> S;
        S.Syx = 0.0; S.Sxx = 0.0; S.Tx1x = 0.0;
#pragma unroll
        for (int k = 0; k < QQ; k++) S.Sxv[k] = 0.0;
#pragma unroll
        for (int k = 0; k < PP; k++) { S.Tx1u[k] = 0.0; S.Tux[k] = 0.0; }
        double Xs = Xu, Vs = Vu;  // t = T-1
        double Pmid = 0.0;        // sum_{t=1}^{T-2} Xs^2 + Vs
        const double termLast = Xs * Xs + Vs;
        if (isfinite(y[T - 1])) {
            S.Syx = y[T - 1] * Xs;
            S.Sxx = termLast;
#pragma unroll
            for (int k = 0; k < QQ; k++) S.Sxv[k] = Xs * v[(T - 1) * QQ + k];
        }
        double term0 = 0.0;
        for (int t = T - 2; t >= 0; t--) {
            const double xu = sx[(2L * t) * sst], vu = sx[(2L * t + 1) * sst];
            double bu = 0.0;
#pragma unroll
            for (int k = 0; k < PP; k++) bu = fma(th.B[k], u[t * PP + k], bu);
            const double Xp1 = th.A * xu + bu;              // identical to the forward value
            const double Vp1 = th.A * vu * th.A + th.Q;
            const double J = vu * th.A / Vp1;
            const double Xn = xu + J * (Xs - Xp1);
            const double Vn = vu + J * (Vs - Vp1) * J;
            S.Tx1x += Xs * Xn + Vs * J;
#pragma unroll
            for (int k = 0; k < PP; k++) {
                const double ut = u[t * PP + k];
                S.Tx1u[k] = fma(Xs, ut, S.Tx1u[k]);
                S.Tux[k] = fma(ut, Xn, S.Tux[k]);
            }
            const double term = Xn * Xn + Vn;
            if (t > 0) Pmid += term; else term0 = term;
            const double yt = y[t];
            if (isfinite(yt)) {
                S.Syx = fma(yt, Xn, S.Syx);
                S.Sxx += term;
#pragma unroll
                for (int k = 0; k < QQ; k++) S.Sxv[k] = fma(Xn, v[t * QQ + k], S.Sxv[k]);
            }
            Xs = Xn;
            Vs = Vn;
        }
        S.Txx = Pmid + term0;
        S.Tx1x1 = Pmid + termLast;
        S.X0 = Xs;
        S.V0 = Vs;
        mstep_update(th, S, sc, T);
    }
    store_theta(th, prm.theta + (long)cell * P, prm.p, prm.q);
    if (prm.liks)
        for (int i = it; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
    prm.lik[cell] = lik;
    prm.n_iter[cell] = it;
    prm.status[cell] = isfinite(lik) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------
// smooth_kernel: Kalman_smoother (src/EM.cpp:22-131) for a batch of thetas, writing the
// full fit.  X / V double as the filtered-state strip during the forward sweep.
// mode 0 = smoother, mode 1 = propagate (src/EM.cpp:295-356: no measurement update).
// ---------------------------------------------------------------------------------------
template <int PP, int QQ>
__global__ __launch_bounds__(64) void smooth_kernel(SmoothParams prm) {
    const int cell = blockIdx.x * 64 + threadIdx.x;
    if (cell >= prm.n_cells) return;
    const int s = prm.series_of_cell[cell];
    const int T = prm.T;
    const double *y = prm.yp + (long)s * T;
    const double *u = prm.up + (long)s * prm.u_stride;
    const double *v = prm.vp + (long)s * prm.v_stride;
    const int P = 6 + prm.p + prm.q;
    Theta<PP, QQ> th;
    load_theta(th, prm.theta + (long)cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
    double *X = prm.X + (long)cell * T, *V = prm.V + (long)cell * T;
    double *Y = prm.Y + (long)cell * T, *J = prm.J + (long)cell * T;
    const bool prop = prm.mode == 1;

    double Xp = th.mu1, Vp = th.V1, Xu = 0.0, Vu = 0.0, acc = 0.0;
    int n_obs = 0;
    for (int t = 0; t < T; t++) {
        if (t > 0) {
            double bu = 0.0;
#pragma unroll
            for (int k = 0; k < PP; k++) bu = fma(th.B[k], u[(t - 1) * PP + k], bu);
            Xp = th.A * Xu + bu;
            Vp = th.A * Vu * th.A + th.Q;
        }
        double dv = 0.0;
#pragma unroll
        for (int k = 0; k < QQ; k++) dv = fma(th.D[k], v[t * QQ + k], dv);
        const double Yp = th.C * Xp + dv;
        const double yt = y[t];
        Xu = Xp;
        Vu = Vp;
        if (isfinite(yt)) {
            const double Sigma = th.C * Vp * th.C + th.R;
            const double delta = yt - Yp;
            acc += delta / Sigma * delta + log(Sigma);
            n_obs++;
            if (!prop) {
                const double K = Vp * th.C / Sigma;
                Xu = Xp + K * delta;
                Vu = (1.0 - K * th.C) * Vp;
            }
        }
        X[t] = Xu;
        V[t] = Vu;
        if (prop) Y[t] = Yp;
    }
    double lik = -0.5 * n_obs * LDSR_LOG_2PI - 0.5 * acc;
    if (prm.stdlik) lik = lik / n_obs;
    prm.lik[cell] = lik;
    if (prop) return;

    double Xs = Xu, Vs = Vu;
    J[T - 1] = Vu * th.A / (th.A * Vu * th.A + th.Q);  // src/EM.cpp:98
    {
        double dv = 0.0;
#pragma unroll
        for (int k = 0; k < QQ; k++) dv = fma(th.D[k], v[(T - 1) * QQ + k], dv);
        Y[T - 1] = th.C * Xs + dv;
    }
    for (int t = T - 2; t >= 0; t--) {
        const double xu = X[t], vu = V[t];
        double bu = 0.0;
#pragma unroll
        for (int k = 0; k < PP; k++) bu = fma(th.B[k], u[t * PP + k], bu);
        const double Xp1 = th.A * xu + bu;
        const double Vp1 = th.A * vu * th.A + th.Q;
        const double Jt = vu * th.A / Vp1;
        Xs = xu + Jt * (Xs - Xp1);
        Vs = vu + Jt * (Vs - Vp1) * Jt;
        double dv = 0.0;
#pragma unroll
        for (int k = 0; k < QQ; k++) dv = fma(th.D[k], v[t * QQ + k], dv);
        X[t] = Xs;
        V[t] = Vs;
        J[t] = Jt;
        Y[t] = th.C * Xs + dv;
    }
}

// ---------------------------------------------------------------------------------------
// mstep_kernel: Mstep (src/EM.cpp:139-229) from a stored fit, sums in ascending t.
// ---------------------------------------------------------------------------------------
template <int PP, int QQ>
__global__ __launch_bounds__(64) void mstep_kernel(SmoothParams prm) {
    const int cell = blockIdx.x * 64 + threadIdx.x;
    if (cell >= prm.n_cells) return;
    const int s = prm.series_of_cell[cell];
    const int T = prm.T;
    const double *y = prm.yp + (long)s * T;
    const double *u = prm.up + (long)s * prm.u_stride;
    const double *v = prm.vp + (long)s * prm.v_stride;
    const int P = 6 + prm.p + prm.q;
    const SeriesConst *sc = prm.sc + s;
    const double *X = prm.X + (long)cell * T, *V = prm.V + (long)cell * T,
                 *J = prm.J + (long)cell * T;
    Sums<PP, QQ> S;
    S.Syx = 0.0; S.Sxx = 0.0; S.Tx1x = 0.0; S.Txx = 0.0; S.Tx1x1 = 0.0;
#pragma unroll
    for (int k = 0; k < QQ; k++) S.Sxv[k] = 0.0;
#pragma unroll
    for (int k = 0; k < PP; k++) { S.Tx1u[k] = 0.0; S.Tux[k] = 0.0; }
    for (int t = 0; t < T; t++) {
        const double x = X[t], vv = V[t];
        const double term = x * x + vv;
        if (isfinite(y[t])) {
            S.Syx = fma(y[t], x, S.Syx);
            S.Sxx += term;
#pragma unroll
            for (int k = 0; k < QQ; k++) S.Sxv[k] = fma(x, v[t * QQ + k], S.Sxv[k]);
        }
        if (t < T - 1) {
            S.Txx += term;
            const double x1 = X[t + 1];
            S.Tx1x += x1 * x + V[t + 1] * J[t];
#pragma unroll
            for (int k = 0; k < PP; k++) {
                const double ut = u[t * PP + k];
                S.Tx1u[k] = fma(x1, ut, S.Tx1u[k]);
                S.Tux[k] = fma(ut, x, S.Tux[k]);
            }
        }
        if (t > 0) S.Tx1x1 += term;
    }
    S.X0 = X[0];
    S.V0 = V[0];
    Theta<PP, QQ> th;
#pragma unroll
    for (int k = 0; k < PP; k++) th.B[k] = 0.0;
#pragma unroll
    for (int k = 0; k < QQ; k++) th.D[k] = 0.0;
    double *out = prm.theta_out + (long)cell * P;
    if (sc->status != 0) {
        for (int k = 0; k < P; k++) out[k] = NAN;
        prm.status[cell] = 2;
        return;
    }
    mstep_update(th, S, sc, T);
    store_theta(th, out, prm.p, prm.q);
    prm.status[cell] = 0;
}

// ---------------------------------------------------------------------------------------
// launchers (dispatch on padded sizes)
// ---------------------------------------------------------------------------------------
#define DISPATCH_PQ(PPv, QQv, CALL)                                                     \
    switch ((PPv) * 16 + (QQv)) {                                                       \
        case 1 * 16 + 1: { constexpr int PP = 1, QQ = 1; CALL; } break;                 \
        case 1 * 16 + 2: { constexpr int PP = 1, QQ = 2; CALL; } break;                 \
        case 1 * 16 + 4: { constexpr int PP = 1, QQ = 4; CALL; } break;                 \
        case 1 * 16 + 8: { constexpr int PP = 1, QQ = 8; CALL; } break;                 \
        case 2 * 16 + 1: { constexpr int PP = 2, QQ = 1; CALL; } break;                 \
        case 2 * 16 + 2: { constexpr int PP = 2, QQ = 2; CALL; } break;                 \
        case 2 * 16 + 4: { constexpr int PP = 2, QQ = 4; CALL; } break;                 \
        case 2 * 16 + 8: { constexpr int PP = 2, QQ = 8; CALL; } break;                 \
        case 4 * 16 + 1: { constexpr int PP = 4, QQ = 1; CALL; } break;                 \
        case 4 * 16 + 2: { constexpr int PP = 4, QQ = 2; CALL; } break;                 \
        case 4 * 16 + 4: { constexpr int PP = 4, QQ = 4; CALL; } break;                 \
        case 4 * 16 + 8: { constexpr int PP = 4, QQ = 8; CALL; } break;                 \
        case 8 * 16 + 1: { constexpr int PP = 8, QQ = 1; CALL; } break;                 \
        case 8 * 16 + 2: { constexpr int PP = 8, QQ = 2; CALL; } break;                 \
        case 8 * 16 + 4: { constexpr int PP = 8, QQ = 4; CALL; } break;                 \
        case 8 * 16 + 8: { constexpr int PP = 8, QQ = 8; CALL; } break;                 \
        default: return hipErrorInvalidValue;                                           \
    }

hipError_t launch_series_prep(const PrepParams &prm, int n_series, hipStream_t stream) {
    hipLaunchKernelGGL(series_prep_kernel, dim3(n_series), dim3(256), 0, stream, prm);
    return hipGetLastError();
}

hipError_t launch_em_serial(const EmParams &prm, int PPv, int QQv, int n_blocks,
                            hipStream_t stream) {
    const size_t lds = (size_t)prm.T * (1 + PPv + QQv) * sizeof(double);
    const bool stage = lds <= 64 * 1024;
    DISPATCH_PQ(PPv, QQv, {
        if (stage)
            hipLaunchKernelGGL((em_serial_kernel<PP, QQ, true>), dim3(n_blocks), dim3(64), lds,
                               stream, prm);
        else
            hipLaunchKernelGGL((em_serial_kernel<PP, QQ, false>), dim3(n_blocks), dim3(64), 0,
                               stream, prm);
    });
    return hipGetLastError();
}

hipError_t launch_smooth(const SmoothParams &prm, int PPv, int QQv, hipStream_t stream) {
    const int nb = (prm.n_cells + 63) / 64;
    DISPATCH_PQ(PPv, QQv, {
        hipLaunchKernelGGL((smooth_kernel<PP, QQ>), dim3(nb), dim3(64), 0, stream, prm);
    });
    return hipGetLastError();
}

hipError_t launch_mstep(const SmoothParams &prm, int PPv, int QQv, hipStream_t stream) {
    const int nb = (prm.n_cells + 63) / 64;
    DISPATCH_PQ(PPv, QQv, {
        hipLaunchKernelGGL((mstep_kernel<PP, QQ>), dim3(nb), dim3(64), 0, stream, prm);
    });
    return hipGetLastError();
}
