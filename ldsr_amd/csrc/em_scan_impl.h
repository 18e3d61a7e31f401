// em_scan_impl.h -- wave-per-cell, parallel-in-time EM kernel (the fast path: T <= 2048, p, q <= 8).
//
// One 64-lane wavefront owns one (series, restart) cell for the whole EM loop.  nl = ceil(T/L)
// lanes are active; lane l owns L (the first rp lanes) or L-1 consecutive time steps, so only
// the last step of a chunk is predicated.  All per-step state lives in that lane's registers
// and nothing but the final theta / lik / n_iter / status ever goes to HBM.  The series
// (y, u, v) is staged once per workgroup into LDS in a chunk-transposed layout [j][lane] so
// that the 64 lanes of a wave read consecutive 8-byte words (conflict-free ds_read_b64).
//
// The reference recursions (/root/reference/src/EM.cpp:70-104) are strictly sequential in t.
// They are compositions of associative maps, so each E-step is done in three phases per
// direction:
//
//  forward (:70-90)   The joint filter state is carried in projective coordinates
//                     (n, d, xt) with Vp = n/d, Xp = xt/d.  One time step is LINEAR in them:
//                         n'  = (A^2 + Q c^2/R) n + Q d
//                         d'  =        (c^2/R) n +   d
//                         xt' = ((A c/R) e + bu c^2/R) n + bu d + A xt
//                     (the matrix of the filter step divided by R: projective coordinates
//                     are scale free) with c = C on observed steps and 0 on missing ones,
//                     e = y - D v, bu = B u.  (F1) every lane multiplies its step matrices;
//                     (scan) a 64-lane Kogge-Stone scan done with DPP row shifts / broadcasts
//                     composes them; (F2) every lane re-runs its steps serially from its exact
//                     entry state with the reference's own expressions and keeps J_t,
//                     g_t = Xu_t - J_t Xp_{t+1}, h_t = Vu_t - J_t^2 Vp_{t+1} in registers.
//  backward (:94-104) Xs_t = J_t Xs_{t+1} + g_t, Vs_t = J_t^2 Vs_{t+1} + h_t are affine maps:
//                     (B1) compose per lane, (scan) reverse scan by DPP + three readlanes,
//                     (B2) serial re-run from the exact entry value, then all M-step sums
//                     (:151-193) in an independent pass.
//  then ONE wave all-reduce of every sum (M-step and likelihood), the stop rule (:272) and the
//  closed-form M-step (ldsr_device.h) redundantly in every lane.
//
// Chunks longer than 16 steps keep J/g/h for their second half only and re-run the first half's
// forward recursion before its backward sweep (register budget: two waves per SIMD).
//
// Reassociation changes results at the 1e-12 level or below (measured against the oracle:
// tools/parity_report.py); iteration counts are identical.
#pragma once
#include "ldsr_device.h"

__device__ __forceinline__ double readlane_d(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double uniform_d(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

// All-reduce N independent sums with ONE dependent chain of 6 cross-lane rounds: every round
// issues all N exchanges before the N adds (a per-value butterfly would serialise 6*N LDS
// round-trips).  The summation tree is fixed, so results are run-to-run deterministic and
// identical in every lane (fp add is commutative).
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&x)[N]) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        double t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = __shfl_xor(x[i], d, 64);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += t[i];
    }
}

// Structured 3x3 step / composite matrix [[m00 m01 0],[m10 m11 0],[m20 m21 m22]].
struct PMat {
    double m00, m01, m10, m11, m20, m21, m22;
};

// r = a * b  (apply b first, then a)
__device__ __forceinline__ PMat pmul(const PMat &a, const PMat &b) {
    PMat r;
    r.m00 = fma(a.m00, b.m00, a.m01 * b.m10);
    r.m01 = fma(a.m00, b.m01, a.m01 * b.m11);
    r.m10 = fma(a.m10, b.m00, a.m11 * b.m10);
    r.m11 = fma(a.m10, b.m01, a.m11 * b.m11);
    r.m20 = fma(a.m20, b.m00, fma(a.m21, b.m10, a.m22 * b.m20));
    r.m21 = fma(a.m20, b.m01, fma(a.m21, b.m11, a.m22 * b.m21));
    r.m22 = a.m22 * b.m22;
    return r;
}

// r = S * b for a step matrix S = [[a00 Q 0],[g 1 0],[s20 bu A]]  (13 flops)
__device__ __forceinline__ PMat pstep(double a00, double Q, double g, double s20, double bu, double A,
                                      const PMat &b) {
    PMat r;
    r.m00 = fma(a00, b.m00, Q * b.m10);
    r.m01 = fma(a00, b.m01, Q * b.m11);
    r.m10 = fma(g, b.m00, b.m10);
    r.m11 = fma(g, b.m01, b.m11);
    r.m20 = fma(s20, b.m00, fma(bu, b.m10, A * b.m20));
    r.m21 = fma(s20, b.m01, fma(bu, b.m11, A * b.m21));
    r.m22 = A * b.m22;
    return r;
}

// Exact power-of-two rescale so that the 2x2 block has max magnitude in [1,2).
__device__ __forceinline__ void prenorm(PMat &m) {
    const double mx = fmax(fmax(fabs(m.m00), fabs(m.m01)), fmax(fabs(m.m10), fabs(m.m11)));
    const int e = 1 - __builtin_amdgcn_frexp_exp(mx);
    m.m00 = __builtin_amdgcn_ldexp(m.m00, e);
    m.m01 = __builtin_amdgcn_ldexp(m.m01, e);
    m.m10 = __builtin_amdgcn_ldexp(m.m10, e);
    m.m11 = __builtin_amdgcn_ldexp(m.m11, e);
    m.m20 = __builtin_amdgcn_ldexp(m.m20, e);
    m.m21 = __builtin_amdgcn_ldexp(m.m21, e);
    m.m22 = __builtin_amdgcn_ldexp(m.m22, e);
}

// Cross-lane move of a double by DPP (no LDS round-trip).  Lanes whose DPP source does not
// exist (outside the 16-lane row, or a row excluded by RM) receive `old`; passing the identity
// element there makes the scan steps unconditional (verified on gfx950 by tools/dpp_probe.hip).
template <int CTRL, int RM>
__device__ __forceinline__ double dppd(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, RM, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, RM, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// partner matrix of a scan round; identity where the partner lane does not exist
template <int CTRL, int RM>
__device__ __forceinline__ PMat pdpp(const PMat &m) {
    PMat r;
    r.m00 = dppd<CTRL, RM>(1.0, m.m00);
    r.m01 = dppd<CTRL, RM>(0.0, m.m01);
    r.m10 = dppd<CTRL, RM>(0.0, m.m10);
    r.m11 = dppd<CTRL, RM>(1.0, m.m11);
    r.m20 = dppd<CTRL, RM>(0.0, m.m20);
    r.m21 = dppd<CTRL, RM>(0.0, m.m21);
    r.m22 = dppd<CTRL, RM>(1.0, m.m22);
    return r;
}

// Lane <-> time mapping.  A cell uses nl = ceil(T/L) lanes; the first rp lanes own L
// consecutive steps and lanes rp..nl-1 own L-1 (T = nl*(L-1) + rp), so steps 0..L-2 of every
// active lane are real and only step L-1 is predicated: the unrolled loops are straight-line
// code.  Lane l starts at t0 = l*(L-1) + min(l, rp).  Requires L*(L-1) <= T <= 64*L (see
// scan_L_for in kernels_scan.hip).


// DENSE = every y_t of the series is observed: the per-step "observed ? a : b" selects vanish.
// GIMG = the series is read from the prepared time-major arrays in global memory (ys/us/vs are
// then this lane's own row pointers) instead of the chunk-transposed LDS image.
template <int PP, int QQ, int L, bool DENSE, bool GIMG>
__device__ __forceinline__ void em_scan_cell(const EmParams &prm, const double *ys,
                                             const double *us, const double *vs, int s, int cell,
                                             int lane, int nl, int rp);

// QUEUE = waves pull cells from the per-series work queue (cells converge at different
// iterations); !QUEUE = wave w of block b owns cell c0 + w (every cell runs exactly niter
// iterations, i.e. tol == 0: nothing to balance, and the queue loop costs ~6 % in spill code).
template <int PP, int QQ, int L, bool QUEUE, bool GIMG>
__global__ __launch_bounds__(512) void em_scan_kernel(EmParams prm) {
    extern __shared__ double smem[];
    // LDS image of the series, chunk-transposed: element (j, lane) of y at ys[j*64 + lane]
    double *ys, *us, *vs;

    const int b = blockIdx.x;
    const int s = prm.blk_series[b];
    const int c0 = prm.blk_cell0[b], nc = prm.blk_ncell[b];   // QUEUE: the series' cells; else the block's
    const int T = prm.T;
    const int lane = threadIdx.x & 63;
    const int nl = (T + L - 1) / L;          // active lanes
    const int rp = T - nl * (L - 1);         // lanes < rp own L steps, the others L-1
    if constexpr (GIMG) {
        // no LDS image: every lane reads its own rows of the prepared arrays (clamped for the
        // idle lanes, which never use them)
        const int t0 = min(lane, nl - 1) * (L - 1) + min(min(lane, nl - 1), rp);
        ys = const_cast<double *>(prm.yz) + (long)s * T + t0;
        us = const_cast<double *>(prm.up) + (long)s * prm.u_stride + (long)t0 * PP;
        vs = const_cast<double *>(prm.vp) + (long)s * prm.v_stride + (long)t0 * QQ;
    } else {
        ys = smem;                  // [L][64]       y, 0 where missing / unused
        us = ys + 64 * L;           // [L][PP][64]   u_t, zero for t = T-1
        vs = us + 64 * L * PP;      // [L][QQ][64]   v_t
        const double *gy = prm.yp + (long)s * T;
        const double *gu = prm.up + (long)s * prm.u_stride;
        const double *gv = prm.vp + (long)s * prm.v_stride;
        for (int i = threadIdx.x; i < 64 * L; i += blockDim.x) {
            const int j = i >> 6, l = i & 63, t = l * (L - 1) + min(l, rp) + j;
            const bool ok = l < nl && (j < L - 1 || l < rp);
            double yv = ok ? gy[t] : 0.0;
            ys[i] = isfinite(yv) ? yv : 0.0;
        }
        for (int i = threadIdx.x; i < 64 * L * PP; i += blockDim.x) {
            const int l = i & 63, jk = i >> 6, j = jk / PP, k = jk - j * PP;
            const int t = l * (L - 1) + min(l, rp) + j;
            const bool ok = l < nl && (j < L - 1 || l < rp);
            us[i] = ok ? gu[(long)t * PP + k] : 0.0;
        }
        for (int i = threadIdx.x; i < 64 * L * QQ; i += blockDim.x) {
            const int l = i & 63, jk = i >> 6, j = jk / QQ, k = jk - j * QQ;
            const int t = l * (L - 1) + min(l, rp) + j;
            const bool ok = l < nl && (j < L - 1 || l < rp);
            vs[i] = ok ? gv[(long)t * QQ + k] : 0.0;
        }
    }
    if constexpr (!GIMG) __syncthreads();
    const bool dense = prm.sc[s].n_obs == T;
    if constexpr (!QUEUE) {
        const int wave = threadIdx.x >> 6;
        if (wave >= nc) return;   // whole wave leaves; no barrier follows
        if (dense)
            em_scan_cell<PP, QQ, L, true, GIMG>(prm, ys, us, vs, s, c0 + wave, lane, nl, rp);
        else
            em_scan_cell<PP, QQ, L, false, GIMG>(prm, ys, us, vs, s, c0 + wave, lane, nl, rp);
    } else {
        // Work queue: every wave pulls cells of this series until the counter passes the
        // series' range (c0 .. c0+nc).  A wave whose cell converges early takes the next one
        // instead of idling, which keeps two waves per SIMD busy when iteration counts differ a
        // lot.  The counter only grows and the loop is bounded, so every wave reaches the exit;
        // which wave computes a cell does not change its result.
        for (int pulls = 0; pulls <= nc; pulls++) {
            int k = 0;
            if (lane == 0) k = atomicAdd(prm.queue + s, 1);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= nc) break;
            if (dense)
                em_scan_cell<PP, QQ, L, true, GIMG>(prm, ys, us, vs, s, c0 + k, lane, nl, rp);
            else
                em_scan_cell<PP, QQ, L, false, GIMG>(prm, ys, us, vs, s, c0 + k, lane, nl, rp);
        }
    }
}

template <int PP, int QQ, int L, bool DENSE, bool GIMG>
__device__ __forceinline__ void em_scan_cell(const EmParams &prm, const double *ys,
                                             const double *us, const double *vs, int s, int cell,
                                             int lane, int nl, int rp) {
    // element (step j, row k) of this lane's chunk of y / u / v
    auto Yat = [&](int j) { return GIMG ? ys[j] : ys[j * 64 + lane]; };
    auto Uat = [&](int j, int k) { return GIMG ? us[j * PP + k] : us[(j * PP + k) * 64 + lane]; };
    auto Vat = [&](int j, int k) { return GIMG ? vs[j * QQ + k] : vs[(j * QQ + k) * 64 + lane]; };
    const int T = prm.T;
    const int P = 6 + prm.p + prm.q;
    const SeriesConst *__restrict__ sc = prm.sc + s;
    const int n_obs = sc->n_obs;
    const bool act = lane < nl;      // this lane owns time steps
    const bool tail = lane < rp;     // ... and its chunk has the L-th step
    const int lastLane = nl - 1;     // owner of step T-1

    // observation mask of this lane's chunk
    unsigned obsmask = 0;
    if (!DENSE && act) {
        const double *gy = prm.yp + (long)s * T;
        const int t0 = lane * (L - 1) + min(lane, rp);
#pragma unroll
        for (int j = 0; j < L; j++) {
            const double yv = (j < L - 1 || tail) ? gy[t0 + j] : NAN;
            if (isfinite(yv)) obsmask |= (1u << j);
        }
    }

    Theta<PP, QQ> th;
    load_theta(th, prm.theta0 + (long)cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
    if (sc->status != 0) {
        if (lane == 0) {
            for (int k = 0; k < P; k++) prm.theta[(long)cell * P + k] = NAN;
            prm.lik[cell] = NAN;
            prm.n_iter[cell] = 0;
            prm.status[cell] = 2;
            if (prm.liks && prm.liks_nanfill)
                for (int i = 0; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
        }
        return;
    }

    double lik = NAN, lik1 = NAN, lik2 = NAN;
    int it = 0;
    // Per-step (J_t, g_t, h_t) kept in registers between the forward and the backward sweep.
    // Chunks longer than 16 steps (T > 1024) would need more than 256 VGPRs and drop to one
    // wave per SIMD, so for them only the second half [HS, L) is kept; the reverse composite is
    // accumulated during the forward sweep and the first half's forward recursion is re-run
    // just before its backward sweep (+~20 % flops, twice the occupancy).
    constexpr int HS = (L > 16) ? L / 2 : 0;   // L in {20, 24, 28, 32}
    constexpr int NS = L - HS;
    double Jv[NS], gv_[NS], hv[NS];

    for (;;) {
        const double A = th.A, C = th.C, Q = th.Q, R = th.R;
        const double A2 = A * A, C2 = C * C;
        const double rR = fast_rcp(R);
        const double C2R = C2 * rR, ACR = A * C * rR, alpha = fma(Q, C2R, A2);

        // e_t = y_t - D v_t (innovation minus C Xp) and bu_t = B u_t of step j of this lane, from LDS
        auto e_at = [&](int j) {
            double e = Yat(j);
#pragma unroll
            for (int k = 0; k < QQ; k++) e = fma(-th.D[k], Vat(j, k), e);
            return e;
        };
        auto bu_at = [&](int j) {
            double bu = 0.0;
#pragma unroll
            for (int k = 0; k < PP; k++) bu = fma(th.B[k], Uat(j, k), bu);
            return bu;
        };

        // ------------------------------------------------ F1: compose this lane's step matrices
        PMat M;
        M.m00 = 1.0; M.m01 = 0.0; M.m10 = 0.0; M.m11 = 1.0; M.m20 = 0.0; M.m21 = 0.0; M.m22 = 1.0;
        auto f1 = [&](int j) {
            const bool o = DENSE || ((obsmask >> j) & 1u);
            const double e = e_at(j), bu = bu_at(j);
            const double a00 = o ? alpha : A2;
            const double g = o ? C2R : 0.0;
            const double s20 = o ? fma(bu, C2R, ACR * e) : 0.0;
            if (j == 0) {
                M.m00 = a00; M.m01 = Q; M.m10 = g; M.m11 = 1.0; M.m20 = s20; M.m21 = bu; M.m22 = A;
            } else {
                M = pstep(a00, Q, g, s20, bu, A, M);
            }
            if ((j & 15) == 15 && j < L - 2) prenorm(M);
        };
        if (act) {
            if (L <= 16) {
#pragma unroll
                for (int j = 0; j < L - 1; j++) f1(j);
            } else {           // no register arrays here: keep long chunks rolled (code size, VGPRs)
                f1(0);
#pragma unroll 4
                for (int j = 1; j < L - 1; j++) f1(j);
            }
            if (tail) f1(L - 1);
            prenorm(M);
        }

        // ------------------------------------------------ forward scan (inclusive, by lane)
        // rows of 16 lanes: Kogge-Stone by DPP row shifts (identity outside the row) ...
        M = pmul(M, pdpp<DPP_ROW_SHR(1), 0xF>(M));
        M = pmul(M, pdpp<DPP_ROW_SHR(2), 0xF>(M));
        M = pmul(M, pdpp<DPP_ROW_SHR(4), 0xF>(M));
        prenorm(M);
        M = pmul(M, pdpp<DPP_ROW_SHR(8), 0xF>(M));
        // ... then row totals: lane 15 -> row 1, lane 47 -> row 3
        M = pmul(M, pdpp<DPP_ROW_BCAST15, 0xA>(M));
        // M composes lanes 0..l (rows 0,1) or 32..l (rows 2,3).  State after this lane's chunk:
        // rows 0,1 apply M to the initial state, rows 2,3 to the state of lane 31 (a matrix-vector
        // product instead of a sixth matrix-matrix round).
        double n_e = fma(M.m00, th.V1, M.m01);
        double d_e = fma(M.m10, th.V1, M.m11);
        double x_e = fma(M.m20, th.V1, fma(M.m22, th.mu1, M.m21));
        {
            const double n_b = dppd<DPP_ROW_BCAST31, 0xC>(th.V1, n_e);
            const double d_b = dppd<DPP_ROW_BCAST31, 0xC>(1.0, d_e);
            const double x_b = dppd<DPP_ROW_BCAST31, 0xC>(th.mu1, x_e);
            n_e = fma(M.m00, n_b, M.m01 * d_b);
            d_e = fma(M.m10, n_b, M.m11 * d_b);
            x_e = fma(M.m20, n_b, fma(M.m21, d_b, M.m22 * x_b));
        }
        // shift by one lane: the entry state of lane l is the exit state of lane l-1; lane 0
        // gets the initial state (V1, 1, mu1)
        n_e = dppd<DPP_WAVE_SHR1, 0xF>(th.V1, n_e);
        d_e = dppd<DPP_WAVE_SHR1, 0xF>(1.0, d_e);
        x_e = dppd<DPP_WAVE_SHR1, 0xF>(th.mu1, x_e);
        double Xp, Vp;
        {
            const double rd = fast_rcp(d_e);
            Vp = n_e * rd;
            Xp = x_e * rd;
        }

        // ------------------------------------------------ F2: serial re-run from the exact entry
        double likq = 0.0, sprod = 1.0, Xu = 0.0, Vu = 0.0;
        int sneg = 0;   // OR of the sign words of every observed Sigma_t
        double sg = fma(C2, Vp, R);     // Sigma_t of the current step (src/EM.cpp:119)
        double r0 = fast_rcp(sg);
        const double Xp0 = Xp, Vp0 = Vp, sg0 = sg, r00 = r0;   // entry state (re-run of [0, HS))
        double Pi = 1.0, G = 0.0, H = 0.0;                     // reverse composite (HS > 0 only)
        auto f2 = [&](int j) {
            const bool o = DENSE || ((obsmask >> j) & 1u);
            const double e = e_at(j), bu = bu_at(j);
            const double r = o ? r0 : 0.0;             // 1/Sigma_t; 0 = "no update" (:82-84)
            const double sl = o ? sg : 1.0;
            sprod *= sl;
            sneg |= __double2hiint(sl);
            const double w = Vp * r;
            const double K = C * w;                    // :86
            if (DENSE) Vu = R * w;                     // (1 - K C) Vp = R Vp / Sigma   :88
            else Vu = fma(-(C2 * w), Vp, Vp);
            const double dl = fma(-C, Xp, e);          // y - Yp
            Xu = fma(K, dl, Xp);                       // :87
            likq = fma(dl * r, dl, likq);              // delta/Sigma*delta  :122
            const double Vp1 = fma(A2, Vu, Q);         // :76
            const double Xp1 = fma(A, Xu, bu);         // :74
            sg = fma(C2, Vp1, R);
            const double z = fast_rcp(sg * Vp1);       // one reciprocal for 1/Vp1 and 1/Sigma_{t+1}
            const double rp1 = sg * z;
            r0 = Vp1 * z;
            const double AVu = A * Vu;
            double J = AVu * rp1;                      // :100
            double g = fma(-J, Xp1, Xu);
            double h = fma(-J, AVu, Vu);
            if (j >= L - 2) {
                // Step T-1 starts the backward recursion: Xs_{T-1} = Xu_{T-1}, Vs_{T-1} = Vu_{T-1}
                // (:94-95).  Expressed as J = 0, g = Xu, h = Vu with a zero terminal value,
                // which also makes the (T-1, T) term of every pair sum vanish.
                const bool fin = (lane == lastLane) && (j == (tail ? L - 1 : L - 2));
                J = fin ? 0.0 : J;
                g = fin ? Xu : g;
                h = fin ? Vu : h;
            }
            if (j >= HS) { Jv[j - HS] = J; gv_[j - HS] = g; hv[j - HS] = h; }
            if (HS > 0) {      // steps arrive in time order: (Pi,G,H) o step_j
                G = fma(Pi, g, G);
                H = fma(Pi * Pi, h, H);
                Pi *= J;
            }
            Xp = Xp1;
            Vp = Vp1;
            if (L > 16) __builtin_amdgcn_sched_barrier(0);
        };
        if (act) {
#pragma unroll
            for (int j = 0; j < L - 1; j++) f2(j);
            if (tail) f2(L - 1);
        }
        const double termLast = readlane_d(fma(Xu, Xu, Vu), lastLane);   // Xs^2 + Vs at T-1

        // The two likelihood sums ride along with the M-step sums in ONE wave reduction at the end
        // of the iteration; the backward sweep of the final iteration is therefore redundant
        // (1 of n_iter sweeps) but every iteration saves 6 dependent cross-lane rounds.
        const double lsp = log(sprod);

        // ------------------------------------------------ B1: compose the reverse affine maps
        auto b1 = [&](int j) {
            const double J = Jv[j];
            G = fma(J, G, gv_[j]);
            H = fma(J * J, H, hv[j]);
            Pi *= J;
        };
        if (HS == 0 && act) {
            if (tail) b1(L - 1);
#pragma unroll
            for (int j = L - 2; j >= 0; j--) b1(j);
        }
        // reverse inclusive scan: lane l composes its map after those of lanes > l.
        // Within rows by DPP row shifts (identity outside the row) ...
#define RSCAN_ROUND(n)                                                     \
        {                                                                  \
            const double Pb = dppd<DPP_ROW_SHL(n), 0xF>(1.0, Pi);          \
            const double Gb = dppd<DPP_ROW_SHL(n), 0xF>(0.0, G);           \
            const double Hb = dppd<DPP_ROW_SHL(n), 0xF>(0.0, H);           \
            G = fma(Pi, Gb, G);                                            \
            H = fma(Pi * Pi, Hb, H);                                       \
            Pi *= Pb;                                                      \
        }
        RSCAN_ROUND(1) RSCAN_ROUND(2) RSCAN_ROUND(4) RSCAN_ROUND(8)
#undef RSCAN_ROUND
        // ... then across rows: lanes 16, 32, 48 hold the composites T1, T2, T3 of rows 1..3;
        // every lane applies the composite of all rows after its own (uniform values)
        {
            const double G3 = readlane_d(G, 48), H3 = readlane_d(H, 48);
            const double P2 = readlane_d(Pi, 32), G2 = readlane_d(G, 32), H2 = readlane_d(H, 32);
            const double P1 = readlane_d(Pi, 16), G1 = readlane_d(G, 16), H1 = readlane_d(H, 16);
            const double G23 = fma(P2, G3, G2), H23 = fma(P2 * P2, H3, H2);   // T2 o T3
            const double G123 = fma(P1, G23, G1), H123 = fma(P1 * P1, H23, H1);              // T1 o T2 o T3
            const int row = lane >> 4;
            const double Gs = row == 0 ? G123 : row == 1 ? G23 : row == 2 ? G3 : 0.0;
            const double Hs = row == 0 ? H123 : row == 1 ? H23 : row == 2 ? H3 : 0.0;
            G = fma(Pi, Gs, G);
            H = fma(Pi * Pi, Hs, H);
        }
        // (G, H) = (Xs, Vs) at the first step of this lane's chunk (terminal value is zero);
        // the entry for lane l is lane l+1's value, zero beyond the last active lane (inactive
        // lanes hold identity maps, so their G = H = 0)
        double Xn = dppd<DPP_WAVE_SHL1, 0xF>(0.0, G);
        double Vn = dppd<DPP_WAVE_SHL1, 0xF>(0.0, H);

        // ------------------------------------------------ B2: serial reverse re-run + M-step sums
        double aSyx = 0.0, aSxx = 0.0, aTx1x = 0.0, aPall = 0.0, term = 0.0;
        double aSxv[QQ], aTx1u[PP], aTux[PP];
#pragma unroll
        for (int k = 0; k < QQ; k++) aSxv[k] = 0.0;
#pragma unroll
        for (int k = 0; k < PP; k++) { aTx1u[k] = 0.0; aTux[k] = 0.0; }
        // One segment = the stored steps [HS, L) (all steps when HS == 0), then for HS > 0 the
        // re-run steps [0, HS).  pass 1 (registers only): the serial recurrence, Xs_t / Vs_t
        // overwrite g_t / h_t;  pass 2: every M-step sum, no dependence between steps so the
        // LDS reads batch freely.
        double XnE = Xn, VnE = Vn;             // Xs, Vs just after the current segment
        auto b2a = [&](int i) {                // i = index into the stored arrays
            const double J = Jv[i];
            const double Xs = fma(J, Xn, gv_[i]);        // :101
            const double Vs = fma(J * J, Vn, hv[i]);     // :102
            gv_[i] = Xs;
            hv[i] = Vs;
            Xn = Xs;
            Vn = Vs;
        };
        auto b2b = [&](int j, int i, bool top) {   // j = step in the chunk, i = storage index
            const bool o = DENSE || ((obsmask >> j) & 1u);
            const double J = Jv[i], Xs = gv_[i], Vs = hv[i];
            const double Xnx = top ? XnE : gv_[top ? i : i + 1];
            const double Vnx = top ? VnE : hv[top ? i : i + 1];
            aTx1x = fma(Xnx, Xs, fma(Vnx, J, aTx1x));   // :180  (zero at t = T-1)
#pragma unroll
            for (int k = 0; k < PP; k++) {
                const double ut = Uat(j, k);                      // zero at t = T-1
                aTx1u[k] = fma(Xnx, ut, aTx1u[k]);                // :190
                aTux[k] = fma(ut, Xs, aTux[k]);                   // :191
            }
            term = fma(Xs, Xs, Vs);
            aPall += term;                                        // :181,:183
            const double xo = o ? Xs : 0.0;
            aSyx = fma(Yat(j), xo, aSyx);                         // :151
            if (!DENSE) aSxx += o ? term : 0.0;                   // :152
#pragma unroll
            for (int k = 0; k < QQ; k++) aSxv[k] = fma(xo, Vat(j, k), aSxv[k]);  // :159
            if (L > 16 && (j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        };
        if (act) {
            if (tail) b2a(NS - 1);
            else { gv_[NS - 1] = XnE; hv[NS - 1] = VnE; }  // "next" of step L-2 for short chunks
#pragma unroll
            for (int i = NS - 2; i >= 0; i--) b2a(i);
            if (tail) b2b(L - 1, NS - 1, true);
#pragma unroll
            for (int i = NS - 2; i >= 0; i--) b2b(HS + i, i, false);
        }
        if (HS > 0) {
            // re-run the forward recursion of steps [0, HS) from the lane's entry state; the
            // likelihood terms of these steps were already accumulated in F2
            double Xq = Xp0, Vq = Vp0, sgq = sg0, rq = r00;
            auto f2r = [&](int j) {
                const bool o = DENSE || ((obsmask >> j) & 1u);
                const double e = e_at(j), bu = bu_at(j);
                const double r = o ? rq : 0.0;
                const double w = Vq * r;
                const double K = C * w;
                double Vuq;
                if (DENSE) Vuq = R * w;
                else Vuq = fma(-(C2 * w), Vq, Vq);
                const double dl = fma(-C, Xq, e);
                const double Xuq = fma(K, dl, Xq);
                const double Vp1 = fma(A2, Vuq, Q);
                const double Xp1 = fma(A, Xuq, bu);
                sgq = fma(C2, Vp1, R);
                const double z = fast_rcp(sgq * Vp1);
                const double rp1 = sgq * z;
                rq = Vp1 * z;
                const double AVu = A * Vuq;
                const double J = AVu * rp1;
                Jv[j] = J;
                gv_[j] = fma(-J, Xp1, Xuq);
                hv[j] = fma(-J, AVu, Vuq);
                Xq = Xp1;
                Vq = Vp1;
                __builtin_amdgcn_sched_barrier(0);   // keep later steps' LDS loads from being hoisted (VGPR pressure)
            };
            if (act) {
#pragma unroll
                for (int j = 0; j < HS; j++) f2r(j);
                XnE = Xn;                      // Xs, Vs at step HS: the entry of this segment
                VnE = Vn;
#pragma unroll
                for (int i = HS - 1; i >= 0; i--) b2a(i);
                b2b(HS - 1, HS - 1, true);
#pragma unroll
                for (int i = HS - 2; i >= 0; i--) b2b(i, i, false);
            }
        }
        const double Xs = Xn, Vs = Vn;         // Xs_t, Vs_t at the first step of the chunk
        Sums<PP, QQ> S;
        {
            constexpr int NR = 5 + (DENSE ? 0 : 1) + QQ + 2 * PP;
            double red[NR];
            red[0] = aSyx; red[1] = aTx1x; red[2] = aPall; red[3] = likq; red[4] = lsp;
            if (!DENSE) red[5] = aSxx;
            constexpr int o0 = 5 + (DENSE ? 0 : 1);
#pragma unroll
            for (int k = 0; k < QQ; k++) red[o0 + k] = aSxv[k];
#pragma unroll
            for (int k = 0; k < PP; k++) { red[o0 + QQ + k] = aTx1u[k]; red[o0 + QQ + PP + k] = aTux[k]; }
            wave_sum_n<NR>(red);
            S.Syx = red[0]; S.Tx1x = red[1];
            S.Sxx = DENSE ? red[2] : red[5];
#pragma unroll
            for (int k = 0; k < QQ; k++) S.Sxv[k] = red[o0 + k];
#pragma unroll
            for (int k = 0; k < PP; k++) { S.Tx1u[k] = red[o0 + QQ + k]; S.Tux[k] = red[o0 + QQ + PP + k]; }
            S.Txx = red[2] - termLast;                  // t = 0 .. T-2
            S.Tx1x1 = red[2] - readlane_d(term, 0);     // t = 1 .. T-1
            // likelihood (:113-124)
            lik2 = lik1;
            lik1 = lik;
            lik = (-0.5 * n_obs * LDSR_LOG_2PI - 0.5 * (red[3] + red[4])) / n_obs;
            if (__any(sneg < 0)) lik = NAN;   // log of a negative Sigma in the reference
        }
        if (prm.liks && lane == 0) prm.liks[(long)cell * prm.niter + it] = lik;
        it++;
        bool stop = it >= prm.niter;
        if (it >= 3 && fabs(lik - lik1) < prm.tol && fabs(lik1 - lik2) < prm.tol) stop = true;  // :272
        if (__builtin_amdgcn_readfirstlane((int)stop)) break;   // theta stays the one that produced this fit
        S.X0 = readlane_d(Xs, 0);                       // :218
        S.V0 = readlane_d(Vs, 0);                       // :219
        mstep_update<PP, QQ, true>(th, S, sc, T);
        // theta is wave-uniform by construction; say so to the compiler (SGPR residency)
        th.A = uniform_d(th.A); th.C = uniform_d(th.C); th.Q = uniform_d(th.Q);
        th.R = uniform_d(th.R); th.mu1 = uniform_d(th.mu1); th.V1 = uniform_d(th.V1);
#pragma unroll
        for (int k = 0; k < PP; k++) th.B[k] = uniform_d(th.B[k]);
#pragma unroll
        for (int k = 0; k < QQ; k++) th.D[k] = uniform_d(th.D[k]);
    }

    if (lane == 0) {
        store_theta(th, prm.theta + (long)cell * P, prm.p, prm.q);
        if (prm.liks && prm.liks_nanfill)
            for (int i = it; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
        prm.lik[cell] = lik;
        prm.n_iter[cell] = it;
        prm.status[cell] = isfinite(lik) ? 0 : 1;
    }
}

// waves per block so that a CU holds ~8 waves given the LDS image of one series
static inline int scan_wpb(int L, int PP, int QQ) {
    const size_t lds = (size_t)64 * L * (1 + PP + QQ) * sizeof(double);
    const int blocks_per_cu = (int)((160 * 1024) / lds);
    const int want = 8;
    int wpb = (want + blocks_per_cu - 1) / (blocks_per_cu > 0 ? blocks_per_cu : 1);
    if (wpb < 2) wpb = 2;
    if (L <= 16 && wpb < 4 && lds > 20 * 1024) wpb = 4;
    const int cap = 8;
    return wpb > cap ? cap : wpb;
}

template <int L>
hipError_t launch_em_scan_L(const EmParams &prm, int PPv, int QQv, int n_blocks, int wpb, bool queue,
                            hipStream_t stream);
