// em_pair_impl.h -- two or four cells per wavefront: the parallel-in-time EM kernel for series of
// 65 to 1024 steps with narrow inputs (LPC = 32 or 16 lanes per cell, T <= LPC L, L <= 32, padded
// p, q <= 4; kernels_scan.hip pair_plan() has the exact ranges).  The text below describes two
// cells per wave (LPC = 32); with LPC = 16 a cell is one 16-lane DPP row, both scans are the four
// row-shift rounds and every per-wave item is shared by four cells.
//
// Why.  In em_scan_kernel (one cell per 64-lane wave, em_scan_impl.h) half of the ~1500 VALU
// instructions of an EM iteration at T = 1000 do not depend on the chunk length: the two
// cross-lane scans, the all-reduce of the M-step sums, the closed-form M-step, the stop rule
// and the log of the likelihood's determinant product are executed once per wave whatever the
// wave holds.  Here a wave holds TWO cells of the same series, one per 32-lane half, each lane
// owning up to 32 consecutive time steps: the per-step work per cell is unchanged (twice the
// steps on half the lanes) while every per-wave item above is shared by two cells and both
// scans and the reduction lose one round.
//
// What makes it fit.  Three per-step values (J_t, g_t, h_t: em_scan_impl.h) must survive from
// the forward to the backward sweep; at L = 32 that is 96 doubles per lane -- 192 VGPRs, too
// many for two waves per SIMD.  J_t and g_t stay in registers; h_t (needed only by the variance
// recursion) goes through a per-wave LDS strip [step][lane] (one ds_write_b64 and one
// ds_read_b64 per step, conflict-free), which with the 32-lane series image fills the CU's 160 KiB
// exactly at eight waves.  Nothing but theta / lik / n_iter / status ever goes to HBM.
//
// Everything else -- projective step matrices, F1 / scan / F2, B1 / reverse scan / B2, the folded
// log-determinant, the fused reduction, mstep_update() -- is em_scan_impl.h's algorithm (same
// reference citations, /root/reference/src/EM.cpp:22-229,245-280), re-mapped to 32-lane halves:
// rows 0,1 of the wave are cell `a`, rows 2,3 are cell `b`; theta is per half (VGPRs).
// Cells are dealt statically (tol == 0) or pulled by each half from the per-series work queue
// (tol > 0: a half whose cell has converged takes the next one while the other half goes on).
#pragma once
#include "em_scan_impl.h"

// lanes per cell LPC = 32 (two cells per wave, T <= 1024) or 16 (FOUR cells per wave, one per DPP
// row, T <= 512): the series image has LPC virtual lanes
__host__ __device__ constexpr long pair_image_doubles(int L, int PP, int QQ, int LPC = 32) {
    return img_doubles(L, LPC, PP, QQ);
}
// LDS strip of one wave: h_t of steps 0 .. L-2 for 64 lanes (the predicated step L-1 keeps its
// h in a register)
__host__ __device__ constexpr long pair_strip_doubles(int L) { return (long)64 * (L - 1); }

// STEADY form (fully observed series, two cells per wave): the first L-1 steps of a series -- the
// chunk of lane 0 -- are also kept one step per lane (`tri`, [pair][lane][2]) for the transient
// block of em_pair_body
__host__ __device__ constexpr long pair_tri_doubles(int PP, int QQ, int LPC = 32) {
    return (long)LPC * 2 * scan_pairs(PP, QQ);
}
#ifndef LDSR_STEADY            // 0: every dense cell takes the generic sweeps (A/B builds)
#define LDSR_STEADY 1
#endif
#ifndef LDSR_STEADY_PF         // steady sweeps: prefetch distance of the LDS image reads, in steps
#define LDSR_STEADY_PF 4
#endif
#ifndef LDSR_STEADY_PF2        // ... and of the strip reads in F2
#define LDSR_STEADY_PF2 6
#endif
#ifndef LDSR_LEAD_CLOSED_VAR    // LEAD pass 1: multipliers and variance offset of a lane's steps in closed form
#define LDSR_LEAD_CLOSED_VAR 1
#endif
#ifndef LDSR_STEADY_SBMASK     // what may still cross the per-step scheduling barriers (0x2 VALU | 0x4 SALU)
#define LDSR_STEADY_SBMASK 0x6
#endif
#ifndef LDSR_STEADY_MIN_L      // shortest chunk whose L-1 transient steps usually reach the fixed point
#define LDSR_STEADY_MIN_L 24
#endif
// read-ahead rings of the generic sweeps (pair_generic_sweeps: SPF): chunks of up to 23 steps, narrow inputs
#ifndef LDSR_PAIR_SPF
#define LDSR_PAIR_SPF 1
#endif
#ifndef LDSR_QUAD_SPF_LONG
#define LDSR_QUAD_SPF_LONG 1
#endif
#ifndef LDSR_PAIR_SPF_MAXL      // (chunks of 24+ steps: the members with the steady form, whose allocation is left alone;
#define LDSR_PAIR_SPF_MAXL 23   //  four cells per wave spill there with the ring)
#endif
__host__ __device__ constexpr bool pair_spf(int PP, int QQ, int L, int LPC = 32) {
    // (four cells per wave have no steady form: their long chunks take the ring where it fits without spills, p + q <= 4)
    return LDSR_PAIR_SPF && PP + QQ <= 8 && (L <= LDSR_PAIR_SPF_MAXL || (LDSR_QUAD_SPF_LONG && LPC == 16 && PP + QQ <= 4));
}
// steps of the transient block: L-1 (the chunk of lane 0 without its predicated step)
__host__ __device__ constexpr int pair_steady_ntr(int L, int LPC) { return L - 1; }
__host__ __device__ constexpr bool pair_steady(int L, int LPC, int PP, int QQ) {
    if (!LDSR_STEADY) return false;
    return LPC == 32 && L >= LDSR_STEADY_MIN_L &&
           (pair_image_doubles(L, PP, QQ, LPC) + 8 * pair_strip_doubles(L) + pair_tri_doubles(PP, QQ, LPC)) * 8 <= 160 * 1024;
}

// Does the (L, LPC, PP, QQ) member leave room for a CU's waves: the series image and eight strips (wide
// inputs -- padded p or q = 8, LEAD forms only: four) within 160 KiB.  ONE rule for the plan
// (kernels_scan.hip pair_plan) and for what is compiled (em_pair_launch.inc).
__host__ __device__ constexpr bool pair_member_fits(int L, int LPC, int PP, int QQ) {
    const bool wide = PP > 4 || QQ > 4;
    return (pair_image_doubles(L, PP, QQ, LPC) + (wide ? 4 : 8) * pair_strip_doubles(L)) * 8 <= 160 * 1024;
}

__device__ __forceinline__ double shfl_d(double x, int src_lane) { return __shfl(x, src_lane, 64); }

// Cross-row step of the reverse scans.  After the four row-shift rounds lane l holds the
// composite (P, G, H) of lanes l .. end of its 16-lane row; every lane then applies the composite
// of all later rows of its cell (first lanes of those rows, read with v_readlane).  P is made
// the full product too (callers whose terminal value is not zero need it).
template <int LPC>
__device__ __forceinline__ void rscan_cross(double &P, double &G, double &H, int lane) {
    if constexpr (LPC == 32) {
        const double G1 = readlane_d(G, 16), H1 = readlane_d(H, 16), P1 = readlane_d(P, 16);
        const double G3 = readlane_d(G, 48), H3 = readlane_d(H, 48), P3 = readlane_d(P, 48);
        const int row = lane >> 4;
        const double Gs = row == 0 ? G1 : row == 2 ? G3 : 0.0;
        const double Hs = row == 0 ? H1 : row == 2 ? H3 : 0.0;
        const double Ps = row == 0 ? P1 : row == 2 ? P3 : 1.0;
        G = fma(P, Gs, G);
        H = fma(P * P, Hs, H);
        P *= Ps;
    }
}

// The closed-form lead's passes walk a lane's nj steps of the (whitened) u_t, [step][lane][PP] in LDS.  A rolled
// loop reads each trip's values and waits for them before it computes: with two waves per SIMD the other wave
// covers the LDS latency, a lone wave -- the drained tail of a run to convergence, or a reference-sized call --
// stands still once per trip (config 4's shape, 16 lone cells: 94 cycles per step of the second pass against 58
// with the SIMD shared).  Here the values of the NEXT four steps are in flight while four steps are computed
// (two register buffers, the loop unrolled over both: no copies); step(j, u) sees the steps in order, so
// results are bit-identical.  nA = rows of the lane's column (reads are clamped to it).  Padded p >= 4 keeps the
// plain loops: with one step per buffer the walk advances by two steps an iteration, the second pass's
// renormalisation every fourth step is no longer a constant of the unrolled copies, and the (4,4) / (4,8) LEAD
// kernels lose 45..50 % (same box: T = 813 (3,3) 8192 cells 1.55 -> 2.24 ms).
#ifndef LDSR_LEAD_PIPE
#define LDSR_LEAD_PIPE 1
#endif
#ifndef LDSR_LEAD_PIPE_MAXP
#define LDSR_LEAD_PIPE_MAXP 2
#endif
__host__ __device__ constexpr bool lead_pipe(int PP) { return LDSR_LEAD_PIPE && PP <= LDSR_LEAD_PIPE_MAXP; }
template <int PP, int LPC, typename F>
__device__ __forceinline__ void lead_walk(const double *lup, int nj, int nA, F &&step) {
    constexpr int G = PP >= 4 ? 1 : 4 / PP;              // steps per buffer: four doubles in flight per buffer (G >= 2: see lead_pipe)
    constexpr long ROW = (long)LPC * PP;                 // doubles between two steps of a lane
    double ua[G][PP], ub[G][PP];
    // rows r0 .. r0+G-1 at immediate offsets from a running pointer (main loop: all of them exist) ...
    auto rd = [&](const double *base, double (&w)[G][PP]) {
#pragma unroll
        for (int i = 0; i < G; i++)
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) w[i][p_] = base[i * ROW + p_];
    };
    // ... or clamped to the lane's last row (head and tail of the walk)
    auto rdc = [&](int j0, double (&w)[G][PP]) {
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int jj = min(j0 + i, nA - 1);
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) w[i][p_] = lup[jj * ROW + p_];
        }
    };
    if (nj <= 0) return;
    int j0 = 0;
    rdc(0, ua);
    const double *base = lup;
    for (; j0 + 3 * G <= nj; j0 += 2 * G, base += 2 * G * ROW) {
        rd(base + G * ROW, ub);
#pragma unroll
        for (int i = 0; i < G; i++) step(j0 + i, ua[i]);
        rd(base + 2 * G * ROW, ua);
#pragma unroll
        for (int i = 0; i < G; i++) step(j0 + G + i, ub[i]);
    }
    // fewer than 3 G steps left: ua holds the first G of them
    double uc[G][PP];
    rdc(j0 + G, ub);
    rdc(j0 + 2 * G, uc);
#pragma unroll
    for (int i = 0; i < G; i++)
        if (j0 + i < nj) step(j0 + i, ua[i]);
#pragma unroll
    for (int i = 0; i < G; i++)
        if (j0 + G + i < nj) step(j0 + G + i, ub[i]);
#pragma unroll
    for (int i = 0; i < G; i++)
        if (j0 + 2 * G + i < nj) step(j0 + 2 * G + i, uc[i]);
}

// Sums and likelihood terms of one E-step, as the sweeps leave them in every lane (reduced over the
// cell's lanes afterwards by em_pair_body)
template <int PP, int QQ>
struct PairSweepOut {
    double aSyx, aTx1x, aPall, aSxx, likq, lsp, tLv, X0v, V0v;
    double aSxv[QQ], aTx1u[PP], aTux[PP];
    int sneg;
};

// The generic sweeps of one E-step for the cell of this lane (any observation mask): F1 / forward
// scan / F2, reverse scan, B2 -- em_scan_impl.h's algorithm on LPC-lane cells, see the head of this
// file.  (x_in, v_in) is the state at the first step of the sweeps (mu1, V1; LEAD: the state at the
// tail's first step).  hs: this lane's column of the wave's h_t strip.
template <int PP, int QQ, int L, int LPC, bool DENSE>
__device__ __forceinline__ void pair_generic_sweeps(PairSweepOut<PP, QQ> &o, const Theta<PP, QQ> &th,
                                                    const double *ys, double *hs, unsigned obsmask,
                                                    int lane, int nl, int rp, double x_t1, double v_t1) {
    constexpr int KV = img_values(PP, QQ);
    const int vl = lane & (LPC - 1);
    auto val = [&](int j, int i) -> double { return ys[img_off(j, i, KV, LPC, L) + vl * 2]; };
    auto Yat = [&](int j) { return val(j, 0); };
    auto Uat = [&](int j, int k) { return val(j, 1 + k); };
    auto Vat = [&](int j, int k) { return val(j, 1 + PP + k); };
    // SPF (em_scan_impl.h scan_spf): short chunks read the image a step or two ahead of its use through a register
    // ring, and B2's first pass the h_t strip -- a lone wave (the drained tail of a run to convergence) otherwise
    // stands still for the LDS latency at every read
    constexpr bool SPF = pair_spf(PP, QQ, L, LPC) && (L <= LDSR_PAIR_SPF_MAXL || DENSE);   // (long chunks: +1.5 % on masked series, -3 .. -11 % on dense ones)
    constexpr int SPFD = scan_pairs(PP, QQ) <= 2 ? 2 : 1;
    constexpr int SPFN = SPFD + 1;
    constexpr int KH2 = 2 * (KV / 2);
    constexpr bool KODD = (KV & 1) != 0;
    struct StepRing {
        double w[SPFN][KH2 > 0 ? KH2 : 1];
        double o[2][2];
    };
    auto ring_rd = [&](double (&w)[KH2 > 0 ? KH2 : 1], double (&o)[2][2], int jn, bool with_odd) {
#pragma unroll
        for (int i = 0; i < KH2; i++) w[i] = val(jn, i);
        if (KODD && with_odd) {
            o[(jn >> 1) & 1][0] = val(jn & ~1, KV - 1);
            o[(jn >> 1) & 1][1] = val(jn | 1, KV - 1);
        }
    };
    const bool act = vl < nl, tail = vl < rp;
    const int lastLane = nl - 1;
    const double A = th.A, C = th.C, Q = th.Q, R = th.R;
    const double A2 = A * A, C2 = C * C;
    const double rR = fast_rcp(R);
    const double C2R = C2 * rR, ACR = A * C * rR, alpha = fma(Q, C2R, A2);
    auto e_at = [&](int j) {
        double e = Yat(j);
#pragma unroll
        for (int q_ = 0; q_ < QQ; q_++) e = fma(-th.D[q_], Vat(j, q_), e);
        return e;
    };
    auto bu_at = [&](int j) {
        double bu = 0.0;
#pragma unroll
        for (int p_ = 0; p_ < PP; p_++) bu = fma(th.B[p_], Uat(j, p_), bu);
        return bu;
    };
    auto e_of = [&](const double (&w)[KH2 > 0 ? KH2 : 1], const double (&o)[2][2], int j) {
        auto vv = [&](int i) { return i < KH2 ? w[i] : o[(j >> 1) & 1][j & 1]; };
        double e = vv(0);
#pragma unroll
        for (int q_ = 0; q_ < QQ; q_++) e = fma(-th.D[q_], vv(1 + PP + q_), e);
        return e;
    };
    auto bu_of = [&](const double (&w)[KH2 > 0 ? KH2 : 1], const double (&o)[2][2], int j) {
        auto vv = [&](int i) { return i < KH2 ? w[i] : o[(j >> 1) & 1][j & 1]; };
        double bu = 0.0;
#pragma unroll
        for (int p_ = 0; p_ < PP; p_++) bu = fma(th.B[p_], vv(1 + p_), bu);
        return bu;
    };
    double Jv[L], gv_[L];
    double hlast = 0.0;      // h of the predicated step L-1
    double likq = 0.0, lsp = 0.0, tLv = 0.0, X0v = 0.0, V0v = 0.0;
    int sneg = 0;
    double aSyx = 0.0, aTx1x = 0.0, aPall = 0.0, aSxx = 0.0;
    double aSxv[QQ], aTx1u[PP], aTux[PP];
#pragma unroll
    for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = 0.0;
#pragma unroll
    for (int p_ = 0; p_ < PP; p_++) { aTx1u[p_] = 0.0; aTux[p_] = 0.0; }

    // ------------------------------------------------ F1: compose this lane's step matrices;
    // e_t and B u_t are handed to F2 through the (not yet live) g_t / J_t slots
    PMat M;
    M.m00 = 1.0; M.m01 = 0.0; M.m10 = 0.0; M.m11 = 1.0; M.m20 = 0.0; M.m21 = 0.0; M.m22 = 1.0;
    if constexpr (DENSE) {
        // Every step has the same 2x2 block Bm = [[alpha, Q],[C2R, 1]]: the chunk's block is a
        // power of it (binary exponentiation) and only the third row needs the per-step
        // recursion, composed from the chunk's last step towards its first (em_scan_impl.h).
        // Chunks are up to 32 steps long here, so the step matrix is first scaled by an exact
        // power of two c = 2^-k with max(alpha, 1) c in [0.5, 1) (projective coordinates are
        // scale free; alpha^31 alone could leave the double range when R is tiny).  Bm is
        // positive with alpha >= Q C2R, so its Perron root lies in [max(alpha, 1), 2 max(alpha, 1)]
        // and the powers of Bm' neither overflow nor underflow.  The row (a, b, r) is carried as
        // (a, b, r c):   a <- a alpha' + b C2R' + (r c) s20_j,   b <- b c + a Q' + (r c) bu_j,
        // r c <- (r c) A'.
        const double mx = fmax(alpha, 1.0);
        const int ke = -__builtin_amdgcn_frexp_exp(mx);
        const double c = __builtin_amdgcn_ldexp(1.0, ke), cinv = __builtin_amdgcn_ldexp(1.0, -ke);
        const double al_ = alpha * c, Q_ = Q * c, C2R_ = C2R * c, A_ = A * c;
        double p00 = al_, p01 = Q_, p10 = C2R_, p11 = c;        // running square Bm'^(2^bit)
        double q00 = 1.0, q01 = 0.0, q10 = 0.0, q11 = 1.0;      // Bm'^(L-1)
        bool have = false;
#pragma unroll
        for (int bit = 0; (1 << bit) <= L - 1; bit++) {
            if ((L - 1) & (1 << bit)) {
                if (!have) { q00 = p00; q01 = p01; q10 = p10; q11 = p11; have = true; }
                else {
                    const double t00 = fma(q00, p00, q01 * p10), t01 = fma(q00, p01, q01 * p11);
                    const double t10 = fma(q10, p00, q11 * p10), t11 = fma(q10, p01, q11 * p11);
                    q00 = t00; q01 = t01; q10 = t10; q11 = t11;
                }
            }
            if ((2 << bit) <= L - 1) {
                const double t00 = fma(p00, p00, p01 * p10), t01 = fma(p00, p01, p01 * p11);
                const double t10 = fma(p10, p00, p11 * p10), t11 = fma(p10, p01, p11 * p11);
                p00 = t00; p01 = t01; p10 = t10; p11 = t11;
            }
        }
        // Bm'^L = Bm'^(L-1) * Bm'
        const double r00 = fma(q00, al_, q01 * C2R_), r01 = fma(q00, Q_, q01 * c);
        const double r10 = fma(q10, al_, q11 * C2R_), r11 = fma(q10, Q_, q11 * c);
        if (act) {
            double ra = 0.0, rb = 0.0, rcc = c;
            auto row_c = [&](int j, double e, double bu) {
                gv_[j] = e; Jv[j] = bu;
                const double s20 = fma(bu, C2R, ACR * e);
                const double na = fma(ra, al_, fma(rb, C2R_, rcc * s20));
                rb = fma(rb, c, fma(ra, Q_, rcc * bu));
                ra = na;
                rcc *= A_;
            };
            auto row = [&](int j) { row_c(j, e_at(j), bu_at(j)); };
            if constexpr (SPF) {
                StepRing r;      // steps L-1 (predicated), L-2, ... 0; step j in slot j % SPFN
#pragma unroll
                for (int d = 0; d < SPFD; d++)
                    if (L - 1 - d >= 0) ring_rd(r.w[(L - 1 - d) % SPFN], r.o, L - 1 - d, d == 0 || ((L - 1 - d) & 1) != 0);
                __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                for (int j = L - 1; j >= 0; j--) {
                    if (j - SPFD >= 0) ring_rd(r.w[(j - SPFD) % SPFN], r.o, j - SPFD, ((j - SPFD) & 1) != 0);
                    if (j < L - 1 || tail) row_c(j, e_of(r.w[j % SPFN], r.o, j), bu_of(r.w[j % SPFN], r.o, j));
                    __builtin_amdgcn_sched_barrier(0x6);
                }
            } else {
                if (tail) row(L - 1);
#pragma unroll
                for (int j = L - 2; j >= 0; j--) row(j);
            }
            M.m00 = tail ? r00 : q00; M.m01 = tail ? r01 : q01;
            M.m10 = tail ? r10 : q10; M.m11 = tail ? r11 : q11;
            M.m20 = ra; M.m21 = rb; M.m22 = rcc * cinv;
            prenorm(M);
        }
    } else if (act) {
        auto f1c = [&](int j, double e, double bu) {
            const bool o = (obsmask >> j) & 1u;
            gv_[j] = e; Jv[j] = bu;
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
        auto f1 = [&](int j) { f1c(j, e_at(j), bu_at(j)); };
        if constexpr (SPF) {
            StepRing r;
#pragma unroll
            for (int d = 0; d < SPFD; d++)
                if (d <= L - 1) ring_rd(r.w[d % SPFN], r.o, d, (d & 1) == 0);
            __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
            for (int j = 0; j < L - 1; j++) {
                if (j + SPFD <= L - 1) ring_rd(r.w[(j + SPFD) % SPFN], r.o, j + SPFD, ((j + SPFD) & 1) == 0);
                f1c(j, e_of(r.w[j % SPFN], r.o, j), bu_of(r.w[j % SPFN], r.o, j));
                __builtin_amdgcn_sched_barrier(0x6);
            }
            if (tail) f1c(L - 1, e_of(r.w[(L - 1) % SPFN], r.o, L - 1), bu_of(r.w[(L - 1) % SPFN], r.o, L - 1));
        } else {
#pragma unroll
            for (int j = 0; j < L - 1; j++) f1(j);
            if (tail) f1(L - 1);
        }
        prenorm(M);
    }

    // ------------------------------------------------ forward scan over the cell's LPC lanes
    M = pmul(M, pdpp<DPP_ROW_SHR(1), 0xF>(M));
    M = pmul(M, pdpp<DPP_ROW_SHR(2), 0xF>(M));
    M = pmul(M, pdpp<DPP_ROW_SHR(4), 0xF>(M));
    prenorm(M);
    M = pmul(M, pdpp<DPP_ROW_SHR(8), 0xF>(M));
    if constexpr (LPC >= 32) M = pmul(M, pdpp<DPP_ROW_BCAST15, 0xA>(M));      // lane 15 -> row 1, lane 47 -> row 3
    // exit state of this lane's chunk, then the entry state = exit state of the lane before
    // (lane 0 of each half: the cell's initial state)
    const double n_in = v_t1, d_in = 1.0, x_in = x_t1;      // (LEAD: the state at the tail's first step)
    double n_e = fma(M.m00, n_in, M.m01 * d_in);
    double d_e = fma(M.m10, n_in, M.m11 * d_in);
    double x_e = fma(M.m20, n_in, fma(M.m21, d_in, M.m22 * x_in));
    n_e = dppd<DPP_WAVE_SHR1, 0xF>(n_in, n_e);
    d_e = dppd<DPP_WAVE_SHR1, 0xF>(d_in, d_e);
    x_e = dppd<DPP_WAVE_SHR1, 0xF>(x_in, x_e);
    if (vl == 0) { n_e = n_in; d_e = d_in; x_e = x_in; }
    double Xp, Vp;
    {
        const double rd = fast_rcp(d_e);
        Vp = n_e * rd;
        Xp = x_e * rd;
    }

    // ------------------------------------------------ F2: serial re-run from the exact entry
    double sprod = 1.0, Xu = 0.0, Vu = 0.0;
    int sexp = 0;
    double sg = fma(C2, Vp, R);
    double r0 = fast_rcp(sg);
    // reverse affine composite of the lane's chunk (B1 of em_scan_impl.h), accumulated in time
    // order while the steps are produced: (Pi, G, H) o step_j -- no second pass over h_t
    double Pi = 1.0, G = 0.0, H = 0.0;
    auto f2 = [&](int j) {
        const bool o = DENSE || ((obsmask >> j) & 1u);
        const double e = gv_[j], bu = Jv[j];       // left there by F1
        const double r = o ? r0 : 0.0;
        const double sl = o ? sg : 1.0;
        sprod *= sl;
        sneg |= __double2hiint(sl);
        if ((j & 7) == 7) {
            sexp += __builtin_amdgcn_frexp_exp(sprod);
            sprod = __builtin_amdgcn_frexp_mant(sprod);
        }
        const double w = Vp * r;
        const double K = C * w;                    // src/EM.cpp:86
        if (DENSE) Vu = R * w;                     // :88
        else Vu = fma(-(C2 * w), Vp, Vp);
        const double dl = fma(-C, Xp, e);
        Xu = fma(K, dl, Xp);                       // :87
        likq = fma(dl * r, dl, likq);              // :122
        const double Vp1 = fma(A2, Vu, Q);         // :76
        const double Xp1 = fma(A, Xu, bu);         // :74
        sg = fma(C2, Vp1, R);
        const double z = fast_rcp(sg * Vp1);
        const double rp1 = sg * z;
        r0 = Vp1 * z;
        const double AVu = A * Vu;
        double J = AVu * rp1;                      // :100
        double g = fma(-J, Xp1, Xu);
        double h = fma(-J, AVu, Vu);
        if (j >= L - 2) {
            // step T-1 starts the backward recursion: J = 0, g = Xu, h = Vu, zero terminal value
            const bool fin = (vl == lastLane) && (j == (tail ? L - 1 : L - 2));
            J = fin ? 0.0 : J;
            g = fin ? Xu : g;
            h = fin ? Vu : h;
        }
        Jv[j] = J; gv_[j] = g;
        if (j < L - 1) hs[j * 64] = h; else hlast = h;
        G = fma(Pi, g, G);
        H = fma(Pi * Pi, h, H);
        Pi *= J;
        Xp = Xp1;
        Vp = Vp1;
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    };
    if (act) {
#pragma unroll
        for (int j = 0; j < L - 1; j++) f2(j);
        if (tail) f2(L - 1);
    }
    tLv = fma(Xu, Xu, Vu);                                                // Xs^2 + Vs at T-1 (in lastLane)
    lsp = fma((double)sexp, 0.69314718055994530942, log_pos(sprod));

    // ------------------------------------------------ reverse scan of the chunk composites
#define RSCAN_ROUND(n)                                                     \
    {                                                                  \
        const double Pb = dpp1<DPP_ROW_SHL(n)>(Pi);                    \
        const double Gb = dppz<DPP_ROW_SHL(n)>(G);                     \
        const double Hb = dppz<DPP_ROW_SHL(n)>(H);                     \
        G = fma(Pi, Gb, G);                                            \
        H = fma(Pi * Pi, Hb, H);                                       \
        Pi *= Pb;                                                      \
    }
    RSCAN_ROUND(1) RSCAN_ROUND(2) RSCAN_ROUND(4) RSCAN_ROUND(8)
#undef RSCAN_ROUND
    rscan_cross<LPC>(Pi, G, H, lane);      // later rows of the cell
    // (G, H) = (Xs, Vs) at the first step of the chunk; the value just after this lane's
    // chunk is the next lane's, and the zero terminal value for the half's last lane
    double Xn = dppd<DPP_WAVE_SHL1, 0xF>(0.0, G);
    double Vn = dppd<DPP_WAVE_SHL1, 0xF>(0.0, H);
    if (vl == LPC - 1) { Xn = 0.0; Vn = 0.0; }

    // ------------------------------------------------ B2: serial reverse re-run + M-step sums
    // pass 1: the recurrence (:101-102); Xs_t overwrites g_t, the variance sums are formed on
    // the fly (Vs_t is not needed again).  pass 2: the sums over Xs_t (no dependence between
    // steps, the LDS reads of the series batch freely).
    const double XnE = Xn;
    auto b2a_h = [&](int j, double h) {
        const bool o = DENSE || ((obsmask >> j) & 1u);
        const double J = Jv[j];
        aTx1x = fma(Vn, J, aTx1x);                  // Vs_{t+1} J_t   (:180; J = 0 at t = T-1)
        const double Xs = fma(J, Xn, gv_[j]);       // :101
        const double Vs = fma(J * J, Vn, h);        // :102
        aPall += Vs;                                // :181,:183
        if (!DENSE) aSxx += o ? Vs : 0.0;           // :152
        gv_[j] = Xs;
        Xn = Xs;
        Vn = Vs;
    };
    auto b2a = [&](int j) { b2a_h(j, (j < L - 1) ? hs[j * 64] : hlast); };
    auto b2b_v = [&](int j, bool top, auto &&vv) {
        const bool o = DENSE || ((obsmask >> j) & 1u);
        const double Xs = gv_[j];
        const double Xnx = top ? XnE : gv_[top ? j : j + 1];
        aTx1x = fma(Xnx, Xs, aTx1x);                // :180
#pragma unroll
        for (int p_ = 0; p_ < PP; p_++) {
            const double ut = vv(1 + p_);           // zero at t = T-1
            aTx1u[p_] = fma(Xnx, ut, aTx1u[p_]);    // :190
            aTux[p_] = fma(ut, Xs, aTux[p_]);       // :191
        }
        aPall = fma(Xs, Xs, aPall);
        const double xo = o ? Xs : 0.0;
        aSyx = fma(vv(0), xo, aSyx);                // :151
        if (!DENSE) aSxx = fma(xo, xo, aSxx);
#pragma unroll
        for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = fma(xo, vv(1 + PP + q_), aSxv[q_]);   // :159
        if (!SPF && (j & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    };
    auto b2b = [&](int j, bool top) { b2b_v(j, top, [&](int k) { return val(j, k); }); };
    if (act) {
        if constexpr (SPF) {
            // pass 1: the strip's h_t two steps ahead; pass 2's first image reads are issued ahead of pass 1
            StepRing r;
#pragma unroll
            for (int d = 0; d < SPFD; d++)
                if (L - 1 - d >= 0) ring_rd(r.w[(L - 1 - d) % SPFN], r.o, L - 1 - d, d == 0 || ((L - 1 - d) & 1) != 0);
            constexpr int HPF = 2;
            double hr[HPF + 1];
#pragma unroll
            for (int d = 0; d < HPF; d++)
                if (L - 2 - d >= 0) hr[(L - 2 - d) % (HPF + 1)] = hs[(L - 2 - d) * 64];
            __builtin_amdgcn_sched_barrier(0x6);
            if (tail) b2a_h(L - 1, hlast);
            else gv_[L - 1] = XnE;
#pragma unroll
            for (int j = L - 2; j >= 0; j--) {
                if (j - HPF >= 0) hr[(j - HPF) % (HPF + 1)] = hs[(j - HPF) * 64];
                b2a_h(j, hr[j % (HPF + 1)]);
                __builtin_amdgcn_sched_barrier(0x6);
            }
#pragma unroll
            for (int j = L - 1; j >= 0; j--) {
                if (j - SPFD >= 0) ring_rd(r.w[(j - SPFD) % SPFN], r.o, j - SPFD, ((j - SPFD) & 1) != 0);
                auto vv = [&](int k) { return k < KH2 ? r.w[j % SPFN][k] : r.o[(j >> 1) & 1][j & 1]; };
                if (j == L - 1) { if (tail) b2b_v(L - 1, true, vv); }
                else b2b_v(j, false, vv);
                __builtin_amdgcn_sched_barrier(0x6);
            }
        } else {
            if (tail) b2a(L - 1);
            else gv_[L - 1] = XnE;             // "next" of step L-2 for chunks without the L-th step
#pragma unroll
            for (int j = L - 2; j >= 0; j--) b2a(j);
            if (tail) b2b(L - 1, true);
#pragma unroll
            for (int j = L - 2; j >= 0; j--) b2b(j, false);
        }
    }
    // Xn, Vn = Xs, Vs at the first step of this lane's chunk
    X0v = Xn; V0v = Vn;

    o.aSyx = aSyx; o.aTx1x = aTx1x; o.aPall = aPall; o.aSxx = aSxx;
    o.likq = likq; o.lsp = lsp; o.tLv = tLv; o.X0v = X0v; o.V0v = V0v; o.sneg = sneg;
#pragma unroll
    for (int q_ = 0; q_ < QQ; q_++) o.aSxv[q_] = aSxv[q_];
#pragma unroll
    for (int p_ = 0; p_ < PP; p_++) { o.aTx1u[p_] = aTx1u[p_]; o.aTux[p_] = aTux[p_]; }
}
template <int PP, int QQ, int L, int LPC, bool DENSE, bool QUEUE, bool LEAD>
__device__ __forceinline__ void em_pair_body(const EmParams &prm, const double *ys, double *hs,
                                             const double *lu, int s, int c0, int nc, int lane,
                                             int wave) {
    static_assert(!(LEAD && DENSE), "a lead of missing steps and a fully observed series exclude each other");
    static_assert(LPC == 32 || LPC == 16, "two or four cells per wave");
    constexpr int CPW = 64 / LPC;                         // cells per wave
    // `half` = which cell of the wave this lane works for (the name dates from LPC = 32)
    const int half = lane / LPC, vl = lane & (LPC - 1), hbase = lane & ~(LPC - 1);
    // LEAD: the first `lead` steps of every series of the launch are unobserved and are handled in
    // closed form (below); the sweeps work on the tail [lead, T) only, all indices tail relative
    const int lead = LEAD ? prm.lead : 0;
    const int T = prm.T - lead;              // steps of the sweeps
    const int P = 6 + prm.p + prm.q;
    const SeriesConst *__restrict__ sc = prm.sc + s;
    const int n_obs = sc->n_obs;
    const int nl = (T + L - 1) / L;          // active lanes of a half
    const int rp = T - nl * (L - 1);         // lanes < rp own L steps, the others L-1
    const bool act = vl < nl;
    const bool tail = vl < rp;
    const int lastLane = nl - 1;             // (within the half) owner of step T-1
    const int t0 = vl * (L - 1) + min(vl, rp);

    unsigned obsmask = 0;
    if (!DENSE && act) {
        const double *gy = prm.yp + (long)s * prm.T + lead;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const double yv = (j < L - 1 || tail) ? gy[t0 + j] : NAN;
            if (isfinite(yv)) obsmask |= (1u << j);
        }
    }

    // this half's cell
    int k = QUEUE ? 0 : CPW * wave + half;
    if constexpr (QUEUE) {
        if (vl == 0) k = atomicAdd(prm.queue + s, 1);
        k = __shfl(k, hbase, 64);
    }
    bool alive = k < nc;
    int cell = c0 + (alive ? k : 0);
    Theta<PP, QQ> th;
    load_theta(th, prm.theta0 + (long)cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
    white_in(th, (SeriesConstK)sc);   // (B, D) -> whitened input coordinates (mstep_update_white)
    double lik = NAN, lik1 = NAN, lik2 = NAN;
    int it = 0;
    int wit = 0;             // wave-uniform iteration count (interrupt poll)
#ifdef LDSR_SCAN_TIMING
    unsigned long long tick_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
    const unsigned long long real0_ = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif

    while (__any(alive)) {
        SCAN_TICK(7)       // M-step, stop rule, loop
        const double A = th.A, Q = th.Q;
        const double A2 = A * A;

        // ------------------------------------------------ LEAD: the all-missing first `lead` steps
        // Over unobserved steps K_t = 0 (src/EM.cpp:82-84): Xp_{t+1} = A Xp_t + B u_t and
        // Vp_{t+1} = A^2 Vp_t + Q are affine with constant multipliers, the smoother's corrections are
        // pure products,  Xs_t - Xp_t = J_t (Xs_{t+1} - Xp_{t+1}),  Vs_t - Vp_t = J_t^2 (Vs_{t+1} - Vp_{t+1}),
        // and  prod_{k=t}^{t1-1} J_k = A^(t1-t) Vp_t / Vp_t1 =: c_t  telescopes (J_k = A Vp_k / Vp_{k+1},
        // :98-100).  So with (delta, eps) = (Xs - Xp, Vs - Vp) at the first step t1 of the tail,
        //     Xs_t = Xp_t + c_t delta,   Vs_t = Vp_t + c_t^2 eps          for every t < t1,
        // so the lead needs no per-step storage, no backward pass and no 3x3 scan: a first pass (here)
        // carries (Xp, Vp) to t1, the sweeps below run on the tail [t1, T) from (Xp_t1, Vp_t1) and leave
        // (delta, eps), and a second pass (after the sweeps) forms Xs_t as it goes and sums the lead's
        // share of every M-step sum (:180-193) directly -- ~16 fp64 operations per step in all against ~85
        // in the masked sweeps.  (Until round 3 the second pass ran BEFORE the sweeps and summed the
        // coefficients of the polynomials in (delta, eps): 7 + 4 p sums and five more operations per step.
        // tools/lead_closed_form_probe.py checks the formulas against the CPU oracle's smoother: 6e-16.)
        // Lane l of the cell owns lead steps [l nA, (l+1) nA); A^(t1-t) is carried as mantissa x
        // 2^exponent (it starts at 2^-thousands for the early lanes and must neither underflow for
        // good nor lose its mantissa on the way up to 1).
        double x_t1 = th.mu1, v_t1 = th.V1, c_first = 1.0;
        constexpr int NLS = 5 + 2 * PP;       // sums of the lead's second pass
        double lS[LEAD ? NLS : 1];
        // what the first pass hands to the second: the lane's entry state, its weight A^(t1-t) / Vp_t1 as
        // mantissa x 2^exponent, and the closed forms of its variance sums
        double l_X = 0.0, l_V = 0.0, l_dm = 0.0, l_rA = 0.0, l_geo = 0.0, l_geo2 = 0.0;
        int l_de = 0, l_nj = 0;
        bool l_closed = false;
        if constexpr (LEAD) {
            const int nA = (lead + LPC - 1) / LPC;
            const int tA = vl * nA;                                  // first lead step of this lane
            const int nj = min(max(lead - tA, 0), nA);               // ... and how many it has
            const double *lup = lu + (long)vl * PP;                  // [step j][lane][PP]
            // pass 1: this lane's composites of x -> A x + B u_t and V -> A^2 V + Q
            // (two cells per wave: both passes read u_t one step ahead -- same box, (4,4) T = 813 2.37 -> 2.30 ms,
            // 50 lone cells 1.46 -> 1.36, (4,8) T = 1024 5.55 -> 5.30; four cells per wave lose 6 % with it)
            constexpr bool LPF = LPC == 32;
            double al = 1.0, bl = 0.0, a2l = 1.0, ql = 0.0;
            bool var_closed = false;
            double geo = 0.0, geo2 = 0.0;
            if constexpr (lead_pipe(PP)) {
                lead_walk<PP, LPC>(lup, nj, nA, [&](int, const double (&un)[PP]) {
                    double bu = 0.0;
#pragma unroll
                    for (int p_ = 0; p_ < PP; p_++) bu = fma(th.B[p_], un[p_], bu);
                    bl = fma(A, bl, bu);
                    if constexpr (!LDSR_LEAD_CLOSED_VAR) {
                        ql = fma(A2, ql, Q);
                        al *= A;
                        a2l *= A2;
                    }
                });
            } else {
                double un[PP];
#pragma unroll
                for (int p_ = 0; p_ < PP; p_++) un[p_] = (LPF && nj > 0) ? lup[p_] : 0.0;
                for (int j = 0; j < nj; j++) {
                    if constexpr (!LPF) {
#pragma unroll
                        for (int p_ = 0; p_ < PP; p_++) un[p_] = lup[(long)j * LPC * PP + p_];
                    }
                    double bu = 0.0;
#pragma unroll
                    for (int p_ = 0; p_ < PP; p_++) bu = fma(th.B[p_], un[p_], bu);
                    if constexpr (LPF) {
                        const int jn = min(j + 1, nj - 1);
#pragma unroll
                        for (int p_ = 0; p_ < PP; p_++) un[p_] = lup[(long)jn * LPC * PP + p_];
                    }
                    bl = fma(A, bl, bu);
                    if constexpr (!LDSR_LEAD_CLOSED_VAR) {
                        ql = fma(A2, ql, Q);
                        al *= A;
                        a2l *= A2;
                    }
                }
            }
            if constexpr (LDSR_LEAD_CLOSED_VAR) {
                // the multipliers and the variance offset of the lane's nj steps do not need the loop:
                // al = A^nj (binary powering), a2l = (A^2)^nj, ql = Q (1 - a2l) / (1 - A^2) -- three of the
                // pass's five operations per step.  Near |A| = 1 the quotient loses digits: there the sum
                // is formed term by term as before.
                double pa = A, pa2 = A2;
#pragma unroll
                for (int bit = 0; bit < 10; bit++) {          // nj <= 8192 / 16
                    if ((nj >> bit) & 1) { al *= pa; a2l *= pa2; }
                    pa *= pa; pa2 *= pa2;
                }
                const double omA2 = 1.0 - A2;
                var_closed = fabs(omA2) > 0.0009765625;
                if (var_closed) {
                    const double romA2 = fast_rcp(omA2);
                    geo = (1.0 - a2l) * romA2;                       // 1 + A^2 + ... + A^(2 (nj - 1))
                    ql = Q * geo;
                    geo2 = ((double)nj - geo) * romA2;               // sum_j of the partial geometric sums
                } else {
                    for (int j = 0; j < nj; j++) ql = fma(A2, ql, Q);
                }
            }
            // inclusive scan of both affine maps over the cell's lanes
#define LSCAN_ROUND(AB, BB, A2B, QB)                                       \
            {                                                              \
                const double ab = AB, bb = BB, a2b = A2B, qb = QB;         \
                bl = fma(al, bb, bl);  al *= ab;                           \
                ql = fma(a2l, qb, ql); a2l *= a2b;                         \
            }
            LSCAN_ROUND(dpp1<DPP_ROW_SHR(1)>(al), dppz<DPP_ROW_SHR(1)>(bl), dpp1<DPP_ROW_SHR(1)>(a2l), dppz<DPP_ROW_SHR(1)>(ql))
            LSCAN_ROUND(dpp1<DPP_ROW_SHR(2)>(al), dppz<DPP_ROW_SHR(2)>(bl), dpp1<DPP_ROW_SHR(2)>(a2l), dppz<DPP_ROW_SHR(2)>(ql))
            LSCAN_ROUND(dpp1<DPP_ROW_SHR(4)>(al), dppz<DPP_ROW_SHR(4)>(bl), dpp1<DPP_ROW_SHR(4)>(a2l), dppz<DPP_ROW_SHR(4)>(ql))
            LSCAN_ROUND(dpp1<DPP_ROW_SHR(8)>(al), dppz<DPP_ROW_SHR(8)>(bl), dpp1<DPP_ROW_SHR(8)>(a2l), dppz<DPP_ROW_SHR(8)>(ql))
            if constexpr (LPC == 32)
                LSCAN_ROUND((dppd<DPP_ROW_BCAST15, 0xA>(1.0, al)), (dppd<DPP_ROW_BCAST15, 0xA>(0.0, bl)),
                            (dppd<DPP_ROW_BCAST15, 0xA>(1.0, a2l)), (dppd<DPP_ROW_BCAST15, 0xA>(0.0, ql)))
#undef LSCAN_ROUND
            double Xl = fma(al, th.mu1, bl), Vl = fma(a2l, th.V1, ql);      // after this lane's steps
            x_t1 = shfl_d(Xl, hbase | (LPC - 1));                          // = at t1 (idle lanes pass through)
            v_t1 = shfl_d(Vl, hbase | (LPC - 1));
            Xl = dppd<DPP_WAVE_SHR1, 0xF>(th.mu1, Xl);                       // entry of this lane
            Vl = dppd<DPP_WAVE_SHR1, 0xF>(th.V1, Vl);
            if (vl == 0) { Xl = th.mu1; Vl = th.V1; }
            const double rV1 = fast_rcp(v_t1);
            // d = A^(t1 - t) at this lane's first step, as m 2^e (|A| clamped away from 0: beyond
            // a few steps the true value is below 1e-300 anyway)
            const double As = fabs(A) < 1e-150 ? copysign(1e-150, A) : A;
            const double rA = fast_rcp(As);
            const int k0 = lead - tA;
            const double xl2 = (double)k0 * (log_pos(fabs(As)) * 1.4426950408889634074);
            const double ef = fmin(fmax(floor(xl2), -1.0e6), 1.0e6);
            double dm = exp2(xl2 - ef);
            int de = (int)ef;
            if (As < 0.0 && (k0 & 1)) dm = -dm;
            // (1 / Vp_t1 rides in the mantissa: one multiplication per step less)
            dm *= rV1;
            de += __builtin_amdgcn_frexp_exp(dm);
            dm = __builtin_amdgcn_frexp_mant(dm);
            c_first = __builtin_amdgcn_ldexp(dm, de) * Vl;
            c_first = shfl_d(c_first, hbase);       // c_0 (lane 0 of the cell)
            l_X = Xl; l_V = Vl; l_dm = dm; l_de = de; l_rA = rA; l_nj = nj;
            l_closed = var_closed; l_geo = geo; l_geo2 = geo2;
        }

        // ------------------------------------------------ outputs of the sweeps (either form)
        double likq = 0.0, lsp = 0.0, tLv = 0.0, X0v = 0.0, V0v = 0.0;
        int sneg = 0;
        // (zeroed where a branch starts to accumulate, not here: as values defined at the top of the
        // iteration they held 2 (5 + q + 2p) registers through the forward sweeps -- 42 at p = 4, q = 8)
        double aSyx, aTx1x, aPall, aSxx = 0.0;
        double aSxv[QQ], aTx1u[PP], aTux[PP];
        auto zero_sums = [&]() {
            aSyx = 0.0; aTx1x = 0.0; aPall = 0.0;
#pragma unroll
            for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = 0.0;
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) { aTx1u[p_] = 0.0; aTux[p_] = 0.0; }
        };

        SCAN_TICK(0)       // iteration constants, first lead pass
        if (alive) {
            PairSweepOut<PP, QQ> o;
            pair_generic_sweeps<PP, QQ, L, LPC, DENSE>(o, th, ys, hs, obsmask, lane, nl, rp, x_t1, v_t1);
            aSyx = o.aSyx; aTx1x = o.aTx1x; aPall = o.aPall; aSxx = o.aSxx;
            likq = o.likq; lsp = o.lsp; tLv = o.tLv; X0v = o.X0v; V0v = o.V0v; sneg = o.sneg;
#pragma unroll
            for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = o.aSxv[q_];
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) { aTx1u[p_] = o.aTx1u[p_]; aTux[p_] = o.aTux[p_]; }
        } else {
            zero_sums();     // (an idle half: nothing ran)
        }

        SCAN_TICK(6)       // the sweeps
        // ------------------------------------------------ LEAD, second pass: Xs_t = Xp_t + c_t delta as it goes
        if constexpr (LEAD) {
            const double dlt = shfl_d(X0v, hbase) - x_t1;            // Xs - Xp at the tail's first step
            const double *lup = lu + (long)vl * PP;
            constexpr bool LPF = LPC == 32;
            const int nj = l_nj;
            const double rA = l_rA;
            double Xl = l_X, Vl = l_V, dm = l_dm;
            int de = l_de;
#pragma unroll
            for (int i = 0; i < NLS; i++) lS[i] = 0.0;
            double c = __builtin_amdgcn_ldexp(dm, de) * Vl;          // c_t, and c1 = c_{t+1}
            double Xs = fma(c, dlt, Xl);
            auto lead2 = [&](int j, const double (&ul)[PP]) {
                double bu = 0.0;
#pragma unroll
                for (int p_ = 0; p_ < PP; p_++) bu = fma(th.B[p_], ul[p_], bu);
                const double Xl1 = fma(A, Xl, bu), Vl1 = fma(A2, Vl, Q);
                dm *= rA;
                // (every 4 steps: the compiler unrolls this loop by four, where (j & 3) == 3 is a constant of
                // each copy -- every 8 cost three selects per renormalisation; exact either way)
                if ((j & 3) == 3) { de += __builtin_amdgcn_frexp_exp(dm); dm = __builtin_amdgcn_frexp_mant(dm); }
                const double c1 = __builtin_amdgcn_ldexp(dm, de) * Vl1;
                const double Xs1 = fma(c1, dlt, Xl1);           // (the last one is Xs at t1 itself: c = 1)
                lS[0] = fma(Xs, Xs, lS[0]);                   // sum Xs^2
                if constexpr (!LDSR_LEAD_CLOSED_VAR) lS[1] += Vl;   // sum Vp
                lS[2] = fma(c, c, lS[2]);                     // sum c^2          (Vs_t = Vp_t + c_t^2 eps)
                lS[3] = fma(Xs1, Xs, lS[3]);                  // sum Xs_{t+1} Xs_t
                lS[4] = fma(c, c1, lS[4]);                    // sum c_t c_{t+1}  (Vs_{t+1} J_t = A Vp_t + c_t c_{t+1} eps)
#pragma unroll
                for (int p_ = 0; p_ < PP; p_++) {
                    lS[5 + p_] = fma(Xs1, ul[p_], lS[5 + p_]);               // sum Xs_{t+1} u_t   (:190)
                    lS[5 + PP + p_] = fma(ul[p_], Xs, lS[5 + PP + p_]);      // sum u_t Xs_t       (:191)
                }
                Xl = Xl1; Vl = Vl1; c = c1; Xs = Xs1;
            };
            if constexpr (lead_pipe(PP)) {
                const int nA = (lead + LPC - 1) / LPC;
                lead_walk<PP, LPC>(lup, nj, nA, lead2);
            } else {
                double un2[PP];
#pragma unroll
                for (int p_ = 0; p_ < PP; p_++) un2[p_] = (LPF && nj > 0) ? lup[p_] : 0.0;
                for (int j = 0; j < nj; j++) {
                    double ul[PP];
                    if constexpr (!LPF) {
#pragma unroll
                        for (int p_ = 0; p_ < PP; p_++) un2[p_] = lup[(long)j * LPC * PP + p_];
                    }
#pragma unroll
                    for (int p_ = 0; p_ < PP; p_++) ul[p_] = un2[p_];
                    if constexpr (LPF) {
                        const int jn = min(j + 1, nj - 1);
#pragma unroll
                        for (int p_ = 0; p_ < PP; p_++) un2[p_] = lup[(long)jn * LPC * PP + p_];
                    }
                    lead2(j, ul);
                }
            }
            if constexpr (LDSR_LEAD_CLOSED_VAR) {
                // sum of Vp over the lane's steps from its entry value V_e:  Vp_j = A^(2j) V_e + Q (1 + .. + A^(2(j-1)))
                if (l_closed) {
                    lS[1] = fma(l_V, l_geo, Q * l_geo2);
                } else {
                    double vv = l_V;
                    for (int j = 0; j < nj; j++) { lS[1] += vv; vv = fma(A2, vv, Q); }
                }
            }
        }
        SCAN_TICK(5)       // (LEAD: the second pass)
        // ------------------------------------------------ one reduction per half, M-step, stop rule
        Sums<PP, QQ> S;
        {
            constexpr int NB = 5 + (DENSE ? 0 : 1);
            constexpr int NT = NB + QQ + 2 * PP;                     // the sweeps' sums
            constexpr int NR = NT + (LEAD ? NLS : 0);                // + the lead's
            constexpr int NSL = (NR + LPC - 1) / LPC;                // live slots per lane after the halving rounds
            static_assert(NSL <= 4, "reduction gather: at most four slots per lane");
            double red[NR];
            if constexpr (LEAD) {
#pragma unroll
                for (int i = 0; i < NLS; i++) red[NT + i] = lS[i];
            }
            red[0] = aSyx; red[1] = aTx1x; red[2] = aPall; red[3] = likq; red[4] = lsp;
            if (!DENSE) red[5] = aSxx;
#pragma unroll
            for (int q_ = 0; q_ < QQ; q_++) red[NB + q_] = aSxv[q_];
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) { red[NB + QQ + p_] = aTx1u[p_]; red[NB + QQ + PP + p_] = aTux[p_]; }
            // recursive halving over the half's 32 lanes (em_scan_impl.h), then every lane fetches
            // the totals from their home lanes of its own half
            red_rounds<NR, LPC / 2>(red, lane);
            {
                // (more values than lanes -- 17 or 18 sums on the 16 lanes of a quad cell -- leave two
                // live slots per lane)
                // (... and the 30 + 21 sums of p = q = 8 with a lead four on 16 lanes)
                double tt[NSL];
#pragma unroll
                for (int k_ = 0; k_ < NSL; k_++) tt[k_] = red[k_];
#pragma unroll
                for (int i = 0; i < NR; i++) red[i] = shfl_d(tt[red_slot(i, NR, LPC)], hbase | red_home(i, NR, LPC));
            }
            S.X0 = shfl_d(X0v, hbase);               // :218
            S.V0 = shfl_d(V0v, hbase);               // :219
            const double termLast = shfl_d(tLv, hbase | lastLane);
            if constexpr (LEAD) {
                // eps = Vs - Vp at the tail's first step closes the lead's variance sums; mu1 / V1 come from t = 0
                const double dlt = S.X0 - x_t1, eps = S.V0 - v_t1;
                const double *l = red + NT;
                red[2] += (l[0] + l[1]) + eps * l[2];                                       // sum Xs^2 + Vs
                red[1] += l[3] + fma(eps, l[4], A * l[1]);                                  // sum Xs' Xs + Vs' J
#pragma unroll
                for (int p_ = 0; p_ < PP; p_++) {
                    red[NB + QQ + p_] += l[5 + p_];                                         // Tx1u
                    red[NB + QQ + PP + p_] += l[5 + PP + p_];                               // Tux
                }
                S.X0 = fma(c_first, dlt, th.mu1);
                S.V0 = fma(c_first * c_first, eps, th.V1);
            }
            const double term0 = fma(S.X0, S.X0, S.V0);
            const unsigned long long negm = __ballot(sneg < 0);
            const bool neg = ((negm >> hbase) & ((1ull << LPC) - 1ull)) != 0;   // log of a negative Sigma
            S.Syx = red[0]; S.Tx1x = red[1];
            S.Sxx = DENSE ? red[2] : red[5];
#pragma unroll
            for (int q_ = 0; q_ < QQ; q_++) S.Sxv[q_] = red[NB + q_];
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) { S.Tx1u[p_] = red[NB + QQ + p_]; S.Tux[p_] = red[NB + QQ + PP + p_]; }
            S.Txx = red[2] - termLast;
            S.Tx1x1 = red[2] - term0;
            lik2 = lik1;
            lik1 = lik;
            lik = (-0.5 * n_obs * LDSR_LOG_2PI - 0.5 * (red[3] + red[4])) / n_obs;   // :113-124
            if (neg) lik = NAN;
        }
        SCAN_TICK(8)       // reduction, likelihood
        int abort_now = 0;
        if (prm.abort && ((++wit) & 63) == 0)      // src/EM.cpp:261-262 polls too
            abort_now = __builtin_amdgcn_readfirstlane(lane == 0 ? ldsr_poll_abort(prm.abort) : 0);
        if (alive && prm.liks && vl == 0) prm.liks[(long)cell * prm.niter + it] = lik;
        it++;
        bool stop = it >= prm.niter || abort_now;
        if (it >= 3 && fabs(lik - lik1) < prm.tol && fabs(lik1 - lik2) < prm.tol) stop = true;   // :272
        if (alive && stop) {
            // theta stays the one that produced this fit (:276-279)
            white_out(th, (SeriesConstK)sc);
            if (vl == 0) {
                store_theta(th, prm.theta + (long)cell * P, prm.p, prm.q);
                if (prm.liks && prm.liks_nanfill)
                    for (int i = it; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
                prm.n_iter[cell] = it;
                prm.lik[cell] = lik;
                prm.status[cell] = (abort_now && it < prm.niter) ? 3 : (isfinite(lik) ? 0 : 1);
#ifdef LDSR_SCAN_TIMING
                tick_[9] = __builtin_amdgcn_s_memrealtime() - real0_;      // -> shader clock = cycles / this x 100 MHz
                if (prm.liks && prm.niter >= 10)
                    for (int k_ = 0; k_ < 10; k_++) prm.liks[(long)cell * prm.niter + k_] = (double)tick_[k_];
#endif
            }
            alive = false;
            if constexpr (QUEUE) {
                if (abort_now) {
                    // drain the queue: what it still holds is marked, not computed (an
                    // LDSR_EINTERRUPTED return never leaves stale numbers that look like results)
                    for (int pulls = 0; pulls <= nc; pulls++) {
                        int kn = 0;
                        if (vl == 0) kn = atomicAdd(prm.queue + s, 1);
                        kn = __shfl(kn, hbase, 64);
                        if (kn >= nc) break;
                        if (vl == 0) mark_cell_interrupted(prm, c0 + kn);
                    }
                }
                if (!abort_now) {
                    int kn = 0;
                    if (vl == 0) kn = atomicAdd(prm.queue + s, 1);
                    kn = __shfl(kn, hbase, 64);
                    if (kn < nc) {
                        alive = true;
                        cell = c0 + kn;
                        load_theta(th, prm.theta0 + (long)cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
                        white_in(th, (SeriesConstK)sc);
                        it = 0;
                        lik = NAN; lik1 = NAN; lik2 = NAN;
                    }
                }
            }
        } else {
            mstep_update_white<PP, QQ>(th, S, (SeriesConstK)sc, T);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// em_pair_body_steady -- fully observed series, two cells per wave, chunks of L >= 24 steps (T = 737..1024):
// BASELINE config 2's kernel.
//
// With every y_t observed the variance side of the filter (Vp_t, K_t, Sigma_t, Vu_t, J_t, h_t:
// src/EM.cpp:76,86,88,100) is the data-independent Riccati recursion, which reaches its fixed point
// geometrically -- within 31 steps for 99.2 % of the (cell, iteration) pairs of BASELINE config 2 (a numpy
// replay of the bench grid; 16 % of the cells fail at theta0, 0.4 % after ten iterations, ONE cell of the
// 4096 for 31 iterations).  The first NTR = L-1 steps of the series (the chunk of lane 0 without its
// predicated step) are done ONE STEP PER LANE (a 2x2 scan of the variance step matrices, an affine scan
// of the means, the reference's expressions from the exact entry state); lane 31 is no transient step,
// its "entry state" is the state at t = NTR and what it evaluates there are the steady constants K,
// 1/Sigma, Vu, J, h, log Sigma.  Verdict, per CELL: one more step leaves Vp unchanged to 2^-48, every
// Sigma of the block is positive, J^2 < 0.8.  Cells that pass run the STEADY SWEEPS on t >= NTR: only the
// mean recursions (affine, constant multipliers), 19 fp64 operations per step where the generic sweeps
// take 50, no h_t strip, the variance sums in closed form.  Cells that fail (slow Riccati convergence: A
// near 1 with a small gain, mostly in the first EM iterations) take the generic sweeps for that iteration.
//
// TWO LOOPS, ONE PER FORM (round 4).  Round 3 ran the generic sweeps as a `noinline` call inside the steady
// loop, once per slow cell and iteration: ~300 scratch accesses around every call (52 MB of spill traffic
// per launch), and a wave with one slow and one fast cell paid transient block + steady sweeps + call.
// Now the wave is a two-state machine.  S loop (here, inlined in the kernel): steady iterations for all
// its cells; left as soon as a cell fails the verdict.  G phase (pair_steady_g_phase, a REAL function:
// its own register allocation, entered once per slow episode, not once per iteration): generic
// iterations for the cells that fail, while a cell that passes WAITS -- its iteration count simply does
// not advance; left when no cell of the wave fails any more.  (Both loops inlined in one function body
// compiled -- and ran, parity-green -- but the allocator then spills ~150 VGPRs in whichever loop it
// considers colder, although each loop alone fits: 38 k cycles per generic iteration against 22 k.)
// Which form a cell's iteration takes is decided by the cell's own theta (pair_var_block is the same
// explicit-fma code on both sides), so its arithmetic never depends on its wave partner or on the order in
// which cells are handed out -- only the time at which it is computed does.  series_prep orders the cells of
// a series by predicted slowness (EmParams.perm, slowest first, dealt across the workgroups): slow cells
// share waves with slow cells, and the work queue hands them out first.

// what the iteration epilogue needs of EmParams (launch-uniform)
struct PairEnv {
    int T, p, q, has_u, has_v, niter, liks_nanfill, n_obs;
    double tol;
    const SeriesConst *sc;      // this series' constants
    int *queue_s;               // this series' work-queue head
    const int *perm;            // position -> cell, or null
    const double *theta0;
    double *theta, *lik, *liks;
    int *n_iter, *status;
    const int *abort;
};
// ... and the state of this lane's cell carried from iteration to iteration
template <int PP, int QQ>
struct PairCarry {
    Theta<PP, QQ> th;
    double lik, lik1, lik2;
    int it, cell, wit;
    bool alive;
#ifdef LDSR_SCAN_TIMING
    unsigned long long tick_[10], last_, real0_;
#endif
};
#ifdef LDSR_SCAN_TIMING
#define PAIR_TICK_REFS auto &tick_ = cs.tick_; auto &last_ = cs.last_;
#else
#define PAIR_TICK_REFS
#endif

__device__ __forceinline__ int pair_cell_at(const PairEnv &E, int c0, int pos) { return E.perm ? E.perm[c0 + pos] : c0 + pos; }

__device__ __forceinline__ void pair_mark_interrupted(const PairEnv &E, int cell) {
    const int P = 6 + E.p + E.q;
    for (int k = 0; k < P; k++) E.theta[(long)cell * P + k] = NAN;
    if (E.liks && E.liks_nanfill)
        for (int i = 0; i < E.niter; i++) E.liks[(long)cell * E.niter + i] = NAN;
    E.n_iter[cell] = 0;
    E.lik[cell] = NAN;
    E.status[cell] = 3;
}

// ---- variance side of the transient block, and the verdict.  Explicit fma / mul only: the S loop and the
// G phase each compile a copy and must decide alike.  (a) inclusive scan of the 2x2 step matrices
// [[alpha, Q],[C2R, 1]] (one and the same for every step: scaled by an exact power of two so that
// max(alpha, 1) c is in [0.5, 1), as the generic dense F1 does; identity beyond step NTR-1), then
// Vp = (p00 V1 + p01) / (p10 V1 + p11) of the lane before; (b) the reference's expressions, as in F2.
struct PairVarBlk { double Vp, sg, r0, K, Vu, AVu, J, Vp1; bool st; };
template <int L>
__device__ __forceinline__ PairVarBlk pair_var_block(double V1, double A, double C, double Q, double R, bool live,
                                                     int vl, int hbase, bool shape_ok) {
    constexpr int LPC = 32, NTR = L - 1;
    PairVarBlk b;
    const double A2 = A * A, C2 = C * C;
    const double C2R = C2 * fast_rcp(R), alpha = fma(Q, C2R, A2);
    const bool trl = vl < NTR;
    double Vp;
    {
        const double mxs = fmax(alpha, 1.0);
        const int ke = -__builtin_amdgcn_frexp_exp(mxs);
        const double cs = __builtin_amdgcn_ldexp(1.0, ke);
        double p00 = trl ? alpha * cs : 1.0, p01 = trl ? Q * cs : 0.0;
        double p10 = trl ? C2R * cs : 0.0, p11 = trl ? cs : 1.0;
#define VSCAN_ROUND(Q00, Q01, Q10, Q11)                                                      \
        {                                                                            \
            const double q00 = Q00, q01 = Q01, q10 = Q10, q11 = Q11;                 \
            const double r00 = fma(p00, q00, p01 * q10), r01 = fma(p00, q01, p01 * q11); \
            const double r10 = fma(p10, q00, p11 * q10), r11 = fma(p10, q01, p11 * q11); \
            p00 = r00; p01 = r01; p10 = r10; p11 = r11;                              \
        }
#define VSCAN_SHR(n) VSCAN_ROUND(dpp1<DPP_ROW_SHR(n)>(p00), dppz<DPP_ROW_SHR(n)>(p01), dppz<DPP_ROW_SHR(n)>(p10), dpp1<DPP_ROW_SHR(n)>(p11))
        VSCAN_SHR(1) VSCAN_SHR(2) VSCAN_SHR(4)
        {   // exact power-of-two rescale (projective coordinates are scale free)
            const double m = fmax(fmax(fabs(p00), fabs(p01)), fmax(fabs(p10), fabs(p11)));
            const int e2 = 1 - __builtin_amdgcn_frexp_exp(m);
            p00 = __builtin_amdgcn_ldexp(p00, e2); p01 = __builtin_amdgcn_ldexp(p01, e2);
            p10 = __builtin_amdgcn_ldexp(p10, e2); p11 = __builtin_amdgcn_ldexp(p11, e2);
        }
        VSCAN_SHR(8)
        VSCAN_ROUND((dppd<DPP_ROW_BCAST15, 0xA>(1.0, p00)), (dppd<DPP_ROW_BCAST15, 0xA>(0.0, p01)),
                    (dppd<DPP_ROW_BCAST15, 0xA>(0.0, p10)), (dppd<DPP_ROW_BCAST15, 0xA>(1.0, p11)))
#undef VSCAN_SHR
#undef VSCAN_ROUND
        double n_e = fma(p00, V1, p01), d_e = fma(p10, V1, p11);
        n_e = dppd<DPP_WAVE_SHR1, 0xF>(V1, n_e);
        d_e = dppd<DPP_WAVE_SHR1, 0xF>(1.0, d_e);
        if (vl == 0) { n_e = V1; d_e = 1.0; }
        Vp = n_e * fast_rcp(d_e);                            // entering step vl (vl >= NTR: t = NTR)
    }
    const double sg = fma(C2, Vp, R);
    const double r0 = fast_rcp(sg);
    const double w = Vp * r0;
    b.K = C * w;                                             // src/EM.cpp:86
    b.Vu = R * w;                                            // :88
    b.Vp1 = fma(A2, b.Vu, Q);                                // :76
    b.AVu = A * b.Vu;
    b.J = b.AVu * fast_rcp(b.Vp1);                           // :100
    b.Vp = Vp; b.sg = sg; b.r0 = r0;
    // fixed point reached, and every Sigma of the block positive (a negative one is the generic
    // sweeps' business: lik = NaN); J^2 < 0.8, which a converged Riccati recursion implies unless V1
    // happens to be the fixed point itself, lets the closed-form variance sums drop J^(2(T-NTR))
    const double dV = b.Vp1 - Vp;
    const bool conv = fabs(dV) <= 3.552713678800501e-15 * fabs(Vp) && Vp > 0.0 && b.J * b.J < 0.8;
    const unsigned long long okm = __ballot(sg > 0.0 && sg < INFINITY);
    const unsigned long long cvm = __ballot(conv);
    constexpr unsigned long long CELL = (1ull << LPC) - 1ull;
    const unsigned long long hm = (okm >> hbase) & CELL, hc = (cvm >> hbase) & CELL;
    b.st = live && hm == CELL && ((hc >> (LPC - 1)) & 1ull) != 0ull && shape_ok;
    return b;
}

// ---- the rest of an iteration for the halves in `active` (the cells whose sweeps just ran): one
// reduction per half, likelihood, stop rule, M-step; a cell that stops stores its result and -- work
// queue -- its half pulls the next one.  The other halves keep their state untouched.
template <int PP, int QQ, bool QUEUE>
__device__ __forceinline__ void pair_steady_finish(const PairEnv &E, PairCarry<PP, QQ> &cs, const PairSweepOut<PP, QQ> &o,
                                                   double addTx1x, double addPall, bool active, int lane, int c0,
                                                   int nc, int lastLane) {
    constexpr int LPC = 32;
    PAIR_TICK_REFS
    const int vl = lane & 31, hbase = lane & 32;
    const int P = 6 + E.p + E.q;
    Sums<PP, QQ> S;
    {
        constexpr int NB = 5;
        constexpr int NR = NB + QQ + 2 * PP;
        double red[NR];
        red[0] = o.aSyx; red[1] = o.aTx1x; red[2] = o.aPall; red[3] = o.likq; red[4] = o.lsp;
#pragma unroll
        for (int q_ = 0; q_ < QQ; q_++) red[NB + q_] = o.aSxv[q_];
#pragma unroll
        for (int p_ = 0; p_ < PP; p_++) { red[NB + QQ + p_] = o.aTx1u[p_]; red[NB + QQ + PP + p_] = o.aTux[p_]; }
        // recursive halving over the half's 32 lanes (em_scan_impl.h), then every lane fetches
        // the totals from their home lanes of its own half
        red_rounds<NR, LPC / 2>(red, lane);
        {
            const double tt = red[0];
#pragma unroll
            for (int i = 0; i < NR; i++) red[i] = shfl_d(tt, hbase | red_home(i, NR, LPC));
        }
        S.X0 = shfl_d(o.X0v, hbase);               // :218
        S.V0 = shfl_d(o.V0v, hbase);               // :219
        const double termLast = shfl_d(o.tLv, hbase | lastLane);
        red[1] += addTx1x; red[2] += addPall;      // (steady form: the closed-form variance sums; else 0)
        const double term0 = fma(S.X0, S.X0, S.V0);
        const unsigned long long negm = __ballot(o.sneg < 0);
        const bool neg = ((negm >> hbase) & ((1ull << LPC) - 1ull)) != 0;   // log of a negative Sigma
        S.Syx = red[0]; S.Tx1x = red[1];
        S.Sxx = red[2];
#pragma unroll
        for (int q_ = 0; q_ < QQ; q_++) S.Sxv[q_] = red[NB + q_];
#pragma unroll
        for (int p_ = 0; p_ < PP; p_++) { S.Tx1u[p_] = red[NB + QQ + p_]; S.Tux[p_] = red[NB + QQ + PP + p_]; }
        S.Txx = red[2] - termLast;
        S.Tx1x1 = red[2] - term0;
        if (active) {
            cs.lik2 = cs.lik1;
            cs.lik1 = cs.lik;
            cs.lik = (-0.5 * E.n_obs * LDSR_LOG_2PI - 0.5 * (red[3] + red[4])) / E.n_obs;   // :113-124
            if (neg) cs.lik = NAN;
        }
    }
    SCAN_TICK(8)       // reduction, likelihood
    int abort_now = 0;
    if (E.abort && ((++cs.wit) & 63) == 0)      // src/EM.cpp:261-262 polls too
        abort_now = __builtin_amdgcn_readfirstlane(lane == 0 ? ldsr_poll_abort(E.abort) : 0);
    if (active && E.liks && vl == 0) E.liks[(long)cs.cell * E.niter + cs.it] = cs.lik;
    if (active) cs.it++;
    bool stop = active && cs.it >= E.niter;
    if (active && cs.it >= 3 && fabs(cs.lik - cs.lik1) < E.tol && fabs(cs.lik1 - cs.lik2) < E.tol) stop = true;   // :272
    if (cs.alive && abort_now) stop = true;       // (a waiting cell stops too)
    if (cs.alive && stop) {
        // theta stays the one that produced this fit (:276-279)
        white_out(cs.th, (SeriesConstK)E.sc);
        if (vl == 0) {
            store_theta(cs.th, E.theta + (long)cs.cell * P, E.p, E.q);
            if (E.liks && E.liks_nanfill)
                for (int i = cs.it; i < E.niter; i++) E.liks[(long)cs.cell * E.niter + i] = NAN;
            E.n_iter[cs.cell] = cs.it;
            E.lik[cs.cell] = cs.lik;
            E.status[cs.cell] = (abort_now && cs.it < E.niter) ? 3 : (isfinite(cs.lik) ? 0 : 1);
#ifdef LDSR_SCAN_TIMING
            tick_[9] = __builtin_amdgcn_s_memrealtime() - cs.real0_;      // -> shader clock = cycles / this x 100 MHz
            if (E.liks && E.niter >= 10)
                for (int k_ = 0; k_ < 10; k_++) E.liks[(long)cs.cell * E.niter + k_] = (double)tick_[k_];
#endif
        }
        cs.alive = false;
        if constexpr (QUEUE) {
            if (abort_now) {
                // drain the queue: what it still holds is marked, not computed (an
                // LDSR_EINTERRUPTED return never leaves stale numbers that look like results)
                for (int pulls = 0; pulls <= nc; pulls++) {
                    int kn = 0;
                    if (vl == 0) kn = atomicAdd(E.queue_s, 1);
                    kn = __shfl(kn, hbase, 64);
                    if (kn >= nc) break;
                    if (vl == 0) pair_mark_interrupted(E, pair_cell_at(E, c0, kn));
                }
            } else {
                int kn = 0;
                if (vl == 0) kn = atomicAdd(E.queue_s, 1);
                kn = __shfl(kn, hbase, 64);
                if (kn < nc) {
                    cs.alive = true;
                    cs.cell = pair_cell_at(E, c0, kn);
                    load_theta(cs.th, E.theta0 + (long)cs.cell * P, E.p, E.q, E.has_u, E.has_v);
                    white_in(cs.th, (SeriesConstK)E.sc);
                    cs.it = 0;
                    cs.lik = NAN; cs.lik1 = NAN; cs.lik2 = NAN;
                }
            }
        }
    } else if (active) {
        mstep_update_white<PP, QQ>(cs.th, S, (SeriesConstK)E.sc, E.T);
    }
}

// ---- G phase: generic iterations for the halves whose cell fails the verdict (`slow`, from the S loop's
// verdict on entry: at least one iteration runs), until no cell of the wave fails.  A real function: the
// generic sweeps keep 244 VGPRs busy, and entered once per slow episode the ~250 scratch accesses of
// the call (callee-saved registers, the caller's live state) are spread over its iterations.  The
// launch-uniform arguments arrive in VGPRs like any other and are made scalar again first.
template <int PP, int QQ, int L, bool QUEUE>
__device__ __attribute__((noinline)) PairCarry<PP, QQ> pair_steady_g_phase(PairEnv Ev, PairCarry<PP, QQ> cs, bool slow,
                                                                           int c0, int nc) {
    constexpr int LPC = 32;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    PAIR_TICK_REFS
    auto ui = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    auto up = [&](auto *ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (decltype(ptr))(((unsigned long long)hi << 32) | lo);
    };
    PairEnv E;
    E.T = ui(Ev.T); E.p = ui(Ev.p); E.q = ui(Ev.q); E.has_u = ui(Ev.has_u); E.has_v = ui(Ev.has_v);
    E.niter = ui(Ev.niter); E.liks_nanfill = ui(Ev.liks_nanfill); E.n_obs = ui(Ev.n_obs);
    E.tol = uniform_d(Ev.tol);
    E.sc = up(Ev.sc); E.queue_s = up(Ev.queue_s); E.perm = up(Ev.perm); E.theta0 = up(Ev.theta0);
    E.theta = up(Ev.theta); E.lik = up(Ev.lik); E.liks = up(Ev.liks); E.n_iter = up(Ev.n_iter);
    E.status = up(Ev.status); E.abort = up(Ev.abort);
    c0 = ui(c0); nc = ui(nc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vl = lane & 31, hbase = lane & 32;
    const int T = E.T;
    const int nl = (T + L - 1) / L, rp = T - nl * (L - 1);
    const bool shape_ok = rp >= 1 && nl > 1;
    const double *ys = smem;
    double *hs = smem + pair_image_doubles(L, PP, QQ, LPC) + (long)wave * pair_strip_doubles(L) + lane;
    do {
        SCAN_TICK(7)
        PairSweepOut<PP, QQ> o;
        pair_generic_sweeps<PP, QQ, L, LPC, true>(o, cs.th, ys, hs, 0u, lane, nl, rp, cs.th.mu1, cs.th.V1);
        SCAN_TICK(6)       // the generic sweeps
        pair_steady_finish<PP, QQ, QUEUE>(E, cs, o, 0.0, 0.0, slow, lane, c0, nc, nl - 1);
        const PairVarBlk vb = pair_var_block<L>(cs.th.V1, cs.th.A, cs.th.C, cs.th.Q, cs.th.R, cs.alive, vl, hbase, shape_ok);
        slow = cs.alive && !vb.st;
    } while (__any(slow));
    return cs;
}

template <int PP, int QQ, int L, bool QUEUE>
__device__ __forceinline__ void em_pair_body_steady(const EmParams &prm, const double *ys, double *hs,
                                                    const double *tri, int s, int c0, int nc, int lane,
                                                    int wave) {
    constexpr int LPC = 32;
    constexpr int KP = scan_pairs(PP, QQ), KV = img_values(PP, QQ);
    constexpr int NTR = L - 1;
    static_assert(NTR < LPC, "the last lane of the cell is the template, not a transient step");
    const int vl = lane & 31, hbase = lane & 32;
    auto val = [&](int j, int i) -> double { return ys[img_off(j, i, KV, LPC, L) + vl * 2]; };
    // all K values of step j of this lane (ds_read_b128 each pair)
    auto ldw = [&](int j, double (&w)[2 * KP]) {
#pragma unroll
        for (int i = 0; i < KV; i++) w[i] = val(j, i);
    };
    // steady sweeps: steps the image is read ahead (a step is KP 16-byte pairs: wide inputs get a
    // shorter ring) and the strip
    constexpr int PF = KP <= 2 ? LDSR_STEADY_PF : (KP <= 4 ? 2 : 1);
    constexpr int PF2 = LDSR_STEADY_PF2;
    const int T = prm.T;
    const int P = 6 + prm.p + prm.q;
    PairEnv E;
    E.T = T; E.p = prm.p; E.q = prm.q; E.has_u = prm.has_u; E.has_v = prm.has_v;
    E.niter = prm.niter; E.liks_nanfill = prm.liks_nanfill; E.tol = prm.tol;
    E.sc = prm.sc + s; E.n_obs = E.sc->n_obs;
    E.queue_s = prm.queue + s; E.perm = prm.perm; E.theta0 = prm.theta0;
    E.theta = prm.theta; E.lik = prm.lik; E.liks = prm.liks; E.n_iter = prm.n_iter; E.status = prm.status;
    E.abort = prm.abort;
    const int nl = (T + L - 1) / L;          // active lanes of a half
    const int rp = T - nl * (L - 1);         // lanes < rp own L steps, the others L-1
    const bool act = vl < nl;
    const bool tail = vl < rp;
    const int lastLane = nl - 1;             // (within the half) owner of step T-1
    const bool shape_ok = rp >= 1 && nl > 1; // lane 0 owns L steps: the block ends on a chunk boundary

    // this half's cell
    PairCarry<PP, QQ> cs;
    int k = QUEUE ? 0 : 2 * wave + (lane >> 5);
    if constexpr (QUEUE) {
        if (vl == 0) k = atomicAdd(E.queue_s, 1);
        k = __shfl(k, hbase, 64);
    }
    cs.alive = k < nc;
    cs.cell = pair_cell_at(E, c0, cs.alive ? k : 0);
    load_theta(cs.th, prm.theta0 + (long)cs.cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
    white_in(cs.th, (SeriesConstK)E.sc);   // (B, D) -> whitened input coordinates (mstep_update_white)
    cs.lik = NAN; cs.lik1 = NAN; cs.lik2 = NAN;
    cs.it = 0;
    cs.wit = 0;              // wave-uniform iteration count (interrupt poll)
#ifdef LDSR_SCAN_TIMING
    for (int k_ = 0; k_ < 10; k_++) cs.tick_[k_] = 0;
    cs.last_ = __builtin_readcyclecounter();
    cs.real0_ = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif
    PAIR_TICK_REFS
    Theta<PP, QQ> &th = cs.th;

    while (__any(cs.alive)) {
        // ================================================================= S loop: steady iterations
        bool slow = false;
        while (__any(cs.alive)) {
            SCAN_TICK(7)       // M-step, stop rule, loop
            const double A = th.A, C = th.C;
            const PairVarBlk vb = pair_var_block<L>(th.V1, A, C, th.Q, th.R, cs.alive, vl, hbase, shape_ok);
            const bool st = vb.st;
            slow = cs.alive && !st;
            if (__builtin_expect(__any(slow), 0)) break;
            // what the sweeps leave in every lane
            double likq = 0.0, lsp = 0.0, tLv = 0.0, X0v = 0.0, V0v = 0.0;
            double addPall = 0.0, addTx1x = 0.0;    // closed-form variance sums of the steady region
            double aSyx = 0.0, aTx1x = 0.0, aPall = 0.0;
            double aSxv[QQ], aTx1u[PP], aTux[PP];
            auto zero_sums = [&]() {
                aSyx = 0.0; aTx1x = 0.0; aPall = 0.0;
#pragma unroll
                for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = 0.0;
#pragma unroll
                for (int p_ = 0; p_ < PP; p_++) { aTx1u[p_] = 0.0; aTux[p_] = 0.0; }
            };
            // ---- mean side of the transient block: Xp_{t+1} = A (1 - K_t C) Xp_t + (A K_t e_t + B u_t) is
            // affine with the gains just found: inclusive scan over the lanes, then the reference's
            // expressions from the exact entry state
            const bool trl = vl < NTR;
            auto tval = [&](int i) -> double { return tri[((i >> 1) * LPC + vl) * 2 + (i & 1)]; };
            double e_t = tval(0), bu_t = 0.0;                       // (tri is zero for vl >= NTR)
#pragma unroll
            for (int q_ = 0; q_ < QQ; q_++) e_t = fma(-th.D[q_], tval(1 + PP + q_), e_t);
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) bu_t = fma(th.B[p_], tval(1 + p_), bu_t);
            const double K = vb.K, J = vb.J, Vu = vb.Vu, AVu = vb.AVu, r0 = vb.r0;
            double Xp;
            {
                const double aKt = A * K;
                double al = trl ? fma(-aKt, C, A) : 1.0, bl = trl ? fma(aKt, e_t, bu_t) : 0.0;
#define MSCAN_ROUND(AB, BB) { const double ab = AB, bb = BB; bl = fma(al, bb, bl); al *= ab; }
                MSCAN_ROUND(dpp1<DPP_ROW_SHR(1)>(al), dppz<DPP_ROW_SHR(1)>(bl))
                MSCAN_ROUND(dpp1<DPP_ROW_SHR(2)>(al), dppz<DPP_ROW_SHR(2)>(bl))
                MSCAN_ROUND(dpp1<DPP_ROW_SHR(4)>(al), dppz<DPP_ROW_SHR(4)>(bl))
                MSCAN_ROUND(dpp1<DPP_ROW_SHR(8)>(al), dppz<DPP_ROW_SHR(8)>(bl))
                MSCAN_ROUND((dppd<DPP_ROW_BCAST15, 0xA>(1.0, al)), (dppd<DPP_ROW_BCAST15, 0xA>(0.0, bl)))
#undef MSCAN_ROUND
                Xp = fma(al, th.mu1, bl);
                Xp = dppd<DPP_WAVE_SHR1, 0xF>(th.mu1, Xp);
                if (vl == 0) Xp = th.mu1;
            }
            const double dl = fma(-C, Xp, e_t);
            const double Xu = fma(K, dl, Xp);                       // :87
            const double Xp1 = fma(A, Xu, bu_t);                    // :74
            const double trJ = J;
            const double trG = fma(-J, Xp1, Xu);
            const double trH = fma(-J, AVu, Vu);
            const double trLq = trl ? dl * r0 * dl : 0.0;           // :122
            const double lg = log_pos(vb.sg);
            const double trLg = trl ? lg : 0.0;
            const int src = hbase | (LPC - 1);
            const double cK = shfl_d(K, src), cJ = shfl_d(J, src), cr = shfl_d(r0, src), cVu = shfl_d(Vu, src);
            const double ch = shfl_d(trH, src), clg = shfl_d(lg, src), X_tr = shfl_d(Xp, src);

            SCAN_TICK(0)       // iteration constants, transient block, verdict
            if (st) {          // (idle halves -- no cell left -- skip the sweeps)
                // ============================================ steady sweeps over t = NTR .. T-1
                // Lane 0 keeps only its predicated step L-1 (= step NTR); lanes 1.. their whole chunks.
                const bool body = act && vl >= 1;
                const bool tail_s = tail;
                const double aK = A * cK, a = fma(-aK, C, A);           // Xp_{t+1} = a Xp_t + (A K e_t + B u_t)
                // a^(L-1), J^(L-1): multipliers of a whole chunk
                double aL = 1.0, JL = 1.0;
                {
                    double sa = a, sj = cJ;
                    bool have = false;
#pragma unroll
                    for (int bit = 0; (1 << bit) <= L - 1; bit++) {
                        if ((L - 1) & (1 << bit)) {
                            if (!have) { aL = sa; JL = sj; have = true; }
                            else { aL *= sa; JL *= sj; }
                        }
                        if ((2 << bit) <= L - 1) { sa *= sa; sj *= sj; }
                    }
                }
                // ---- F1: chunk composite of the affine mean recursion; e_t, B u_t left for F2.
                // The series image is read PF steps ahead through an explicit register ring pinned by
                // scheduling barriers: left alone, the scheduler issued each ds_read_b128 right before
                // its use (one read in flight, 35 ns exposed per read: the sweeps were LDS-latency bound).
                // (the strip is free here: the steady sweeps have no h_t; J_t's registers stay unused, which
                // is what makes room for the read-ahead rings at two waves per SIMD)
                double gv_[L];           // e_t, then g_t of this lane's steps
                double al = 1.0, bl = 0.0, buLast = 0.0;
                auto f1s = [&](int j, const double (&w)[2 * KP]) {
                    double e = w[0], bu = 0.0;
#pragma unroll
                    for (int q_ = 0; q_ < QQ; q_++) e = fma(-th.D[q_], w[1 + PP + q_], e);
#pragma unroll
                    for (int p_ = 0; p_ < PP; p_++) bu = fma(th.B[p_], w[1 + p_], bu);
                    gv_[j] = e;
                    if (j < L - 1) hs[j * 64] = bu; else buLast = bu;   // B u_t waits for F2 in the wave's LDS strip
                    bl = fma(a, bl, fma(aK, e, bu));
                };
                {
                    constexpr bool PRET = KP <= 4;       // the predicated step's values read ahead too (narrow inputs)
                    double Wt[2 * KP], W[PF][2 * KP];
                    if constexpr (PRET) ldw(L - 1, Wt);
#pragma unroll
                    for (int d = 0; d < PF; d++) ldw(d, W[d]);
                    __builtin_amdgcn_sched_barrier(LDSR_STEADY_SBMASK);
                    if (body) {
#pragma unroll
                        for (int j = 0; j < L - 1; j++) {
                            f1s(j, W[j % PF]);
                            if (j + PF < L - 1) ldw(j + PF, W[j % PF]);
                            __builtin_amdgcn_sched_barrier(LDSR_STEADY_SBMASK);
                        }
                        al = aL;
                    }
                    if (tail_s) {
                        if constexpr (!PRET) ldw(L - 1, Wt);
                        f1s(L - 1, Wt);
                        al *= a;
                    }
                }
                // ---- inclusive scan over the cell's lanes, then the entry state of this lane
#define SSCAN_ROUND(AB, BB) { const double ab = AB, bb = BB; bl = fma(al, bb, bl); al *= ab; }
                SSCAN_ROUND(dpp1<DPP_ROW_SHR(1)>(al), dppz<DPP_ROW_SHR(1)>(bl))
                SSCAN_ROUND(dpp1<DPP_ROW_SHR(2)>(al), dppz<DPP_ROW_SHR(2)>(bl))
                SSCAN_ROUND(dpp1<DPP_ROW_SHR(4)>(al), dppz<DPP_ROW_SHR(4)>(bl))
                SSCAN_ROUND(dpp1<DPP_ROW_SHR(8)>(al), dppz<DPP_ROW_SHR(8)>(bl))
                SSCAN_ROUND((dppd<DPP_ROW_BCAST15, 0xA>(1.0, al)), (dppd<DPP_ROW_BCAST15, 0xA>(0.0, bl)))
#undef SSCAN_ROUND
                SCAN_TICK(1)   // steady F1
                double Xq = fma(al, X_tr, bl);                           // after this lane's steps
                Xq = dppd<DPP_WAVE_SHR1, 0xF>(X_tr, Xq);
                if (vl == 0) Xq = X_tr;
                // ---- F2: the reference's mean expressions with the steady gains
                double lq = 0.0, Xuq = 0.0;
                auto f2s = [&](int j, double bu) {
                    const double e = gv_[j];
                    const double dlq = fma(-C, Xq, e);
                    lq = fma(dlq, dlq, lq);                            // :122 (times 1/Sigma below)
                    Xuq = fma(cK, dlq, Xq);                            // :87
                    const double Xq1 = fma(A, Xuq, bu);                // :74
                    double g = fma(-cJ, Xq1, Xuq);
                    if (j >= L - 2) {
                        const bool fin = (vl == lastLane) && (j == (tail_s ? L - 1 : L - 2));
                        g = fin ? Xuq : g;                             // step T-1: Xs = Xu
                    }
                    gv_[j] = g;
                    Xq = Xq1;
                };
                if (body) {
                    double U[PF2];
#pragma unroll
                    for (int d = 0; d < PF2; d++) U[d] = hs[d * 64];
                    __builtin_amdgcn_sched_barrier(LDSR_STEADY_SBMASK);
#pragma unroll
                    for (int j = 0; j < L - 1; j++) {
                        f2s(j, U[j % PF2]);
                        if (j + PF2 < L - 1) U[j % PF2] = hs[(j + PF2) * 64];
                        __builtin_amdgcn_sched_barrier(LDSR_STEADY_SBMASK);
                    }
                }
                if (tail_s) f2s(L - 1, buLast);
                SCAN_TICK(2)   // forward scan, steady F2
                // B2's first reads of the image are issued here, ahead of the reverse scan
                constexpr bool PRET2 = KP <= 4;
                double Vt[2 * KP], V[PF][2 * KP];
                if constexpr (PRET2) {
                    ldw(L - 1, Vt);
#pragma unroll
                    for (int d = 0; d < PF; d++) ldw(L - 2 - d, V[d]);
                }
                __builtin_amdgcn_sched_barrier(LDSR_STEADY_SBMASK);
                tLv = fma(Xuq, Xuq, cVu);
                const int nst = (body ? L - 1 : 0) + (tail_s ? 1 : 0);    // steady steps of this lane
                likq = fma(cr, lq, trLq);
                lsp = fma((double)nst, clg, trLg);
                // ---- reverse composite of the chunk (constant multiplier J), reverse scan
                double Pi = 1.0, G = 0.0;
                if (tail_s) { G = gv_[L - 1]; Pi = cJ; }
                if (body) {
#pragma unroll
                    for (int j = L - 2; j >= 0; j--) G = fma(cJ, G, gv_[j]);
                    Pi *= JL;
                }
#define RSCAN_ROUND(n) { const double Pb = dpp1<DPP_ROW_SHL(n)>(Pi), Gb = dppz<DPP_ROW_SHL(n)>(G); G = fma(Pi, Gb, G); Pi *= Pb; }
                RSCAN_ROUND(1) RSCAN_ROUND(2) RSCAN_ROUND(4) RSCAN_ROUND(8)
#undef RSCAN_ROUND
                {
                    double Hdummy = 0.0;
                    rscan_cross<LPC>(Pi, G, Hdummy, lane);
                }
                double Xn = dppd<DPP_WAVE_SHL1, 0xF>(0.0, G);
                if (vl == LPC - 1) Xn = 0.0;
                const double XsS = shfl_d(G, hbase);                    // Xs at t = NTR (step L-1 of lane 0)
                SCAN_TICK(3)   // reverse composite and scan
                // ---- B2: Xs_t = J Xs_{t+1} + g_t and the sums over Xs in ONE pass (no variance chain)
                zero_sums();
                auto b2s = [&](int j, const double (&w)[2 * KP]) {
                    const double Xs = fma(cJ, Xn, gv_[j]);             // :101
                    aTx1x = fma(Xn, Xs, aTx1x);                        // :180 (Xn = 0 after step T-1)
#pragma unroll
                    for (int p_ = 0; p_ < PP; p_++) {
                        const double ut = w[1 + p_];                   // zero at t = T-1
                        aTx1u[p_] = fma(Xn, ut, aTx1u[p_]);            // :190
                        aTux[p_] = fma(ut, Xs, aTux[p_]);              // :191
                    }
                    aPall = fma(Xs, Xs, aPall);
                    aSyx = fma(w[0], Xs, aSyx);                        // :151
#pragma unroll
                    for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = fma(Xs, w[1 + PP + q_], aSxv[q_]);   // :159
                    Xn = Xs;
                };
                if constexpr (!PRET2) {
                    if (tail_s) ldw(L - 1, Vt);
#pragma unroll
                    for (int d = 0; d < PF; d++) ldw(L - 2 - d, V[d]);
                }
                if (tail_s) b2s(L - 1, Vt);
                if (body) {
#pragma unroll
                    for (int j = L - 2; j >= 0; j--) {
                        const int d = (L - 2 - j) % PF;
                        b2s(j, V[d]);
                        if (j - PF >= 0) ldw(j - PF, V[d]);
                        __builtin_amdgcn_sched_barrier(LDSR_STEADY_SBMASK);
                    }
                }
                SCAN_TICK(4)   // steady B2
                // ---- smoothed variances of the steady region in closed form:  Vs_{T-1} = Vu,
                // Vs_t = rho Vs_{t+1} + h with rho = J^2  =>  Vs_{T-1-k} = Vs* + (Vu - Vs*) rho^k
                const int N = T - NTR;                                   // steps NTR .. T-1
                const double rho = cJ * cJ;
                const double romr = fast_rcp(1.0 - rho);
                const double Vss = ch * romr;
                const double dV = cVu - Vss;
                // (rho < 0.8 is part of the verdict and N >= 500: rho^(N-1) < 1e-48 is dropped)
                const double VsS = Vss;                                  // Vs at t = NTR
                const double sumVs = fma(dV, romr, (double)N * Vss);     // sum_{t >= NTR} Vs_t
                addPall = sumVs;                                          // :181,:183
                addTx1x = cJ * (sumVs - VsS);                             // sum_{t=NTR}^{T-2} Vs_{t+1} J_t  (:180)
                // ---- transient block backwards: composite of steps vl .. NTR-1 applied to (XsS, VsS)
                {
                    // (the block's y, u, v are read from LDS AGAIN: through a lane index the compiler cannot
                    // match with the forward block's, or it keeps all 1 + p + q values alive -- in scratch,
                    // for wide inputs -- across the steady sweeps)
                    int vl2 = vl;
                    asm volatile("" : "+v"(vl2));
                    auto tval2 = [&](int i) -> double { return tri[((i >> 1) * LPC + vl2) * 2 + (i & 1)]; };
                    double Pt = trl ? trJ : 1.0, Gt = trl ? trG : 0.0, Ht = trl ? trH : 0.0;
#define RSCAN_ROUND(n)                                                     \
                    {                                                          \
                        const double Pb = dpp1<DPP_ROW_SHL(n)>(Pt);            \
                        const double Gb = dppz<DPP_ROW_SHL(n)>(Gt);            \
                        const double Hb = dppz<DPP_ROW_SHL(n)>(Ht);            \
                        Gt = fma(Pt, Gb, Gt);                                  \
                        Ht = fma(Pt * Pt, Hb, Ht);                             \
                        Pt *= Pb;                                              \
                    }
                    RSCAN_ROUND(1) RSCAN_ROUND(2) RSCAN_ROUND(4) RSCAN_ROUND(8)
#undef RSCAN_ROUND
                    rscan_cross<LPC>(Pt, Gt, Ht, lane);     // (full products: the terminal value at t = NTR is not zero)
                    const double XsT = fma(Pt, XsS, Gt), VsT = fma(Pt * Pt, VsS, Ht);   // at step vl (vl >= NTR: at NTR)
                    const double XsN = dppd<DPP_WAVE_SHL1, 0xF>(0.0, XsT);
                    const double VsN = dppd<DPP_WAVE_SHL1, 0xF>(0.0, VsT);
                    if (trl) {
                        aTx1x = fma(XsN, XsT, fma(VsN, trJ, aTx1x));     // :180
#pragma unroll
                        for (int p_ = 0; p_ < PP; p_++) {
                            const double ut = tval2(1 + p_);
                            aTx1u[p_] = fma(XsN, ut, aTx1u[p_]);
                            aTux[p_] = fma(ut, XsT, aTux[p_]);
                        }
                        aPall += fma(XsT, XsT, VsT);
                        aSyx = fma(tval2(0), XsT, aSyx);
#pragma unroll
                        for (int q_ = 0; q_ < QQ; q_++) aSxv[q_] = fma(XsT, tval2(1 + PP + q_), aSxv[q_]);
                    }
                    X0v = XsT; V0v = VsT;                                // lane 0: Xs_0, Vs_0
                }
            } else {
                zero_sums();       // (an idle half: nothing ran)
            }
            SCAN_TICK(5)       // closed-form variance sums, transient block backwards
            PairSweepOut<PP, QQ> o;
            o.aSyx = aSyx; o.aTx1x = aTx1x; o.aPall = aPall; o.aSxx = 0.0;
            o.likq = likq; o.lsp = lsp; o.tLv = tLv; o.X0v = X0v; o.V0v = V0v; o.sneg = 0;
#pragma unroll
            for (int q_ = 0; q_ < QQ; q_++) o.aSxv[q_] = aSxv[q_];
#pragma unroll
            for (int p_ = 0; p_ < PP; p_++) { o.aTx1u[p_] = aTx1u[p_]; o.aTux[p_] = aTux[p_]; }
            pair_steady_finish<PP, QQ, QUEUE>(E, cs, o, addTx1x, addPall, st, lane, c0, nc, lastLane);
        }
        if (__builtin_expect(!__any(slow), 1)) break;      // every cell of the wave is done
        // ================================================================= G phase: generic iterations
        cs = pair_steady_g_phase<PP, QQ, L, QUEUE>(E, cs, slow, c0, nc);
    }
}

// One workgroup = up to 8 waves = up to 16 (LPC = 32) or 32 (LPC = 16) cells of ONE series.
// Static: wave w owns cells c0 + CPW w .. c0 + CPW w + CPW - 1 of the block (nc of them), CPW = 64 / LPC.  QUEUE: (c0, nc) is the series' whole range and
// every half pulls cells from the per-series counter.
template <int PP, int QQ, int L, int LPC, bool QUEUE, bool LEAD = false>
__global__ __launch_bounds__(512) void em_pair_kernel(EmParams prm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr long IMG = pair_image_doubles(L, PP, QQ, LPC);
    const int b = blockIdx.x;
    const int s = prm.blk_series[b];
    const int c0 = prm.blk_cell0[b], nc = prm.blk_ncell[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *gimg = prm.img2 + (long)s * prm.img2_stride;
    for (int i = threadIdx.x; i < (int)IMG; i += blockDim.x) smem[i] = gimg[i];
    // LEAD: the (whitened) u_t of the all-missing first prm.lead steps, [step of the lane][lane][PP],
    // behind the strips
    constexpr long STRIP = pair_strip_doubles(L);
    double *lu = smem + IMG + (long)(blockDim.x >> 6) * STRIP;
    if constexpr (LEAD) {
        const int n3 = ((prm.lead + LPC - 1) / LPC) * LPC * PP;
        const double *g3 = prm.img3 + (long)s * prm.img3_stride;
        for (int i = threadIdx.x; i < n3; i += blockDim.x) lu[i] = g3[i];
    }
    // STEADY: the first L-1 steps (lane 0's chunk) once more, one step per lane, zero beyond
    constexpr bool STEADY = !LEAD && pair_steady(L, LPC, PP, QQ);
    const double *tri = lu;
    if constexpr (STEADY) {
        constexpr int KP = scan_pairs(PP, QQ);
        constexpr int NTR = pair_steady_ntr(L, LPC);
        for (int i = threadIdx.x; i < KP * LPC * 2; i += blockDim.x) {
            const int c = i & 1, l = (i >> 1) % LPC, m = (i >> 1) / LPC;      // step l = step l % L of lane l / L
            const int vi = 2 * m + c;                                         // (an odd K leaves the last half-pair zero)
            lu[i] = (l < NTR && vi < img_values(PP, QQ)) ? gimg[img_off(l % L, vi, img_values(PP, QQ), LPC, L) + (l / L) * 2] : 0.0;
        }
    }
    __syncthreads();
    const SeriesConst *sc = prm.sc + s;
    if (sc->status != 0) {
        // singular Svv / Tuu: every cell of the block (static) or of the series (queue: block 0
        // of the series' blocks is enough, the others repeat the same writes) ends with status 2
        const int P = 6 + prm.p + prm.q;
        for (int c = threadIdx.x; c < nc; c += blockDim.x) {
            const int cell = c0 + c;
            for (int k = 0; k < P; k++) prm.theta[(long)cell * P + k] = NAN;
            prm.n_iter[cell] = 0;
            if (prm.liks && prm.liks_nanfill)
                for (int i = 0; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
            prm.lik[cell] = NAN;
            prm.status[cell] = 2;
        }
        return;
    }
    if (!QUEUE && (64 / LPC) * wave >= nc) return;    // whole wave leaves; no barrier follows
    double *hs = smem + IMG + (long)wave * STRIP + lane;
    const bool dense = sc->n_obs == prm.T;
    if constexpr (LEAD) {
        em_pair_body<PP, QQ, L, LPC, false, QUEUE, true>(prm, smem, hs, lu, s, c0, nc, lane, wave);
    } else {
        if (dense) {
            if constexpr (STEADY) em_pair_body_steady<PP, QQ, L, QUEUE>(prm, smem, hs, tri, s, c0, nc, lane, wave);
            else em_pair_body<PP, QQ, L, LPC, true, QUEUE, false>(prm, smem, hs, lu, s, c0, nc, lane, wave);
        } else {
            em_pair_body<PP, QQ, L, LPC, false, QUEUE, false>(prm, smem, hs, lu, s, c0, nc, lane, wave);
        }
    }
}

struct PairPlan {
    int L = 0;
    int lpc = 0;        // lanes per cell: 32 (two cells per wave) or 16 (four)
    int wpb = 0;        // waves per workgroup
    bool ok = false;
};
PairPlan pair_plan(int T, int PP, int QQ, int lpc, bool lead_form = false);

template <int L, int LPC>
hipError_t launch_em_pair_L(const EmParams &prm, int PPv, int QQv, int n_blocks, int wpb, bool queue,
                            hipStream_t stream);
// doubles of the lead image of one series (LEAD kernels)
__host__ __device__ constexpr long pair_lead_doubles(int lead, int LPC, int PP) {
    return (long)((lead + LPC - 1) / LPC) * LPC * PP;
}
