// em_scan_impl.h -- wave-per-cell, parallel-in-time EM kernel (the fast path: T <= 8192, p, q <= 8).
//
// W (1, 2 or 4) 64-lane wavefronts own one (series, restart) cell for the whole EM loop: a cell
// is spread over NL = 64 W "virtual lanes".  nl = ceil(T/L) of them are active; virtual lane l
// owns L (the first rp lanes) or L-1 consecutive time steps, so only the last step of a chunk is
// predicated.  All per-step state lives in that lane's registers and nothing but the final theta
// / lik / n_iter / status ever goes to HBM.  The series (y, u, v) comes as a chunk-transposed
// image [step][value pair][virtual lane][2] built once per launch by series_prep_kernel; the
// kernel copies it into LDS (flat coalesced copy; the 64 lanes of a wave then read consecutive
// 16-byte words: conflict-free ds_read_b128) or, when it exceeds the 160 KiB of a CU (GIMG), reads the very same
// layout straight from global memory (coalesced 512-byte rows, L2 resident).
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
//                     (B1) compose per lane, (scan) reverse scan by DPP + readlanes,
//                     (B2) serial re-run from the exact entry value, then all M-step sums
//                     (:151-193) in an independent pass.
//  then ONE wave all-reduce of every sum (M-step and likelihood), the stop rule (:272) and the
//  closed-form M-step (ldsr_device.h) redundantly in every lane.
//
// W > 1 (T > 2048): each wave scans its own 64 lanes exactly as above and the W waves of the
// cell exchange three small records per iteration through LDS (workgroup = one cell, so plain
// s_barrier; the series image is always the global one, so nothing but registers limits how many
// cells a CU holds): the wave's forward composite matrix, its reverse affine composite, and its partial
// sums.  Every wave then forms the same totals in the same order, so theta, lik and the stop
// decision are bit-identical in all waves of the cell.
//
// Chunks longer than 16 steps keep J/g/h for their second half only and re-run the first half's
// forward recursion before its backward sweep (register budget: two waves per SIMD).  Wide
// inputs (padded p + q >= 12: nine or more LDS words per step) are compiled for one wave per
// SIMD (512 registers) and carry e_t, B u_t from F1 to F2 in registers instead of re-forming them.
//
// FIT: the same machinery run for exactly one E-step at the given thetas, writing the full fit
// X, Y, V, J (Kalman_smoother, src/EM.cpp:22-131) and optionally penalized_likelihood
// (R/LDS_GA.R:28-44) -- the smoother of the winners' fits and of GA populations.
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

// Sum N independent per-lane values over the WIDTH (64 or 32) lanes of a group by recursive
// halving: in the round with exchange distance D a lane keeps one half of its live values, hands
// the other half to its partner (lane ^ D) and adds what it receives, so the number of live values
// per lane halves every round -- N/2 + N/4 + ... ~ N exchanges and fp64 adds in all, where a
// butterfly all-reduce of every value (round 1's form) took N log2(WIDTH) of each (cfg3: 25 instead
// of 150 adds and 50 instead of 300 ds_bpermute per EM iteration).  fp64 adds are what the kernels'
// time is made of; the selects that pick the halves are 32-bit moves.  The total of value v ends in
// slot 0 of lane red_home(v) of the group; callers broadcast it from there (v_readlane, or
// ds_bpermute where the two halves of a wave hold different cells).  The summation tree is
// fixed: results are run-to-run deterministic and identical in every lane.
__host__ __device__ constexpr int red_home(int v, int n, int width) {
    int lane = 0;
    for (int d = width / 2; d >= 1; d >>= 1) {
        const int nk = (n + 1) / 2;
        if (v >= nk) { lane |= d; v -= nk; }
        n = nk;
    }
    return lane;
}
// ... and the slot it ends in (0 unless there are more values than lanes in the group)
__host__ __device__ constexpr int red_slot(int v, int n, int width) {
    for (int d = width / 2; d >= 1; d >>= 1) {
        const int nk = (n + 1) / 2;
        if (v >= nk) v -= nk;
        n = nk;
    }
    return v;
}
template <int N, int D>
__device__ __forceinline__ void red_rounds(double *x, int lane);   // (below, after the DPP helpers)
// Butterfly all-reduce (round 1's form): N log2(64) adds, but no select / broadcast code -- kept
// for the FIT instantiations, which run once per winner and whose register budget (they also
// carry the fit's output pointers) the halving form's temporaries overflowed into scratch.
template <int N>
__device__ __forceinline__ void wave_sum_butterfly(double (&x)[N]) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        double t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = __shfl_xor(x[i], d, 64);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] += t[i];
    }
}
// wave-uniform totals of x[0..N) over all 64 lanes (left in x, the same in every lane)
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&x)[N]) {
    red_rounds<N, 32>(x, (int)(threadIdx.x & 63));
    const double t0 = x[0];
#pragma unroll
    for (int v = 0; v < N; v++) x[v] = readlane_d(t0, red_home(v, N, 64));   // constant lanes after unrolling
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
#ifndef LDSR_DPP_BOUND_CTRL  // zero-identity scan operands through bound_ctrl (no `old` initialisation)
#define LDSR_DPP_BOUND_CTRL 1
#endif
// Same with zero as the out-of-range value: bound_ctrl makes the hardware supply the 0, so no
// `old` register has to be initialised first (full row / bank masks only).
template <int CTRL>
__device__ __forceinline__ double dppz(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// ... and with 1.0 as the out-of-range value: its low word is 0 (hardware-supplied), only the
// high word 0x3FF00000 needs an `old` register.
template <int CTRL>
__device__ __forceinline__ double dpp1(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(src), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// The exchange of a halving round.  Distances 32 and 16 are what v_permlane32_swap / v_permlane16_swap
// (gfx950) do natively: with the kept value in the first operand and the handed-over one in the second,
// the swap leaves (own, partner's) of the half each lane keeps in the two registers -- no selects, no
// LDS.  Distances 8, 4, 2, 1 stay inside a row of 16 lanes: DPP moves (row_ror:8, two bank-masked
// row shifts by 4, quad_perm).  A ds_bpermute round trip is ~35 ns for a lone wave, a DPP move ~2 ns;
// six rounds an iteration (tools/scan_sections.py: 1438 of 7175 cycles at T = 213 before).  The
// operands of every add are the same as before, so results are bit-identical.
#ifndef LDSR_RED_DPP
#define LDSR_RED_DPP 1
#endif
#ifndef LDSR_SCAN_HI_BASE      // LDS images beyond 64 KiB: a second base register for the far half
#define LDSR_SCAN_HI_BASE 1
#endif
template <int D>
__device__ __forceinline__ int xor_dpp_w(int v) {
    static_assert(D == 1 || D == 2 || D == 4 || D == 8, "row-local distances only");
    if constexpr (D == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
    else if constexpr (D == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);    // row_ror:8
    else {   // banks 0, 2 read four lanes up, banks 1, 3 four lanes down
        const int t = __builtin_amdgcn_update_dpp(v, v, DPP_ROW_SHL(4), 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, v, DPP_ROW_SHR(4), 0xF, 0xA, false);
    }
}
template <int D>
__device__ __forceinline__ double xor_dpp(double v) {
    return __hiloint2double(xor_dpp_w<D>(__double2hiint(v)), xor_dpp_w<D>(__double2loint(v)));
}
template <int N, int D>
__device__ __forceinline__ void red_rounds(double *x, int lane) {
    if constexpr (D >= 1) {
        constexpr int NK = (N + 1) / 2, NF = N - NK;
        if constexpr (LDSR_RED_DPP && D >= 16) {
#pragma unroll
            for (int i = 0; i < NK; i++) {
                const double a = x[i], b = (i < NF) ? x[NK + i] : 0.0;   // (an odd count leaves one padded slot)
                unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
                unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
                if constexpr (D == 32) {
                    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
                    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
                    alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
                } else {
                    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
                    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
                    alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
                }
                x[i] = __hiloint2double((int)ahi, (int)alo) + __hiloint2double((int)bhi, (int)blo);
            }
        } else {
            const bool up = (lane & D) != 0;
            double snd[NK], kp[NK];
#pragma unroll
            for (int i = 0; i < NK; i++) {
                const double hi = (i < NF) ? x[NK + i] : 0.0;
                snd[i] = up ? x[i] : hi;
                kp[i] = up ? hi : x[i];
            }
#pragma unroll
            for (int i = 0; i < NK; i++) {
                if constexpr (LDSR_RED_DPP && D <= 8) snd[i] = xor_dpp<D>(snd[i]);
                else snd[i] = __shfl_xor(snd[i], D, 64);
            }
#pragma unroll
            for (int i = 0; i < NK; i++) x[i] = kp[i] + snd[i];
        }
        red_rounds<NK, D / 2>(x, lane);
    }
}

// partner matrix of a scan round; identity where the partner lane does not exist
template <int CTRL, int RM>
__device__ __forceinline__ PMat pdpp(const PMat &m) {
    PMat r;
    if constexpr (RM == 0xF && LDSR_DPP_BOUND_CTRL) {     // zero entries / words: hardware-supplied 0
        r.m00 = dpp1<CTRL>(m.m00);
        r.m11 = dpp1<CTRL>(m.m11);
        r.m22 = dpp1<CTRL>(m.m22);
        r.m01 = dppz<CTRL>(m.m01);
        r.m10 = dppz<CTRL>(m.m10);
        r.m20 = dppz<CTRL>(m.m20);
        r.m21 = dppz<CTRL>(m.m21);
    } else {
        r.m00 = dppd<CTRL, RM>(1.0, m.m00);
        r.m11 = dppd<CTRL, RM>(1.0, m.m11);
        r.m22 = dppd<CTRL, RM>(1.0, m.m22);
        r.m01 = dppd<CTRL, RM>(0.0, m.m01);
        r.m10 = dppd<CTRL, RM>(0.0, m.m10);
        r.m20 = dppd<CTRL, RM>(0.0, m.m20);
        r.m21 = dppd<CTRL, RM>(0.0, m.m21);
    }
    return r;
}

// Layout constants shared by the kernel, series_prep_kernel and the host.
//
// Lane <-> time mapping.  A cell uses nl = ceil(T/L) of its NL = 64 W virtual lanes; the first rp
// own L consecutive steps and lanes rp..nl-1 own L-1 (T = nl*(L-1) + rp), so steps 0..L-2 of
// every active lane are real and only step L-1 is predicated: the unrolled loops are
// straight-line code.  Virtual lane l starts at t0 = l*(L-1) + min(l, rp).  Requires
// L*(L-1) <= T <= NL*L (see scan_plan in kernels_scan.hip).
//
// Series image: the K = 1 + PP + QQ values of one time step -- value 0 = y (0 where missing /
// unused), values 1..PP = u (zero for t = T-1), values 1+PP.. = v -- are stored in PAIRS, 16 bytes
// per lane: pair m < KH = K/2 of step j of virtual lane l at doubles [((j*KH + m)*NL + l)*2 + {0,1}].
// Two values then come with ONE ds_read_b128 (4 LDS cycles per wave) where two separate 8-byte rows
// were fused by the compiler into ds_read2st64_b64 (8 cycles): the LDS pipe, shared by the 8 waves of
// a CU, was 65-85 % busy with those.
// An odd K leaves one value per step over.  Until round 3 it sat in a half-empty pair of its own: read
// alone it is a ds_read_b64 at a 16-byte lane stride -- lanes l and l+16 of a 32-lane group on the same
// banks, a 2-way conflict -- and the compiler fused the reads of two steps into ds_read2st64_b64 (8 LDS
// cycles + 8 of conflicts): 11.8 % of config 3's LDS cycles were bank conflicts
// (profiles/r03_cfg3_pmc_sq2.csv).  Now the odd values of steps 2jj and 2jj+1 share a pair, in a block
// behind the full pairs: [L*KH*NL*2 + (jj*NL + l)*2 + (j & 1)] -- one conflict-free ds_read_b128 per
// TWO steps, and the image is K instead of K+1 doubles per lane and step (config 3: 104 KB, was 112).
__host__ __device__ constexpr int scan_pairs(int PP, int QQ) { return (1 + PP + QQ + 1) / 2; }
// (img_values / img_doubles / img_off: ldsr_device.h, shared with series_prep_kernel)
__host__ __device__ constexpr long scan_image_doubles(int L, int W, int PP, int QQ) {
    return img_doubles(L, 64 * W, PP, QQ);
}
// per-wave exchange records of a multi-wave cell (doubles): forward composite (8), reverse
// composite (4), partial sums (XCH_SUMS), plus one slot for the queue pull
#define XCH_SUMS 40
__host__ __device__ constexpr int scan_xch_doubles(int W) { return W > 1 ? W * (8 + 4 + XCH_SUMS) + 2 : 0; }

// padded p + q >= 12: one wave per SIMD with the 512-register budget (LDSR_WIDE_OCC1), and
// e_t / B u_t kept in registers from F1 to F2 (LDSR_WIDE_EBR); both switchable for A/B builds
#ifndef LDSR_WIDE_OCC1
#define LDSR_WIDE_OCC1 0
#endif
#ifndef LDSR_WIDE_EBR
#define LDSR_WIDE_EBR 0
#endif
#ifndef LDSR_WIDE_SB      // scheduling barrier after every step of the wide kernels' sweeps
#define LDSR_WIDE_SB 0
#endif
#ifndef LDSR_EB_ALIAS        // F1 hands e_t, B u_t to F2 through the (not yet live) g_t / h_t slots
#define LDSR_EB_ALIAS 1
#endif
#ifndef LDSR_F2_BARRIER_EVERY   // scheduling barrier every n steps of F2 where the hand-over is on (0 = none)
#define LDSR_F2_BARRIER_EVERY 8
#endif
#ifndef LDSR_DENSE_F1_POW    // dense series: chunk composite = power of the 2x2 block + row recursion
#define LDSR_DENSE_F1_POW 1
#endif
#ifndef LDSR_DENSE_F1_POW_MAXP   // widest padded inputs that take that form
#define LDSR_DENSE_F1_POW_MAXP 4
#endif
#ifndef LDSR_DENSE_F1_POW_MAXQ   // (8 since round 3: with the far LDS base the reversed read order no longer costs
#define LDSR_DENSE_F1_POW_MAXQ 8  //  more than the flops save -- same box (4,8) -0.9 %, (2,8) -2.3 %, (1,8) -0.9 %)
#endif
#ifndef LDSR_GIMG_PREFETCH   // the same pipeline for the global image (same-box A/B: 2-5 % slower -- off)
#define LDSR_GIMG_PREFETCH 0
#endif
#ifndef LDSR_SCAN_PREFETCH   // round 2's one-step pipeline of the LDS reads in the long-chunk sweeps (members without the
#define LDSR_SCAN_PREFETCH 0  // ring below: it issues the next step's reads at the top of a step that then waits for its own -- off)
#endif
#ifndef LDSR_SCAN_SPF        // the image reads of the sweeps run a step or two ahead of their use (scan_spf below)
#define LDSR_SCAN_SPF 1
#endif
#ifndef LDSR_SCAN_SPF_GIMG   // ... and the global image of the multi-wave cells (T > 2048): raw buffer loads, a deeper ring
#define LDSR_SCAN_SPF_GIMG 1
#endif
#ifndef LDSR_SCAN_SPF_WIDE_LONG   // ... and wide inputs in the half-stored chunks of 17 .. 24 steps
#define LDSR_SCAN_SPF_WIDE_LONG 1
#endif
#ifndef LDSR_SCAN_SPF_MAXL   // longest chunk with the read-ahead (chunks beyond 16 steps: F2, the re-run of the first half
#define LDSR_SCAN_SPF_MAXL 32 // and both segments of B2 read through the ring too; LDS images only)
#endif
#ifndef LDSR_SCAN_SPF_MAXPQ  // widest padded p + q that takes the read-ahead at every chunk length (register budget: two
#define LDSR_SCAN_SPF_MAXPQ 8 // waves per SIMD); wider inputs -- the ring is ~50 VGPRs there -- only with chunks of <= 4 steps
#endif
__host__ __device__ constexpr bool scan_wide(int PP, int QQ) { return LDSR_WIDE_OCC1 && PP + QQ >= 12; }
// Read-ahead of the series image in the sweeps (em_scan_cell: SPF).  Left alone the scheduler issues every
// ds_read_b128 right before its use and waits for it.  Short chunks (<= 16 steps): with two waves per SIMD the
// other wave covers that, a LONE wave -- the reference's own call shape, LDS_reconstruction(num.restarts = 50): 50
// waves on 1024 SIMDs -- stands still for the LDS latency ~55 times in F1 and again in B2 (T = 813, p = q = 3:
// 4015 + 3121 of 12 908 cycles per iteration for 830 instructions, profiles/r03_scan_sections.txt).  Long chunks
// (a full scheduling barrier behind every step) and the global image of the multi-wave cells were latency bound
// with the device full as well: T = 2000 (3,3) -27 %, T = 2049 .. 8192 1.6x .. 2.4x (EXPERIMENTS.md R4.10, R4.15, R4.17).
// Which members take the ring is a matter of registers (two waves per SIMD, tools/resource_usage.py):
__host__ __device__ constexpr bool scan_spf(int PP, int QQ, int L, int W) {
    if (!LDSR_SCAN_SPF) return false;
    if (PP + QQ <= LDSR_SCAN_SPF_MAXPQ) return (W == 1 || LDSR_SCAN_SPF_GIMG) && L <= LDSR_SCAN_SPF_MAXL;
    // wide inputs (the ring is 40 .. 60 VGPRs): chunks of <= 4 steps, and the half-stored long chunks that have the room --
    // one wave per cell, 17 .. 24 steps with padded p + q <= 10, 20 steps up to 12 (tools/resource_usage.py: no spills)
    return W == 1 && (L <= 4 || (LDSR_SCAN_SPF_WIDE_LONG && ((L > 16 && L <= 24 && PP + QQ <= 10) || (L == 20 && PP + QQ <= 12))));
}
__host__ __device__ constexpr bool scan_ebr(int PP, int QQ) { return LDSR_WIDE_EBR && PP + QQ >= 12; }
__host__ __device__ constexpr bool scan_sb(int PP, int QQ) { return (LDSR_WIDE_SB || LDSR_WIDE_EBR) && PP + QQ >= 12; }

// DENSE = every y_t of the series is observed: the per-step "observed ? a : b" selects vanish.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int PP, int QQ, int L, int W, bool DENSE, bool FIT, bool GIMG>
__device__ __forceinline__ bool em_scan_cell(const EmParams &prm, const double *ys,
                                             __amdgpu_buffer_rsrc_t rs, double *xch,
                                             int s, int cell, int lane, int wv, int nl, int rp, int &wit);

// A cell the work queue handed out after the host raised the interrupt flag: not computed, but
// marked, so that an LDSR_EINTERRUPTED return never leaves stale numbers that look like results.
__device__ __forceinline__ void mark_cell_interrupted(const EmParams &prm, int cell) {
    const int P = 6 + prm.p + prm.q;
    for (int k = 0; k < P; k++) prm.theta[(long)cell * P + k] = NAN;
    if (prm.liks && prm.liks_nanfill)
        for (int i = 0; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
    prm.n_iter[cell] = 0;
    prm.lik[cell] = NAN;
    prm.status[cell] = 3;
}

// QUEUE = waves pull cells from the per-series work queue (cells converge at different
// iterations); !QUEUE = wave w of block b owns cell c0 + w (every cell runs exactly niter
// iterations, i.e. tol == 0: nothing to balance, and the queue loop costs ~6 % in spill code).
// GIMG = the series image is read from global memory instead of being copied to LDS.
// W > 1: the workgroup is ONE group of W waves working on one cell at a time.
template <int PP, int QQ, int L, int W, bool QUEUE, bool GIMG, bool FIT>
__global__ __launch_bounds__(scan_wide(PP, QQ) ? 256 : 512) void em_scan_kernel(EmParams prm) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr long IMG = scan_image_doubles(L, W, PP, QQ);
    const int b = blockIdx.x;
    const int s = prm.blk_series[b];
    const int c0 = prm.blk_cell0[b], nc = prm.blk_ncell[b];   // QUEUE: the series' cells; else the block's
    const int T = prm.T;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nl = (T + L - 1) / L;          // active virtual lanes
    const int rp = T - nl * (L - 1);         // lanes < rp own L steps, the others L-1
    const double *gimg = prm.img + (long)s * prm.img_stride;
    const double *ys;
    double *xch = nullptr;
    if constexpr (GIMG) {
        ys = gimg;
        if constexpr (W > 1) xch = smem;
    } else {
        for (int i = threadIdx.x; i < (int)IMG; i += blockDim.x) smem[i] = gimg[i];
        ys = smem;
        if constexpr (W > 1) xch = smem + IMG;
        __syncthreads();
    }
    // global image: buffer resource over the series image (out-of-range reads return 0)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gimg, 0, (int)(IMG * sizeof(double)), 0x00020000);
    const bool dense = prm.sc[s].n_obs == T && !FIT && W == 1;   // FIT / multi-wave: generic path only
    // The host's interrupt flag is polled every 64 EM iterations OF THE WAVE GROUP (not of the
    // cell: cells that converge in fewer would never poll); once it is seen, the group computes
    // nothing more and marks whatever the queue still hands it.
    int wit = 0;
    bool aborted = false;
    if constexpr (W > 1) {
        // one group: every wave of the workgroup works on the same cell
        int *qslot = reinterpret_cast<int *>(xch + W * (8 + 4 + XCH_SUMS));
        if constexpr (!QUEUE) {
            if (nc > 0) em_scan_cell<PP, QQ, L, W, false, FIT, GIMG>(prm, ys, rs, xch, s, c0, lane, wave, nl, rp, wit);
        } else {
            for (int pulls = 0; pulls <= nc; pulls++) {
                if (threadIdx.x == 0) *qslot = atomicAdd(prm.queue + s, 1);
                __syncthreads();
                const int k = __builtin_amdgcn_readfirstlane(*(volatile int *)qslot);
                __syncthreads();              // everyone has read the slot before the next pull
                if (k >= nc) break;
                if (aborted) { if (threadIdx.x == 0) mark_cell_interrupted(prm, c0 + k); continue; }
                aborted = em_scan_cell<PP, QQ, L, W, false, FIT, GIMG>(prm, ys, rs, xch, s, c0 + k, lane, wave, nl, rp, wit);
            }
        }
    } else if constexpr (!QUEUE) {
        if (wave >= nc) return;   // whole wave leaves; no barrier follows
        if (dense)
            em_scan_cell<PP, QQ, L, 1, true, FIT, GIMG>(prm, ys, rs, xch, s, c0 + wave, lane, 0, nl, rp, wit);
        else
            em_scan_cell<PP, QQ, L, 1, false, FIT, GIMG>(prm, ys, rs, xch, s, c0 + wave, lane, 0, nl, rp, wit);
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
            if (aborted) { if (lane == 0) mark_cell_interrupted(prm, c0 + k); continue; }
            if (dense)
                aborted = em_scan_cell<PP, QQ, L, 1, true, FIT, GIMG>(prm, ys, rs, xch, s, c0 + k, lane, 0, nl, rp, wit);
            else
                aborted = em_scan_cell<PP, QQ, L, 1, false, FIT, GIMG>(prm, ys, rs, xch, s, c0 + k, lane, 0, nl, rp, wit);
        }
    }
}

// LDSR_SCAN_TIMING (tools/scan_sections.py): shader-clock cycles of every section of an EM iteration,
// summed over the cell's iterations and left in the first eight slots of the cell's likelihood trace
#ifdef LDSR_SCAN_TIMING
#define SCAN_TICK(k)                                                          \
    {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const unsigned long long now_ = __builtin_readcyclecounter();         \
        tick_[k] += now_ - last_;                                             \
        last_ = now_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
    }
#elif defined(LDSR_SCAN_MARK)
// (compile-only: section boundaries as comments in the listing, for instruction counts per section)
#define SCAN_TICK(k)                                                          \
    {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
        asm volatile("; SECTION_MARK " #k);                                   \
        __builtin_amdgcn_sched_barrier(0);                                    \
    }
#else
#define SCAN_TICK(k)
#endif

template <int PP, int QQ, int L, int W, bool DENSE, bool FIT, bool GIMG>
__device__ __forceinline__ bool em_scan_cell(const EmParams &prm, const double *ys,
                                             __amdgpu_buffer_rsrc_t rs, double *xch,
                                             int s, int cell, int lane, int wv, int nl, int rp, int &wit) {
    constexpr int NL = 64 * W;
    constexpr bool EBR = scan_ebr(PP, QQ);    // e_t, B u_t stay in registers from F1 to F2
    constexpr bool SB = scan_sb(PP, QQ);
    const int vl = wv * 64 + lane;            // virtual lane
    // element (step j, row k) of this lane's chunk of y / u / v
    // value i of step j of this lane (0 = y, 1.. = u, 1+PP.. = v): see the image layout above.
    // Global image: raw buffer loads -- ONE VGPR holds the lane's byte offset and the pair offset
    // travels in the scalar soffset operand (plain global loads made the compiler keep a 64-bit
    // address pair per 4 KiB window and spill ~400 VGPRs).
    constexpr int KV = img_values(PP, QQ);
    constexpr bool BIGIMG = !GIMG && LDSR_SCAN_HI_BASE && (long)scan_image_doubles(L, W, PP, QQ) * 8 > 65536;
    int hi_pairs = 4096;       // (in 16-byte pairs, so that the far base keeps the alignment ds_read_b128 needs)
    if constexpr (BIGIMG) asm volatile("" : "+v"(hi_pairs));
    const double *ys_hi = ys + 2 * hi_pairs;
    auto val = [&](int j, int i) -> double {
        if constexpr (GIMG) {
            const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rs, vl * 16, img_off(j, i, KV, NL, L) * 8, 0);
            return __hiloint2double((int)w.y, (int)w.x);
        } else {
            // Images beyond 64 KiB: a ds_read's immediate offset has 16 bits, and the compiler formed the
            // address of every far element with its own v_add_u32 (80 per EM iteration at (4,8), L = 16).
            // A second base 64 KiB up, opaque to the compiler, serves the far half with immediates again.
            const int e = img_off(j, i, KV, NL, L);
            if (BIGIMG && e >= 8192) return ys_hi[e - 8192 + vl * 2];
            return ys[e + vl * 2];
        }
    };
    auto Yat = [&](int j) { return val(j, 0); };
    auto Uat = [&](int j, int k) { return val(j, 1 + k); };
    auto Vat = [&](int j, int k) { return val(j, 1 + PP + k); };
    // SPF: the K values of a step are read SPFD steps ahead of their use into an explicit register ring pinned by
    // scheduling barriers (the full pairs of step j in slot j % SPFN; the odd value of an odd K comes in ONE
    // 16-byte read for the two steps 2 jj, 2 jj + 1 that share its pair: slot jj % OSL).
    constexpr bool SPF = scan_spf(PP, QQ, L, W) && (!GIMG || (LDSR_SCAN_SPF_GIMG && PP + QQ <= LDSR_SCAN_SPF_MAXPQ)) && !scan_ebr(PP, QQ);
    // (global image: an L2 round trip is several steps long)
    constexpr int SPFD = GIMG ? (scan_pairs(PP, QQ) <= 2 ? 3 : 2) : (scan_pairs(PP, QQ) <= 2 ? 2 : 1);
    constexpr int KH2 = 2 * (KV / 2);              // values held in full pairs
    constexpr bool KODD = (KV & 1) != 0;
    constexpr int SPFN = SPFD + 1;                 // ring slots: step j is used while steps j+1 .. j+SPFD are in flight
    constexpr int OSL = SPFD > 2 ? 4 : 2;          // ... and those of the odd values' pairs (a pair serves two steps)
    struct StepRing {
        double w[SPFN][KH2 > 0 ? KH2 : 1];
        double o[OSL][2];
    };
    auto ring_rd = [&](double (&w)[KH2 > 0 ? KH2 : 1], double (&o)[OSL][2], int jn, bool with_odd) {
        if constexpr (GIMG) {      // one 16-byte buffer load per pair
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int m = 0; m < KH2 / 2; m++) {
                const u32x4_t q4 = __builtin_amdgcn_raw_buffer_load_b128(rs, vl * 16, img_off(jn, 2 * m, KV, NL, L) * 8, 0);
                w[2 * m] = __hiloint2double((int)q4.y, (int)q4.x);
                w[2 * m + 1] = __hiloint2double((int)q4.w, (int)q4.z);
            }
            if (KODD && with_odd) {
                const u32x4_t q4 = __builtin_amdgcn_raw_buffer_load_b128(rs, vl * 16, img_off(jn & ~1, KV - 1, KV, NL, L) * 8, 0);
                o[(jn >> 1) & (OSL - 1)][0] = __hiloint2double((int)q4.y, (int)q4.x);
                o[(jn >> 1) & (OSL - 1)][1] = __hiloint2double((int)q4.w, (int)q4.z);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KH2; i++) w[i] = val(jn, i);
            if (KODD && with_odd) {
                o[(jn >> 1) & (OSL - 1)][0] = val(jn & ~1, KV - 1);
                o[(jn >> 1) & (OSL - 1)][1] = val(jn | 1, KV - 1);
            }
        }
    };
    const int T = prm.T;
    const int P = 6 + prm.p + prm.q;
    const SeriesConst *__restrict__ sc = prm.sc + s;
    const int n_obs = sc->n_obs;
    const bool act = vl < nl;        // this lane owns time steps
    const bool tail = vl < rp;       // ... and its chunk has the L-th step
    const int lastW = (nl - 1) >> 6, lastLane = (nl - 1) & 63;   // owner of step T-1
    const bool lastOwner = W == 1 || wv == lastW;
    const int t0 = vl * (L - 1) + min(vl, rp);

    // observation mask of this lane's chunk
    unsigned obsmask = 0;
    if (!DENSE && act) {
        const double *gy = prm.yp + (long)s * T;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const double yv = (j < L - 1 || tail) ? gy[t0 + j] : NAN;
            if (isfinite(yv)) obsmask |= (1u << j);
        }
    }

    Theta<PP, QQ> th;
    load_theta(th, prm.theta0 + (long)cell * P, prm.p, prm.q, prm.has_u, prm.has_v);
    if (sc->status != 0) {     // uniform over the cell's waves: no barrier is skipped unevenly
        if (lane == 0 && wv == 0) {
            if constexpr (!FIT) {
                for (int k = 0; k < P; k++) prm.theta[(long)cell * P + k] = NAN;
                prm.n_iter[cell] = 0;
                if (prm.liks && prm.liks_nanfill)
                    for (int i = 0; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
            }
            prm.lik[cell] = NAN;
            prm.status[cell] = 2;
            if (FIT && prm.pen) prm.pen[cell] = NAN;
        }
        if constexpr (FIT) {
            if (act)
                for (int j = 0; j < L; j++)
                    if (j < L - 1 || tail) {
                        const long o = (long)cell * T + t0 + j;
                        if (prm.fitX) prm.fitX[o] = NAN;
                        if (prm.fitY) prm.fitY[o] = NAN;
                        if (prm.fitV) prm.fitV[o] = NAN;
                        if (prm.fitJ) prm.fitJ[o] = NAN;
                    }
        }
        return false;
    }

    white_in(th, (SeriesConstK)sc);   // (B, D) -> whitened input coordinates (mstep_update_white)
    double lik = NAN, lik1 = NAN, lik2 = NAN;
    int it = 0;
    bool interrupted = false;   // the host raised the interrupt flag (src/EM.cpp:261-262 polls too)
    // Per-step (J_t, g_t, h_t) kept in registers between the forward and the backward sweep.
    // Chunks longer than 16 steps (T > 1024) would need more than 256 VGPRs and drop to one
    // wave per SIMD, so for them only the second half [HS, L) is kept; the reverse composite is
    // accumulated during the forward sweep and the first half's forward recursion is re-run
    // just before its backward sweep (+~20 % flops, twice the occupancy).
    constexpr int HS = (L > 16) ? L / 2 : 0;   // L in {20, 24, 28, 32}
    constexpr int NS = L - HS;
    double Jv[NS], gv_[NS], hv[NS];
    double ev[EBR ? L : 1], buv[EBR ? L : 1];
    // Short chunks (everything stored): F1 leaves e_t = y_t - D v_t and B u_t in the slots of
    // g_t / h_t, which are not live before F2 reaches step t -- F2 neither re-reads the series
    // from LDS nor re-forms the q + p products.  No extra registers on paper; in practice the
    // scheduler postponed the J / g / h arithmetic of many steps to shorten the recurrence's
    // critical path, which kept (Xu, Vu, Xp1, z, Sigma) of all those steps alive and spilled 24-73
    // VGPRs -- a scheduling barrier every 8 steps of F2 (LDSR_F2_BARRIER_EVERY) bounds that: no
    // spills on the narrow kernels, 21-23 on (4,8).  Same-box A/B of hand-over + barrier: cfg2
    // 1.537 -> 1.41 ms, cfg3 6.08 -> 5.24, cfg5 9.18 -> 8.47, masked (1,2) 1.71 -> 1.61.
    constexpr bool EBA = (L <= 16) && !EBR && LDSR_EB_ALIAS;
    double Jfin = 0.0;      // FIT: J[T-1] of src/EM.cpp:98 (the backward recursion itself uses 0)
#ifdef LDSR_SCAN_TIMING
    unsigned long long tick_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_readcyclecounter();
#endif

    for (;;) {
        SCAN_TICK(7)       // M-step, theta broadcast, loop overhead
        const double A = th.A, C = th.C, Q = th.Q, R = th.R;
        const double A2 = A * A, C2 = C * C;
        const double rR = fast_rcp(R);
        const double C2R = C2 * rR, ACR = A * C * rR, alpha = fma(Q, C2R, A2);

        // e_t = y_t - D v_t (innovation minus C Xp) and bu_t = B u_t of step j of this lane
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

        // ... and the same from a step's values held in registers (SPF): value i of step j
        auto e_of = [&](const double (&w)[KH2 > 0 ? KH2 : 1], const double (&o)[OSL][2], int j) {
            auto vv = [&](int i) { return i < KH2 ? w[i] : o[(j >> 1) & (OSL - 1)][j & 1]; };
            double e = vv(0);
#pragma unroll
            for (int k = 0; k < QQ; k++) e = fma(-th.D[k], vv(1 + PP + k), e);
            return e;
        };
        auto bu_of = [&](const double (&w)[KH2 > 0 ? KH2 : 1], const double (&o)[OSL][2], int j) {
            auto vv = [&](int i) { return i < KH2 ? w[i] : o[(j >> 1) & (OSL - 1)][j & 1]; };
            double bu = 0.0;
#pragma unroll
            for (int k = 0; k < PP; k++) bu = fma(th.B[k], vv(1 + k), bu);
            return bu;
        };

        // ------------------------------------------------ F1: compose this lane's step matrices
        PMat M;
        M.m00 = 1.0; M.m01 = 0.0; M.m10 = 0.0; M.m11 = 1.0; M.m20 = 0.0; M.m21 = 0.0; M.m22 = 1.0;
        double eT = 0.0, buT = 0.0;      // SPF: e_t, B u_t of the predicated step L-1, kept for F2
        auto f1c = [&](int j, double e, double bu) {
            const bool o = DENSE || ((obsmask >> j) & 1u);
            if (SPF && j == L - 1) { eT = e; buT = bu; }
            if constexpr (EBR) { ev[j] = e; buv[j] = bu; }
            if (EBA && j < L - 1) { gv_[j] = e; hv[j] = bu; }   // (the predicated L-th step re-reads)
            const double a00 = o ? alpha : A2;
            const double g = o ? C2R : 0.0;
            const double s20 = o ? fma(bu, C2R, ACR * e) : 0.0;
            if (j == 0) {
                M.m00 = a00; M.m01 = Q; M.m10 = g; M.m11 = 1.0; M.m20 = s20; M.m21 = bu; M.m22 = A;
            } else {
                M = pstep(a00, Q, g, s20, bu, A, M);
            }
            if ((j & 15) == 15 && j < L - 2) prenorm(M);
            if constexpr (SB) __builtin_amdgcn_sched_barrier(0);
        };
        auto f1 = [&](int j) { f1c(j, e_at(j), bu_at(j)); };
        if constexpr (DENSE && L <= 16 && PP <= LDSR_DENSE_F1_POW_MAXP && QQ <= LDSR_DENSE_F1_POW_MAXQ && LDSR_DENSE_F1_POW) {
            // Fully observed series: every step has the SAME 2x2 block Bm = [[alpha, Q],[C2R, 1]],
            // so the chunk's 2x2 block is a power of it -- lane independent, by binary
            // exponentiation -- and only the third row (a, b, c) of the composite needs the
            // per-step recursion.  Composing from the chunk's last step towards its first,
            //   (a, b, c) <- (a, b, c) * S_j = (a alpha + b C2R + c s20_j,  a Q + b + c bu_j,  c A),
            // touches nothing but the row: 11 flops per step instead of 18.  Same-box A/B: +2 % at
            // (1,2), +1.5 % at (1,4) and (4,4); round 2 measured -6 % at (1,8) and -1 % at (4,8) (eight-row
            // inputs: the reversed read order cost more than the flops saved) and kept QQ <= 4; with the
            // far LDS base of round 3 it is +1..2 % there too: PP <= 4, QQ <= 8.
            double p00 = alpha, p01 = Q, p10 = C2R, p11 = 1.0;      // running square Bm^(2^bit)
            double q00 = 1.0, q01 = 0.0, q10 = 0.0, q11 = 1.0;      // Bm^(L-1)
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
            // Bm^L = Bm^(L-1) * Bm
            const double r00 = fma(q00, alpha, q01 * C2R), r01 = fma(q00, Q, q01);
            const double r10 = fma(q10, alpha, q11 * C2R), r11 = fma(q10, Q, q11);
            if (act) {
                // row of the last step (lanes with the L-th step), else the identity row
                double ra = 0.0, rb = 0.0, rc = 1.0;
                auto row_step = [&](int j, double e, double bu) {
                    if constexpr (EBA) { gv_[j] = e; hv[j] = bu; }
                    const double s20 = fma(bu, C2R, ACR * e);
                    const double na = fma(ra, alpha, fma(rb, C2R, rc * s20));
                    rb = fma(ra, Q, fma(rc, bu, rb));
                    ra = na;
                    rc *= A;
                };
                if constexpr (SPF) {
                    // steps L-1 (predicated), L-2, ... 0: the reads run SPFD steps ahead (step j in slot j % SPFN)
                    StepRing r;
#pragma unroll
                    for (int d = 0; d < SPFD; d++)
                        if (L - 1 - d >= 0) ring_rd(r.w[(L - 1 - d) % SPFN], r.o, L - 1 - d, d == 0 || ((L - 1 - d) & 1) != 0);
                    __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                    for (int j = L - 1; j >= 0; j--) {
                        if (j - SPFD >= 0) ring_rd(r.w[(j - SPFD) % SPFN], r.o, j - SPFD, ((j - SPFD) & 1) != 0);
                        if (j == L - 1) {
                            if (tail) {
                                const double e = e_of(r.w[j % SPFN], r.o, j), bu = bu_of(r.w[j % SPFN], r.o, j);
                                eT = e; buT = bu;
                                ra = fma(bu, C2R, ACR * e); rb = bu; rc = A;
                            }
                        } else {
                            row_step(j, e_of(r.w[j % SPFN], r.o, j), bu_of(r.w[j % SPFN], r.o, j));
                        }
                        __builtin_amdgcn_sched_barrier(0x6);
                    }
                } else {
                    if (tail) {
                        const double e = e_at(L - 1), bu = bu_at(L - 1);
                        ra = fma(bu, C2R, ACR * e); rb = bu; rc = A;
                    }
#pragma unroll
                    for (int j = L - 2; j >= 0; j--) row_step(j, e_at(j), bu_at(j));
                }
                M.m00 = tail ? r00 : q00; M.m01 = tail ? r01 : q01;
                M.m10 = tail ? r10 : q10; M.m11 = tail ? r11 : q11;
                M.m20 = ra; M.m21 = rb; M.m22 = rc;
                prenorm(M);
            }
        } else if (act) {
            if constexpr (SPF) {
                // steps 0 .. L-2, then the predicated step L-1: the reads run SPFD steps ahead
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
            } else if (L <= 16 || EBR) {
#pragma unroll
                for (int j = 0; j < L - 1; j++) f1(j);
            } else {           // no register arrays here: keep long chunks rolled (code size, VGPRs)
                f1(0);
#pragma unroll 4
                for (int j = 1; j < L - 1; j++) f1(j);
            }
            if (!SPF && tail) f1(L - 1);
            prenorm(M);
        }

        SCAN_TICK(0)       // F1
        // ------------------------------------------------ forward scan (inclusive, by lane)
        // rows of 16 lanes: Kogge-Stone by DPP row shifts (identity outside the row) ...
        M = pmul(M, pdpp<DPP_ROW_SHR(1), 0xF>(M));
        M = pmul(M, pdpp<DPP_ROW_SHR(2), 0xF>(M));
        M = pmul(M, pdpp<DPP_ROW_SHR(4), 0xF>(M));
        prenorm(M);
        M = pmul(M, pdpp<DPP_ROW_SHR(8), 0xF>(M));
        // ... then row totals: lane 15 -> row 1, lane 47 -> row 3
        M = pmul(M, pdpp<DPP_ROW_BCAST15, 0xA>(M));
        // M composes lanes 0..l (rows 0,1) or 32..l (rows 2,3) of this wave.
        // State entering the wave: the initial state, or (W > 1) the composites of the waves
        // before this one applied to it.
        double n_in = th.V1, d_in = 1.0, x_in = th.mu1;
        if constexpr (W > 1) {
            // this wave's composite = (lanes 32..63) o (lanes 0..31), formed in lane 63
            PMat Cw = pmul(M, pdpp<DPP_ROW_BCAST31, 0xC>(M));
            prenorm(Cw);
            double *fx = xch + wv * 8;
            if (lane == 63) {
                fx[0] = Cw.m00; fx[1] = Cw.m01; fx[2] = Cw.m10; fx[3] = Cw.m11;
                fx[4] = Cw.m20; fx[5] = Cw.m21; fx[6] = Cw.m22;
            }
            __syncthreads();
            for (int k = 0; k < wv; k++) {
                const double *c = xch + k * 8;
                const double nn = fma(c[0], n_in, c[1] * d_in);
                const double dd = fma(c[2], n_in, c[3] * d_in);
                const double xx = fma(c[4], n_in, fma(c[5], d_in, c[6] * x_in));
                n_in = nn; d_in = dd; x_in = xx;
            }
            n_in = uniform_d(n_in); d_in = uniform_d(d_in); x_in = uniform_d(x_in);
        }
        // State after this lane's chunk: rows 0,1 apply M to the wave's entry state, rows 2,3 to
        // the state of lane 31 (a matrix-vector product instead of a sixth matrix-matrix round).
        double n_e = fma(M.m00, n_in, M.m01 * d_in);
        double d_e = fma(M.m10, n_in, M.m11 * d_in);
        double x_e = fma(M.m20, n_in, fma(M.m21, d_in, M.m22 * x_in));
        {
            const double n_b = dppd<DPP_ROW_BCAST31, 0xC>(n_in, n_e);
            const double d_b = dppd<DPP_ROW_BCAST31, 0xC>(d_in, d_e);
            const double x_b = dppd<DPP_ROW_BCAST31, 0xC>(x_in, x_e);
            n_e = fma(M.m00, n_b, M.m01 * d_b);
            d_e = fma(M.m10, n_b, M.m11 * d_b);
            x_e = fma(M.m20, n_b, fma(M.m21, d_b, M.m22 * x_b));
        }
        // shift by one lane: the entry state of lane l is the exit state of lane l-1; lane 0
        // gets the wave's entry state
        n_e = dppd<DPP_WAVE_SHR1, 0xF>(n_in, n_e);
        d_e = dppd<DPP_WAVE_SHR1, 0xF>(d_in, d_e);
        x_e = dppd<DPP_WAVE_SHR1, 0xF>(x_in, x_e);
        double Xp, Vp;
        {
            const double rd = fast_rcp(d_e);
            Vp = n_e * rd;
            Xp = x_e * rd;
        }

        SCAN_TICK(1)       // forward scan
        // ------------------------------------------------ F2: serial re-run from the exact entry
        // log-determinant: running product of the observed Sigma_t, folded into (mantissa,
        // exponent) every 8 steps so that it can neither overflow nor underflow (:122 sums logs)
        double likq = 0.0, sprod = 1.0, Xu = 0.0, Vu = 0.0;
        int sexp = 0;
        int sneg = 0;   // OR of the sign words of every observed Sigma_t
        double sg = fma(C2, Vp, R);     // Sigma_t of the current step (src/EM.cpp:119)
        double r0 = fast_rcp(sg);
        const double Xp0 = Xp, Vp0 = Vp, sg0 = sg, r00 = r0;   // entry state (re-run of [0, HS))
        double Pi = 1.0, G = 0.0, H = 0.0;                     // reverse composite (HS > 0 only)
        // Long chunks end every step with a scheduling barrier (register pressure), which would
        // also pin each step's LDS reads right before their use: software-pipeline them one step
        // ahead instead (PF) -- the reads of step j+1 are issued at the top of step j.
        constexpr bool PF = L > 16 && !EBR && (LDSR_SCAN_PREFETCH || (GIMG && LDSR_GIMG_PREFETCH));
        double e_nx = 0.0, bu_nx = 0.0;
        if (PF && act) { e_nx = e_at(0); bu_nx = bu_at(0); }
        auto f2c = [&](int j, double e, double bu) {
            const bool o = DENSE || ((obsmask >> j) & 1u);
            const double r = o ? r0 : 0.0;             // 1/Sigma_t; 0 = "no update" (:82-84)
            const double sl = o ? sg : 1.0;
            sprod *= sl;
            sneg |= __double2hiint(sl);
            if ((j & 7) == 7) {
                sexp += __builtin_amdgcn_frexp_exp(sprod);
                sprod = __builtin_amdgcn_frexp_mant(sprod);
            }
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
                const bool fin = lastOwner && (lane == lastLane) && (j == (tail ? L - 1 : L - 2));
                if (FIT && fin) Jfin = J;              // J[T-1] = Vu A / (A Vu A + Q)   :98
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
            if (L > 16 || SB) __builtin_amdgcn_sched_barrier(0);
            if (EBA && LDSR_F2_BARRIER_EVERY > 0 && (j % LDSR_F2_BARRIER_EVERY) == LDSR_F2_BARRIER_EVERY - 1)
                __builtin_amdgcn_sched_barrier(0);
        };
        auto f2 = [&](int j) {
            double e, bu;
            if constexpr (PF) {
                e = e_nx; bu = bu_nx;
                if (j + 1 < L) { e_nx = e_at(j + 1); bu_nx = bu_at(j + 1); }
            } else if (EBA && j < L - 1) {
                e = gv_[j]; bu = hv[j];                // left there by F1; overwritten below by g_t, h_t
            } else if (SPF) {
                e = eT; bu = buT;                      // (the predicated step: kept by F1)
            } else {
                e = EBR ? ev[j] : e_at(j);
                bu = EBR ? buv[j] : bu_at(j);
            }
            f2c(j, e, bu);
        };
        if (act) {
            if constexpr (SPF && L > 16) {
                // long chunks: F2 reads the image itself (no hand-over from F1) -- through the ring
                StepRing r;
#pragma unroll
                for (int d = 0; d < SPFD; d++) ring_rd(r.w[d % SPFN], r.o, d, (d & 1) == 0);
                __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                for (int j = 0; j < L - 1; j++) {
                    if (j + SPFD <= L - 1) ring_rd(r.w[(j + SPFD) % SPFN], r.o, j + SPFD, ((j + SPFD) & 1) == 0);
                    f2c(j, e_of(r.w[j % SPFN], r.o, j), bu_of(r.w[j % SPFN], r.o, j));
                }
                if (tail) f2c(L - 1, e_of(r.w[(L - 1) % SPFN], r.o, L - 1), bu_of(r.w[(L - 1) % SPFN], r.o, L - 1));
            } else {
#pragma unroll
                for (int j = 0; j < L - 1; j++) f2(j);
                if (tail) f2(L - 1);
            }
        }
        // Xs^2 + Vs at T-1 (0 in the waves that do not own it)
        double termLast = readlane_d(fma(Xu, Xu, Vu), lastLane);
        if (W > 1 && !lastOwner) termLast = 0.0;

        // The two likelihood sums ride along with the M-step sums in ONE wave reduction at the end
        // of the iteration; the backward sweep of the final iteration is therefore redundant
        // (1 of n_iter sweeps) but every iteration saves 6 dependent cross-lane rounds.
        const double lsp = fma((double)sexp, 0.69314718055994530942, log_pos(sprod));

        SCAN_TICK(2)       // F2
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
            const double Pb = LDSR_DPP_BOUND_CTRL ? dpp1<DPP_ROW_SHL(n)>(Pi) : dppd<DPP_ROW_SHL(n), 0xF>(1.0, Pi); \
            const double Gb = LDSR_DPP_BOUND_CTRL ? dppz<DPP_ROW_SHL(n)>(G) : dppd<DPP_ROW_SHL(n), 0xF>(0.0, G); \
            const double Hb = LDSR_DPP_BOUND_CTRL ? dppz<DPP_ROW_SHL(n)>(H) : dppd<DPP_ROW_SHL(n), 0xF>(0.0, H); \
            G = fma(Pi, Gb, G);                                            \
            H = fma(Pi * Pi, Hb, H);                                       \
            Pi *= Pb;                                                      \
        }
        RSCAN_ROUND(1) RSCAN_ROUND(2) RSCAN_ROUND(4) RSCAN_ROUND(8)
#undef RSCAN_ROUND
        // ... then across rows: lanes 16, 32, 48 hold the composites T1, T2, T3 of rows 1..3;
        // every lane applies the composite of all rows after its own (uniform values)
        double Xt = 0.0, Vt = 0.0;     // Xs, Vs just after this wave's last step (0 = terminal)
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
            if constexpr (W > 1) {
                // (G, H) assume a zero value after the wave's last lane; the true one comes from
                // the waves after this one: publish (product of J, G, H) of the whole wave (lane 0)
                const double P3 = readlane_d(Pi, 48);
                const double P23 = P2 * P3, P123 = P1 * P23;
                const double Ptot = Pi * (row == 0 ? P123 : row == 1 ? P23 : row == 2 ? P3 : 1.0);
                double *bx = xch + W * 8 + wv * 4;
                if (lane == 0) { bx[0] = Ptot; bx[1] = G; bx[2] = H; }
                __syncthreads();
                for (int k = W - 1; k > wv; k--) {
                    const double *c = xch + W * 8 + k * 4;
                    Xt = fma(c[0], Xt, c[1]);
                    Vt = fma(c[0] * c[0], Vt, c[2]);
                }
                Xt = uniform_d(Xt); Vt = uniform_d(Vt);
                G = fma(Ptot, Xt, G);
                H = fma(Ptot * Ptot, Vt, H);
            }
        }
        // (G, H) = (Xs, Vs) at the first step of this lane's chunk; the entry for lane l is lane
        // l+1's value, and the value after the wave for lane 63 (inactive lanes hold identity
        // maps, so they pass the terminal value through)
        double Xn = dppd<DPP_WAVE_SHL1, 0xF>(Xt, G);
        double Vn = dppd<DPP_WAVE_SHL1, 0xF>(Vt, H);

        SCAN_TICK(3)       // B1 + reverse scan
        // ------------------------------------------------ B2: serial reverse re-run + M-step sums
        double aSyx = 0.0, aSxx = 0.0, aTx1x = 0.0, aPall = 0.0, term = 0.0, aSsq = 0.0;
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
        auto b2b_v = [&](int j, int i, bool top, auto &&vv) {   // j = step in the chunk, i = storage index, vv(k) = value k of the step
            const bool o = DENSE || ((obsmask >> j) & 1u);
            const double J = Jv[i], Xs = gv_[i], Vs = hv[i];
            const double Xnx = top ? XnE : gv_[top ? i : i + 1];
            const double Vnx = top ? VnE : hv[top ? i : i + 1];
            aTx1x = fma(Xnx, Xs, fma(Vnx, J, aTx1x));   // :180  (zero at t = T-1)
            double bu = 0.0;
#pragma unroll
            for (int k = 0; k < PP; k++) {
                const double ut = vv(1 + k);                      // zero at t = T-1
                aTx1u[k] = fma(Xnx, ut, aTx1u[k]);                // :190
                aTux[k] = fma(ut, Xs, aTux[k]);                   // :191
                if (FIT) bu = fma(th.B[k], ut, bu);
            }
            term = fma(Xs, Xs, Vs);
            aPall += term;                                        // :181,:183
            const double xo = o ? Xs : 0.0;
            aSyx = fma(vv(0), xo, aSyx);                          // :151
            if (!DENSE) aSxx += o ? term : 0.0;                   // :152
            double dv = 0.0;
#pragma unroll
            for (int k = 0; k < QQ; k++) {
                const double vt = vv(1 + PP + k);
                aSxv[k] = fma(xo, vt, aSxv[k]);                   // :159
                if (FIT) dv = fma(th.D[k], vt, dv);
            }
            if constexpr (FIT) {
                // the fit of src/EM.cpp:126-130 at time t0 + j, and the penalty term of
                // R/LDS_GA.R:34-40 (steps t < T-1 only)
                const bool fin = lastOwner && (lane == lastLane) && (j == (tail ? L - 1 : L - 2));
                const long o_ = (long)cell * T + t0 + j;
                if (prm.fitX) prm.fitX[o_] = Xs;
                if (prm.fitV) prm.fitV[o_] = Vs;
                if (prm.fitJ) prm.fitJ[o_] = fin ? Jfin : J;
                if (prm.fitY) prm.fitY[o_] = fma(th.C, Xs, dv);   // :106-110
                const double d = Xnx - fma(th.A, Xs, bu);
                aSsq = fin ? aSsq : fma(d, d, aSsq);
            }
            if ((L > 16 && (j & 3) == 3) || SB) __builtin_amdgcn_sched_barrier(0);
        };
        auto b2b = [&](int j, int i, bool top) { b2b_v(j, i, top, [&](int k) { return val(j, k); }); };
        if (act) {
            if constexpr (SPF && HS > 0) {
                // long chunks, stored half [HS, L): steps L-1 (predicated) .. HS through the ring
                if (tail) b2a(NS - 1);
                else { gv_[NS - 1] = XnE; hv[NS - 1] = VnE; }
#pragma unroll
                for (int i = NS - 2; i >= 0; i--) b2a(i);
                StepRing r;
#pragma unroll
                for (int d = 0; d < SPFD; d++) ring_rd(r.w[(L - 1 - d) % SPFN], r.o, L - 1 - d, d == 0 || ((L - 1 - d) & 1) != 0);
                __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                for (int j = L - 1; j >= HS; j--) {
                    if (j - SPFD >= HS) ring_rd(r.w[(j - SPFD) % SPFN], r.o, j - SPFD, ((j - SPFD) & 1) != 0);
                    auto vv = [&](int k) { return k < KH2 ? r.w[j % SPFN][k] : r.o[(j >> 1) & (OSL - 1)][j & 1]; };
                    if (j == L - 1) { if (tail) b2b_v(L - 1, NS - 1, true, vv); }
                    else b2b_v(j, j - HS, false, vv);
                    __builtin_amdgcn_sched_barrier(0x6);
                }
            } else if constexpr (SPF) {
                // pass 2 reads the image again, last step first: the first reads are issued ahead of pass 1
                static_assert(HS == 0, "short chunks only");
                StepRing r;
#pragma unroll
                for (int d = 0; d < SPFD; d++)
                    if (L - 1 - d >= 0) ring_rd(r.w[(L - 1 - d) % SPFN], r.o, L - 1 - d, d == 0 || ((L - 1 - d) & 1) != 0);
                __builtin_amdgcn_sched_barrier(0x6);
                if (tail) b2a(NS - 1);
                else { gv_[NS - 1] = XnE; hv[NS - 1] = VnE; }
#pragma unroll
                for (int i = NS - 2; i >= 0; i--) b2a(i);
                __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                for (int j = L - 1; j >= 0; j--) {
                    if (j - SPFD >= 0) ring_rd(r.w[(j - SPFD) % SPFN], r.o, j - SPFD, ((j - SPFD) & 1) != 0);
                    auto vv = [&](int k) { return k < KH2 ? r.w[j % SPFN][k] : r.o[(j >> 1) & (OSL - 1)][j & 1]; };
                    if (j == L - 1) { if (tail) b2b_v(L - 1, NS - 1, true, vv); }
                    else b2b_v(j, j, false, vv);
                    __builtin_amdgcn_sched_barrier(0x6);
                }
            } else {
                if (tail) b2a(NS - 1);
                else { gv_[NS - 1] = XnE; hv[NS - 1] = VnE; }  // "next" of step L-2 for short chunks
#pragma unroll
                for (int i = NS - 2; i >= 0; i--) b2a(i);
                if (tail) b2b(L - 1, NS - 1, true);
#pragma unroll
                for (int i = NS - 2; i >= 0; i--) b2b(HS + i, i, false);
            }
        }
        if (HS > 0) {
            // re-run the forward recursion of steps [0, HS) from the lane's entry state; the
            // likelihood terms of these steps were already accumulated in F2
            double Xq = Xp0, Vq = Vp0, sgq = sg0, rq = r00;
            if (PF && act) { e_nx = e_at(0); bu_nx = bu_at(0); }
            auto f2rc = [&](int j, double e, double bu) {
                const bool o = DENSE || ((obsmask >> j) & 1u);
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
            auto f2r = [&](int j) {
                double e, bu;
                if constexpr (PF) {
                    e = e_nx; bu = bu_nx;
                    if (j + 1 < HS) { e_nx = e_at(j + 1); bu_nx = bu_at(j + 1); }
                } else {
                    e = EBR ? ev[j] : e_at(j);
                    bu = EBR ? buv[j] : bu_at(j);
                }
                f2rc(j, e, bu);
            };
            if (act) {
                if constexpr (SPF) {
                    StepRing r;
#pragma unroll
                    for (int d = 0; d < SPFD; d++) ring_rd(r.w[d % SPFN], r.o, d, (d & 1) == 0);
                    __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                    for (int j = 0; j < HS; j++) {
                        if (j + SPFD < HS) ring_rd(r.w[(j + SPFD) % SPFN], r.o, j + SPFD, ((j + SPFD) & 1) == 0);
                        f2rc(j, e_of(r.w[j % SPFN], r.o, j), bu_of(r.w[j % SPFN], r.o, j));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < HS; j++) f2r(j);
                }
                XnE = Xn;                      // Xs, Vs at step HS: the entry of this segment
                VnE = Vn;
#pragma unroll
                for (int i = HS - 1; i >= 0; i--) b2a(i);
                if constexpr (SPF) {
                    StepRing r;
#pragma unroll
                    for (int d = 0; d < SPFD; d++) ring_rd(r.w[(HS - 1 - d) % SPFN], r.o, HS - 1 - d, d == 0 || ((HS - 1 - d) & 1) != 0);
                    __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                    for (int j = HS - 1; j >= 0; j--) {
                        if (j - SPFD >= 0) ring_rd(r.w[(j - SPFD) % SPFN], r.o, j - SPFD, ((j - SPFD) & 1) != 0);
                        b2b_v(j, j, j == HS - 1, [&](int k) { return k < KH2 ? r.w[j % SPFN][k] : r.o[(j >> 1) & (OSL - 1)][j & 1]; });
                        __builtin_amdgcn_sched_barrier(0x6);
                    }
                } else {
                    b2b(HS - 1, HS - 1, true);
#pragma unroll
                    for (int i = HS - 2; i >= 0; i--) b2b(i, i, false);
                }
            }
        }
        const double Xs = Xn, Vs = Vn;         // Xs_t, Vs_t at the first step of the chunk
        SCAN_TICK(4)       // B2
        Sums<PP, QQ> S;
        double ssq = 0.0;
        {
            constexpr int NB = 5 + (DENSE ? 0 : 1);          // fixed part
            constexpr int NR = NB + QQ + 2 * PP + (FIT ? 1 : 0);
            static_assert(NR + 6 <= XCH_SUMS, "exchange record too small");
            double red[NR];
            red[0] = aSyx; red[1] = aTx1x; red[2] = aPall; red[3] = likq; red[4] = lsp;
            if (!DENSE) red[5] = aSxx;
#pragma unroll
            for (int k = 0; k < QQ; k++) red[NB + k] = aSxv[k];
#pragma unroll
            for (int k = 0; k < PP; k++) { red[NB + QQ + k] = aTx1u[k]; red[NB + QQ + PP + k] = aTux[k]; }
            if (FIT) red[NR - 1] = aSsq;
            if constexpr (FIT) wave_sum_butterfly<NR>(red);
            else wave_sum_n<NR>(red);
            double term0 = readlane_d(term, 0);        // Xs^2 + Vs at t = 0 (wave 0)
            S.X0 = readlane_d(Xs, 0);                  // :218  (wave 0)
            S.V0 = readlane_d(Vs, 0);                  // :219
            bool neg = __any(sneg < 0);                // log of a negative Sigma in the reference
            int abort_now = 0;
            if (!FIT && prm.abort && ((++wit) & 63) == 0) {
                // every wave of a multi-wave cell must see the SAME value (barrier counts):
                // wave 0 polls, the flag travels with the sums record
                if (wv == 0) abort_now = __builtin_amdgcn_readfirstlane(lane == 0 ? ldsr_poll_abort(prm.abort) : 0);
            }
            if constexpr (W > 1) {
                // partial sums of the W waves -> totals, formed by every wave in the same order
                double *sx = xch + W * 12 + wv * XCH_SUMS;
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NR; i++) sx[i] = red[i];
                    sx[NR] = termLast; sx[NR + 1] = term0; sx[NR + 2] = S.X0; sx[NR + 3] = S.V0;
                    sx[NR + 4] = neg ? 1.0 : 0.0;
                    if (wv == 0) sx[NR + 5] = abort_now ? 1.0 : 0.0;
                }
                __syncthreads();
                const double *s0 = xch + W * 12;
#pragma unroll
                for (int i = 0; i < NR; i++) {
                    double a = s0[i];
                    for (int k = 1; k < W; k++) a += s0[k * XCH_SUMS + i];
                    red[i] = uniform_d(a);
                }
                double tl = 0.0, ng = 0.0;
                for (int k = 0; k < W; k++) { tl += s0[k * XCH_SUMS + NR]; ng += s0[k * XCH_SUMS + NR + 4]; }
                termLast = uniform_d(tl);
                neg = __builtin_amdgcn_readfirstlane((int)(ng > 0.0)) != 0;
                term0 = uniform_d(s0[NR + 1]);
                S.X0 = uniform_d(s0[NR + 2]);
                S.V0 = uniform_d(s0[NR + 3]);
                abort_now = __builtin_amdgcn_readfirstlane((int)(s0[NR + 5] > 0.0));
            }
            interrupted = abort_now != 0;
            S.Syx = red[0]; S.Tx1x = red[1];
            S.Sxx = DENSE ? red[2] : red[5];
#pragma unroll
            for (int k = 0; k < QQ; k++) S.Sxv[k] = red[NB + k];
#pragma unroll
            for (int k = 0; k < PP; k++) { S.Tx1u[k] = red[NB + QQ + k]; S.Tux[k] = red[NB + QQ + PP + k]; }
            if (FIT) ssq = red[NR - 1];
            S.Txx = red[2] - termLast;                  // t = 0 .. T-2
            S.Tx1x1 = red[2] - term0;                   // t = 1 .. T-1
            // likelihood (:113-124)
            lik2 = lik1;
            lik1 = lik;
            lik = -0.5 * n_obs * LDSR_LOG_2PI - 0.5 * (red[3] + red[4]);
            if (!FIT || prm.stdlik) lik = lik / n_obs;
            if (neg) lik = NAN;
            if constexpr (FIT) {
                if (prm.pen && lane == 0 && wv == 0) {
                    const double full = (prm.stdlik ? lik * n_obs : lik);
                    prm.pen[cell] = full - prm.lambda * ssq;
                }
            }
        }
        SCAN_TICK(5)       // reduction, likelihood
        if constexpr (FIT) {
            it = 1;
            break;
        }
        if (prm.liks && lane == 0 && wv == 0) prm.liks[(long)cell * prm.niter + it] = lik;
        it++;
        bool stop = it >= prm.niter || interrupted;
        if (it >= 3 && fabs(lik - lik1) < prm.tol && fabs(lik1 - lik2) < prm.tol) stop = true;  // :272
        if (__builtin_amdgcn_readfirstlane((int)stop)) break;   // theta stays the one that produced this fit
        mstep_update_white<PP, QQ>(th, S, (SeriesConstK)sc, T);
        // theta is wave-uniform by construction; say so to the compiler (SGPR residency)
        th.A = uniform_d(th.A); th.C = uniform_d(th.C); th.Q = uniform_d(th.Q);
        th.R = uniform_d(th.R); th.mu1 = uniform_d(th.mu1); th.V1 = uniform_d(th.V1);
#pragma unroll
        for (int k = 0; k < PP; k++) th.B[k] = uniform_d(th.B[k]);
#pragma unroll
        for (int k = 0; k < QQ; k++) th.D[k] = uniform_d(th.D[k]);
    }

    if (lane == 0 && wv == 0) {
        if constexpr (!FIT) {
            white_out(th, (SeriesConstK)sc);
            store_theta(th, prm.theta + (long)cell * P, prm.p, prm.q);
            if (prm.liks && prm.liks_nanfill)
                for (int i = it; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
            prm.n_iter[cell] = it;
#ifdef LDSR_SCAN_TIMING
            if (prm.liks && prm.niter >= 8)
                for (int k = 0; k < 8; k++) prm.liks[(long)cell * prm.niter + k] = (double)tick_[k];
#endif
        }
        prm.lik[cell] = lik;
        prm.status[cell] = (interrupted && it < prm.niter) ? 3 : (isfinite(lik) ? 0 : 1);
    }
    return interrupted;
}

// Does the (L, W, PP, QQ) member read its series image from global memory?  When the image does not fit a
// CU's LDS (L >= 20 only: the short-chunk images always fit) and for every multi-wave cell.  ONE rule for
// the plan (kernels_scan.hip) and for what is compiled (em_scan_launch.inc): a member exists in the form
// the plan uses and in no other.
__host__ __device__ constexpr bool scan_image_fits_lds(int L, int W, int PP, int QQ) {
    return (scan_image_doubles(L, W, PP, QQ) + scan_xch_doubles(W)) * 8 <= 160 * 1024;
}
__host__ __device__ constexpr bool scan_uses_gimg(int L, int W, int PP, int QQ) {
    return L >= 20 && (W > 1 || !scan_image_fits_lds(L, W, PP, QQ));
}

// Launch plan of a (T, PP, QQ) shape: chunk length, waves per cell, cells per workgroup, and
// whether the series image is read from global memory (kernels_scan.hip).
struct ScanPlan {
    int L = 0, W = 0;
    int cpb = 0;        // cells (= wave groups) per workgroup
    bool gimg = false;
    bool ok = false;
};
ScanPlan scan_plan(int T, int PP, int QQ);

template <int L, int W>
hipError_t launch_em_scan_LW(const EmParams &prm, int PPv, int QQv, int n_blocks, int cpb,
                             bool queue, bool gimg, bool fit, hipStream_t stream);
