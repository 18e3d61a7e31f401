// ldsr_device.h -- device-side types shared by the gfx950 kernels of libldsr_hip.so.
//
// Model (reference: /root/reference/src/EM.cpp:20 "one dimensional state and output"):
//   x_{t+1} = A x_t + B u_t + w_t,  w ~ N(0,Q)         y_t = C x_t + D v_t + e_t,  e ~ N(0,R)
// p = rows of u, q = rows of v.  Kernels are instantiated on padded sizes PP, QQ in
// {1,2,4,8,16} (scan kernel: up to 8); padded rows of u / v are zero, so they contribute nothing to any sum, and the
// per-series inverse blocks are padded with identity so the padded B / D entries solve to 0.
// An absent u (v) is the same thing with p (q) = 1 and an all-zero row, which reproduces
// the reference's absent-input branches (src/EM.cpp:71-75,172,212-213) exactly.
#pragma once
#include <hip/hip_runtime.h>

#define LDSR_MAXPQ 16   // widest u / v (rows); the scan kernel is instantiated up to 8

// Theta-independent statistics of one (series, NA mask): computed once by series_prep_kernel.
// The reference recomputes Svv, Syv, Tuu on every Mstep call (src/EM.cpp:158-161,193).
struct SeriesConst {
    int n_obs;    // number of finite y_t           (src/EM.cpp:113-114,147)
    int status;   // 0, or LDSR_CELL_SINGULAR if Svv / Tuu cannot be inverted
    int t_first_obs, t_last_obs;
    double Syy;                             // sum_obs y_t^2
    double Syv[LDSR_MAXPQ];                 // sum_obs y_t v_t            (:158)
    double wv[LDSR_MAXPQ];                  // Svv^{-1} Syv
    double Svv_inv[LDSR_MAXPQ * LDSR_MAXPQ];  // (sum_obs v_t v_t')^{-1}   (:161), identity padded
    double Tuu_inv[LDSR_MAXPQ * LDSR_MAXPQ];  // (sum_{t<T-1} u_t u_t')^{-1} (:193), identity padded
    double rn_obs, rTm1;                      // 1 / n_obs, 1 / (T - 1): the divisors of R (:177) and Q (:210)
    // Whitened inputs of the scan / pair kernels (their series images hold Lv^{-1} v_t and
    // Lu^{-1} u_t): Cholesky factors Svv = Lv Lv', Tuu = Lu Lu' (lower, identity padded), their
    // inverses, and Lv^{-1} Syv.  See mstep_update_white().
    double Lv[LDSR_MAXPQ * LDSR_MAXPQ], Lv_inv[LDSR_MAXPQ * LDSR_MAXPQ];
    double Lu[LDSR_MAXPQ * LDSR_MAXPQ], Lu_inv[LDSR_MAXPQ * LDSR_MAXPQ];
    double Syv_w[LDSR_MAXPQ];
};

template <int PP, int QQ>
struct Theta {
    double A, C, Q, R, mu1, V1;
    double B[PP], D[QQ];
};

// E-step sufficient statistics consumed by the M-step (src/EM.cpp:151-152,159,180-183,190-191).
template <int PP, int QQ>
struct Sums {
    double Syx;       // sum_obs y_t Xs_t
    double Sxx;       // sum_obs Xs_t^2 + Vs_t
    double Sxv[QQ];   // sum_obs Xs_t v_t
    double Tx1x;      // sum_{t=0}^{T-2} Xs_{t+1} Xs_t + Vs_{t+1} J_t
    double Txx;       // sum_{t=0}^{T-2} Xs_t^2 + Vs_t
    double Tx1x1;     // sum_{t=1}^{T-1} Xs_t^2 + Vs_t
    double Tx1u[PP];  // sum_{t=0}^{T-2} Xs_{t+1} u_t
    double Tux[PP];   // sum_{t=0}^{T-2} u_t Xs_t
    double X0, V0;    // Xs_0, Vs_0  -> mu1, V1 (:218-219)
};

template <int PP, int QQ>
__device__ __forceinline__ void load_theta(Theta<PP, QQ> &th, const double *__restrict__ g, int p,
                                           int q, bool has_u, bool has_v) {
    th.A = g[0];
#pragma unroll
    for (int k = 0; k < PP; k++) th.B[k] = (has_u && k < p) ? g[1 + k] : 0.0;
    th.C = g[1 + p];
#pragma unroll
    for (int k = 0; k < QQ; k++) th.D[k] = (has_v && k < q) ? g[2 + p + k] : 0.0;
    th.Q = g[2 + p + q];
    th.R = g[3 + p + q];
    th.mu1 = g[4 + p + q];
    th.V1 = g[5 + p + q];
}

template <int PP, int QQ>
__device__ __forceinline__ void store_theta(const Theta<PP, QQ> &th, double *__restrict__ g, int p,
                                            int q) {
    g[0] = th.A;
#pragma unroll
    for (int k = 0; k < PP; k++)
        if (k < p) g[1 + k] = th.B[k];
    g[1 + p] = th.C;
#pragma unroll
    for (int k = 0; k < QQ; k++)
        if (k < q) g[2 + p + k] = th.D[k];
    g[2 + p + q] = th.Q;
    g[3 + p + q] = th.R;
    g[4 + p + q] = th.mu1;
    g[5 + p + q] = th.V1;
}

// 1/x to ~1 ulp: v_rcp_f64 seed (24 bits, e = 1 - x r <= 4.6e-8) + ONE third-order step
// r (1 + e + e^2) -- three FMAs where two Newton steps take four, same result quality: 99.98 % of
// results are the correctly rounded reciprocal, max error 1.1e-16 (tools/rcp_probe.hip on MI355X;
// the neglected e^3 is 1e-22).  fp64 FMAs are what the EM kernels' time is made of
// (tools/valu_rate_probe.hip: 2.06 ns per wave-instruction per SIMD against 0.85 for 32-bit ops).
__device__ __forceinline__ double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// log(x) of a finite positive double as (exponent, mantissa): x = m 2^k with m in [sqrt(1/2),
// sqrt(2)), log x = k ln 2 + log m, and log m = f - (f^2/2 - s (f^2/2 + R(s^2))) with f = m - 1,
// s = f / (2 + f): the classic fdlibm e_log.c scheme and its degree-7 minimax R (error < 1 ulp),
// restated without the special cases -- the argument here is the folded product of the innovation
// variances (em_scan_impl.h), and a zero / negative / non-finite one only has to give a
// non-finite likelihood (0 -> -inf, NaN -> NaN; a negative Sigma is flagged separately).
// ~25 fp64 operations where the library log takes ~50 plus its case analysis.
__device__ __forceinline__ double log_pos(double x) {
    int k = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s_ = f * fast_rcp(2.0 + f);
    const double z = s_ * s_, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double hfsq = 0.5 * f * f;
    const double lm = f - (hfsq - s_ * (hfsq + (t1 + t2)));
    const double r = fma((double)k, 0.69314718055994530942, lm);
    return x > 0.0 && x < INFINITY ? r : (x == 0.0 ? -INFINITY : (x > 0.0 ? INFINITY : NAN));
}

// Closed-form M-step (src/EM.cpp:139-229).  The reference solves
//   [C D] = [Syx Syv] inv([[Sxx Sxv],[Svx Svv]])   and   [A B] = [Tx1x Tx1u] inv([[Txx Txu],[Tux Tuu]])
// with a dense inverse each call.  Svv and Tuu do not depend on theta, so their inverses are
// per-series constants and the two systems reduce to a scalar Schur complement:
//   zv = Svv^{-1} Sxv',  C = (Syx - Syv zv) / (Sxx - Sxv zv),  D = Svv^{-1} Syv' - C zv
//   zu = Tuu^{-1} Tux,   A = (Tx1x - Tx1u zu) / (Txx - Txu zu), B = Tuu^{-1} Tx1u' - A zu
// R uses the algebraic form of ((y - yhat) y')/n  (:177) and Q is the reference's (:210).
// FAST: the four divisions become multiplications by rcp+Newton reciprocals (scan kernel).
// SCP: pointer to the per-series constants.  The scan kernel passes a constant-address-space
// pointer (SeriesConstK): the record is written by series_prep_kernel before the EM kernel starts
// and never changes, and only loads from that address space are scalarised -- s_load into SGPRs,
// many in flight -- inside a loop that also stores to global memory.  Through a plain pointer
// they became vector loads waited for pair by pair: 8 (q = 4) to 40 (p = 4, q = 8) dependent
// L2 round trips per EM iteration.
typedef const __attribute__((address_space(4))) SeriesConst *SeriesConstK;

template <int PP, int QQ, bool FAST = false, typename SCP = const SeriesConst *>
__device__ __forceinline__ void mstep_update(Theta<PP, QQ> &th, const Sums<PP, QQ> &S, SCP sc, int T) {
    double zv[QQ];
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < QQ; l++) a = fma(sc->Svv_inv[k * LDSR_MAXPQ + l], S.Sxv[l], a);
        zv[k] = a;
    }
    double numC = S.Syx, denC = S.Sxx;
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        numC = fma(-sc->Syv[k], zv[k], numC);
        denC = fma(-S.Sxv[k], zv[k], denC);
    }
    const double C = FAST ? numC * fast_rcp(denC) : numC / denC;
    double racc = fma(-C, S.Syx, sc->Syy);
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        const double d = fma(-C, zv[k], sc->wv[k]);
        th.D[k] = d;
        racc = fma(-d, sc->Syv[k], racc);
    }
    th.C = C;
    th.R = FAST ? racc * sc->rn_obs : racc / (double)sc->n_obs;

    double zu[PP], ru[PP];
#pragma unroll
    for (int k = 0; k < PP; k++) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int l = 0; l < PP; l++) {
            const double w = sc->Tuu_inv[k * LDSR_MAXPQ + l];
            a = fma(w, S.Tux[l], a);
            b = fma(w, S.Tx1u[l], b);
        }
        zu[k] = a;
        ru[k] = b;
    }
    double numA = S.Tx1x, denA = S.Txx;
#pragma unroll
    for (int k = 0; k < PP; k++) {
        numA = fma(-S.Tx1u[k], zu[k], numA);
        denA = fma(-S.Tux[k], zu[k], denA);
    }
    const double A = FAST ? numA * fast_rcp(denA) : numA / denA;
    double qacc = fma(-A, S.Tx1x, S.Tx1x1);
#pragma unroll
    for (int k = 0; k < PP; k++) {
        const double b = fma(-A, zu[k], ru[k]);
        th.B[k] = b;
        qacc = fma(-b, S.Tx1u[k], qacc);
    }
    th.A = A;
    th.Q = FAST ? qacc * sc->rTm1 : qacc / (double)(T - 1);
    th.mu1 = S.X0;
    th.V1 = S.V0;
}

// The same M-step in WHITENED input coordinates (scan / pair kernels).  The regression of y on
// (x, v) and of x_{t+1} on (x_t, u_t) is invariant under an invertible change of basis of the
// exogenous inputs: with v~_t = Lv^{-1} v_t (Svv = Lv Lv') the observed second moment of v~ is the
// identity, D~ = D Lv gives D~ v~_t = D v_t, and the formulas above lose both matrix-vector
// products (zv = Sxv~, Svv^{-1} Syv = Syv~): q^2 + 2 p^2 fp64 FMAs and as many scalar loads per EM
// iteration (cfg3: 96 of each) for two q x q products per CELL (white_in at load, white_out at
// store).  The series image is built from whitened inputs by series_prep_kernel; sums over it are
// Sxv~, Tux~, Tx1u~ by construction.  theta inside the EM loop holds (B~, D~).
template <int PP, int QQ, typename SCP = const SeriesConst *>
__device__ __forceinline__ void mstep_update_white(Theta<PP, QQ> &th, const Sums<PP, QQ> &S, SCP sc, int T) {
    double numC = S.Syx, denC = S.Sxx;
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        numC = fma(-sc->Syv_w[k], S.Sxv[k], numC);
        denC = fma(-S.Sxv[k], S.Sxv[k], denC);
    }
    const double C = numC * fast_rcp(denC);
    double racc = fma(-C, S.Syx, sc->Syy);
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        const double d = fma(-C, S.Sxv[k], sc->Syv_w[k]);
        th.D[k] = d;
        racc = fma(-d, sc->Syv_w[k], racc);
    }
    th.C = C;
    th.R = racc * sc->rn_obs;
    double numA = S.Tx1x, denA = S.Txx;
#pragma unroll
    for (int k = 0; k < PP; k++) {
        numA = fma(-S.Tx1u[k], S.Tux[k], numA);
        denA = fma(-S.Tux[k], S.Tux[k], denA);
    }
    const double A = numA * fast_rcp(denA);
    double qacc = fma(-A, S.Tx1x, S.Tx1x1);
#pragma unroll
    for (int k = 0; k < PP; k++) {
        const double b = fma(-A, S.Tux[k], S.Tx1u[k]);
        th.B[k] = b;
        qacc = fma(-b, S.Tx1u[k], qacc);
    }
    th.A = A;
    th.Q = qacc * sc->rTm1;
    th.mu1 = S.X0;
    th.V1 = S.V0;
}

// theta (B, D) -> whitened (B~, D~) = (B Lu, D Lv), and back with the inverse factors
template <int PP, int QQ, typename SCP>
__device__ __forceinline__ void white_in(Theta<PP, QQ> &th, SCP sc) {
    double d[QQ], b[PP];
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        double a = 0.0;
#pragma unroll
        for (int j = k; j < QQ; j++) a = fma(th.D[j], sc->Lv[j * LDSR_MAXPQ + k], a);
        d[k] = a;
    }
#pragma unroll
    for (int k = 0; k < PP; k++) {
        double a = 0.0;
#pragma unroll
        for (int j = k; j < PP; j++) a = fma(th.B[j], sc->Lu[j * LDSR_MAXPQ + k], a);
        b[k] = a;
    }
#pragma unroll
    for (int k = 0; k < QQ; k++) th.D[k] = d[k];
#pragma unroll
    for (int k = 0; k < PP; k++) th.B[k] = b[k];
}
template <int PP, int QQ, typename SCP>
__device__ __forceinline__ void white_out(Theta<PP, QQ> &th, SCP sc) {
    double d[QQ], b[PP];
#pragma unroll
    for (int k = 0; k < QQ; k++) {
        double a = 0.0;
#pragma unroll
        for (int j = k; j < QQ; j++) a = fma(th.D[j], sc->Lv_inv[j * LDSR_MAXPQ + k], a);
        d[k] = a;
    }
#pragma unroll
    for (int k = 0; k < PP; k++) {
        double a = 0.0;
#pragma unroll
        for (int j = k; j < PP; j++) a = fma(th.B[j], sc->Lu_inv[j * LDSR_MAXPQ + k], a);
        b[k] = a;
    }
#pragma unroll
    for (int k = 0; k < QQ; k++) th.D[k] = d[k];
#pragma unroll
    for (int k = 0; k < PP; k++) th.B[k] = b[k];
}

// Kernel argument block shared by the EM kernels.
struct EmParams {
    int T, p, q, has_u, has_v, niter, n_cells;
    int liks_nanfill;        // pad liks[cell][n_iter..niter) with NaN (the batch ABI); 0 = leave as is
    double tol;
    const double *yp;        // [n_series][T]       NaN = missing
    const double *yz;        // [n_series][T]       same with 0 where missing (scan kernel, global-image variant)
    const double *up;        // [n_series or 1][T][PP]  zero padded, row T-1 zeroed
    const double *vp;        // [n_series or 1][T][QQ]
    long u_stride, v_stride; // doubles per series (0 when shared)
    const double *img;       // scan kernel: [n_series] chunk-transposed series images (see em_scan_impl.h)
    long img_stride;         // doubles per series image
    const double *img2;      // pair kernel (em_pair_impl.h): [n_series] images in its 32-lane layout, or null
    long img2_stride;
    int lead;                // pair kernel: all-missing first steps of every series handled in closed form (0 = none)
    const double *img3;      // ... and their (whitened) u_t, [n_series][step of the lane][lane][PP]
    long img3_stride;
    const SeriesConst *sc;   // [n_series]
    const int *blk_series, *blk_cell0, *blk_ncell;  // block table
    int *queue;              // scan kernel: per-series cell counter (zeroed by series_prep_kernel)
    const int *perm;         // pair kernel, steady form: position -> cell (slowest cells first), or null
    const double *theta0;    // [n_cells][6+p+q]
    double *theta, *lik, *liks;
    int *n_iter, *status;
    const int *abort;        // host-pinned interrupt flag polled every 64 EM iterations, or null
    double *scratch;         // serial kernel: [T][2][scratch_stride]
    long scratch_stride;
    // FIT variant of the scan kernel (one E-step at theta0, the full fit written out)
    double *fitX, *fitY, *fitV, *fitJ;   // [n_cells][T], any may be null
    double *pen;             // lik(stdlik = FALSE) - lambda * ssq  (R/LDS_GA.R:28-44), may be null
    double lambda;
    int stdlik;              // divide lik by n_obs (src/EM.cpp:124)
};

// One system-scope load of the interrupt flag (pinned host memory written by the waiting host
// thread): bypasses the GPU caches, costs one PCIe round trip -- hence only every 64 iterations.
__device__ __forceinline__ int ldsr_poll_abort(const int *flag) {
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Series-image layout of the scan / pair kernels (described in em_scan_impl.h)
__host__ __device__ constexpr int img_values(int PP, int QQ) { return 1 + PP + QQ; }
// doubles of an image of NL virtual lanes with chunks of L steps
__host__ __device__ constexpr long img_doubles(int L, int NL, int PP, int QQ) {
    const int K = img_values(PP, QQ);
    return (long)NL * 2 * (L * (K / 2) + (K & 1) * ((L + 1) / 2));
}
// offset (doubles) of value i of step j, to which the lane adds 2 l
__host__ __device__ constexpr int img_off(int j, int i, int K, int NL, int L) {
    const int KH = K / 2;
    return i < 2 * KH ? ((j * KH + (i >> 1)) * NL) * 2 + (i & 1)
                      : (L * KH + (j >> 1)) * NL * 2 + (j & 1);
}

#define LDSR_LOG_2PI 1.8378770664093454835606594728112
