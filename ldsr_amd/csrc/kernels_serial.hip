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
#include <cstdio>

#include "ldsr_kernels.h"

// ---------------------------------------------------------------------------------------
// series_prep_kernel: one block per series.  Builds the zero-padded time-major copies of
// u and v, and the theta-independent statistics (SeriesConst).
// ---------------------------------------------------------------------------------------
// The small dense factorizations of series_prep, done by ONE WAVE (lane = column j, four row groups):
// one thread doing them alone walks O(n^3) dependent LDS round trips -- 50 us of a 120 us series_prep at
// q = 8, and the 16 x 16 padded products behind them another 25 us whatever q is (round 4).  Every
// element sees the same operations in the same order as in the one-thread form, so the results are
// unchanged.  `volatile`: the lanes exchange values through LDS in program order (one wave: lockstep,
// in-order LDS), which per-thread alias analysis must not second-guess.
typedef volatile double *vdp;

// Gauss-Jordan with partial pivoting on an n x n matrix (stride LDSR_MAXPQ, n <= 16) in place; entries
// outside n x n are left untouched (identity padding); `inv`: LDS workspace.  False if a pivot is 0.
__device__ static bool invert_small(double *a_, double *inv_, int n, int lane) {
    vdp a = a_, inv = inv_;
    constexpr int M = LDSR_MAXPQ;
    const int j = lane & 15, rg = lane >> 4;
    const bool col = j < n;
    for (int r = rg; r < n; r += 4)
        if (col) inv[r * M + j] = (r == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; c++) {
        int piv = c;                                       // (every lane finds the same pivot)
        double best = fabs(a[c * M + c]);
        for (int r = c + 1; r < n; r++) {
            const double m = fabs(a[r * M + c]);
            if (m > best) { best = m; piv = r; }
        }
        if (!(best > 0.0) || !isfinite(best)) return false;
        if (piv != c && rg == 0 && col) {
            double t = a[c * M + j];
            a[c * M + j] = a[piv * M + j];
            a[piv * M + j] = t;
            t = inv[c * M + j];
            inv[c * M + j] = inv[piv * M + j];
            inv[piv * M + j] = t;
        }
        const double d = 1.0 / a[c * M + c];               // (read by every lane before lane c rescales it)
        if (rg == 0 && col) {
            a[c * M + j] = a[c * M + j] * d;
            inv[c * M + j] = inv[c * M + j] * d;
        }
        for (int r = rg; r < n; r += 4) {
            if (r == c || !col) continue;
            const double f = a[r * M + c];                 // (read by the row's lanes before lane c clears it)
            a[r * M + j] = a[r * M + j] - f * a[c * M + j];
            inv[r * M + j] = inv[r * M + j] - f * inv[c * M + j];
        }
    }
    for (int r = rg; r < n; r += 4)
        if (col) a[r * M + j] = inv[r * M + j];
    return true;
}

// Cholesky factor of the n x n SPD matrix `a` (stride LDSR_MAXPQ): on return `l` holds the lower
// factor and `li` its inverse, both zero above the diagonal inside n x n and untouched
// (identity padding) outside.  One wave: lane i owns row i of l, then column i of li.  False if a
// pivot is not positive.
__device__ static bool chol_small(const double *a_, double *l_, double *li_, int n, int lane) {
    const volatile double *a = a_;
    vdp l = l_, li = li_;
    constexpr int M = LDSR_MAXPQ;
    const int i = lane;
    if (i < n)
        for (int j = 0; j < n; j++) { l[i * M + j] = 0.0; li[i * M + j] = 0.0; }
    for (int j = 0; j < n; j++) {
        double d = a[j * M + j];                           // (every lane: the same value)
        for (int k = 0; k < j; k++) d -= l[j * M + k] * l[j * M + k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        const double ljj = sqrt(d);
        if (i == j) l[j * M + j] = ljj;
        if (i > j && i < n) {
            double e = a[i * M + j];
            for (int k = 0; k < j; k++) e -= l[i * M + k] * l[j * M + k];
            l[i * M + j] = e / ljj;
        }
    }
    if (i < n) {                                           // forward substitution, column c = i of the inverse
        const int c = i;
        for (int r = c; r < n; r++) {
            double e = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; k++) e -= l[r * M + k] * li[k * M + c];
            li[r * M + c] = e / l[r * M + r];
        }
    }
    return true;
}

__device__ __forceinline__ double prep_wave_sum(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}


// ---------------------------------------------------------------------------------------
// Cell order for the steady form of the pair kernel (em_pair_impl.h em_pair_body_steady).
// A fully observed cell takes the steady sweeps in an EM iteration when the prior-variance
// recursion Vp_{t+1} = (alpha Vp_t + Q) / (g Vp_t + 1), g = C^2 / R, alpha = A^2 + Q g
// (src/EM.cpp:76,86-88 without missing steps) has settled within the kernel's transient block.
// The recursion is ONE Moebius map iterated: with its fixed points V+ > 0 > V- and
// rho = lambda- / lambda+ = A^2 / lambda+^2 (eigenvalues of [[alpha, Q],[g, 1]]) the cross ratio
// (V_t - V+) / (V_t - V-) = rho^t (V_1 - V+) / (V_1 - V-) gives the number of steps until the
// relative distance to V+ is below 2^-48 in closed form.  Cells that need more than the block
// run generic iterations first (16 % of BASELINE config 2's cells at theta0, for 2..31 iterations).
// Sorting the cells by that prediction puts slow cells into the same waves (a wave of two slow
// cells runs ONE generic iteration for both; a slow and a fast cell make the fast one wait) and
// lets the work queue hand out the longest jobs first.  Only the schedule depends on the order:
// a cell's arithmetic does not (tests/test_gpu_pair.py).
__device__ static double steady_settle_steps(double A, double C, double Q, double R, double V1) {
    const double A2 = A * A, g = C * C / R, alpha = A2 + Q * g;
    const double d = alpha - 1.0, disc = d * d + 4.0 * Q * g, sq = sqrt(disc);
    const double lp = 0.5 * (alpha + 1.0 + sq), rho = A2 / (lp * lp);
    if (!(rho < 1.0) || !(disc >= 0.0) || !(R > 0.0) || !(Q > 0.0) || !(V1 > 0.0)) return 1.0e9;
    if (!(rho > 0.0)) return 1.0;
    const double Vp = d >= 0.0 ? (d + sq) / (2.0 * g) : 2.0 * Q / (sq - d);      // V+ (no cancellation either way)
    if (!(Vp > 0.0) || !(Vp < 1.0e300)) return 1.0e9;
    const double J = A * R / (C * C * Vp + R);                                   // steady J = A Vu / Vp (:100)
    if (!(J * J < 0.8)) return 1.0e9;                                            // (the kernel's verdict asks for it)
    const double r = 2.0 * sq / (2.0 * g * V1 - d + sq);                         // (V+ - V-) / (V1 - V-)
    const double dev0 = fabs(V1 - Vp) * fabs(r) / Vp;
    if (!(dev0 > 3.552713678800501e-15)) return 0.0;
    return log(3.552713678800501e-15 / dev0) / log(rho);
}

// One workgroup (1024 threads) per series: stable counting sort of the cells into 8 buckets of
// predicted slowness, slowest first; sorted index -> position by the launch's schedule.
__device__ static void order_cells(const PrepParams &prm, int s) {
    constexpr int NB = 8, NW = 16;
    __shared__ int hist[NB], start[NB], wcnt[NW][NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
    const int base = prm.cell_off[s], n = prm.cell_off[s + 1] - base;
    const int P = 6 + prm.p + prm.q;
    const double ntr = (double)prm.order_ntr;
    if (tid < NB) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const double *t = prm.theta0 + (long)(base + i) * P;
        const double st = steady_settle_steps(t[0], t[1 + prm.p], t[2 + prm.p + prm.q], t[3 + prm.p + prm.q], t[5 + prm.p + prm.q]);
        const double x = st / ntr;
        const int b = x > 16.0 ? 0 : x > 8.0 ? 1 : x > 4.0 ? 2 : x > 2.0 ? 3 : x > 1.4 ? 4 : x > 1.05 ? 5 : x > 0.8 ? 6 : 7;
        prm.perm_key[base + i] = b;
        atomicAdd(&hist[b], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int b = 0; b < NB; b++) { start[b] = a; a += hist[b]; }
    }
    __syncthreads();
    // position of sorted index sg (slowest first).  Work queue: the queue hands out positions in
    // order.  Static schedule: workgroup b, wave w, half h works on position b cpb + 2 w + h; the
    // sorted PAIRS are dealt across the workgroups (pair 0 -> wave 0 of workgroup 0, pair 1 -> wave 0
    // of workgroup 1, ...), so every CU gets its share of the slow waves.
    const int cpb = prm.order_cpb;
    const int nb = cpb > 0 ? (n + cpb - 1) / cpb : 1;
    const int cells_last = cpb > 0 ? n - (nb - 1) * cpb : 0;
    auto position = [&](int sg) {
        if (cpb <= 0) return sg;
        int w = 0, rowstart = 0;
        for (;; w++) {
            const int cw = min(2, max(0, cells_last - 2 * w));
            const int rowsize = 2 * (nb - 1) + cw;
            if (sg < rowstart + rowsize || 2 * (w + 1) >= cpb) break;
            rowstart += rowsize;
        }
        const int idx = sg - rowstart;
        const int b = idx < 2 * (nb - 1) ? (idx >> 1) : nb - 1;
        const int h = idx < 2 * (nb - 1) ? (idx & 1) : idx - 2 * (nb - 1);
        return b * cpb + 2 * w + h;
    };
    for (int t0 = 0; t0 < n; t0 += NT) {
        const int i = t0 + tid;
        const int b = i < n ? prm.perm_key[base + i] : -1;
        int rank = 0;
#pragma unroll
        for (int bb = 0; bb < NB; bb++) {
            const unsigned long long m = __ballot(b == bb);
            if (lane == 0) wcnt[wave][bb] = __popcll(m);
            if (b == bb) rank = __popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (i < n) {
            int o = start[b] + rank;
            for (int w = 0; w < wave; w++) o += wcnt[w][b];
            prm.perm[base + position(o)] = base + i;
        }
        __syncthreads();
        if (tid < NB) {
            int a = 0;
            for (int w = 0; w < NW; w++) a += wcnt[w][tid];
            start[tid] += a;
        }
        __syncthreads();
    }
}

// One workgroup (16 waves) per series.  Every theta-independent statistic is a wave-parallel
// strided sum over t followed by a shuffle reduction (fixed summation tree: deterministic);
// statistics are dealt round-robin to the waves.
__global__ __launch_bounds__(1024) void series_prep_kernel(PrepParams prm) {
    if (prm.perm && (int)blockIdx.x >= prm.n_series) {      // the second half of the grid orders the cells
        order_cells(prm, (int)blockIdx.x - prm.n_series);
        return;
    }
    const int s = blockIdx.x;
    const int T = prm.T, p = prm.p, q = prm.q, PP = prm.PP, QQ = prm.QQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, n_waves = NT >> 6;        // 1024 threads: the loops below are latency bound
    const double *y = prm.y + (long)s * T;
    const bool own_uv = (!prm.shared_uv) || s == 0;
    const double *u = prm.u ? prm.u + (prm.shared_uv ? 0 : (long)s * T * p) : nullptr;
    const double *v = prm.v ? prm.v + (prm.shared_uv ? 0 : (long)s * T * q) : nullptr;
    double *yp = prm.yp + (long)s * T;
    double *up = prm.up + (prm.shared_uv ? 0 : (long)s * T * PP);
    double *vp = prm.vp + (prm.shared_uv ? 0 : (long)s * T * QQ);

    __shared__ SeriesConst sc;
    __shared__ double inv_ws[LDSR_MAXPQ * LDSR_MAXPQ];

    if (tid == 0 && prm.queue) prm.queue[s] = 0;
    // prepared copies
    for (int t = tid; t < T; t += NT) {
        const double yv = y[t];
        yp[t] = yv;
        prm.yz[(long)s * T + t] = isfinite(yv) ? yv : 0.0;
    }
    if (own_uv) {
        for (int i = tid; i < T * PP; i += NT) {
            const int t = i / PP, k = i - t * PP;
            // u[:,T-1] is never read by the reference (src/EM.cpp:74,190-193): zero it
            up[i] = (u && k < p && t < T - 1) ? u[(long)t * p + k] : 0.0;
        }
        for (int i = tid; i < T * QQ; i += NT) {
            const int t = i / QQ, k = i - t * QQ;
            vp[i] = (v && k < q) ? v[(long)t * q + k] : 0.0;
        }
    }
    // identity / zero padding of the statistics
    for (int i = tid; i < LDSR_MAXPQ * LDSR_MAXPQ; i += NT) {
        const double id = ((i / LDSR_MAXPQ) == (i % LDSR_MAXPQ)) ? 1.0 : 0.0;
        sc.Svv_inv[i] = id;
        sc.Tuu_inv[i] = id;
        sc.Lv[i] = id; sc.Lv_inv[i] = id; sc.Lu[i] = id; sc.Lu_inv[i] = id;
    }
    if (tid < LDSR_MAXPQ) { sc.Syv[tid] = 0.0; sc.wv[tid] = 0.0; sc.Syv_w[tid] = 0.0; }
    __syncthreads();

    // statistic ids: [0, q*q) Svv(k,l); [.., +p*p) Tuu(k,l); [.., +q) Syv(k); last: Syy + counts
    const int nqq = v ? q * q : 0, npp = u ? p * p : 0, nq = v ? q : 0;
    const int n_stat = nqq + npp + nq + 1;
    for (int id = wave; id < n_stat; id += n_waves) {
        double acc = 0.0;
        if (id < nqq) {
            const int k = id / q, l = id - k * q;
            if (l < k) continue;                     // symmetric: (l, k) is written by (k, l)
            for (int t = lane; t < T; t += 64)
                acc += isfinite(y[t]) ? v[(long)t * q + k] * v[(long)t * q + l] : 0.0;   // :161
            acc = prep_wave_sum(acc);
            if (lane == 0) { sc.Svv_inv[k * LDSR_MAXPQ + l] = acc; sc.Svv_inv[l * LDSR_MAXPQ + k] = acc; }
        } else if (id < nqq + npp) {
            const int j = id - nqq, k = j / p, l = j - k * p;
            if (l < k) continue;
            for (int t = lane; t < T - 1; t += 64) acc += u[(long)t * p + k] * u[(long)t * p + l];   // :193
            acc = prep_wave_sum(acc);
            if (lane == 0) { sc.Tuu_inv[k * LDSR_MAXPQ + l] = acc; sc.Tuu_inv[l * LDSR_MAXPQ + k] = acc; }
        } else if (id < nqq + npp + nq) {
            const int k = id - nqq - npp;
            for (int t = lane; t < T; t += 64) {
                const double yv = y[t];
                acc += isfinite(yv) ? yv * v[(long)t * q + k] : 0.0;   // :158
            }
            acc = prep_wave_sum(acc);
            if (lane == 0) sc.Syv[k] = acc;
        } else {
            int n = 0, first = T, last = -1;
            for (int t = lane; t < T; t += 64) {
                const double yv = y[t];
                const bool o = isfinite(yv);
                acc += o ? yv * yv : 0.0;
                n += o ? 1 : 0;
                first = (o && t < first) ? t : first;
                last = o ? t : last;
            }
            acc = prep_wave_sum(acc);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                n += __shfl_xor(n, d, 64);
                first = min(first, __shfl_xor(first, d, 64));
                last = max(last, __shfl_xor(last, d, 64));
            }
            if (lane == 0) {
                sc.Syy = acc;
                sc.n_obs = n;
                sc.t_first_obs = n ? first : -1;
                sc.t_last_obs = last;
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        bool ok = sc.n_obs > 0;
        // Cholesky factors of Svv / Tuu for the whitened images (before they are inverted in place)
        if (v) ok = chol_small(sc.Svv_inv, sc.Lv, sc.Lv_inv, q, lane) && ok;
        if (u) ok = chol_small(sc.Tuu_inv, sc.Lu, sc.Lu_inv, p, lane) && ok;
        if (v) ok = invert_small(sc.Svv_inv, inv_ws, q, lane) && ok;
        if (u) ok = invert_small(sc.Tuu_inv, inv_ws, p, lane) && ok;
        // a caller-supplied lead (ldsr_em_batch_device_lead) that is not all-missing in this series
        // would silently drop observations: refuse the series like a singular one
        if (prm.img3 && prm.lead > 0 && sc.n_obs > 0 && sc.t_first_obs < prm.lead) ok = false;
        if (lane == 0) {
            sc.status = ok ? 0 : 2;
            sc.rn_obs = 1.0 / (double)sc.n_obs;
            sc.rTm1 = 1.0 / (double)(T - 1);
        }
        // Lv^{-1} Syv and Svv^{-1} Syv: row kk by lane kk (left-to-right sums over the padded 16 columns,
        // as the one-thread loops had them)
        if (lane < LDSR_MAXPQ) {
            const int kk = lane;
            vdp Lvi = sc.Lv_inv, Sv = sc.Svv_inv, Syv = sc.Syv;
            double a = 0.0;
            for (int ll = 0; ll <= kk; ll++) a += Lvi[kk * LDSR_MAXPQ + ll] * Syv[ll];
            sc.Syv_w[kk] = a;
            double b = 0.0;
            for (int ll = 0; ll < LDSR_MAXPQ; ll++) b += Sv[kk * LDSR_MAXPQ + ll] * Syv[ll];
            sc.wv[kk] = b;
        }
    }
    __syncthreads();
    if (sc.status == 0) {
        // chunk-transposed images (layout: em_scan_impl.h) of y and the WHITENED inputs
        // (ldsr_device.h mstep_update_white; built after the factors): the K = 1+PP+QQ values of a step in
        // pairs (em_scan_impl.h img_off: pair m of step j of virtual lane l at [((j*KH + m)*NL + l)*2 + {0,1}],
        // the odd value of an odd K paired over two steps behind them); 0 where missing / beyond the chunk / padding.  img: the scan kernel's (64 W lanes); img2: the pair
        // kernel's (32 lanes, its own chunk length).
        // toff: the image covers the steps [toff, T) (the pair kernel's LEAD form), else 0
        auto build_image = [&](double *im, int L, int NL, int toff) {
            const int K = 1 + PP + QQ, KH = K / 2;
            const int Tt = T - toff;
            const int nl = (Tt + L - 1) / L, rp = Tt - nl * (L - 1);
            const int n_pair = NL * L * KH * 2;                       // the full pairs, then the odd values' block
            const int n_all = (int)img_doubles(L, NL, PP, QQ);
            for (int e = tid; e < n_all; e += NT) {
                int i, j, l;
                if (e < n_pair) {
                    const int h = e & 1, jm = (e >> 1) / NL, m = jm % KH;
                    l = (e >> 1) % NL; j = jm / KH; i = 2 * m + h;
                } else {                                              // value K-1 of steps 2 jj and 2 jj + 1
                    const int r = e - n_pair;
                    l = (r >> 1) % NL; j = 2 * ((r >> 1) / NL) + (r & 1); i = K - 1;
                }
                const int t = toff + l * (L - 1) + min(l, rp) + j;
                const bool ok = l < nl && j < L && (j < L - 1 || l < rp);
                double val = 0.0;
                if (ok) {
                    if (i == 0) {
                        const double yv = y[t];
                        val = isfinite(yv) ? yv : 0.0;
                    } else if (i <= PP) {             // whitened inputs: row k of Lu^{-1} u_t / Lv^{-1} v_t
                        const int k = i - 1;
                        if (u && k < p && t < T - 1)
                            for (int j_ = 0; j_ <= k; j_++) val = fma(sc.Lu_inv[k * LDSR_MAXPQ + j_], u[(long)t * p + j_], val);
                    } else if (i < K) {
                        const int k = i - 1 - PP;
                        if (v && k < q)
                            for (int j_ = 0; j_ <= k; j_++) val = fma(sc.Lv_inv[k * LDSR_MAXPQ + j_], v[(long)t * q + j_], val);
                    }
                }
                im[e] = val;
            }
        };
        if (prm.img) build_image(prm.img + (long)s * prm.img_stride, prm.L, prm.NL, 0);
        if (prm.img2) build_image(prm.img2 + (long)s * prm.img2_stride, prm.L2, prm.NL2, prm.img3 ? prm.lead : 0);
        if (prm.img3) {
            // whitened u_t of the lead: lane l owns steps [l nA, (l+1) nA), element [(j NL2 + l) PP + k]
            double *im3 = prm.img3 + (long)s * prm.img3_stride;
            const int NL2 = prm.NL2, nA = (prm.lead + NL2 - 1) / NL2;
            for (int e = tid; e < nA * NL2 * PP; e += NT) {
                const int k = e % PP, l = (e / PP) % NL2, j = e / (PP * NL2);
                const int t = l * nA + j;
                double val = 0.0;
                if (t < prm.lead && j < nA && u && k < p && t < T - 1)
                    for (int jj = 0; jj <= k; jj++) val = fma(sc.Lu_inv[k * LDSR_MAXPQ + jj], u[(long)t * p + jj], val);
                im3[e] = val;
            }
        }
    }
    {   // cooperative copy of the result to global memory
        const int *src = reinterpret_cast<const int *>(&sc);
        int *dst = reinterpret_cast<int *>(prm.sc + s);
        for (int i = tid; i < (int)(sizeof(SeriesConst) / sizeof(int)); i += NT) dst[i] = src[i];
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
        if (prm.liks && prm.liks_nanfill)
            for (int i = 0; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
        return;
    }

    double lik = NAN, lik1 = NAN, lik2 = NAN;
    int it = 0;
    bool interrupted = false;
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
        if (prm.abort && (it & 63) == 0 && ldsr_poll_abort(prm.abort)) { interrupted = true; break; }

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
    if (prm.liks && prm.liks_nanfill)
        for (int i = it; i < prm.niter; i++) prm.liks[(long)cell * prm.niter + i] = NAN;
    prm.lik[cell] = lik;
    prm.n_iter[cell] = it;
    prm.status[cell] = interrupted ? 3 : (isfinite(lik) ? 0 : 1);
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
    double ssq = 0.0;   // sum_t (Xs_{t+1} - A Xs_t - B u_t)^2, the penalty of R/LDS_GA.R:34-40
    const bool full = !prm.scalar_only;
    if (full) {
        J[T - 1] = Vu * th.A / (th.A * Vu * th.A + th.Q);  // src/EM.cpp:98
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
        const double Xs1 = Xs;
        Xs = xu + Jt * (Xs - Xp1);
        Vs = vu + Jt * (Vs - Vp1) * Jt;
        {
            const double d = Xs1 - th.A * Xs - bu;
            ssq += d * d;
        }
        if (full) {
            double dv = 0.0;
#pragma unroll
            for (int k = 0; k < QQ; k++) dv = fma(th.D[k], v[t * QQ + k], dv);
            X[t] = Xs;
            V[t] = Vs;
            J[t] = Jt;
            Y[t] = th.C * Xs + dv;
        }
    }
    if (prm.pen) prm.pen[cell] = lik - prm.lambda * ssq;
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
    switch ((PPv) * 32 + (QQv)) {                                                       \
        case 1 * 32 + 1: { constexpr int PP = 1, QQ = 1; CALL; } break;               \
        case 1 * 32 + 2: { constexpr int PP = 1, QQ = 2; CALL; } break;               \
        case 1 * 32 + 4: { constexpr int PP = 1, QQ = 4; CALL; } break;               \
        case 1 * 32 + 8: { constexpr int PP = 1, QQ = 8; CALL; } break;               \
        case 1 * 32 + 16: { constexpr int PP = 1, QQ = 16; CALL; } break;               \
        case 2 * 32 + 1: { constexpr int PP = 2, QQ = 1; CALL; } break;               \
        case 2 * 32 + 2: { constexpr int PP = 2, QQ = 2; CALL; } break;               \
        case 2 * 32 + 4: { constexpr int PP = 2, QQ = 4; CALL; } break;               \
        case 2 * 32 + 8: { constexpr int PP = 2, QQ = 8; CALL; } break;               \
        case 2 * 32 + 16: { constexpr int PP = 2, QQ = 16; CALL; } break;               \
        case 4 * 32 + 1: { constexpr int PP = 4, QQ = 1; CALL; } break;               \
        case 4 * 32 + 2: { constexpr int PP = 4, QQ = 2; CALL; } break;               \
        case 4 * 32 + 4: { constexpr int PP = 4, QQ = 4; CALL; } break;               \
        case 4 * 32 + 8: { constexpr int PP = 4, QQ = 8; CALL; } break;               \
        case 4 * 32 + 16: { constexpr int PP = 4, QQ = 16; CALL; } break;               \
        case 8 * 32 + 1: { constexpr int PP = 8, QQ = 1; CALL; } break;               \
        case 8 * 32 + 2: { constexpr int PP = 8, QQ = 2; CALL; } break;               \
        case 8 * 32 + 4: { constexpr int PP = 8, QQ = 4; CALL; } break;               \
        case 8 * 32 + 8: { constexpr int PP = 8, QQ = 8; CALL; } break;               \
        case 8 * 32 + 16: { constexpr int PP = 8, QQ = 16; CALL; } break;               \
        case 16 * 32 + 1: { constexpr int PP = 16, QQ = 1; CALL; } break;               \
        case 16 * 32 + 2: { constexpr int PP = 16, QQ = 2; CALL; } break;               \
        case 16 * 32 + 4: { constexpr int PP = 16, QQ = 4; CALL; } break;               \
        case 16 * 32 + 8: { constexpr int PP = 16, QQ = 8; CALL; } break;               \
        case 16 * 32 + 16: { constexpr int PP = 16, QQ = 16; CALL; } break;               \
        default: return hipErrorInvalidValue;                                           \
    }

// ---------------------------------------------------------------------------------------
// gather_winners_kernel: one block per winner; copies its theta / theta0 rows and its
// likelihood trace (NaN padded beyond n_iter) into the compact winner arrays.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void gather_winners_kernel(GatherParams prm) {
    const int i = blockIdx.x;
    const long c = prm.cell[i];
    const bool has = c >= 0;
    for (int k = threadIdx.x; k < prm.P; k += 128) {
        prm.theta_w[(long)i * prm.P + k] = has ? prm.theta[c * prm.P + k] : NAN;
        prm.theta0_w[(long)i * prm.P + k] = has ? prm.theta0[c * prm.P + k] : NAN;
    }
    if (prm.liks) {
        const int n = has ? prm.n_iter[c] : 0;
        for (int k = threadIdx.x; k < prm.niter; k += 128)
            prm.liks_w[(long)i * prm.niter + k] = k < n ? prm.liks[c * prm.niter + k] : NAN;
    }
    if (threadIdx.x == 0) {
        if (prm.lik_w) prm.lik_w[i] = has ? prm.lik[c] : NAN;
        if (prm.n_iter_w) prm.n_iter_w[i] = has ? prm.n_iter[c] : 0;
        if (prm.blk) {
            prm.blk[i] = i;                          // series of the block
            prm.blk[prm.n_w + i] = i;                // its first cell: row i of the winner arrays
            prm.blk[2 * prm.n_w + i] = has ? 1 : 0;  // no winner: the block has nothing to do
        }
    }
}

hipError_t launch_gather_winners(const GatherParams &prm, hipStream_t stream) {
    hipLaunchKernelGGL(gather_winners_kernel, dim3(prm.n_w), dim3(128), 0, stream, prm);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// select_winners_kernel: the reference's selection rule per series -- highest lik among the
// restarts with C > 0 if there is one, else among all; NaN ignored; first index on ties.
// Same result as the host routine ldsr_select_restart (max is order independent, ties go to
// the lowest index).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_winners_kernel(SelectParams prm) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const int a = prm.off[s], b = prm.off[s + 1];
    __shared__ double sl[256];
    __shared__ int si[256];
    int any = 0;
    for (int c = a + tid; c < b; c += 256) any |= prm.theta[(long)c * prm.P + prm.c_index] > 0 ? 1 : 0;
    any = __syncthreads_or(any);
    double best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = a + tid; c < b; c += 256) {
        const double lk = prm.lik[c];
        if (isnan(lk)) continue;
        if (any && !(prm.theta[(long)c * prm.P + prm.c_index] > 0)) continue;
        if (lk > best || (lk == best && c < bi)) { best = lk; bi = c; }
    }
    sl[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) {
            const double ol = sl[tid + d];
            const int oi = si[tid + d];
            if (oi != 0x7fffffff && (si[tid] == 0x7fffffff || ol > sl[tid] || (ol == sl[tid] && oi < si[tid]))) {
                sl[tid] = ol;
                si[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) prm.winner[s] = si[0] == 0x7fffffff ? -1 : si[0];
}

hipError_t launch_select_winners(const SelectParams &prm, hipStream_t stream) {
    hipLaunchKernelGGL(select_winners_kernel, dim3(prm.n_series), dim3(256), 0, stream, prm);
    return hipGetLastError();
}

void em_serial_kernel_name(int T, int PP, int QQ, char *buf, size_t len) {
    const bool stage = (size_t)T * (1 + PP + QQ) * sizeof(double) <= 64 * 1024;
    snprintf(buf, len, "em_serial_kernel<%d, %d, %s>", PP, QQ, stage ? "true" : "false");
}

hipError_t launch_series_prep(const PrepParams &prm, int n_series, hipStream_t stream) {
    hipLaunchKernelGGL(series_prep_kernel, dim3(prm.perm ? 2 * n_series : n_series), dim3(1024), 0, stream, prm);
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
