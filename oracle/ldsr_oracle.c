/*
 * ldsr_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see ldsr_oracle.h).
 *
 * Scalar fp64 restatement of the reference algorithm in the reference's operation
 * order.  Each function cites the lines of /root/reference/src/EM.cpp it follows.
 * Armadillo calls are restated by their mathematical definition:
 *   inv(M)          -> Gauss-Jordan inverse with partial pivoting, then an explicit
 *                      product with the inverse (the reference forms P1 * inv(P2))
 *   accu / products -> plain left-to-right sums
 *   find_finite     -> isfinite();  NumericMatrix::is_na -> isnan()
 * Build: gcc -O2 -std=c99 -fPIC -shared (no -ffast-math), see oracle/Makefile.
 */
#include "ldsr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const double ORACLE_PI = 3.141592653589793238463; /* src/EM.cpp:3 */

int oracle_theta_len(int p, int q) { return 6 + p + q; }

/* packed-theta accessors */
#define TH_A(th) ((th)[0])
#define TH_B(th) ((th) + 1)
#define TH_C(th, p) ((th)[1 + (p)])
#define TH_D(th, p) ((th) + 2 + (p))
#define TH_Q(th, p, q) ((th)[2 + (p) + (q)])
#define TH_R(th, p, q) ((th)[3 + (p) + (q)])
#define TH_MU1(th, p, q) ((th)[4 + (p) + (q)])
#define TH_V1(th, p, q) ((th)[5 + (p) + (q)])

static double dotk(const double *a, const double *b, int n) {
    double s = 0.0;
    for (int k = 0; k < n; k++) s += a[k] * b[k];
    return s;
}

/* arma::inv restated: Gauss-Jordan with partial pivoting.  a: n x n row-major, in place.
 * Returns 0 on success, 1 if a pivot is exactly zero (arma::inv would throw). */
static int invert(double *a, int n) {
    double *w = (double *)malloc(sizeof(double) * (size_t)n * 2 * (size_t)n);
    if (!w) return 1;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            w[i * 2 * n + j] = a[i * n + j];
            w[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = fabs(w[c * 2 * n + c]);
        for (int r = c + 1; r < n; r++) {
            double m = fabs(w[r * 2 * n + c]);
            if (m > best) { best = m; piv = r; }
        }
        if (!(best > 0.0)) { free(w); return 1; }
        if (piv != c) {
            for (int j = 0; j < 2 * n; j++) {
                double t = w[c * 2 * n + j];
                w[c * 2 * n + j] = w[piv * 2 * n + j];
                w[piv * 2 * n + j] = t;
            }
        }
        double d = w[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) w[c * 2 * n + j] /= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = w[r * 2 * n + c];
            if (f != 0.0)
                for (int j = 0; j < 2 * n; j++) w[r * 2 * n + j] -= f * w[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) a[i * n + j] = w[i * 2 * n + n + j];
    free(w);
    return 0;
}

/* ---- Kalman_smoother: src/EM.cpp:22-131 ------------------------------------------- */
int oracle_kalman_smoother(int T, int p, int q, const double *y, const double *u,
                           const double *v, int has_u, int has_v, const double *theta,
                           int stdlik, double *X, double *Y, double *V, double *J,
                           double *lik_out) {
    if (T < 2 || p < 1 || q < 1) return LDSR_ORACLE_EINVAL;
    const double A = TH_A(theta), C = TH_C(theta, p), Q = TH_Q(theta, p, q),
                 R = TH_R(theta, p, q), x1 = TH_MU1(theta, p, q), V1 = TH_V1(theta, p, q);
    const double *B = TH_B(theta), *D = TH_D(theta, p);
    double *buf = (double *)malloc(sizeof(double) * 5 * (size_t)T);
    if (!buf) return LDSR_ORACLE_EINVAL;
    double *Xp = buf, *Vp = buf + T, *Yp = buf + 2 * T, *Xu = buf + 3 * T, *Vu = buf + 4 * T;

    /* :48-54 */
    Xp[0] = x1;
    Vp[0] = V1;
    Yp[0] = has_v ? C * Xp[0] + dotk(D, v, q) : C * Xp[0];
    /* :61-68 */
    if (isnan(y[0])) {
        Xu[0] = Xp[0];
        Vu[0] = Vp[0];
    } else {
        double K = Vp[0] * C * (1.0 / (C * Vp[0] * C + R));
        Xu[0] = Xp[0] + K * (y[0] - Yp[0]);
        Vu[0] = (1.0 - K * C) * Vp[0];
    }
    /* :70-90 */
    for (int t = 1; t < T; t++) {
        Xp[t] = has_u ? A * Xu[t - 1] + dotk(B, u + (size_t)(t - 1) * p, p) : A * Xu[t - 1];
        Vp[t] = A * Vu[t - 1] * A + Q;
        Yp[t] = has_v ? C * Xp[t] + dotk(D, v + (size_t)t * q, q) : C * Xp[t];
        if (isnan(y[t])) {
            Xu[t] = Xp[t];
            Vu[t] = Vp[t];
        } else {
            double K = Vp[t] * C * (1.0 / (C * Vp[t] * C + R));
            Xu[t] = Xp[t] + K * (y[t] - Yp[t]);
            Vu[t] = (1.0 - K * C) * Vp[t];
        }
    }
    /* :94-104 */
    for (int t = 0; t < T; t++) { X[t] = Xu[t]; V[t] = Vu[t]; J[t] = 0.0; }
    J[T - 1] = Vu[T - 1] * A * (1.0 / (A * Vu[T - 1] * A + Q));
    for (int t = T - 2; t >= 0; t--) {
        J[t] = Vu[t] * A * (1.0 / Vp[t + 1]);
        X[t] = Xu[t] + J[t] * (X[t + 1] - Xp[t + 1]);
        V[t] = Vu[t] + J[t] * (V[t + 1] - Vp[t + 1]) * J[t];
    }
    /* :106-110 */
    for (int t = 0; t < T; t++) Y[t] = has_v ? C * X[t] + dotk(D, v + (size_t)t * q, q) : C * X[t];
    /* :113-124 */
    int n_obs = 0;
    double acc = 0.0;
    for (int t = 0; t < T; t++) {
        if (!isfinite(y[t])) continue;
        n_obs++;
        double delta = y[t] - Yp[t];
        double Sigma = C * Vp[t] * C + R;
        acc += delta / Sigma * delta + log(Sigma);
    }
    double lik = -0.5 * n_obs * log(2 * ORACLE_PI) - 0.5 * acc;
    if (stdlik) lik = lik / n_obs;
    *lik_out = lik;
    free(buf);
    return LDSR_ORACLE_OK;
}

/* ---- Mstep: src/EM.cpp:139-229 ---------------------------------------------------- */
int oracle_mstep(int T, int p, int q, const double *y, const double *u, const double *v,
                 int has_u, int has_v, const double *X, const double *V, const double *J,
                 double *th) {
    if (T < 2 || p < 1 || q < 1) return LDSR_ORACLE_EINVAL;
    int rc = LDSR_ORACLE_OK;
    const int nq = 1 + q, np = 1 + p;
    double *P1 = (double *)calloc((size_t)nq, sizeof(double));
    double *P2 = (double *)calloc((size_t)nq * nq, sizeof(double));
    double *P3 = (double *)calloc((size_t)np, sizeof(double));
    double *P4 = (double *)calloc((size_t)np * np, sizeof(double));
    double *CD = (double *)calloc((size_t)nq, sizeof(double));
    double *AB = (double *)calloc((size_t)np, sizeof(double));

    /* :147-152 */
    int n_obs = 0;
    double Syx = 0.0, Sxx_xx = 0.0, Sxx_v = 0.0;
    for (int t = 0; t < T; t++) {
        if (!isfinite(y[t])) continue;
        n_obs++;
        Syx += y[t] * X[t];
        Sxx_xx += X[t] * X[t];
        Sxx_v += V[t];
    }
    double Sxx = Sxx_xx + Sxx_v;
    double Cn;
    double *Dn = TH_D(th, p);
    for (int k = 0; k < q; k++) Dn[k] = 0.0; /* :154 */
    if (has_v) {
        /* :158-169 */
        P1[0] = Syx;
        P2[0] = Sxx;
        for (int t = 0; t < T; t++) {
            if (!isfinite(y[t])) continue;
            const double *vt = v + (size_t)t * q;
            for (int k = 0; k < q; k++) {
                P1[1 + k] += y[t] * vt[k];        /* Syv */
                P2[0 * nq + 1 + k] += X[t] * vt[k]; /* Sxv */
                for (int l = 0; l < q; l++) P2[(1 + k) * nq + 1 + l] += vt[k] * vt[l]; /* Svv */
            }
        }
        for (int k = 0; k < q; k++) P2[(1 + k) * nq + 0] = P2[0 * nq + 1 + k]; /* Svx */
        if (invert(P2, nq)) { rc = LDSR_ORACLE_ESINGULAR; goto done; }
        for (int j = 0; j < nq; j++) {
            double s = 0.0;
            for (int i = 0; i < nq; i++) s += P1[i] * P2[i * nq + j];
            CD[j] = s;
        }
        Cn = CD[0];
        for (int k = 0; k < q; k++) Dn[k] = CD[1 + k];
    } else {
        Cn = Syx * (1.0 / Sxx); /* :172 */
    }
    /* :170-177  R = ((y - y_hat) * y') / n_obs over observed columns */
    double Racc = 0.0;
    for (int t = 0; t < T; t++) {
        if (!isfinite(y[t])) continue;
        double yhat = has_v ? Cn * X[t] + dotk(Dn, v + (size_t)t * q, q) : Cn * X[t];
        Racc += (y[t] - yhat) * y[t];
    }
    double Rn = Racc / n_obs;

    /* :180-183 */
    double Tx1x_a = 0.0, Tx1x_b = 0.0, Txx_a = 0.0, Txx_b = 0.0, T11_a = 0.0, T11_b = 0.0;
    for (int t = 1; t < T; t++) { Tx1x_a += X[t] * X[t - 1]; Tx1x_b += V[t] * J[t - 1]; }
    for (int t = 0; t < T - 1; t++) { Txx_a += X[t] * X[t]; Txx_b += V[t]; }
    for (int t = 1; t < T; t++) { T11_a += X[t] * X[t]; T11_b += V[t]; }
    double Tx1x = Tx1x_a + Tx1x_b, Txx = Txx_a + Txx_b, Tx1x1 = T11_a + T11_b;

    double An, Qn;
    double *Bn = TH_B(th);
    for (int k = 0; k < p; k++) Bn[k] = 0.0; /* :186 */
    if (has_u) {
        /* :190-210 */
        P3[0] = Tx1x;
        P4[0] = Txx;
        for (int t = 0; t < T - 1; t++) {
            const double *ut = u + (size_t)t * p;
            for (int k = 0; k < p; k++) {
                P3[1 + k] += X[t + 1] * ut[k];           /* Tx1u */
                P4[(1 + k) * np + 0] += ut[k] * X[t];    /* Tux  */
                for (int l = 0; l < p; l++) P4[(1 + k) * np + 1 + l] += ut[k] * ut[l]; /* Tuu */
            }
        }
        for (int k = 0; k < p; k++) P4[0 * np + 1 + k] = P4[(1 + k) * np + 0]; /* Txu */
        double Tx1u_keep[64];
        double *Tx1u = (p <= 64) ? Tx1u_keep : (double *)malloc(sizeof(double) * (size_t)p);
        for (int k = 0; k < p; k++) Tx1u[k] = P3[1 + k];
        if (invert(P4, np)) {
            if (Tx1u != Tx1u_keep) free(Tx1u);
            rc = LDSR_ORACLE_ESINGULAR;
            goto done;
        }
        for (int j = 0; j < np; j++) {
            double s = 0.0;
            for (int i = 0; i < np; i++) s += P3[i] * P4[i * np + j];
            AB[j] = s;
        }
        An = AB[0];
        for (int k = 0; k < p; k++) Bn[k] = AB[1 + k];
        Qn = (Tx1x1 - An * Tx1x - dotk(Bn, Tx1u, p)) / (T - 1);
        if (Tx1u != Tx1u_keep) free(Tx1u);
    } else {
        An = Tx1x * (1.0 / Txx);            /* :212 */
        Qn = (Tx1x1 - An * Tx1x) / (T - 1); /* :213 */
    }
    TH_A(th) = An;
    TH_C(th, p) = Cn;
    TH_Q(th, p, q) = Qn;
    TH_R(th, p, q) = Rn;
    TH_MU1(th, p, q) = X[0]; /* :218 */
    TH_V1(th, p, q) = V[0];  /* :219 */
done:
    free(P1); free(P2); free(P3); free(P4); free(CD); free(AB);
    return rc;
}

/* ---- LDS_EM: src/EM.cpp:245-280 --------------------------------------------------- */
int oracle_lds_em(int T, int p, int q, const double *y, const double *u, const double *v,
                  int has_u, int has_v, const double *theta0, int niter, double tol,
                  double *theta_out, double *Xo, double *Yo, double *Vo, double *Jo,
                  double *liks, int *n_iter, double *lik_out) {
    if (niter < 2 || T < 2) return LDSR_ORACLE_EINVAL; /* lik[1] would be out of bounds */
    const int P = oracle_theta_len(p, q);
    double *buf = (double *)malloc(sizeof(double) * (4 * (size_t)T + 2 * (size_t)P + (size_t)niter));
    if (!buf) return LDSR_ORACLE_EINVAL;
    double *X = buf, *Y = buf + T, *V = buf + 2 * T, *J = buf + 3 * T;
    double *theta = buf + 4 * T, *tnew = theta + P, *lik = tnew + P;
    int rc;
    memcpy(theta, theta0, sizeof(double) * P);
    /* i = 0 (:251-253) */
    rc = oracle_kalman_smoother(T, p, q, y, u, v, has_u, has_v, theta, 1, X, Y, V, J, &lik[0]);
    if (rc) goto done;
    rc = oracle_mstep(T, p, q, y, u, v, has_u, has_v, X, V, J, tnew);
    if (rc) goto done;
    memcpy(theta, tnew, sizeof(double) * P);
    /* i = 1 (:255-256) */
    rc = oracle_kalman_smoother(T, p, q, y, u, v, has_u, has_v, theta, 1, X, Y, V, J, &lik[1]);
    if (rc) goto done;
    int lastIter = 2;
    for (int i = 2; i < niter; i++) { /* :259-275 */
        rc = oracle_mstep(T, p, q, y, u, v, has_u, has_v, X, V, J, tnew);
        if (rc) goto done;
        memcpy(theta, tnew, sizeof(double) * P);
        rc = oracle_kalman_smoother(T, p, q, y, u, v, has_u, has_v, theta, 1, X, Y, V, J, &lik[i]);
        if (rc) goto done;
        lastIter++;
        if (fabs(lik[i] - lik[i - 1]) < tol && fabs(lik[i - 1] - lik[i - 2]) < tol) break;
    }
    memcpy(theta_out, theta, sizeof(double) * P);
    if (Xo) memcpy(Xo, X, sizeof(double) * T);
    if (Yo) memcpy(Yo, Y, sizeof(double) * T);
    if (Vo) memcpy(Vo, V, sizeof(double) * T);
    if (Jo) memcpy(Jo, J, sizeof(double) * T);
    if (liks) memcpy(liks, lik, sizeof(double) * lastIter);
    if (n_iter) *n_iter = lastIter;
    if (lik_out) *lik_out = lik[lastIter - 1];
done:
    free(buf);
    return rc;
}

/* ---- propagate: src/EM.cpp:295-356 ------------------------------------------------ */
int oracle_propagate(int T, int p, int q, const double *theta, const double *u,
                     const double *v, int has_u, int has_v, const double *y, int stdlik,
                     double *X, double *Y, double *V, double *lik_out) {
    if (T < 1 || p < 1 || q < 1) return LDSR_ORACLE_EINVAL;
    const double A = TH_A(theta), C = TH_C(theta, p), Q = TH_Q(theta, p, q),
                 R = TH_R(theta, p, q), x1 = TH_MU1(theta, p, q), V1 = TH_V1(theta, p, q);
    const double *B = TH_B(theta), *D = TH_D(theta, p);
    X[0] = x1;  /* :318-319 */
    V[0] = V1;
    for (int t = 1; t < T; t++) { /* :322-329 */
        X[t] = has_u ? A * X[t - 1] + dotk(B, u + (size_t)(t - 1) * p, p) : A * X[t - 1];
        V[t] = A * V[t - 1] * A + Q;
    }
    for (int t = 0; t < T; t++) /* :332-336 */
        Y[t] = has_v ? C * X[t] + dotk(D, v + (size_t)t * q, q) : C * X[t];
    int n_obs = 0; /* :339-350 */
    double acc = 0.0;
    for (int t = 0; t < T; t++) {
        if (!isfinite(y[t])) continue;
        n_obs++;
        double delta = y[t] - Y[t];
        double Sigma = C * V[t] * C + R;
        acc += delta / Sigma * delta + log(Sigma);
    }
    double lik = -0.5 * n_obs * log(2 * ORACLE_PI) - 0.5 * acc;
    if (stdlik) lik = lik / n_obs;
    *lik_out = lik;
    return LDSR_ORACLE_OK;
}

/* ---- restart fan-out: R/LDS_reconstruction.R:46 (one LDS_EM per init) ------------- */
typedef struct {
    int T, p, q, has_u, has_v, niter, lo, hi;
    double tol;
    const double *y_all, *u_all, *v_all, *theta0;
    const int *series_of_cell;
    double *theta, *lik;
    int *n_iter, *status;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *jb = (batch_job *)arg;
    const int P = oracle_theta_len(jb->p, jb->q);
    for (int c = jb->lo; c < jb->hi; c++) {
        int s = jb->series_of_cell ? jb->series_of_cell[c] : 0;
        const double *y = jb->y_all + (size_t)s * jb->T;
        const double *u = jb->u_all ? jb->u_all + (size_t)s * jb->T * jb->p : NULL;
        const double *v = jb->v_all ? jb->v_all + (size_t)s * jb->T * jb->q : NULL;
        int it = 0;
        double lk = NAN;
        int rc = oracle_lds_em(jb->T, jb->p, jb->q, y, u, v, jb->has_u, jb->has_v,
                               jb->theta0 + (size_t)c * P, jb->niter, jb->tol,
                               jb->theta + (size_t)c * P, NULL, NULL, NULL, NULL, NULL, &it, &lk);
        jb->status[c] = rc;
        jb->n_iter[c] = rc ? 0 : it;
        jb->lik[c] = rc ? NAN : lk;
        if (rc) for (int k = 0; k < P; k++) jb->theta[(size_t)c * P + k] = NAN;
    }
    return NULL;
}

int oracle_em_batch(int n_series, int T, int p, int q, const double *y_all,
                    const double *u_all, const double *v_all, int has_u, int has_v,
                    int n_cells, const int *series_of_cell, const double *theta0, int niter,
                    double tol, int n_threads, double *theta, double *lik, int *n_iter,
                    int *status) {
    if (n_series < 1 || n_cells < 0 || niter < 2 || T < 2) return LDSR_ORACLE_EINVAL;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_cells) n_threads = n_cells > 0 ? n_cells : 1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    batch_job *jobs = (batch_job *)malloc(sizeof(batch_job) * (size_t)n_threads);
    for (int i = 0; i < n_threads; i++) {
        batch_job *jb = &jobs[i];
        jb->T = T; jb->p = p; jb->q = q; jb->has_u = has_u; jb->has_v = has_v;
        jb->niter = niter; jb->tol = tol;
        jb->lo = (int)((long long)n_cells * i / n_threads);
        jb->hi = (int)((long long)n_cells * (i + 1) / n_threads);
        jb->y_all = y_all; jb->u_all = u_all; jb->v_all = v_all; jb->theta0 = theta0;
        jb->series_of_cell = series_of_cell;
        jb->theta = theta; jb->lik = lik; jb->n_iter = n_iter; jb->status = status;
        if (n_threads == 1) batch_worker(jb);
        else pthread_create(&tid[i], NULL, batch_worker, jb);
    }
    if (n_threads > 1)
        for (int i = 0; i < n_threads; i++) pthread_join(tid[i], NULL);
    free(tid);
    free(jobs);
    return LDSR_ORACLE_OK;
}

/* ---- selection: R/LDS_reconstruction.R:50-58 -------------------------------------- */
int oracle_select(int n, const double *lik, const double *C) {
    int best = -1;
    int any_pos = 0;
    for (int i = 0; i < n; i++)
        if (C[i] > 0) { any_pos = 1; break; }
    for (int i = 0; i < n; i++) {
        if (isnan(lik[i])) continue;
        if (any_pos && !(C[i] > 0)) continue;
        if (best < 0 || lik[i] > lik[best]) best = i;
    }
    /* R: if every C>0 model has NaN lik, max(..., na.rm=TRUE) is -Inf and nothing matches */
    return best;
}
