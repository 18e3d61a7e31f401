/*
 * ldsr_oracle.h -- CPU oracle for the ldsr EM/Kalman hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar fp64 restatement, in the reference's
 * own operation order, of /root/reference/src/EM.cpp (Kalman_smoother :22-131,
 * Mstep :139-229, LDS_EM :245-280, propagate :295-356) and of the restart selection
 * rule of /root/reference/R/LDS_reconstruction.R:50-58.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product path
 * (ldsr_amd/, include/ldsr_hip.h) never links or calls it.
 *
 * Parity pin: the reference's own known-answer test tests/testthat/test-LDS-EM.R:21-41
 * (11 numbers at 1e-6) is reproduced by tests/test_oracle_golden.py.  The reference
 * itself (RcppArmadillo + R) cannot be built here, see DESIGN.md.
 *
 * Conventions (shared with include/ldsr_hip.h):
 *   y      [T]        NaN = missing observation
 *   u      [T*p]      column-major p x T as R stores it: u[t*p + k]; ignored if !has_u
 *   v      [T*q]      column-major q x T: v[t*q + k];               ignored if !has_v
 *   theta  [6+p+q]    packed  A, B[0..p-1], C, D[0..q-1], Q, R, mu1, V1
 */
#ifndef LDSR_ORACLE_H
#define LDSR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define LDSR_ORACLE_OK 0
#define LDSR_ORACLE_EINVAL 1   /* bad argument (niter < 2, T < 2, ...) */
#define LDSR_ORACLE_ESINGULAR 2 /* an M-step system was exactly singular (arma::inv throws) */

int oracle_theta_len(int p, int q);

/* src/EM.cpp:22-131.  X,Y,V,J: [T] outputs. */
int oracle_kalman_smoother(int T, int p, int q, const double *y, const double *u,
                           const double *v, int has_u, int has_v, const double *theta,
                           int stdlik, double *X, double *Y, double *V, double *J,
                           double *lik);

/* src/EM.cpp:139-229.  theta_out: [6+p+q]. */
int oracle_mstep(int T, int p, int q, const double *y, const double *u, const double *v,
                 int has_u, int has_v, const double *X, const double *V, const double *J,
                 double *theta_out);

/* src/EM.cpp:245-280.  liks: [niter] (first *n_iter entries valid); X,Y,V,J may be NULL. */
int oracle_lds_em(int T, int p, int q, const double *y, const double *u, const double *v,
                  int has_u, int has_v, const double *theta0, int niter, double tol,
                  double *theta_out, double *X, double *Y, double *V, double *J,
                  double *liks, int *n_iter, double *lik);

/* src/EM.cpp:295-356.  X,Y,V: [T] outputs. */
int oracle_propagate(int T, int p, int q, const double *theta, const double *u,
                     const double *v, int has_u, int has_v, const double *y, int stdlik,
                     double *X, double *Y, double *V, double *lik);

/* The foreach fan-out of R/LDS_reconstruction.R:46 for cells that may belong to
 * different series (all series share T,p,q):  y_all [n_series*T], u_all [n_series*T*p],
 * v_all [n_series*T*q], series_of_cell [n_cells], theta0 [n_cells*P].
 * Outputs theta [n_cells*P], lik [n_cells], n_iter [n_cells], status [n_cells].
 * n_threads host threads, static contiguous partition of the cells. */
int oracle_em_batch(int n_series, int T, int p, int q, const double *y_all,
                    const double *u_all, const double *v_all, int has_u, int has_v,
                    int n_cells, const int *series_of_cell, const double *theta0, int niter,
                    double tol, int n_threads, double *theta, double *lik, int *n_iter,
                    int *status);

/* R/LDS_reconstruction.R:50-58: among models with C > 0 take the max lik (NaN ignored);
 * if none has C > 0, which.max(liks).  First index on ties.  Returns -1 if nothing
 * is selectable (all lik NaN). */
int oracle_select(int n, const double *lik, const double *C);

#ifdef __cplusplus
}
#endif
#endif
