/*
 * ldsr_hip.h -- C ABI of libldsr_hip.so, the MI355X (gfx950) engine for ldsr's
 * EM/Kalman restart path.  Plain pointers and sizes only; no torch / Rcpp types.
 *
 * What each entry point replaces in the reference (paths under /root/reference):
 *
 *   ldsr_em_restart_grid     LDS_EM_restart (R/LDS_reconstruction.R:42-62) for one series, and
 *                            the fold loop of cvLDS (R/LDS_reconstruction.R:373-375 ->
 *                            one_lds_cv :270-285) for several: the foreach fan-out :46, the
 *                            selection :50-58 and the winner's model {theta, fit, liks, lik} in
 *                            ONE call -- only the winners' fit / liks cross PCIe
 *   ldsr_em_restart_groups   the ensemble loop R/LDS_reconstruction.R:242-246 (members differ in
 *                            p, q: tests/testthat/test-ensemble.R:4-5): one grid per member,
 *                            run concurrently
 *   ldsr_em_batch            the bare fan-out  foreach(theta0 = init) %dopar% LDS_EM(...)
 *                            of R/LDS_reconstruction.R:46, i.e. n_cells calls of
 *                            _ldsr_LDS_EM (src/RcppExports.cpp:40-53 -> src/EM.cpp:245-280),
 *                            batched over restarts and over series / CV folds
 *   ldsr_em_batch_multi      same, sharded over several GPUs by host threads (no collective)
 *   ldsr_em_batch_device     same, operands already resident in HBM (bench + torch callers)
 *   ldsr_smooth_batch        _ldsr_Kalman_smoother (src/RcppExports.cpp:11-24 ->
 *                            src/EM.cpp:22-131), one E-step per (series, theta) cell;
 *                            also yields the winner's `fit` list of LDS_EM (src/EM.cpp:276-279)
 *   ldsr_mstep_batch         _ldsr_Mstep (src/RcppExports.cpp:26-38 -> src/EM.cpp:139-229)
 *   ldsr_propagate_batch     _ldsr_propagate (src/RcppExports.cpp:56-69 -> src/EM.cpp:295-356)
 *   ldsr_penalized_lik_batch penalized_likelihood of R/LDS_GA.R:28-44 for a whole GA population
 *   ldsr_select_restart      the argmax-with-C>0 rule of R/LDS_reconstruction.R:50-58
 *
 * Data conventions (identical to the bytes R hands to .Call):
 *   y      [n_series][T]        double, NaN / NA_real_ = missing
 *   u      [n_series][T][p]     double; an R p x T matrix is column-major, i.e. exactly
 *                               this time-major layout (u[t*p + k]).  NULL = input absent
 *                               (the reference's 1x1 `matrix(0)` sentinel, src/EM.cpp:71)
 *   v      [n_series][T][q]     likewise (src/EM.cpp:77)
 *   shared_uv != 0              u and v hold ONE series ([T][p], [T][q]) shared by every
 *                               y series (cvLDS folds: same inputs, different NA masks,
 *                               R/LDS_reconstruction.R:274)
 *   theta  [n_cells][6+p+q]     packed  A, B[p], C, D[q], Q, R, mu1, V1  (list order of
 *                               src/EM.cpp:221-228).  With u (v) absent p (q) is 1, the
 *                               B (D) slot is ignored on input and returned as 0, as
 *                               B.zeros(1, p) at src/EM.cpp:186 (:154)
 *   cell_offsets [n_series+1]   HOST array: cells [cell_offsets[s], cell_offsets[s+1]) are
 *                               the restarts of series s
 *
 * Limits of this build: 1 <= p, q <= 16 (larger returns LDSR_EUNSUPPORTED; the scan kernel
 * covers p, q <= 8 = every BASELINE config and the reference's tests, wider inputs run on the
 * serial kernel), T >= 2, niter >= 2 (the reference indexes lik[1]
 * unconditionally, src/EM.cpp:256).
 *
 * Every function returns LDSR_OK (0) or an error code; ldsr_last_error() gives the
 * message of the calling thread's last failure.
 *
 * Ownership: the library never keeps a pointer of the caller after a call returns.  The
 * host-pointer entry points cache their device buffers, pinned staging buffers and one stream
 * per concurrent caller and device (grow-only arenas; a call on a busy device gets its own), so
 * repeated calls do no hipMalloc / hipFree; ldsr_shutdown() frees everything.  All entry
 * points may be called from several threads.
 */
#ifndef LDSR_HIP_H
#define LDSR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDSR_OK 0
#define LDSR_EINVAL 1        /* bad argument */
#define LDSR_EUNSUPPORTED 2  /* p or q above the compiled limit */
#define LDSR_EHIP 3          /* a HIP runtime call failed (no device, OOM, launch error) */
#define LDSR_EINTERRUPTED 4  /* the interrupt callback asked to stop; outputs are incomplete */

/* per-cell status words */
#define LDSR_CELL_OK 0
#define LDSR_CELL_NONFINITE 1 /* final lik is NaN/Inf (tolerated by the reference's selection, na.rm) */
#define LDSR_CELL_SINGULAR 2  /* Svv or Tuu of the series is singular: arma::inv would throw */
#define LDSR_CELL_INTERRUPTED 3 /* stopped early by the interrupt callback (theta = last E-step's) */

/* algorithm selector */
#define LDSR_ALGO_AUTO 0
#define LDSR_ALGO_SERIAL 1 /* one thread per cell, sequential in time (any T) */
#define LDSR_ALGO_SCAN 2   /* one to four wavefronts per cell, parallel-in-time scans (T <= 8192,
                              p, q <= 8); AUTO picks it whenever it applies and PAIR does not */
#define LDSR_ALGO_PAIR 3   /* the same scans with TWO cells per wavefront (one per 32-lane half):
                              65 <= T <= 1024, p, q <= 4, narrower ranges of T for the wider inputs
                              (ldsr_em_plan tells); AUTO's first choice where it applies -- with
                              tol > 0 only for fully observed series.  Its LEAD form
                              (a closed-form all-missing lead, tails <= 512 steps: ldsr_em_plan_lead)
                              also exists for padded p or q = 8 */
#define LDSR_ALGO_QUAD 4   /* FOUR cells per wavefront (one per 16-lane DPP row): 65 <= T <= 512,
                              p, q <= 4; AUTO's first choice there for launches that fill the device */

const char *ldsr_last_error(void);
const char *ldsr_version(void);
/* first 16 hex digits of the SHA-256 over the library's sources (every .hip, .h and .inc file of
 * ldsr_amd/csrc, its Makefile and this header) it was built from: a caller that has the tree can tell a
 * stale binary (ldsr_amd/_lib.py refuses one) */
const char *ldsr_source_hash(void);
int ldsr_device_count(void);
void ldsr_shutdown(void);

/* LDS_EM_restart for a whole grid of series / CV folds in one call: runs every (series,
 * restart) cell (cells are cut into contiguous slices over devices[0..n_devices), host threads,
 * no collective), picks each series' winner by the reference's rule (highest lik among
 * restarts with C > 0 if any, NaN ignored, first index on ties), and returns per series the
 * winner's model exactly as LDS_EM returns it (src/EM.cpp:276-279):
 *   winner   [n_series]          global cell index, or -1 if nothing is selectable
 *   theta_w  [n_series][6+p+q]   lik_w [n_series]   n_iter_w [n_series]
 *   liks_w   [n_series][niter]   the winner's likelihood trace, NaN beyond n_iter_w[s]
 *   X, Y, V, J [n_series][T]     the winner's fit (Kalman_smoother at theta_w, stdlik = TRUE)
 * Rows of a series without a winner are NaN.  liks_w, X, Y, V, J may each be NULL.
 * The per-cell results (theta_all [n_cells][6+p+q], lik_all, n_iter_all, status_all) are
 * optional: pass NULL to leave them on the device side of PCIe. */
int ldsr_em_restart_grid(int n_devices, const int *devices, int n_series, int T, int p, int q,
                         const double *y, const double *u, const double *v, int shared_uv,
                         const int *cell_offsets, const double *theta0, int niter, double tol,
                         int algo, double *theta_all, double *lik_all, int *n_iter_all,
                         int *status_all, int *winner, double *theta_w, double *lik_w,
                         int *n_iter_w, double *liks_w, double *X, double *Y, double *V,
                         double *J);

/* Heterogeneous ensembles (R/LDS_reconstruction.R:242-246: list members differ in p and q):
 * one ldsr_em_restart_grid per group, groups run concurrently (one host thread and one stream
 * each; group g starts on devices[g % n_devices]).  Fields are the arguments of
 * ldsr_em_restart_grid; rc receives each group's return code. */
typedef struct ldsr_group {
    int n_series, T, p, q, shared_uv;
    const double *y, *u, *v;
    const int *cell_offsets;
    const double *theta0;
    double *theta_all, *lik_all;
    int *n_iter_all, *status_all;
    int *winner;
    double *theta_w, *lik_w;
    int *n_iter_w;
    double *liks_w, *X, *Y, *V, *J;
    int rc;
} ldsr_group;
int ldsr_em_restart_groups(int n_devices, const int *devices, int n_groups, ldsr_group *groups,
                           int niter, double tol, int algo);

/* Batched LDS_EM.  Host pointers; copies in, runs on `device`, copies out.
 * liks may be NULL; otherwise [n_cells][niter], entries beyond n_iter[c] are NaN. */
int ldsr_em_batch(int device, int n_series, int T, int p, int q, const double *y,
                  const double *u, const double *v, int shared_uv, const int *cell_offsets,
                  const double *theta0, int niter, double tol, int algo,
                  double *theta, double *lik, int *n_iter, int *status, double *liks);

/* Same over several GPUs of one node: the cell grid is cut into n_devices contiguous slices,
 * one host thread per listed device, no collective (restarts never communicate,
 * R/LDS_reconstruction.R:46).  devices[] may repeat an id. */
int ldsr_em_batch_multi(int n_devices, const int *devices, int n_series, int T, int p, int q,
                        const double *y, const double *u, const double *v, int shared_uv,
                        const int *cell_offsets, const double *theta0, int niter, double tol,
                        int algo, double *theta, double *lik, int *n_iter, int *status,
                        double *liks);

/* Same with DEVICE pointers (cell_offsets stays a host array and may be reused or freed as
 * soon as the call returns).  Asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 * default stream): the call only enqueues work -- the block table travels through a pinned
 * staging ring owned by the library -- and never waits for the device.  workspace: device
 * buffer of at least ldsr_em_workspace_bytes(...) bytes, 256-byte aligned; it holds the
 * prepared series, the block table and the work-queue heads of THIS call, so calls that may
 * overlap in time (different streams) need separate workspaces. */
size_t ldsr_em_workspace_bytes(int n_series, int T, int p, int q, int n_cells, int algo);
int ldsr_em_batch_device(int device, void *stream, int n_series, int T, int p, int q,
                         const double *d_y, const double *d_u, const double *d_v,
                         int shared_uv, const int *cell_offsets, const double *d_theta0,
                         int niter, double tol, int algo, double *d_theta, double *d_lik,
                         int *d_n_iter, int *d_status, double *d_liks, void *d_workspace,
                         size_t workspace_bytes);

/* Which kernel a call with these arguments launches: writes its name as rocprofv3 prints it
 * (e.g. "em_pair_kernel<1, 2, 32, false>") into buf and returns the resolved algorithm
 * (LDSR_ALGO_SERIAL / LDSR_ALGO_SCAN / LDSR_ALGO_PAIR), or a negative value for unsupported arguments.
 * With LDSR_ALGO_AUTO it assumes a launch large enough to fill the device (below ~3600 cells the
 * entries keep the scan kernel, whose four-cell workgroups spread over more CUs), and with tol > 0
 * it reports what ldsr_em_batch_device runs (the scan kernel); the host-pointer entries additionally
 * take the pair kernel when every series is fully observed. */
int ldsr_em_plan(int T, int p, int q, int niter, double tol, int algo, char *buf, size_t len);

/* The same two with one more piece of knowledge about the data: the first lead_steps time steps of
 * EVERY series are missing (paleo-type series: centuries before the instrumental period; 0 = none
 * or unknown).  LDSR_ALGO_AUTO then handles that lead in closed form and sweeps only the tail
 * (em_pair_impl.h, LEAD).  lead_steps = -1 says the opposite: every y_t of every series is
 * observed -- AUTO then keeps the pair / quad kernels with tol > 0 as the host-pointer entries do
 * for such series (a wrong claim costs speed, not correctness: the kernels look at the mask
 * themselves).  The host-pointer entries find both facts in y. */
int ldsr_em_batch_device_lead(int device, void *stream, int n_series, int T, int p, int q,
                              const double *d_y, const double *d_u, const double *d_v,
                              int shared_uv, const int *cell_offsets, const double *d_theta0,
                              int niter, double tol, int algo, double *d_theta, double *d_lik,
                              int *d_n_iter, int *d_status, double *d_liks, void *d_workspace,
                              size_t workspace_bytes, int lead_steps);
int ldsr_em_plan_lead(int T, int p, int q, int niter, double tol, int algo, int lead_steps,
                      char *buf, size_t len);
/* The kernel one Kalman_smoother pass of this shape runs: LDSR_ALGO_SCAN and the name of the scan
 * kernel's FIT form, or LDSR_ALGO_SERIAL (empty name) beyond its shapes; -1 for bad arguments. */
int ldsr_smooth_plan(int T, int p, int q, char *buf, size_t len);
/* Names (as rocprofv3 prints them, one per line) of every instantiation of the parallel-in-time kernel
 * families compiled into this library; returns the buffer length needed.  The library is built to hold
 * exactly what the plans above can return (tests/test_abi_and_host.py). */
size_t ldsr_kernel_inventory(char *buf, size_t len);
/* Name of the EM kernel of the most recent EM launch on `device` (what AUTO actually chose for that
 * launch's size and data); -1 if there was none. */
int ldsr_last_em_kernel(int device, char *buf, size_t len);

/* Batched Kalman_smoother: one E-step for each cell's theta.  Host pointers.
 * X, Y, V, J: [n_cells][T] (any may be NULL); lik: [n_cells].  stdlik as src/EM.cpp:124. */
int ldsr_smooth_batch(int device, int n_series, int T, int p, int q, const double *y,
                      const double *u, const double *v, int shared_uv,
                      const int *cell_offsets, const double *theta, int stdlik, double *X,
                      double *Y, double *V, double *J, double *lik);

/* Batched Mstep: fit (X, V, J: [n_cells][T]) -> theta [n_cells][6+p+q].  Host pointers. */
int ldsr_mstep_batch(int device, int n_series, int T, int p, int q, const double *y,
                     const double *u, const double *v, int shared_uv,
                     const int *cell_offsets, const double *X, const double *V,
                     const double *J, double *theta, int *status);

/* Batched propagate (no measurement update).  X, Y, V: [n_cells][T]; lik [n_cells]. */
int ldsr_propagate_batch(int device, int n_series, int T, int p, int q, const double *y,
                         const double *u, const double *v, int shared_uv,
                         const int *cell_offsets, const double *theta, int stdlik, double *X,
                         double *Y, double *V, double *lik);

/* Batched penalized_likelihood (R/LDS_GA.R:28-44, the GA / BFGS fitness): for every theta,
 * Kalman_smoother(..., stdlik = FALSE)$lik - lambda * sum_t (Xs[t+1] - A Xs[t] - B u[t])^2.
 * pl: [n_cells].  Only the scalar leaves the device. */
int ldsr_penalized_lik_batch(int device, int n_series, int T, int p, int q, const double *y,
                             const double *u, const double *v, int shared_uv,
                             const int *cell_offsets, const double *theta, double lambda,
                             double *pl);

/* User interrupts during a run (the reference calls Rcpp::checkUserInterrupt() every 100 EM
 * iterations, src/EM.cpp:261-262).  While a callback is registered, the thread that called an
 * EM entry point (ldsr_em_batch[_multi], ldsr_em_restart_grid / _groups) invokes it about once
 * per millisecond while it waits for the device -- never from the library's worker threads; a
 * non-zero return raises a host-pinned flag that every running EM kernel polls every 64
 * iterations: cells stop with status LDSR_CELL_INTERRUPTED and the call returns
 * LDSR_EINTERRUPTED.  NULL unregisters.  The R shim registers R_CheckUserInterrupt wrapped in
 * R_ToplevelExec. */
int ldsr_set_interrupt_callback(int (*callback)(void *), void *arg);

/* Optional kernel timer.  While enabled, every ldsr_em_batch_device call brackets its EM
 * kernel with HIP events on the launch stream; collect() waits for them and returns the summed
 * kernel time and the number of launches since the last collect / enable. */
void ldsr_profile_enable(int on);
int ldsr_profile_collect(double *total_ms, int *n_launches);

/* Restart selection on the host: index of the winning cell among n, or -1. */
int ldsr_select_restart(int n, const double *lik, const double *theta, int p, int q);

/* The reference's five skill metrics (/root/reference/src/utils.cpp: NSE :13, nRMSE :36, corr :49,
 * KGE :68, RE :93; .Call entries _ldsr_NSE .. _ldsr_RE, src/RcppExports.cpp:71-130,137-141).  Host
 * code on n points -- cross-validation folds of 12..46 values: they are here because a DLL that
 * replaces ldsr.so must keep them registered, not because they are hot.  Same argument order as
 * the reference: model output first, observation second. */
double ldsr_metric_nse(int n, const double *yhat, const double *y);
double ldsr_metric_nrmse(int n, const double *yhat, const double *y, double norm_const);
double ldsr_metric_corr(int n, const double *x, const double *y);
double ldsr_metric_kge(int n, const double *yhat, const double *y);
double ldsr_metric_re(int n, const double *yhat, const double *y, double yc_bar);

#ifdef __cplusplus
}
#endif
#endif
