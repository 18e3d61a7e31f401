/*
 * ldsrhip_call.c -- R .Call shim over the C ABI of include/ldsr_hip.h (side-car DLL "ldsrhip").
 *
 * Pure marshalling: SEXP -> plain pointers -> libldsr_hip.so -> SEXP.  Entry points and what
 * they replace in the reference (paths under /root/reference):
 *
 *   ldsrhip_LDS_EM_batch(y, u, v, init, niter, tol)
 *       the per-restart path  foreach(theta0 = init) %dopar% LDS_EM(...)  + the selection of
 *       LDS_EM_restart (R/LDS_reconstruction.R:46-58): ONE call for all restarts
 *   ldsrhip_LDS_EM_grid(Y, u, v, inits, niter, tol)
 *       the fold loop of cvLDS (R/LDS_reconstruction.R:373-375 -> one_lds_cv :270-285): Y is a
 *       T x F matrix, one column per fold (y with that fold's points set to NA, :274); inits is
 *       a list of F init lists (fresh make_init per fold, :275); ONE call for folds x restarts,
 *       returns a list of F winning models
 *   ldsrhip_LDS_EM_batch_raw / ldsrhip_LDS_EM_grid_raw
 *       the same with raw = propagate(theta, u, v, y) of every winner appended: return.raw of
 *       LDS_reconstruction (R/LDS_reconstruction.R:219-222) and use.raw of one_lds_cv (:279-281)
 *   ldsrhip_LDS_EM_groups(Y, us, vs, inits, niter, tol, raw)
 *       the ensemble loops -- foreach(i = seq_along(u)) of LDS_reconstruction (:242-246) and the
 *       nested folds x members loop of cvLDS (:377-381): M members (own p, q) x F folds in ONE call,
 *       members run concurrently (ldsr_em_restart_groups)
 *   ldsrhip_LDS_EM / ldsrhip_Kalman_smoother / ldsrhip_Mstep / ldsrhip_propagate
 *       the four numeric entries of the registration table src/RcppExports.cpp:132-143
 *       (_ldsr_LDS_EM :40, _ldsr_Kalman_smoother :11, _ldsr_Mstep :26, _ldsr_propagate :56),
 *       same arguments and return shapes, for a full replacement of ldsr.so's EM path
 *
 * Argument conventions are the reference's (src/RcppExports.cpp:44-49): y 1xT REALSXP with NA =
 * missing; u, v  pxT / qxT column-major REALSXP, or the 1x1 sentinel `matrix(0)` for an absent
 * input (detected by ncol == 1, as src/EM.cpp:50,71 do); init = list of theta lists looked up BY
 * NAME (src/EM.cpp:25-32); niter integer or double; tol double; stdlik logical.
 *
 * Build where R is installed (not possible in the build container: no R headers):
 *   R CMD SHLIB -o ldsrhip.so ldsrhip_call.c -L<repo>/ldsr_amd -lldsr_hip -I<repo>/include
 * Never longjmps across a HIP call: all device work happens inside ldsr_* calls that return
 * status codes; Rf_error is raised only after they have returned (device buffers are cached by
 * the library and released by ldsr_shutdown() at unload).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "ldsr_hip.h"

static const char *const THETA_NAMES[8] = {"A", "B", "C", "D", "Q", "R", "mu1", "V1"};

static SEXP list_get(SEXP lst, const char *name) {
    SEXP nm = Rf_getAttrib(lst, R_NamesSymbol);
    if (nm != R_NilValue)
        for (R_xlen_t i = 0; i < Rf_xlength(lst); i++)
            if (strcmp(CHAR(STRING_ELT(nm, i)), name) == 0) return VECTOR_ELT(lst, i);
    Rf_error("element '%s' not found", name);
    return R_NilValue;
}

static SEXP mat(int nr, int nc, const double *src) { /* caller PROTECTs */
    SEXP m = Rf_allocMatrix(REALSXP, nr, nc);
    memcpy(REAL(m), src, sizeof(double) * (size_t)nr * nc);
    return m;
}

static SEXP named_list(int n, const char *const *names, SEXP *out_names) { /* both PROTECTed by the caller's count */
    SEXP out = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
    for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    Rf_setAttrib(out, R_NamesSymbol, nm);
    *out_names = nm;
    return out;
}

/* the series arguments shared by every entry: dimensions and absent-input sentinels */
typedef struct {
    int T, p, q, has_u, has_v;
    const double *u, *v;
} inputs_t;

static inputs_t get_inputs(SEXP u, SEXP v, int T) {   /* u, v: already through real_arg() */
    inputs_t in;
    if (!Rf_isReal(u) || !Rf_isReal(v)) Rf_error("u, v must be double matrices");
    in.T = T;
    in.has_u = Rf_ncols(u) > 1;
    in.has_v = Rf_ncols(v) > 1;
    in.p = Rf_nrows(u);
    in.q = Rf_nrows(v);
    if ((in.has_u && Rf_ncols(u) != T) || (in.has_v && Rf_ncols(v) != T))
        Rf_error("u, v must have ncol(y) columns");
    in.u = in.has_u ? REAL(u) : NULL;
    in.v = in.has_v ? REAL(v) : NULL;
    return in;
}

/* named theta list -> packed [A, B(p), C, D(q), Q, R, mu1, V1] */
static void pack_theta(SEXP th, int p, int q, double *dst) {
    const int nc[8] = {1, p, 1, q, 1, 1, 1, 1};
    int o = 0;
    for (int i = 0; i < 8; i++) {
        SEXP e = list_get(th, THETA_NAMES[i]);
        if (!Rf_isReal(e) || Rf_xlength(e) != nc[i]) Rf_error("theta$%s has the wrong length", THETA_NAMES[i]);
        memcpy(dst + o, REAL(e), sizeof(double) * nc[i]);
        o += nc[i];
    }
}

/* packed theta -> named list of matrices (src/EM.cpp:221-228) */
static SEXP theta_to_list(const double *th, int p, int q) {
    const int nc[8] = {1, p, 1, q, 1, 1, 1, 1};
    SEXP nm, out = named_list(8, THETA_NAMES, &nm);
    int o = 0;
    for (int i = 0; i < 8; i++) {
        SET_VECTOR_ELT(out, i, mat(1, nc[i], th + o));
        o += nc[i];
    }
    UNPROTECT(2);
    return out;
}

/* list(X, Y, V, J, lik) of src/EM.cpp:126-130 (J may be NULL: propagate, :352-355) */
static SEXP fit_to_list(int T, const double *X, const double *Y, const double *V, const double *J,
                        double lik) {
    static const char *const n5[5] = {"X", "Y", "V", "J", "lik"};
    static const char *const n4[4] = {"X", "Y", "V", "lik"};
    SEXP nm, out = named_list(J ? 5 : 4, J ? n5 : n4, &nm);
    int k = 0;
    SET_VECTOR_ELT(out, k++, mat(1, T, X));
    SET_VECTOR_ELT(out, k++, mat(1, T, Y));
    SET_VECTOR_ELT(out, k++, mat(1, T, V));
    if (J) SET_VECTOR_ELT(out, k++, mat(1, T, J));
    SET_VECTOR_ELT(out, k, Rf_ScalarReal(lik));
    UNPROTECT(2);
    return out;
}

/* list(theta, fit, liks, lik, index) -- LDS_EM's return (src/EM.cpp:276-279) + the winner's index */
static SEXP model_to_list(const inputs_t *in, const double *theta, const double *X, const double *Y,
                          const double *V, const double *J, const double *liks, int n_iter,
                          double lik, int index1) {
    static const char *const nms[5] = {"theta", "fit", "liks", "lik", "index"};
    SEXP nm, out = named_list(5, nms, &nm);
    SET_VECTOR_ELT(out, 0, theta_to_list(theta, in->p, in->q));
    SET_VECTOR_ELT(out, 1, fit_to_list(in->T, X, Y, V, J, lik));
    SET_VECTOR_ELT(out, 2, mat(n_iter, 1, liks)); /* arma::vec -> n x 1 */
    SET_VECTOR_ELT(out, 3, Rf_ScalarReal(lik));
    SET_VECTOR_ELT(out, 4, Rf_ScalarInteger(index1));
    UNPROTECT(2);
    return out;
}

/* User interrupts while the GPU works: the library calls this about once per millisecond from the
 * .Call's own thread; R_CheckUserInterrupt would longjmp, so it runs inside R_ToplevelExec, which
 * reports a pending interrupt as FALSE instead (the reference polls every 100 EM iterations,
 * src/EM.cpp:261-262). */
static void check_interrupt(void *unused) { (void)unused; R_CheckUserInterrupt(); }
static int poll_interrupt(void *unused) { (void)unused; return R_ToplevelExec(check_interrupt, NULL) == FALSE; }

static int *all_devices(int n_cells, int *n_dev) {
    int n = ldsr_device_count(); /* restarts shard over every GPU of the node, no collective */
    if (n < 1) Rf_error("ldsrhip: no ROCm device visible");
    if (n > n_cells) n = n_cells;
    int *devs = (int *)R_alloc(n, sizeof(int));
    for (int d = 0; d < n; d++) devs[d] = d;
    *n_dev = n;
    return devs;
}

/* double view of a numeric argument: the reference's Rcpp / arma parameters coerce integer and
 * logical vectors silently (NSE(1:10, obs) works), so does the shim.  Whatever it allocates is
 * PROTECTed and counted in *np. */
static SEXP real_arg(SEXP x, int *np, const char *what) {
    if (Rf_isReal(x)) return x;
    if (TYPEOF(x) != INTSXP && TYPEOF(x) != LGLSXP) Rf_error("%s must be numeric", what);
    SEXP r = PROTECT(Rf_coerceVector(x, REALSXP));
    (*np)++;
    return r;
}

/* model list of one winner plus, on request, raw = propagate(theta, u, v, y) (src/EM.cpp:295-356:
 * list(X, Y, V, lik)) -- what one_lds_cv returns under use.raw (R/LDS_reconstruction.R:279-281) and
 * LDS_reconstruction attaches under return.raw (:219-222) */
static SEXP model_with_raw(SEXP m, int T, const double *rX, const double *rY, const double *rV, double rlik) {
    static const char *const nms[6] = {"theta", "fit", "liks", "lik", "index", "raw"};
    SEXP nm, m2 = named_list(6, nms, &nm);
    for (int i = 0; i < 5; i++) SET_VECTOR_ELT(m2, i, VECTOR_ELT(m, i));
    SET_VECTOR_ELT(m2, 5, fit_to_list(T, rX, rY, rV, NULL, rlik));
    UNPROTECT(2);
    return m2;
}

/* Shared body of the batch (one series), grid (F folds) and groups (M ensemble members x F folds)
 * entries.  Y: [F][T] (every member sees the same y: R/LDS_reconstruction.R:242-246, :377-381);
 * ins[m]: member m's inputs; inits[m * F + f]: the init list of member m, fold f.  Returns a list of
 * M lists of F models.  One member: ONE ldsr_em_restart_grid call; several: ldsr_em_restart_groups
 * runs the members concurrently (one host thread, stream and arena each). */
static SEXP em_members(const double *Y, int F, int M, const inputs_t *ins, SEXP *inits, int niter,
                       double tol, int want_all, int want_raw) {
    const int T = ins[0].T;
    if (niter < 2) Rf_error("niter must be >= 2");
    ldsr_group *g = (ldsr_group *)R_alloc((size_t)M, sizeof(ldsr_group));
    int **offs = (int **)R_alloc((size_t)M, sizeof(int *));
    int n_max = 1;
    for (int m = 0; m < M; m++) {
        const inputs_t *in = &ins[m];
        const int p = in->p, q = in->q, P = 6 + p + q;
        int *off = offs[m] = (int *)R_alloc((size_t)F + 1, sizeof(int));
        off[0] = 0;
        for (int f = 0; f < F; f++) {
            const int n = (int)Rf_xlength(inits[m * F + f]);
            if (n < 1) Rf_error("init is empty");
            off[f + 1] = off[f] + n;
        }
        const int n = off[F];
        if (n > n_max) n_max = n;
        double *th0 = (double *)R_alloc((size_t)n * P, sizeof(double));
        for (int f = 0; f < F; f++)
            for (int c = off[f]; c < off[f + 1]; c++)
                pack_theta(VECTOR_ELT(inits[m * F + f], c - off[f]), p, q, th0 + (size_t)c * P);
        memset(&g[m], 0, sizeof(ldsr_group));
        g[m].n_series = F; g[m].T = T; g[m].p = p; g[m].q = q; g[m].shared_uv = 1;
        g[m].y = Y; g[m].u = in->u; g[m].v = in->v;
        g[m].cell_offsets = off;
        g[m].theta0 = th0;
        g[m].status_all = (int *)R_alloc(n, sizeof(int));
        if (want_all) {
            g[m].theta_all = (double *)R_alloc((size_t)n * P, sizeof(double));
            g[m].lik_all = (double *)R_alloc(n, sizeof(double));
            g[m].n_iter_all = (int *)R_alloc(n, sizeof(int));
        }
        g[m].winner = (int *)R_alloc(F, sizeof(int));
        g[m].n_iter_w = (int *)R_alloc(F, sizeof(int));
        g[m].theta_w = (double *)R_alloc((size_t)F * P, sizeof(double));
        g[m].lik_w = (double *)R_alloc(F, sizeof(double));
        g[m].liks_w = (double *)R_alloc((size_t)F * niter, sizeof(double));
        g[m].X = (double *)R_alloc((size_t)4 * F * T, sizeof(double));
        g[m].Y = g[m].X + (size_t)F * T; g[m].V = g[m].Y + (size_t)F * T; g[m].J = g[m].V + (size_t)F * T;
    }
    R_CheckUserInterrupt();
    int n_dev;
    int *devs = all_devices(M > 1 ? M * n_max : n_max, &n_dev);
    (void)ldsr_set_interrupt_callback(poll_interrupt, NULL); /* polled during the run, see above */
    int rc;
    if (M == 1)
        rc = ldsr_em_restart_grid(n_dev, devs, F, T, g[0].p, g[0].q, Y, g[0].u, g[0].v, 1, g[0].cell_offsets,
                                  g[0].theta0, niter, tol, LDSR_ALGO_AUTO, g[0].theta_all, g[0].lik_all,
                                  g[0].n_iter_all, g[0].status_all, g[0].winner, g[0].theta_w, g[0].lik_w,
                                  g[0].n_iter_w, g[0].liks_w, g[0].X, g[0].Y, g[0].V, g[0].J);
    else
        rc = ldsr_em_restart_groups(n_dev, devs, M, g, niter, tol, LDSR_ALGO_AUTO);
    if (rc == LDSR_EINTERRUPTED) Rf_error("ldsrhip: interrupted by the user");
    if (rc != LDSR_OK) Rf_error("%s: %s", M == 1 ? "ldsr_em_restart_grid" : "ldsr_em_restart_groups", ldsr_last_error());
    for (int m = 0; m < M; m++) {
        for (int c = 0; c < offs[m][F]; c++)
            if (g[m].status_all[c] == LDSR_CELL_SINGULAR) Rf_error("inv(): matrix is singular"); /* arma::inv throws */
        for (int f = 0; f < F; f++)
            if (g[m].winner[f] < 0) Rf_error("no restart produced a finite likelihood (fold %d)", f + 1);
    }
    /* raw: propagate() at every winner's theta -- F cells of one more (small) batched call per member */
    double *raw = NULL, *rlik = NULL;
    if (want_raw) {
        raw = (double *)R_alloc((size_t)3 * M * F * T, sizeof(double));
        rlik = (double *)R_alloc((size_t)M * F, sizeof(double));
        int *off1 = (int *)R_alloc((size_t)F + 1, sizeof(int));
        for (int f = 0; f <= F; f++) off1[f] = f;
        for (int m = 0; m < M; m++) {
            double *rX = raw + (size_t)3 * m * F * T, *rY = rX + (size_t)F * T, *rV = rY + (size_t)F * T;
            const int rcp = ldsr_propagate_batch(devs[m % n_dev], F, T, g[m].p, g[m].q, Y, g[m].u, g[m].v, 1, off1,
                                                 g[m].theta_w, 1, rX, rY, rV, rlik + (size_t)m * F);
            if (rcp != LDSR_OK) Rf_error("ldsr_propagate_batch: %s", ldsr_last_error());
        }
    }

    SEXP outer = PROTECT(Rf_allocVector(VECSXP, M));
    for (int m = 0; m < M; m++) {
        const int P = 6 + g[m].p + g[m].q;
        SEXP out = PROTECT(Rf_allocVector(VECSXP, F));
        for (int f = 0; f < F; f++) {
            SEXP mod = PROTECT(model_to_list(&ins[m], g[m].theta_w + (size_t)f * P, g[m].X + (size_t)f * T,
                                             g[m].Y + (size_t)f * T, g[m].V + (size_t)f * T, g[m].J + (size_t)f * T,
                                             g[m].liks_w + (size_t)f * niter, g[m].n_iter_w[f], g[m].lik_w[f],
                                             g[m].winner[f] - offs[m][f] + 1));
            if (want_raw) {
                const double *rX = raw + (size_t)3 * m * F * T, *rY = rX + (size_t)F * T, *rV = rY + (size_t)F * T;
                mod = model_with_raw(mod, T, rX + (size_t)f * T, rY + (size_t)f * T, rV + (size_t)f * T, rlik[(size_t)m * F + f]);
            }
            SET_VECTOR_ELT(out, f, mod);
            UNPROTECT(1);
        }
        if (want_all) { /* per-restart summary of the member's first series: list(lik, C, n_iter, status) */
            const int n = offs[m][1], p = g[m].p;
            static const char *const anm[4] = {"lik", "C", "n_iter", "status"};
            SEXP nm, all = named_list(4, anm, &nm);
            SEXP a_lik = PROTECT(Rf_allocVector(REALSXP, n)), a_C = PROTECT(Rf_allocVector(REALSXP, n));
            SEXP a_it = PROTECT(Rf_allocVector(INTSXP, n)), a_st = PROTECT(Rf_allocVector(INTSXP, n));
            for (int c = 0; c < n; c++) {
                REAL(a_lik)[c] = g[m].lik_all[c];
                REAL(a_C)[c] = g[m].theta_all[(size_t)c * P + 1 + p];
                INTEGER(a_it)[c] = g[m].n_iter_all[c];
                INTEGER(a_st)[c] = g[m].status_all[c];
            }
            SET_VECTOR_ELT(all, 0, a_lik); SET_VECTOR_ELT(all, 1, a_C);
            SET_VECTOR_ELT(all, 2, a_it); SET_VECTOR_ELT(all, 3, a_st);
            /* append `all` to the first model */
            SEXP m0 = VECTOR_ELT(out, 0);
            const int k0 = (int)Rf_xlength(m0);
            SEXP nm0 = Rf_getAttrib(m0, R_NamesSymbol);
            SEXP m2 = PROTECT(Rf_allocVector(VECSXP, k0 + 1)), nm2 = PROTECT(Rf_allocVector(STRSXP, k0 + 1));
            for (int i = 0; i < k0; i++) {
                SET_VECTOR_ELT(m2, i, VECTOR_ELT(m0, i));
                SET_STRING_ELT(nm2, i, STRING_ELT(nm0, i));
            }
            SET_VECTOR_ELT(m2, k0, all);
            SET_STRING_ELT(nm2, k0, Rf_mkChar("all"));
            Rf_setAttrib(m2, R_NamesSymbol, nm2);
            SET_VECTOR_ELT(out, 0, m2);
            UNPROTECT(8);
        }
        SET_VECTOR_ELT(outer, m, out);
        UNPROTECT(1);
    }
    UNPROTECT(1);
    return outer;
}

static int flag_arg(SEXP x) { return Rf_asLogical(x) != 0; }

/* .Call("ldsrhip_LDS_EM_batch", y, u, v, init, niter, tol):
 * list(theta, fit = list(X, Y, V, J, lik), liks, lik, index, all = list(lik, C, n_iter, status)) */
static SEXP em_batch_entry(SEXP y, SEXP u, SEXP v, SEXP init, SEXP niterS, SEXP tolS, int want_raw) {
    int np = 0;
    y = real_arg(y, &np, "y"); u = real_arg(u, &np, "u"); v = real_arg(v, &np, "v");
    const inputs_t in = get_inputs(u, v, Rf_ncols(y));
    SEXP res = PROTECT(em_members(REAL(y), 1, 1, &in, &init, Rf_asInteger(niterS), Rf_asReal(tolS), 1, want_raw));
    SEXP out = VECTOR_ELT(VECTOR_ELT(res, 0), 0);
    UNPROTECT(1 + np);
    return out;
}
SEXP ldsrhip_LDS_EM_batch(SEXP y, SEXP u, SEXP v, SEXP init, SEXP niterS, SEXP tolS) {
    return em_batch_entry(y, u, v, init, niterS, tolS, 0);
}
/* ... and with raw = propagate(theta, u, v, y) of the winner appended (return.raw, R/LDS_reconstruction.R:219-222) */
SEXP ldsrhip_LDS_EM_batch_raw(SEXP y, SEXP u, SEXP v, SEXP init, SEXP niterS, SEXP tolS) {
    return em_batch_entry(y, u, v, init, niterS, tolS, 1);
}

/* .Call("ldsrhip_LDS_EM_grid", Y, u, v, inits, niter, tol): Y is T x F (one column per fold),
 * inits a list of F init lists; returns a list of F models list(theta, fit, liks, lik, index). */
static SEXP em_grid_entry(SEXP Y, SEXP u, SEXP v, SEXP inits, SEXP niterS, SEXP tolS, int want_raw) {
    int np = 0;
    Y = real_arg(Y, &np, "Y"); u = real_arg(u, &np, "u"); v = real_arg(v, &np, "v");
    const int F = Rf_ncols(Y);
    if (TYPEOF(inits) != VECSXP || (int)Rf_xlength(inits) != F) Rf_error("inits must have one init list per column of Y");
    const inputs_t in = get_inputs(u, v, Rf_nrows(Y));
    SEXP *il = (SEXP *)R_alloc(F, sizeof(SEXP));
    for (int f = 0; f < F; f++) il[f] = VECTOR_ELT(inits, f);
    SEXP res = PROTECT(em_members(REAL(Y), F, 1, &in, il, Rf_asInteger(niterS), Rf_asReal(tolS), 0, want_raw));
    SEXP out = VECTOR_ELT(res, 0);
    UNPROTECT(1 + np);
    return out;
}
SEXP ldsrhip_LDS_EM_grid(SEXP Y, SEXP u, SEXP v, SEXP inits, SEXP niterS, SEXP tolS) {
    return em_grid_entry(Y, u, v, inits, niterS, tolS, 0);
}
/* ... each model with raw = propagate() at its theta (one_lds_cv's use.raw, R/LDS_reconstruction.R:279-281) */
SEXP ldsrhip_LDS_EM_grid_raw(SEXP Y, SEXP u, SEXP v, SEXP inits, SEXP niterS, SEXP tolS) {
    return em_grid_entry(Y, u, v, inits, niterS, tolS, 1);
}

/* .Call("ldsrhip_LDS_EM_groups", Y, us, vs, inits, niter, tol, raw): the ensemble loops of the
 * reference in ONE call -- LDS_reconstruction's  foreach(i = seq_along(u))  (R/LDS_reconstruction.R:242-246)
 * and cvLDS's nested  foreach(z = Z) %:% foreach(i = seq_along(u))  (:377-381).  Y is T x F (F = 1 for a
 * reconstruction, one column per fold for cvLDS); us, vs are lists of M input matrices (members may
 * differ in p and q: tests/testthat/test-ensemble.R:4-5; matrix(0) = absent); inits is a list of M
 * lists of F init lists.  Returns a list of M lists of F models (with $raw when raw is TRUE).
 * The members run concurrently on the device(s) (ldsr_em_restart_groups). */
SEXP ldsrhip_LDS_EM_groups(SEXP Y, SEXP us, SEXP vs, SEXP inits, SEXP niterS, SEXP tolS, SEXP rawS) {
    int np = 0;
    Y = real_arg(Y, &np, "Y");
    const int F = Rf_ncols(Y), T = Rf_nrows(Y);
    if (TYPEOF(us) != VECSXP || TYPEOF(vs) != VECSXP || TYPEOF(inits) != VECSXP)
        Rf_error("us, vs and inits must be lists (one element per ensemble member)");
    const int M = (int)Rf_xlength(us);
    if (M < 1 || (int)Rf_xlength(vs) != M || (int)Rf_xlength(inits) != M)
        Rf_error("us, vs and inits must have one element per ensemble member");
    inputs_t *ins = (inputs_t *)R_alloc((size_t)M, sizeof(inputs_t));
    SEXP *il = (SEXP *)R_alloc((size_t)M * F, sizeof(SEXP));
    for (int m = 0; m < M; m++) {
        SEXP u = real_arg(VECTOR_ELT(us, m), &np, "u"), v = real_arg(VECTOR_ELT(vs, m), &np, "v");
        ins[m] = get_inputs(u, v, T);
        SEXP im = VECTOR_ELT(inits, m);
        if (TYPEOF(im) != VECSXP || (int)Rf_xlength(im) != F)
            Rf_error("inits[[%d]] must hold one init list per column of Y", m + 1);
        for (int f = 0; f < F; f++) il[m * F + f] = VECTOR_ELT(im, f);
    }
    SEXP res = PROTECT(em_members(REAL(Y), F, M, ins, il, Rf_asInteger(niterS), Rf_asReal(tolS), 0, flag_arg(rawS)));
    UNPROTECT(1 + np);
    return res;
}

/* .Call("ldsrhip_LDS_EM", y, u, v, theta0, niter, tol)  ==  _ldsr_LDS_EM (src/RcppExports.cpp:40-53) */
SEXP ldsrhip_LDS_EM(SEXP y, SEXP u, SEXP v, SEXP theta0, SEXP niterS, SEXP tolS) {
    int np = 0;
    y = real_arg(y, &np, "y"); u = real_arg(u, &np, "u"); v = real_arg(v, &np, "v");
    const int T = Rf_ncols(y), niter = Rf_asInteger(niterS);
    const inputs_t in = get_inputs(u, v, T);
    const int P = 6 + in.p + in.q, off[2] = {0, 1};
    if (niter < 2) Rf_error("niter must be >= 2");
    double *th0 = (double *)R_alloc((size_t)2 * P, sizeof(double)), *th = th0 + P;
    pack_theta(theta0, in.p, in.q, th0);
    double *liks = (double *)R_alloc((size_t)niter + 4 * (size_t)T, sizeof(double));
    double *X = liks + niter, *Y = X + T, *V = Y + T, *J = V + T, lik, flik;
    int n_iter, status;
    R_CheckUserInterrupt();
    (void)ldsr_set_interrupt_callback(poll_interrupt, NULL);
    int rc = ldsr_em_batch(0, 1, T, in.p, in.q, REAL(y), in.u, in.v, 0, off, th0, niter, Rf_asReal(tolS),
                           LDSR_ALGO_AUTO, th, &lik, &n_iter, &status, liks);
    if (rc == LDSR_EINTERRUPTED) Rf_error("ldsrhip: interrupted by the user");
    if (rc != LDSR_OK) Rf_error("ldsr_em_batch: %s", ldsr_last_error());
    if (status == LDSR_CELL_SINGULAR) Rf_error("inv(): matrix is singular");
    rc = ldsr_smooth_batch(0, 1, T, in.p, in.q, REAL(y), in.u, in.v, 0, off, th, 1, X, Y, V, J, &flik);
    if (rc != LDSR_OK) Rf_error("ldsr_smooth_batch: %s", ldsr_last_error());
    SEXP m = PROTECT(model_to_list(&in, th, X, Y, V, J, liks, n_iter, lik, 1));
    static const char *const nms[4] = {"theta", "fit", "liks", "lik"};
    SEXP nm, out = named_list(4, nms, &nm);
    for (int i = 0; i < 4; i++) SET_VECTOR_ELT(out, i, VECTOR_ELT(m, i));
    UNPROTECT(3 + np);
    return out;
}

/* .Call("ldsrhip_Kalman_smoother", y, u, v, theta, stdlik)  ==  _ldsr_Kalman_smoother (:11-24) */
SEXP ldsrhip_Kalman_smoother(SEXP y, SEXP u, SEXP v, SEXP theta, SEXP stdlikS) {
    int np = 0;
    y = real_arg(y, &np, "y"); u = real_arg(u, &np, "u"); v = real_arg(v, &np, "v");
    const int T = Rf_ncols(y), off[2] = {0, 1};
    const inputs_t in = get_inputs(u, v, T);
    double *th = (double *)R_alloc((size_t)(6 + in.p + in.q) + 4 * (size_t)T, sizeof(double));
    double *X = th + 6 + in.p + in.q, *Y = X + T, *V = Y + T, *J = V + T, lik;
    pack_theta(theta, in.p, in.q, th);
    const int rc = ldsr_smooth_batch(0, 1, T, in.p, in.q, REAL(y), in.u, in.v, 0, off, th,
                                     Rf_asLogical(stdlikS) != 0, X, Y, V, J, &lik);
    if (rc != LDSR_OK) Rf_error("ldsr_smooth_batch: %s", ldsr_last_error());
    SEXP out = fit_to_list(T, X, Y, V, J, lik);
    UNPROTECT(np);
    return out;
}

/* .Call("ldsrhip_propagate", theta, u, v, y, stdlik)  ==  _ldsr_propagate (:56-69) */
SEXP ldsrhip_propagate(SEXP theta, SEXP u, SEXP v, SEXP y, SEXP stdlikS) {
    int np = 0;
    y = real_arg(y, &np, "y"); u = real_arg(u, &np, "u"); v = real_arg(v, &np, "v");
    const int T = Rf_ncols(y), off[2] = {0, 1};
    const inputs_t in = get_inputs(u, v, T);
    double *th = (double *)R_alloc((size_t)(6 + in.p + in.q) + 3 * (size_t)T, sizeof(double));
    double *X = th + 6 + in.p + in.q, *Y = X + T, *V = Y + T, lik;
    pack_theta(theta, in.p, in.q, th);
    const int rc = ldsr_propagate_batch(0, 1, T, in.p, in.q, REAL(y), in.u, in.v, 0, off, th,
                                        Rf_asLogical(stdlikS) != 0, X, Y, V, &lik);
    if (rc != LDSR_OK) Rf_error("ldsr_propagate_batch: %s", ldsr_last_error());
    SEXP out = fit_to_list(T, X, Y, V, NULL, lik);
    UNPROTECT(np);
    return out;
}

/* .Call("ldsrhip_Mstep", y, u, v, fit)  ==  _ldsr_Mstep (:26-38) */
SEXP ldsrhip_Mstep(SEXP y, SEXP u, SEXP v, SEXP fit) {
    int np = 0;
    y = real_arg(y, &np, "y"); u = real_arg(u, &np, "u"); v = real_arg(v, &np, "v");
    const int T = Rf_ncols(y), off[2] = {0, 1};
    const inputs_t in = get_inputs(u, v, T);
    SEXP X = list_get(fit, "X"), V = list_get(fit, "V"), J = list_get(fit, "J");
    if (!Rf_isReal(X) || !Rf_isReal(V) || !Rf_isReal(J) || Rf_xlength(X) != T || Rf_xlength(V) != T ||
        Rf_xlength(J) != T)
        Rf_error("fit$X, fit$V, fit$J must be double vectors of length ncol(y)");
    double *th = (double *)R_alloc((size_t)(6 + in.p + in.q), sizeof(double));
    int status;
    const int rc = ldsr_mstep_batch(0, 1, T, in.p, in.q, REAL(y), in.u, in.v, 0, off, REAL(X), REAL(V),
                                    REAL(J), th, &status);
    if (rc != LDSR_OK) Rf_error("ldsr_mstep_batch: %s", ldsr_last_error());
    if (status == LDSR_CELL_SINGULAR) Rf_error("inv(): matrix is singular");
    SEXP out = theta_to_list(th, in.p, in.q);
    UNPROTECT(np);
    return out;
}

/* ---- the reference's five metric entries (src/RcppExports.cpp:71-130): host code of the library */
/* (integer / logical vectors are coerced like Rcpp's NumericVector parameters do: NSE(1:10, obs)) */
static int two_reals(SEXP *a, SEXP *b, int *np) {
    *a = real_arg(*a, np, "the first argument");
    *b = real_arg(*b, np, "the second argument");
    if (Rf_xlength(*a) != Rf_xlength(*b) || Rf_xlength(*a) < 1)
        Rf_error("two numeric vectors of one (positive) length expected");
    return (int)Rf_xlength(*a);
}
#define METRIC2(NAME, FN)                                                      \
    SEXP NAME(SEXP a, SEXP b) {                                                \
        int np = 0;                                                            \
        const int n = two_reals(&a, &b, &np);                                  \
        const double r = FN(n, REAL(a), REAL(b));                              \
        UNPROTECT(np);                                                         \
        return Rf_ScalarReal(r);                                               \
    }
#define METRIC3(NAME, FN)                                                      \
    SEXP NAME(SEXP a, SEXP b, SEXP c) {                                        \
        int np = 0;                                                            \
        const int n = two_reals(&a, &b, &np);                                  \
        const double r = FN(n, REAL(a), REAL(b), Rf_asReal(c));                \
        UNPROTECT(np);                                                         \
        return Rf_ScalarReal(r);                                               \
    }
METRIC2(ldsrhip_NSE, ldsr_metric_nse)
METRIC3(ldsrhip_nRMSE, ldsr_metric_nrmse)
METRIC2(ldsrhip_corr, ldsr_metric_corr)
METRIC2(ldsrhip_KGE, ldsr_metric_kge)
METRIC3(ldsrhip_RE, ldsr_metric_re)

static const R_CallMethodDef CallEntries[] = {
    {"ldsrhip_LDS_EM_batch", (DL_FUNC)&ldsrhip_LDS_EM_batch, 6},
    {"ldsrhip_LDS_EM_grid", (DL_FUNC)&ldsrhip_LDS_EM_grid, 6},
    {"ldsrhip_LDS_EM_batch_raw", (DL_FUNC)&ldsrhip_LDS_EM_batch_raw, 6},
    {"ldsrhip_LDS_EM_grid_raw", (DL_FUNC)&ldsrhip_LDS_EM_grid_raw, 6},
    {"ldsrhip_LDS_EM_groups", (DL_FUNC)&ldsrhip_LDS_EM_groups, 7},
    {"ldsrhip_LDS_EM", (DL_FUNC)&ldsrhip_LDS_EM, 6},
    {"ldsrhip_Kalman_smoother", (DL_FUNC)&ldsrhip_Kalman_smoother, 5},
    {"ldsrhip_propagate", (DL_FUNC)&ldsrhip_propagate, 5},
    {"ldsrhip_Mstep", (DL_FUNC)&ldsrhip_Mstep, 4},
    {NULL, NULL, 0}};

void R_init_ldsrhip(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_ldsrhip(DllInfo *dll) {
    (void)dll;
    ldsr_shutdown();
}

/* Deployment shape (ii), full replacement: built as ldsr.so this same object answers R's
 * `useDynLib(ldsr, .registration = TRUE)` (NAMESPACE:29) with EXACTLY the reference's table --
 * the nine names and arities of src/RcppExports.cpp:132-143 -- so R/RcppExports.R:15-59 keeps
 * working unchanged, plus the two batched entries the replaced LDS_EM_restart / cvLDS call. */
static const R_CallMethodDef CallEntriesLdsr[] = {
    {"_ldsr_Kalman_smoother", (DL_FUNC)&ldsrhip_Kalman_smoother, 5},
    {"_ldsr_Mstep", (DL_FUNC)&ldsrhip_Mstep, 4},
    {"_ldsr_LDS_EM", (DL_FUNC)&ldsrhip_LDS_EM, 6},
    {"_ldsr_propagate", (DL_FUNC)&ldsrhip_propagate, 5},
    {"_ldsr_NSE", (DL_FUNC)&ldsrhip_NSE, 2},
    {"_ldsr_nRMSE", (DL_FUNC)&ldsrhip_nRMSE, 3},
    {"_ldsr_corr", (DL_FUNC)&ldsrhip_corr, 2},
    {"_ldsr_KGE", (DL_FUNC)&ldsrhip_KGE, 2},
    {"_ldsr_RE", (DL_FUNC)&ldsrhip_RE, 3},
    {"ldsrhip_LDS_EM_batch", (DL_FUNC)&ldsrhip_LDS_EM_batch, 6},
    {"ldsrhip_LDS_EM_grid", (DL_FUNC)&ldsrhip_LDS_EM_grid, 6},
    {"ldsrhip_LDS_EM_batch_raw", (DL_FUNC)&ldsrhip_LDS_EM_batch_raw, 6},
    {"ldsrhip_LDS_EM_grid_raw", (DL_FUNC)&ldsrhip_LDS_EM_grid_raw, 6},
    {"ldsrhip_LDS_EM_groups", (DL_FUNC)&ldsrhip_LDS_EM_groups, 7},
    {NULL, NULL, 0}};

void R_init_ldsr(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntriesLdsr, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_ldsr(DllInfo *dll) {
    (void)dll;
    ldsr_shutdown();
}
