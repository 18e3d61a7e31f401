/*
 * ldsrhip_call.c -- R .Call shim over the C ABI of include/ldsr_hip.h (side-car DLL "ldsrhip").
 *
 * Pure marshalling: SEXP -> plain pointers -> libldsr_hip.so -> SEXP.  It replaces the
 * per-restart path  foreach(theta0 = init) %dopar% LDS_EM(...)  of the reference
 * (R/LDS_reconstruction.R:46 -> R/RcppExports.R:41-43 -> src/RcppExports.cpp:40-53) with ONE
 * call for all restarts.  Argument conventions are the reference's (src/RcppExports.cpp:44-49):
 * y 1xT REALSXP with NA = missing; u, v  pxT / qxT column-major REALSXP, or the 1x1 sentinel
 * `matrix(0)` for an absent input (detected by ncol == 1, as src/EM.cpp:50,71 do); init = list
 * of theta lists looked up BY NAME (src/EM.cpp:25-32); niter integer or double; tol double.
 *
 * Build where R is installed (not possible in the build container: no R headers):
 *   R CMD SHLIB -o ldsrhip.so ldsrhip_call.c -L<repo>/ldsr_amd -lldsr_hip -I<repo>/include
 * Never longjmps across a HIP call: all device work happens inside ldsr_* calls that return
 * status codes; Rf_error is raised only after they have returned and freed device memory.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "ldsr_hip.h"

static SEXP list_get(SEXP lst, const char *name) {
    SEXP nm = Rf_getAttrib(lst, R_NamesSymbol);
    for (R_xlen_t i = 0; i < Rf_xlength(lst); i++)
        if (strcmp(CHAR(STRING_ELT(nm, i)), name) == 0) return VECTOR_ELT(lst, i);
    Rf_error("theta: element '%s' not found", name);
    return R_NilValue;
}

static SEXP mat(int nr, int nc, const double *src) { /* caller PROTECTs */
    SEXP m = Rf_allocMatrix(REALSXP, nr, nc);
    memcpy(REAL(m), src, sizeof(double) * (size_t)nr * nc);
    return m;
}

/* packed theta [A, B(p), C, D(q), Q, R, mu1, V1] -> named list of matrices (src/EM.cpp:221-228) */
static SEXP theta_to_list(const double *th, int p, int q) {
    static const char *nms[] = {"A", "B", "C", "D", "Q", "R", "mu1", "V1"};
    const int nc[] = {1, p, 1, q, 1, 1, 1, 1};
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 8)), names = PROTECT(Rf_allocVector(STRSXP, 8));
    int o = 0;
    for (int i = 0; i < 8; i++) {
        SET_VECTOR_ELT(out, i, mat(1, nc[i], th + o));
        SET_STRING_ELT(names, i, Rf_mkChar(nms[i]));
        o += nc[i];
    }
    Rf_setAttrib(out, R_NamesSymbol, names);
    UNPROTECT(2);
    return out;
}

/* .Call("ldsrhip_LDS_EM_batch", y, u, v, init, niter, tol):
 * list(theta, fit = list(X, Y, V, J, lik), liks, lik, index, all = list(lik, C, n_iter, status)) */
SEXP ldsrhip_LDS_EM_batch(SEXP y, SEXP u, SEXP v, SEXP init, SEXP niterS, SEXP tolS) {
    if (!Rf_isReal(y) || !Rf_isReal(u) || !Rf_isReal(v)) Rf_error("y, u, v must be double matrices");
    const int T = Rf_ncols(y);
    const int has_u = Rf_ncols(u) > 1, has_v = Rf_ncols(v) > 1;
    const int p = Rf_nrows(u), q = Rf_nrows(v), P = 6 + p + q;
    if ((has_u && Rf_ncols(u) != T) || (has_v && Rf_ncols(v) != T)) Rf_error("u, v must have ncol(y) columns");
    const int n = (int)Rf_xlength(init), niter = Rf_asInteger(niterS);
    const double tol = Rf_asReal(tolS);
    if (n < 1) Rf_error("init is empty");
    double *th0 = (double *)R_alloc((size_t)n * P, sizeof(double));
    static const char *nms[] = {"A", "B", "C", "D", "Q", "R", "mu1", "V1"};
    const int nc[] = {1, p, 1, q, 1, 1, 1, 1};
    for (int c = 0; c < n; c++) {
        SEXP th = VECTOR_ELT(init, c);
        int o = 0;
        for (int i = 0; i < 8; i++) {
            SEXP e = list_get(th, nms[i]);
            if (!Rf_isReal(e) || Rf_xlength(e) != nc[i]) Rf_error("theta$%s has the wrong length", nms[i]);
            memcpy(th0 + (size_t)c * P + o, REAL(e), sizeof(double) * nc[i]);
            o += nc[i];
        }
    }
    double *theta = (double *)R_alloc((size_t)n * P, sizeof(double));
    double *lik = (double *)R_alloc(n, sizeof(double));
    double *liks = (double *)R_alloc((size_t)n * niter, sizeof(double));
    int *n_iter = (int *)R_alloc(n, sizeof(int)), *status = (int *)R_alloc(n, sizeof(int));
    const int off[2] = {0, n};
    R_CheckUserInterrupt(); /* the reference polls every 100 iterations (src/EM.cpp:261-262) */
    int n_dev = ldsr_device_count(); /* restarts shard over every GPU of the node, no collective */
    if (n_dev < 1) Rf_error("ldsrhip: no ROCm device visible");
    if (n_dev > n) n_dev = n;
    int *devs = (int *)R_alloc(n_dev, sizeof(int));
    for (int d = 0; d < n_dev; d++) devs[d] = d;
    int rc = ldsr_em_batch_multi(n_dev, devs, 1, T, p, q, REAL(y), has_u ? REAL(u) : NULL,
                                 has_v ? REAL(v) : NULL, 0, off, th0, niter, tol, LDSR_ALGO_AUTO,
                                 theta, lik, n_iter, status, liks);
    if (rc != LDSR_OK) Rf_error("ldsr_em_batch_multi: %s", ldsr_last_error());
    for (int c = 0; c < n; c++)
        if (status[c] == LDSR_CELL_SINGULAR) Rf_error("inv(): matrix is singular"); /* arma::inv throws */
    const int k = ldsr_select_restart(n, lik, theta, p, q); /* R/LDS_reconstruction.R:50-58 */
    if (k < 0) Rf_error("no restart produced a finite likelihood");
    double *X = (double *)R_alloc((size_t)4 * T, sizeof(double)), *Y = X + T, *V = Y + T, *J = V + T;
    double flik;
    rc = ldsr_smooth_batch(0, 1, T, p, q, REAL(y), has_u ? REAL(u) : NULL, has_v ? REAL(v) : NULL, 0,
                           (const int[]){0, 1}, theta + (size_t)k * P, 1, X, Y, V, J, &flik);
    if (rc != LDSR_OK) Rf_error("ldsr_smooth_batch: %s", ldsr_last_error());

    SEXP fit = PROTECT(Rf_allocVector(VECSXP, 5)), fn = PROTECT(Rf_allocVector(STRSXP, 5));
    const char *fnm[] = {"X", "Y", "V", "J", "lik"};
    const double *fv[] = {X, Y, V, J};
    for (int i = 0; i < 4; i++) SET_VECTOR_ELT(fit, i, mat(1, T, fv[i]));
    SET_VECTOR_ELT(fit, 4, Rf_ScalarReal(flik));
    for (int i = 0; i < 5; i++) SET_STRING_ELT(fn, i, Rf_mkChar(fnm[i]));
    Rf_setAttrib(fit, R_NamesSymbol, fn);

    SEXP all = PROTECT(Rf_allocVector(VECSXP, 4)), an = PROTECT(Rf_allocVector(STRSXP, 4));
    SEXP a_lik = PROTECT(Rf_allocVector(REALSXP, n)), a_C = PROTECT(Rf_allocVector(REALSXP, n));
    SEXP a_it = PROTECT(Rf_allocVector(INTSXP, n)), a_st = PROTECT(Rf_allocVector(INTSXP, n));
    for (int c = 0; c < n; c++) {
        REAL(a_lik)[c] = lik[c];
        REAL(a_C)[c] = theta[(size_t)c * P + 1 + p];
        INTEGER(a_it)[c] = n_iter[c];
        INTEGER(a_st)[c] = status[c];
    }
    const char *anm[] = {"lik", "C", "n_iter", "status"};
    SEXP av[] = {a_lik, a_C, a_it, a_st};
    for (int i = 0; i < 4; i++) { SET_VECTOR_ELT(all, i, av[i]); SET_STRING_ELT(an, i, Rf_mkChar(anm[i])); }
    Rf_setAttrib(all, R_NamesSymbol, an);

    SEXP out = PROTECT(Rf_allocVector(VECSXP, 6)), on = PROTECT(Rf_allocVector(STRSXP, 6));
    const char *onm[] = {"theta", "fit", "liks", "lik", "index", "all"};
    SET_VECTOR_ELT(out, 0, theta_to_list(theta + (size_t)k * P, p, q));
    SET_VECTOR_ELT(out, 1, fit);
    SET_VECTOR_ELT(out, 2, mat(n_iter[k], 1, liks + (size_t)k * niter)); /* arma::vec -> n x 1 */
    SET_VECTOR_ELT(out, 3, Rf_ScalarReal(lik[k]));
    SET_VECTOR_ELT(out, 4, Rf_ScalarInteger(k + 1));
    SET_VECTOR_ELT(out, 5, all);
    for (int i = 0; i < 6; i++) SET_STRING_ELT(on, i, Rf_mkChar(onm[i]));
    Rf_setAttrib(out, R_NamesSymbol, on);
    UNPROTECT(10);
    return out;
}

static const R_CallMethodDef CallEntries[] = {
    {"ldsrhip_LDS_EM_batch", (DL_FUNC)&ldsrhip_LDS_EM_batch, 6},
    {NULL, NULL, 0}};

void R_init_ldsrhip(DllInfo *dll) {
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_ldsrhip(DllInfo *dll) {
    (void)dll;
    ldsr_shutdown();
}
