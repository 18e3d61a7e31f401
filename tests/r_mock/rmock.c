/*
 * rmock.c -- a miniature of R's C API: exactly the entry points declared in
 * tests/r_api_stub/{Rinternals.h,R_ext/Rdynload.h}, implemented well enough to EXECUTE the
 * .Call shim ldsr_amd/r_shim/ldsrhip_call.c in tests (R itself is not installed in this image).
 * TEST INFRASTRUCTURE ONLY.  Semantics kept from R: objects are typed vectors with optional
 * names / dim, Rf_error does not return (longjmp to the .Call boundary), R_alloc memory lives
 * until the call ends, the PROTECT stack must balance by the time a call returns normally.
 * Not kept: garbage collection (objects live until rmock_reset()), so a missing PROTECT cannot
 * be detected here -- only an unbalanced count can.
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#define CHARSXP 9

struct SEXPREC {
    int type;
    R_xlen_t len;
    int nrow, ncol; /* 0, 0 = no dim attribute */
    void *data;
    SEXP names;
};

static struct SEXPREC nil_obj = {0, 0, 0, 0, NULL, NULL}, names_sym = {1, 0, 0, 0, NULL, NULL};
SEXP R_NilValue = &nil_obj, R_NamesSymbol = &names_sym;

static void **g_objs = NULL;
static size_t g_nobj = 0, g_cap = 0;
static int g_protect = 0, g_interrupt_polls = 0;
static jmp_buf g_jmp;
static int g_jmp_armed = 0;
void Rf_error(const char *fmt, ...);
static char g_errmsg[512];
static const R_CallMethodDef *g_table = NULL;

static void *track(void *p) {
    if (g_nobj == g_cap) {
        g_cap = g_cap ? 2 * g_cap : 1024;
        g_objs = (void **)realloc(g_objs, g_cap * sizeof(void *));
    }
    g_objs[g_nobj++] = p;
    return p;
}

static SEXP new_obj(int type, R_xlen_t n, size_t elt) {
    SEXP s = (SEXP)track(calloc(1, sizeof(struct SEXPREC)));
    s->type = type;
    s->len = n;
    s->data = track(calloc(n > 0 ? (size_t)n : 1, elt));
    s->names = R_NilValue;
    return s;
}

SEXP Rf_allocVector(unsigned int type, R_xlen_t n) {
    switch (type) {
        case REALSXP: return new_obj(REALSXP, n, sizeof(double));
        case INTSXP: case LGLSXP: return new_obj((int)type, n, sizeof(int));
        case STRSXP: case VECSXP: {
            SEXP s = new_obj((int)type, n, sizeof(SEXP));
            for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue;
            return s;
        }
        default: Rf_error("rmock: allocVector of unsupported type %u", type);
    }
    return R_NilValue;
}

SEXP Rf_allocMatrix(unsigned int type, int nr, int nc) {
    SEXP s = Rf_allocVector(type, (R_xlen_t)nr * nc);
    s->nrow = nr;
    s->ncol = nc;
    return s;
}

SEXP Rf_mkChar(const char *c) {
    SEXP s = new_obj(CHARSXP, (R_xlen_t)strlen(c), 1);
    free(s->data);
    g_objs[g_nobj - 1] = s->data = strdup(c);
    return s;
}
const char *CHAR(SEXP s) { return (const char *)s->data; }
double *REAL(SEXP s) { if (s->type != REALSXP) Rf_error("rmock: REAL() on a non-double"); return (double *)s->data; }
int *INTEGER(SEXP s) { if (s->type != INTSXP && s->type != LGLSXP) Rf_error("rmock: INTEGER() on a non-integer"); return (int *)s->data; }
R_xlen_t Rf_xlength(SEXP s) { return s->len; }
int Rf_isReal(SEXP s) { return s->type == REALSXP; }
int TYPEOF(SEXP s) { return s->type; }
/* integer / logical -> double, dim kept; NA_INTEGER (INT_MIN) -> NA_real_ (a NaN) */
SEXP Rf_coerceVector(SEXP s, unsigned int type) {
    if (type != REALSXP || (s->type != INTSXP && s->type != LGLSXP)) Rf_error("rmock: unsupported coerceVector");
    SEXP r = Rf_allocVector(REALSXP, s->len);
    r->nrow = s->nrow;
    r->ncol = s->ncol;
    for (R_xlen_t i = 0; i < s->len; i++) {
        const int v = ((int *)s->data)[i];
        ((double *)r->data)[i] = v == (-2147483647 - 1) ? (0.0 / 0.0) : (double)v;
    }
    return r;
}
int Rf_ncols(SEXP s) { return s->nrow || s->ncol ? s->ncol : 1; }
int Rf_nrows(SEXP s) { return s->nrow || s->ncol ? s->nrow : (int)s->len; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { if (s->type != STRSXP || i >= s->len) Rf_error("rmock: bad STRING_ELT"); return ((SEXP *)s->data)[i]; }
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) { if (s->type != VECSXP || i >= s->len) Rf_error("rmock: bad VECTOR_ELT"); return ((SEXP *)s->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v) { if (s->type != VECSXP || i >= s->len) Rf_error("rmock: bad SET_VECTOR_ELT"); ((SEXP *)s->data)[i] = v; return v; }
void SET_STRING_ELT(SEXP s, R_xlen_t i, SEXP v) { if (s->type != STRSXP || i >= s->len || v->type != CHARSXP) Rf_error("rmock: bad SET_STRING_ELT"); ((SEXP *)s->data)[i] = v; }
SEXP Rf_getAttrib(SEXP s, SEXP what) { return what == R_NamesSymbol ? s->names : R_NilValue; }
SEXP Rf_setAttrib(SEXP s, SEXP what, SEXP v) {
    if (what == R_NamesSymbol) {
        if (v->type != STRSXP || v->len != s->len) Rf_error("rmock: names of the wrong length");
        s->names = v;
    }
    return v;
}
static double scalar_of(SEXP s) {
    if (s->len < 1) Rf_error("rmock: scalar expected");
    return s->type == REALSXP ? ((double *)s->data)[0] : (double)((int *)s->data)[0];
}
int Rf_asInteger(SEXP s) { return (int)scalar_of(s); }
double Rf_asReal(SEXP s) { return scalar_of(s); }
int Rf_asLogical(SEXP s) { return scalar_of(s) != 0.0; }
SEXP Rf_ScalarReal(double x) { SEXP s = Rf_allocVector(REALSXP, 1); REAL(s)[0] = x; return s; }
SEXP Rf_ScalarInteger(int x) { SEXP s = Rf_allocVector(INTSXP, 1); INTEGER(s)[0] = x; return s; }
SEXP Rf_protect(SEXP s) { g_protect++; return s; }
void Rf_unprotect(int n) { g_protect -= n; }
char *R_alloc(size_t n, int size) { return (char *)track(calloc(n ? n : 1, (size_t)size)); }
/* Scripted user interrupt: the (n+1)-th poll from now behaves like a pending Ctrl-C -- inside
 * R_ToplevelExec it unwinds to it (which then returns FALSE), elsewhere it is an R error. */
static jmp_buf g_tl_jmp;
static int g_tl_armed = 0, g_intr_after = -1;
void rmock_interrupt_after(int n_polls) { g_intr_after = n_polls < 0 ? -1 : g_interrupt_polls + n_polls; }
void R_CheckUserInterrupt(void) {
    g_interrupt_polls++;
    if (g_intr_after >= 0 && g_interrupt_polls > g_intr_after) {
        g_intr_after = -1;
        if (g_tl_armed) longjmp(g_tl_jmp, 1);
        Rf_error("interrupt");
    }
}
int R_ToplevelExec(void (*fun)(void *), void *data) {
    g_tl_armed = 1;
    if (setjmp(g_tl_jmp)) {
        g_tl_armed = 0;
        return 0;
    }
    fun(data);
    g_tl_armed = 0;
    return 1;
}

void Rf_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_errmsg, sizeof(g_errmsg), fmt, ap);
    va_end(ap);
    if (!g_jmp_armed) {
        fprintf(stderr, "rmock: Rf_error outside a call: %s\n", g_errmsg);
        abort();
    }
    longjmp(g_jmp, 1);
}

int R_registerRoutines(DllInfo *dll, const void *c, const R_CallMethodDef *call, const void *f, const void *e) {
    (void)dll; (void)c; (void)f; (void)e;
    g_table = call;
    return 1;
}
int R_useDynamicSymbols(DllInfo *dll, int v) { (void)dll; return v; }

/* ---- test harness (called from Python through ctypes) ------------------------------------ */
void R_init_ldsrhip(DllInfo *dll);
void R_unload_ldsrhip(DllInfo *dll);

void R_init_ldsr(DllInfo *dll);
void R_unload_ldsr(DllInfo *dll);
int rmock_init(void) { R_init_ldsrhip(NULL); return g_table != NULL; }
void rmock_unload(void) { R_unload_ldsrhip(NULL); }
/* the same object loaded under the name ldsr.so: R would call R_init_ldsr */
int rmock_init_as_ldsr(void) { R_init_ldsr(NULL); return g_table != NULL; }
int rmock_n_routines(void) { int n = 0; while (g_table && g_table[n].name) n++; return n; }
const char *rmock_routine_name(int i) { return g_table[i].name; }
int rmock_routine_nargs(int i) { return g_table[i].numArgs; }
const char *rmock_last_error(void) { return g_errmsg; }
int rmock_protect_depth(void) { return g_protect; }
int rmock_interrupt_polls(void) { return g_interrupt_polls; }
void rmock_reset(void) { /* frees every object: call between tests, never while results are in use */
    for (size_t i = 0; i < g_nobj; i++) free(g_objs[i]);
    g_nobj = 0;
    g_protect = 0;
}

typedef SEXP (*fn2)(SEXP, SEXP);
typedef SEXP (*fn3)(SEXP, SEXP, SEXP);
typedef SEXP (*fn4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);

/* .Call(name, args...): 0 = returned normally (*res set), 1 = Rf_error (message in
 * rmock_last_error()), 2 = no such routine / wrong arity.  Like R, an error unwinds the PROTECT
 * stack to its depth at the call. */
int rmock_call(const char *name, int nargs, SEXP *a, SEXP *res) {
    const R_CallMethodDef *d = g_table;
    while (d && d->name && strcmp(d->name, name) != 0) d++;
    if (!d || !d->name || d->numArgs != nargs) return 2;
    const int depth = g_protect;
    g_errmsg[0] = 0;
    g_jmp_armed = 1;
    if (setjmp(g_jmp)) {
        g_jmp_armed = 0;
        g_protect = depth;
        return 1;
    }
    if (nargs == 2) *res = ((fn2)d->fun)(a[0], a[1]);
    else if (nargs == 3) *res = ((fn3)d->fun)(a[0], a[1], a[2]);
    else if (nargs == 4) *res = ((fn4)d->fun)(a[0], a[1], a[2], a[3]);
    else if (nargs == 5) *res = ((fn5)d->fun)(a[0], a[1], a[2], a[3], a[4]);
    else if (nargs == 6) *res = ((fn6)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5]);
    else if (nargs == 7) *res = ((fn7)d->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
    else { g_jmp_armed = 0; return 2; }
    g_jmp_armed = 0;
    return 0;
}

/* constructors / accessors for the Python side */
SEXP rmock_real_matrix(int nr, int nc, const double *src) {
    SEXP m = Rf_allocMatrix(REALSXP, nr, nc);
    memcpy(m->data, src, sizeof(double) * (size_t)nr * nc);
    return m;
}
SEXP rmock_int_matrix(int nr, int nc, const int *src) {   /* an R integer matrix (1:10 and friends) */
    SEXP m = Rf_allocMatrix(INTSXP, nr, nc);
    memcpy(m->data, src, sizeof(int) * (size_t)nr * nc);
    return m;
}
SEXP rmock_scalar(int type, double v) {
    SEXP s = Rf_allocVector((unsigned)type, 1);
    if (type == REALSXP) ((double *)s->data)[0] = v; else ((int *)s->data)[0] = (int)v;
    return s;
}
SEXP rmock_list(int n, int named) {
    SEXP l = Rf_allocVector(VECSXP, n);
    if (named) {
        SEXP nm = Rf_allocVector(STRSXP, n);
        for (int i = 0; i < n; i++) ((SEXP *)nm->data)[i] = Rf_mkChar("");
        l->names = nm;
    }
    return l;
}
void rmock_list_set(SEXP l, int i, const char *name, SEXP v) {
    ((SEXP *)l->data)[i] = v;
    if (name && l->names != R_NilValue) ((SEXP *)l->names->data)[i] = Rf_mkChar(name);
}
int rmock_type(SEXP s) { return s->type; }
long rmock_len(SEXP s) { return (long)s->len; }
int rmock_nrow(SEXP s) { return s->nrow; }
int rmock_ncol(SEXP s) { return s->ncol; }
void *rmock_data(SEXP s) { return s->data; }
SEXP rmock_elt(SEXP l, int i) { return ((SEXP *)l->data)[i]; }
const char *rmock_name(SEXP l, int i) { return l->names == R_NilValue ? "" : (const char *)((SEXP *)l->names->data)[i]->data; }
