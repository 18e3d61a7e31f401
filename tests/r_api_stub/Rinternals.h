/* Minimal DECLARATIONS of the R C API used by ldsr_amd/r_shim/ldsrhip_call.c.  R is not
 * installed in this image, so the shim is checked two ways: a syntax-only compile against these
 * declarations (tests/test_r_shim_syntax.py) and an EXECUTED run against tests/r_mock/rmock.c, a
 * miniature implementation of exactly these entry points (tests/test_r_shim_mock.py).  Neither
 * is a substitute for testing under R. */
#ifndef R_STUB_RINTERNALS_H
#define R_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef void *(*DL_FUNC)(void);
#define LGLSXP 10
#define REALSXP 14
#define INTSXP 13
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol;
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
R_xlen_t Rf_xlength(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
const char *CHAR(SEXP);
SEXP Rf_mkChar(const char *);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
double *REAL(SEXP);
int *INTEGER(SEXP);
int Rf_isReal(SEXP);
int TYPEOF(SEXP);
SEXP Rf_coerceVector(SEXP, unsigned int);
int Rf_ncols(SEXP);
int Rf_nrows(SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_asLogical(SEXP);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarInteger(int);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char *, ...);
char *R_alloc(size_t, int);
void R_CheckUserInterrupt(void);
int R_ToplevelExec(void (*fun)(void *), void *data);   /* Rboolean in R: FALSE if fun was interrupted */
#endif
