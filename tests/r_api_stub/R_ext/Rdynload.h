/* see ../Rinternals.h: syntax-check stub only */
#ifndef R_STUB_RDYNLOAD_H
#define R_STUB_RDYNLOAD_H
typedef struct _DllInfo DllInfo;
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
int R_useDynamicSymbols(DllInfo *, int);
#define FALSE 0
#endif
