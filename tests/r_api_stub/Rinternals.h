/* tests/r_api_stub/Rinternals.h -- see R.h in this directory: declarations only, for a syntax check of rshim/bwgr_shim.c */
#ifndef BWGR_R_API_STUB_RINTERNALS_H
#define BWGR_R_API_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_DimSymbol, R_NamesSymbol;
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
int TYPEOF(SEXP);
double *REAL(SEXP);
int *INTEGER(SEXP);
R_xlen_t XLENGTH(SEXP);
int Rf_length(SEXP);
SEXP Rf_allocVector(unsigned int type, R_xlen_t n);
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol);
SEXP Rf_coerceVector(SEXP, unsigned int type);
SEXP Rf_ScalarReal(double);
SEXP Rf_mkChar(const char *);
double Rf_asReal(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
Rboolean Rf_isNull(SEXP);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP);
void R_ClearExternalPtr(SEXP);
typedef void (*R_CFinalizer_t)(SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean onexit);
#endif
