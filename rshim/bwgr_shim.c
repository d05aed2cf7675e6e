/*
 * rshim/bwgr_shim.c -- the reference-side binding a bWGR maintainer would add: a thin .Call shim over the C ABI of
 * libbwgr_hip.so (include/bwgr.h).  SOURCE ONLY: R, Rinternals.h and libR do not exist in the build image, so this file
 * is not compiled or tested there.  Build where R exists with
 *     R CMD SHLIB bwgr_shim.c -I../include -L../bwgr_amd -lbwgr_hip -o bwgrhip.so
 *
 * It replaces the Rcpp-generated glue for the hot path:
 *     _bWGR_KMUP      src/RcppExports.cpp:16-31      -> bwgrhip_KMUP     (8 args)
 *     _bWGR_KMUP2     src/RcppExports.cpp:34-50      -> bwgrhip_KMUP2    (9 args)
 *     _bWGR_BayesA..  src/RcppExports.cpp:177-290    -> bwgrhip_Bayes    (model + 7 args)
 * and adds bwgrhip_wgr (R/wgr.R:2-169 as one device-resident call) plus panel handles so that X is staged in HBM
 * once instead of being converted SEXP -> Eigen::MatrixXf on every call (src/RcppExports.cpp:20).
 *
 * Conventions kept from the reference: inputs are never written (Rcpp passes by value, :20-27); outputs are fresh
 * vectors under PROTECT; errors become R conditions (BEGIN_RCPP/END_RCPP, :17,30) via Rf_error with
 * bwgr_last_error(); the seed is drawn from R's stream between GetRNGstate/PutRNGstate (Rcpp::RNGScope, :19) so
 * set.seed() governs repeatability.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Random.h>
#include <stdint.h>
#include <string.h>
#include "bwgr.h"

static void chk(int rc) { if (rc != BWGR_OK) Rf_error("bwgr: %s", bwgr_last_error()); }

static uint64_t seed_from_R(void) {
  GetRNGstate();
  uint64_t hi = (uint64_t)(unif_rand() * 4294967296.0), lo = (uint64_t)(unif_rand() * 4294967296.0);
  PutRNGstate();
  return (hi << 32) | lo;
}

static void panel_finalizer(SEXP ext) {
  bwgr_panel *P = (bwgr_panel *)R_ExternalPtrAddr(ext);
  if (P) { bwgr_panel_destroy(P); R_ClearExternalPtr(ext); }
}

/* X: numeric (double) or integer matrix, column-major as R stores it.  Integer matrices within int8 range are
 * staged as int8 genotypes, everything else as float (the narrowing Rcpp performs on every call). */
SEXP bwgrhip_panel(SEXP X, SEXP device) {
  SEXP dim = Rf_getAttrib(X, R_DimSymbol);
  if (Rf_length(dim) != 2) Rf_error("X must be a matrix");
  const int64_t n = INTEGER(dim)[0], p = INTEGER(dim)[1];
  bwgr_panel *P = NULL;
  if (TYPEOF(X) == INTSXP) {
    const int *xi = INTEGER(X);
    int8_t *x8 = (int8_t *)R_alloc((size_t)n * p, 1);
    int ok = 1;
    for (int64_t k = 0; k < n * p; k++) { if (xi[k] < -128 || xi[k] > 127) { ok = 0; break; } x8[k] = (int8_t)xi[k]; }
    if (ok) chk(bwgr_panel_create(&P, x8, BWGR_X_I8, BWGR_HOST, n, p, n, Rf_asInteger(device), 0, 0));
    else X = Rf_coerceVector(X, REALSXP);
  }
  if (!P) chk(bwgr_panel_create(&P, REAL(X), BWGR_X_F64, BWGR_HOST, n, p, n, Rf_asInteger(device), 0, 0));
  SEXP ext = PROTECT(R_MakeExternalPtr(P, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(ext, panel_finalizer, TRUE);
  UNPROTECT(1);
  return ext;
}

static bwgr_panel *panel_of(SEXP ext) {
  bwgr_panel *P = (bwgr_panel *)R_ExternalPtrAddr(ext);
  if (!P) Rf_error("bwgr: panel was freed");
  return P;
}
static float *to_float(SEXP v, R_xlen_t n) {
  float *f = (float *)R_alloc((size_t)n, sizeof(float));
  const double *d = REAL(v);
  for (R_xlen_t k = 0; k < n; k++) f[k] = (float)d[k];
  return f;
}
static SEXP from_float(const float *f, R_xlen_t n) {
  SEXP v = PROTECT(Rf_allocVector(REALSXP, n));
  for (R_xlen_t k = 0; k < n; k++) REAL(v)[k] = (double)f[k];
  UNPROTECT(1);
  return v;
}
static SEXP named_list(int n, const char **names) {
  SEXP l = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
  for (int k = 0; k < n; k++) SET_STRING_ELT(nm, k, Rf_mkChar(names[k]));
  Rf_setAttrib(l, R_NamesSymbol, nm);
  UNPROTECT(2);
  return l;
}

/* KMUP(X,b,d,xx,e,L,Ve,pi) -> list(b=,d=,e=)            src/Rcpp20260726ai.cpp:12-38 */
SEXP bwgrhip_KMUP(SEXP panel, SEXP b, SEXP d, SEXP xx, SEXP e, SEXP L, SEXP Ve, SEXP pi, SEXP iter) {
  bwgr_panel *P = panel_of(panel);
  int64_t info[8]; chk(bwgr_panel_info(P, info));
  const R_xlen_t n = info[0], p = info[1];
  if (XLENGTH(b) != p || XLENGTH(d) != p || XLENGTH(xx) != p || XLENGTH(L) != p || XLENGTH(e) != n) Rf_error("KMUP: length mismatch");
  float *fb = to_float(b, p), *fd = to_float(d, p), *fxx = to_float(xx, p), *fe = to_float(e, n), *fL = to_float(L, p);
  chk(bwgr_kmup(P, fb, fd, fxx, fe, fL, (float)Rf_asReal(Ve), (float)Rf_asReal(pi), seed_from_R(), (uint32_t)Rf_asInteger(iter), BWGR_RNG_PHILOX));
  const char *nm[] = {"b", "d", "e"};
  SEXP out = PROTECT(named_list(3, nm));
  SET_VECTOR_ELT(out, 0, from_float(fb, p)); SET_VECTOR_ELT(out, 1, from_float(fd, p)); SET_VECTOR_ELT(out, 2, from_float(fe, n));
  UNPROTECT(1);
  return out;
}

/* BayesA/B/C/L/RR/Cpi/Dpi(y,X,it,bi,[pi,]df,R2)          src/Rcpp20260726ai.cpp:589-987; return lists :631-634,
 * :694-698, :916-920 (names and order kept) */
SEXP bwgrhip_Bayes(SEXP model, SEXP y, SEXP panel, SEXP it, SEXP bi, SEXP pi, SEXP df, SEXP R2) {
  bwgr_panel *P = panel_of(panel);
  int64_t info[8]; chk(bwgr_panel_info(P, info));
  const R_xlen_t n = info[0], p = info[1];
  const int m = Rf_asInteger(model);
  if (XLENGTH(y) != n) Rf_error("length(y) must equal nrow(X)");
  const int per = (m == BWGR_BAYESA || m == BWGR_BAYESB || m == BWGR_BAYESL || m == BWGR_BAYESDPI);
  float *fy = to_float(y, n);
  float *B = (float *)R_alloc(p, 4), *D = (float *)R_alloc(p, 4), *hat = (float *)R_alloc(n, 4), *VB = (float *)R_alloc(per ? p : 1, 4), *PV = (float *)R_alloc(p, 4);
  float mu, ve, h2, MSx, Pi;
  chk(bwgr_bayes(P, m, fy, (float)Rf_asReal(it), (float)Rf_asReal(bi), (float)Rf_asReal(pi), (float)Rf_asReal(df), (float)Rf_asReal(R2),
                 seed_from_R(), BWGR_RNG_PHILOX, &mu, B, D, hat, VB, &ve, &h2, &MSx, &Pi, PV));
  SEXP out;
  if (m == BWGR_BAYESA || m == BWGR_BAYESL || m == BWGR_BAYESRR) {
    const char *nm[] = {"mu", "b", "hat", "vb", "ve", "h2", "MSx"};
    out = PROTECT(named_list(7, nm));
    SET_VECTOR_ELT(out, 0, Rf_ScalarReal(mu)); SET_VECTOR_ELT(out, 1, from_float(B, p)); SET_VECTOR_ELT(out, 2, from_float(hat, n));
    SET_VECTOR_ELT(out, 3, from_float(VB, per ? p : 1)); SET_VECTOR_ELT(out, 4, Rf_ScalarReal(ve)); SET_VECTOR_ELT(out, 5, Rf_ScalarReal(h2));
    SET_VECTOR_ELT(out, 6, Rf_ScalarReal(MSx));
  } else if (m == BWGR_BAYESB || m == BWGR_BAYESC) {
    const char *nm[] = {"mu", "b", "d", "hat", "vb", "ve", "h2", "MSx"};
    out = PROTECT(named_list(8, nm));
    SET_VECTOR_ELT(out, 0, Rf_ScalarReal(mu)); SET_VECTOR_ELT(out, 1, from_float(B, p)); SET_VECTOR_ELT(out, 2, from_float(D, p));
    SET_VECTOR_ELT(out, 3, from_float(hat, n)); SET_VECTOR_ELT(out, 4, from_float(VB, per ? p : 1)); SET_VECTOR_ELT(out, 5, Rf_ScalarReal(ve));
    SET_VECTOR_ELT(out, 6, Rf_ScalarReal(h2)); SET_VECTOR_ELT(out, 7, Rf_ScalarReal(MSx));
  } else {
    const char *nm[] = {"mu", "b", "d", "pi", "hat", "h2", "vb", "ve", "PVAL"};
    out = PROTECT(named_list(9, nm));
    SET_VECTOR_ELT(out, 0, Rf_ScalarReal(mu)); SET_VECTOR_ELT(out, 1, from_float(B, p)); SET_VECTOR_ELT(out, 2, from_float(D, p));
    SET_VECTOR_ELT(out, 3, Rf_ScalarReal(Pi)); SET_VECTOR_ELT(out, 4, from_float(hat, n)); SET_VECTOR_ELT(out, 5, Rf_ScalarReal(h2));
    SET_VECTOR_ELT(out, 6, from_float(VB, per ? p : 1)); SET_VECTOR_ELT(out, 7, Rf_ScalarReal(ve)); SET_VECTOR_ELT(out, 8, from_float(PV, p));
  }
  UNPROTECT(1);
  return out;
}

/* BayesA2 / BayesB2 / BayesRR2(y,X1,X2,it,bi,[pi,]df,R2)  src/Rcpp20260726ai.cpp:990-1218; return lists :1054-1057,
 * :1146-1149, :1216-1219 (names and order kept) */
SEXP bwgrhip_Bayes2(SEXP model, SEXP y, SEXP panel1, SEXP panel2, SEXP it, SEXP bi, SEXP pi, SEXP df, SEXP R2) {
  bwgr_panel *P1 = panel_of(panel1), *P2 = panel_of(panel2);
  int64_t i1[8], i2[8]; chk(bwgr_panel_info(P1, i1)); chk(bwgr_panel_info(P2, i2));
  const R_xlen_t n = i1[0], p1 = i1[1], p2 = i2[1];
  const int m = Rf_asInteger(model);
  if (XLENGTH(y) != n || i2[0] != i1[0]) Rf_error("length(y), nrow(X1) and nrow(X2) must agree");
  const int per = (m != BWGR_BAYESRR);
  float *fy = to_float(y, n);
  float *B1 = (float *)R_alloc(p1, 4), *D1 = (float *)R_alloc(p1, 4), *V1 = (float *)R_alloc(per ? p1 : 1, 4);
  float *B2 = (float *)R_alloc(p2, 4), *D2 = (float *)R_alloc(p2, 4), *V2 = (float *)R_alloc(per ? p2 : 1, 4), *hat = (float *)R_alloc(n, 4);
  float mu, ve, h2;
  chk(bwgr_bayes2(P1, P2, m, fy, (float)Rf_asReal(it), (float)Rf_asReal(bi), (float)Rf_asReal(pi), (float)Rf_asReal(df), (float)Rf_asReal(R2),
                  seed_from_R(), BWGR_RNG_PHILOX, &mu, B1, D1, V1, B2, D2, V2, &ve, hat, &h2));
  SEXP out;
  if (m == BWGR_BAYESB) {
    const char *nm[] = {"mu", "b1", "d1", "vb1", "b2", "d2", "vb2", "ve", "hat", "h2"};
    out = PROTECT(named_list(10, nm));
    SET_VECTOR_ELT(out, 0, Rf_ScalarReal(mu)); SET_VECTOR_ELT(out, 1, from_float(B1, p1)); SET_VECTOR_ELT(out, 2, from_float(D1, p1));
    SET_VECTOR_ELT(out, 3, from_float(V1, p1)); SET_VECTOR_ELT(out, 4, from_float(B2, p2)); SET_VECTOR_ELT(out, 5, from_float(D2, p2));
    SET_VECTOR_ELT(out, 6, from_float(V2, p2)); SET_VECTOR_ELT(out, 7, Rf_ScalarReal(ve)); SET_VECTOR_ELT(out, 8, from_float(hat, n));
    SET_VECTOR_ELT(out, 9, Rf_ScalarReal(h2));
  } else {
    const char *nm[] = {"hat", "mu", "b1", "b2", "vb1", "vb2", "ve", "h2"};
    out = PROTECT(named_list(8, nm));
    SET_VECTOR_ELT(out, 0, from_float(hat, n)); SET_VECTOR_ELT(out, 1, Rf_ScalarReal(mu)); SET_VECTOR_ELT(out, 2, from_float(B1, p1));
    SET_VECTOR_ELT(out, 3, from_float(B2, p2)); SET_VECTOR_ELT(out, 4, from_float(V1, per ? p1 : 1)); SET_VECTOR_ELT(out, 5, from_float(V2, per ? p2 : 1));
    SET_VECTOR_ELT(out, 6, Rf_ScalarReal(ve)); SET_VECTOR_ELT(out, 7, Rf_ScalarReal(h2));
  }
  UNPROTECT(1);
  return out;
}

/* EM / Gauss-Seidel family: emRR, emBA, emBB, emBC, emBCpi, emDE, emBL, emEN, emML  src/Rcpp20260726ai.cpp:80-521, :1502-1550;
 * return lists :122-127, :181-187, :240-247, :298-304, :347-353, :394, :453-459, :514-520, :1545-1549 (names and order kept).
 * model = BWGR_EM_*; par = Pi or alpha; D = emML's weights or NULL */
SEXP bwgrhip_em(SEXP model, SEXP y, SEXP panel, SEXP df, SEXP R2, SEXP par, SEXP D) {
  bwgr_panel *P = panel_of(panel);
  int64_t info[8]; chk(bwgr_panel_info(P, info));
  const R_xlen_t n = info[0], p = info[1];
  const int m = Rf_asInteger(model);
  if (XLENGTH(y) != n) Rf_error("length(y) must equal nrow(gen)");
  float *fy = to_float(y, n), *fD = Rf_isNull(D) ? NULL : to_float(D, p);
  float *B = (float *)R_alloc(p, 4), *Dv = (float *)R_alloc(p, 4), *V = (float *)R_alloc(p, 4), *hat = (float *)R_alloc(n, 4);
  float mu, s[6]; int iters;
  chk(bwgr_em(P, m, fy, (float)Rf_asReal(df), (float)Rf_asReal(R2), (float)Rf_asReal(par), fD, 0, &mu, B, Dv, hat, V, s, &iters));
  SEXP out;
#define EM_SET(k_, v_) SET_VECTOR_ELT(out, k_, v_)
  if (m == BWGR_EM_RR) {
    const char *nm[] = {"mu", "b", "hat", "Va", "Ve", "h2"}; out = PROTECT(named_list(6, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(hat, n)); EM_SET(3, Rf_ScalarReal(s[0])); EM_SET(4, Rf_ScalarReal(s[1])); EM_SET(5, Rf_ScalarReal(s[2]));
  } else if (m == BWGR_EM_BA || m == BWGR_EM_DE) {
    const char *nm[] = {"mu", "b", "hat", "Vb", "Ve", "h2"}; out = PROTECT(named_list(6, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(hat, n)); EM_SET(3, from_float(V, p)); EM_SET(4, Rf_ScalarReal(s[1])); EM_SET(5, Rf_ScalarReal(s[2]));
  } else if (m == BWGR_EM_BB) {
    const char *nm[] = {"mu", "b", "d", "hat", "Vb", "Ve", "h2"}; out = PROTECT(named_list(7, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(Dv, p)); EM_SET(3, from_float(hat, n)); EM_SET(4, from_float(V, p)); EM_SET(5, Rf_ScalarReal(s[1])); EM_SET(6, Rf_ScalarReal(s[2]));
  } else if (m == BWGR_EM_BC) {
    const char *nm[] = {"mu", "b", "d", "hat", "Vg", "Va", "Ve", "h2"}; out = PROTECT(named_list(8, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(Dv, p)); EM_SET(3, from_float(hat, n)); EM_SET(4, Rf_ScalarReal(s[3])); EM_SET(5, Rf_ScalarReal(s[0])); EM_SET(6, Rf_ScalarReal(s[1])); EM_SET(7, Rf_ScalarReal(s[2]));
  } else if (m == BWGR_EM_BCPI) {
    const char *nm[] = {"mu", "b", "d", "pi", "hat", "Vg", "Va", "Ve", "h2"}; out = PROTECT(named_list(9, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(Dv, p)); EM_SET(3, Rf_ScalarReal(s[4])); EM_SET(4, from_float(hat, n)); EM_SET(5, Rf_ScalarReal(s[3])); EM_SET(6, Rf_ScalarReal(s[0])); EM_SET(7, Rf_ScalarReal(s[1])); EM_SET(8, Rf_ScalarReal(s[2]));
  } else if (m == BWGR_EM_BL) {
    const char *nm[] = {"mu", "b", "hat", "h2"}; out = PROTECT(named_list(4, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(hat, n)); EM_SET(3, Rf_ScalarReal(s[2]));
  } else if (m == BWGR_EM_LASSO) {
    const char *nm[] = {"mu", "b", "h2", "hat", "Lmb"}; out = PROTECT(named_list(5, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, Rf_ScalarReal(s[2])); EM_SET(3, from_float(hat, n)); EM_SET(4, Rf_ScalarReal(s[0]));
  } else if (m == BWGR_EM_EN) {
    const char *nm[] = {"mu", "b", "hat", "Va", "Ve", "h2"}; out = PROTECT(named_list(6, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(hat, n)); EM_SET(3, Rf_ScalarReal(s[0])); EM_SET(4, Rf_ScalarReal(s[1])); EM_SET(5, Rf_ScalarReal(s[2]));
  } else {
    const char *nm[] = {"mu", "b", "hat", "h2", "Vb", "Va", "Ve"}; out = PROTECT(named_list(7, nm));
    EM_SET(0, Rf_ScalarReal(mu)); EM_SET(1, from_float(B, p)); EM_SET(2, from_float(hat, n)); EM_SET(3, Rf_ScalarReal(s[2])); EM_SET(4, Rf_ScalarReal(s[0])); EM_SET(5, Rf_ScalarReal(s[3])); EM_SET(6, Rf_ScalarReal(s[1]));
  }
#undef EM_SET
  UNPROTECT(1);
  return out;
}

/* wgr(y,X,it,bi,th,bag=1,rp=FALSE,iv,de,pi,df,R2,eigK=NULL)    R/wgr.R:2-169 -> list(mu,b,Vb,d,Ve,hat,cxx), :155-168 */
SEXP bwgrhip_wgr(SEXP y, SEXP panel, SEXP it, SEXP bi, SEXP th, SEXP iv, SEXP de, SEXP pi, SEXP df, SEXP R2, SEXP U, SEXP V, SEXP bag, SEXP rp) {
  bwgr_panel *P = panel_of(panel);
  int64_t info[8]; chk(bwgr_panel_info(P, info));
  const R_xlen_t n = info[0], p = info[1];
  if (XLENGTH(y) != n) Rf_error("length(y) must equal nrow(X)");
  const int per = Rf_asLogical(iv) || Rf_asLogical(de);
  SEXP b = PROTECT(Rf_allocVector(REALSXP, p)), d = PROTECT(Rf_allocVector(REALSXP, p)), Vb = PROTECT(Rf_allocVector(REALSXP, per ? p : 1));
  SEXP hat = PROTECT(Rf_allocMatrix(REALSXP, (int)n, 1));
  double mu, Ve, cxx, Vk = 0;
  const int has_k = !Rf_isNull(U);
  const int64_t pk = has_k ? INTEGER(Rf_getAttrib(U, R_DimSymbol))[1] : 0;
  SEXP u = PROTECT(Rf_allocMatrix(REALSXP, (int)n, 1));
  chk(bwgr_wgr_ex(P, REAL(y), Rf_asInteger(it), Rf_asInteger(bi), Rf_asInteger(th), Rf_asLogical(iv), Rf_asLogical(de), Rf_asReal(pi),
                  Rf_asReal(df), Rf_asReal(R2), seed_from_R(), BWGR_RNG_PHILOX, has_k ? REAL(U) : NULL, has_k ? REAL(V) : NULL, pk,
                  Rf_asReal(bag), Rf_asLogical(rp), &mu, REAL(b), REAL(Vb), REAL(d), &Ve, REAL(hat), &cxx, REAL(u), &Vk));
  SEXP out;
  if (has_k) {   /* list(mu,b,Vb,d,Ve,hat,u,Vk,cxx), R/wgr.R:157-160 */
    const char *nm[] = {"mu", "b", "Vb", "d", "Ve", "hat", "u", "Vk", "cxx"};
    out = PROTECT(named_list(9, nm));
    SET_VECTOR_ELT(out, 6, u); SET_VECTOR_ELT(out, 7, Rf_ScalarReal(Vk)); SET_VECTOR_ELT(out, 8, Rf_ScalarReal(cxx));
  } else {
    const char *nm[] = {"mu", "b", "Vb", "d", "Ve", "hat", "cxx"};
    out = PROTECT(named_list(7, nm));
    SET_VECTOR_ELT(out, 6, Rf_ScalarReal(cxx));
  }
  SET_VECTOR_ELT(out, 0, Rf_ScalarReal(mu)); SET_VECTOR_ELT(out, 1, b); SET_VECTOR_ELT(out, 2, Vb); SET_VECTOR_ELT(out, 3, d);
  SET_VECTOR_ELT(out, 4, Rf_ScalarReal(Ve)); SET_VECTOR_ELT(out, 5, hat);
  UNPROTECT(6);
  return out;
}

/* KMUP2(X,Use,b,d,xx,E,L,Ve,pi) -> list(b=,d=,e=)     src/Rcpp20260726ai.cpp:41-77; Use holds 0-based row ids (R/wgr.R:68) */
SEXP bwgrhip_KMUP2(SEXP panel, SEXP Use, SEXP b, SEXP d, SEXP xx, SEXP E, SEXP L, SEXP Ve, SEXP pi, SEXP iter) {
  bwgr_panel *P = panel_of(panel);
  int64_t info[8]; chk(bwgr_panel_info(P, info));
  const R_xlen_t n0 = info[0], p = info[1], n = XLENGTH(Use);
  if (XLENGTH(b) != p || XLENGTH(d) != p || XLENGTH(xx) != p || XLENGTH(L) != p || XLENGTH(E) != n0) Rf_error("KMUP2: length mismatch");
  int *use = (int *)R_alloc((size_t)n, sizeof(int));
  const double *ud = REAL(Use);                            /* the reference takes Use as a float vector and truncates, :50 */
  for (R_xlen_t k = 0; k < n; k++) use[k] = (int)ud[k];
  float *fb = to_float(b, p), *fd = to_float(d, p), *fxx = to_float(xx, p), *fE = to_float(E, n0), *fL = to_float(L, p);
  float *fe = (float *)R_alloc((size_t)n, sizeof(float));
  chk(bwgr_kmup2(P, use, (int64_t)n, fb, fd, fxx, fE, fe, fL, (float)Rf_asReal(Ve), (float)Rf_asReal(pi), seed_from_R(), (uint32_t)Rf_asInteger(iter), BWGR_RNG_PHILOX));
  const char *nm[] = {"b", "d", "e"};
  SEXP out = PROTECT(named_list(3, nm));
  SET_VECTOR_ELT(out, 0, from_float(fb, p)); SET_VECTOR_ELT(out, 1, from_float(fd, p)); SET_VECTOR_ELT(out, 2, from_float(fe, n));
  UNPROTECT(1);
  return out;
}

static const R_CallMethodDef CallEntries[] = {   /* as src/RcppExports.cpp:1152-1228 registers _bWGR_* */
  {"bwgrhip_panel", (DL_FUNC)&bwgrhip_panel, 2}, {"bwgrhip_KMUP", (DL_FUNC)&bwgrhip_KMUP, 9}, {"bwgrhip_KMUP2", (DL_FUNC)&bwgrhip_KMUP2, 10},
  {"bwgrhip_Bayes", (DL_FUNC)&bwgrhip_Bayes, 8}, {"bwgrhip_Bayes2", (DL_FUNC)&bwgrhip_Bayes2, 9},
  {"bwgrhip_wgr", (DL_FUNC)&bwgrhip_wgr, 14}, {"bwgrhip_em", (DL_FUNC)&bwgrhip_em, 7}, {NULL, NULL, 0}};

void R_init_bwgrhip(DllInfo *dll) {              /* as R_init_bWGR, src/RcppExports.cpp:1230-1233 */
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
