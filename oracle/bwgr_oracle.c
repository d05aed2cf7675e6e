/*
 * oracle/bwgr_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the bWGR Gibbs hot path, used as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the product
 * (bwgr_amd/, include/) may include, link or call this file.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or known answers for this
 * path and cannot be built or run here (needs R, Rcpp, RcppEigen and
 * R nmath, none present; /root/reference/src/Rcpp20260726ai.cpp:3).  This file is therefore
 * pinned only by (a) being a line-by-line restatement of the cited reference lines,
 * (b) the analytic invariants in tests/test_oracle_invariants.py, (c) Philox known-answer
 * vectors for the RNG.  Third-party arithmetic it stands in for: Eigen (via RcppEigen,
 * version unpinned in /root/reference/DESCRIPTION:13-15) reductions dot/squaredNorm/mean/
 * GEMV, and R nmath rnorm/rchisq/rbinom (replaced by the counter-based contract in
 * bwgr_rng.h: same distributions, different stream).
 *
 * The file is compiled twice (see Makefile):
 *   -DACC_T=double -DSUF=_w   "wide":     every Eigen reduction accumulates in double, the
 *                                          residual vector e (and the e1/e2 temporaries) is carried
 *                                          in double, the marker's dot product and conditional mean
 *                                          stay in double (only the drawn effect is rounded to float,
 *                                          as stored), and the Bernoulli log-odds uses the un-rounded
 *                                          norm difference.  This is the parity target for the GPU:
 *                                          the same algorithm with the float round-off of the
 *                                          n-vector arithmetic removed, so that it is independent
 *                                          of summation order and of how markers are blocked.
 *   -DACC_T=float  -DSUF=_f   "faithful": reductions accumulate in float (8 interleaved
 *                                          partial sums, as Eigen's packet reduction does) and
 *                                          the norms are rounded to float before subtraction,
 *                                          exactly as the reference's types dictate.  This is
 *                                          the CPU baseline that bench.py times.
 * All scalar arithmetic (effects, variances, lambda, probabilities) is float in both, as in the
 * reference's float locals; in the faithful flavour e is float too (Eigen::VectorXf).  The two
 * flavours are compared with each other in tests/test_oracle_invariants.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bwgr_rng.h"

#ifndef ACC_T
#define ACC_T double
#define SUF _w
#define ACC_WIDE 1
#endif
#ifdef ACC_WIDE
typedef double E_T;   /* residual vector element */
#else
typedef float E_T;
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUF)

enum { M_BAYESA = 0, M_BAYESB = 1, M_BAYESC = 2, M_BAYESL = 3, M_BAYESRR = 4, M_BAYESCPI = 5, M_BAYESDPI = 6 };

/* ---- Eigen-shaped reductions ------------------------------------------------------------ */
static inline ACC_T red8(const ACC_T s[8]) {
  return ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
}
/* X.col(j).dot(e) */
static float v_dot(const float *x, const E_T *e, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)x[i + l] * (ACC_T)e[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)x[i] * (ACC_T)e[i];
  return (float)red8(s);
}
/* the same dot kept in the accumulator type */
static ACC_T v_dot_acc(const float *x, const E_T *e, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)x[i + l] * (ACC_T)e[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)x[i] * (ACC_T)e[i];
  return red8(s);
}
/* b1 = rnorm((X_j.e + xx_j b0)/(xx_j + lambda_j), sd)  -- src/Rcpp20260726ai.cpp:20, :615, :670 ...
 * faithful: every intermediate is a float, as the reference's types dictate.
 * wide:     the dot product and the conditional mean stay in double (xx_j*b0, the denominator and sd are the
 *           reference's floats); only b1 itself is rounded to float, as the reference stores it. */
static inline float draw_b1_add(const float *xj, const E_T *e, int64_t n, float addend, float den, float sd, double z) {
#ifdef ACC_WIDE
  const double mean = (v_dot_acc(xj, e, n) + (double)addend) / (double)den;
  return (float)(mean + (double)sd * z);
#else
  const float mean = ((float)v_dot_acc(xj, e, n) + addend) / den;
  return (float)((double)mean + (double)sd * z);
#endif
}
static inline float draw_b1(const float *xj, const E_T *e, int64_t n, float xxj, float b0, float den, float sd, double z) {
  return draw_b1_add(xj, e, n, xxj * b0, den, sd, z);
}
/* v.squaredNorm() kept in the accumulator type (caller rounds) */
static ACC_T v_sqnorm_acc(const float *v, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)v[i + l] * (ACC_T)v[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)v[i] * (ACC_T)v[i];
  return red8(s);
}
static float v_sqnorm(const float *v, int64_t n) { return (float)v_sqnorm_acc(v, n); }
static ACC_T v_sum_acc(const float *v, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)v[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)v[i];
  return red8(s);
}
/* the same reductions over the residual vector (E_T) */
static ACC_T e_sqnorm_acc(const E_T *v, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)v[i + l] * (ACC_T)v[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)v[i] * (ACC_T)v[i];
  return red8(s);
}
static float e_sqnorm(const E_T *v, int64_t n) { return (float)e_sqnorm_acc(v, n); }
static float e_mean(const E_T *v, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)v[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)v[i];
  return (float)(red8(s) / (ACC_T)n);
}
/* v.mean(): Eigen = sum()/size() in the scalar type */
static float v_mean(const float *v, int64_t n) { return (float)(v_sum_acc(v, n) / (ACC_T)n); }

/* fvar, /root/reference/src/Rcpp20260726ai.cpp:7-9 */
static float v_fvar(const float *x, int64_t n) {
  const float m = v_mean(x, n);
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = 0; i < n; i++) {
    float dev = x[i] - m;
    float sq = dev * dev;
    s[i & 7] += (ACC_T)sq;
  }
  return (float)(red8(s) / (ACC_T)(float)(n - 1));
}
/* out = e - x*db   (Eigen: e - X.col(j)*(scalar), element-wise in float) */
static void v_axpy_to(E_T *out, const E_T *e, const float *x, float db, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    E_T t = (E_T)x[i] * (E_T)db;
    out[i] = e[i] - t;
  }
}
/* e -= x*db */
static void v_axpy(E_T *e, const float *x, float db, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    E_T t = (E_T)x[i] * (E_T)db;
    e[i] = e[i] - t;
  }
}
static inline float f_exp(float x) { return (float)exp((double)x); } /* std::exp(float) */

/* R::rnorm(mu, sigma) = mu + sigma*norm_rand() in double, narrowed to float on assignment */
static inline float draw_norm(float mu, float sd, double z) { return (float)((double)mu + (double)sd * z); }

/* ---- exported helpers --------------------------------------------------------------------- */
float FN(oracle_fvar)(const float *x, int64_t n) { return v_fvar(x, n); }
float FN(oracle_dot)(const float *x, const float *e, int64_t n) {
  E_T *t = (E_T *)malloc(sizeof(E_T) * n); for (int64_t i = 0; i < n; i++) t[i] = e[i];
  float r = v_dot(x, t, n); free(t); return r;
}

/* setup block shared by all seven samplers, /root/reference/src/Rcpp20260726ai.cpp:593-598
 * (identical at :644-649, 707-712, 767-772, 817-822, 863-868, 929-934) */
void FN(oracle_stats)(const float *X, int64_t n, int64_t p, int64_t ldx, float *xx, float *vx, float *MSx) {
  for (int64_t j = 0; j < p; j++) {
    xx[j] = v_sqnorm(X + j * ldx, n);
    vx[j] = v_fvar(X + j * ldx, n);
  }
  *MSx = (float)v_sum_acc(vx, p);
}

/* philox known-answer access for tests */
void FN(oracle_philox)(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { orng_philox4x32_10(ctr, key, out); }
/* variates: kind 0 normal, 1 uniform, 2 chisq(nu) */
double FN(oracle_variate)(uint64_t seed, int mode, int kind, double nu, uint32_t marker, uint32_t iter, uint32_t purpose, uint32_t k) {
  orng_t g = { seed, mode };
  if (kind == 0) return orng_normal(&g, marker, iter, purpose, k);
  if (kind == 1) return orng_uniform(&g, marker, iter, purpose, k);
  return orng_chisq(&g, nu, marker, iter, purpose);
}

/* ORNG_RSTREAM (bwgr_rstream.h, UNVERIFIED restatement of R's stream): set.seed(seed), and single draws for the self-tests --
 * kind 0 unif_rand, 1 norm_rand, 2 exp_rand, 3 rgamma(par, 1), 4 rchisq(par), 5 rbinom(1, par).  Each flavour of this file has its own stream. */
void FN(oracle_rstream_seed)(uint32_t seed) { ors_set_seed(seed); }
double FN(oracle_rstream_draw)(int kind, double par) {
  switch (kind) {
    case 0: return ors_unif_rand();
    case 1: return ors_norm_rand();
    case 2: return ors_exp_rand();
    case 3: return ors_rgamma(par, 1.0);
    case 4: return ors_rchisq(par);
    default: return (double)ors_rbinom1(par);
  }
}

/* ---- KMUP: one Gibbs sweep, /root/reference/src/Rcpp20260726ai.cpp:12-38 ----------------------
 * stable = 0: literal cj/(cj+dj) (underflows to 0/0 = NaN for 0.5*|e|^2/sqrt(Ve) >~ 103, then
 *             every marker takes the else branch, :25-31);
 * stable = 1: the algebraically identical 1/(1 + pi/(1-pi)*exp(C(|e2|^2-|e1|^2))), the form the
 *             reference itself uses in BayesB (:673-674).  The GPU implements stable = 1. */
int FN(oracle_kmup)(const float *X, int64_t n, int64_t p, int64_t ldx, float *b, float *d, const float *xx,
                    float *e_io, const float *L, float Ve, float pi, uint64_t seed, uint32_t iter, int rng_mode,
                    int stable, uint32_t marker0) {
  orng_t g = { seed, rng_mode };
  E_T *e = (E_T *)malloc(sizeof(E_T) * n), *e1 = (E_T *)malloc(sizeof(E_T) * n), *e2 = (E_T *)malloc(sizeof(E_T) * n);
  if (!e || !e1 || !e2) return 1;
  for (int64_t i = 0; i < n; i++) e[i] = e1[i] = e2[i] = (E_T)e_io[i];            /* :14-15 */
  float b0, b1, b2, cj, dj, pj;
  float C = -0.5f / sqrtf(Ve);                                                   /* :17 */
  for (int64_t j = 0; j < p; j++) {                                              /* :18 */
    const float *xj = X + j * ldx;
    uint32_t mk = marker0 + (uint32_t)j;   /* RNG counters carry global marker ids (0 offset = the reference's single panel) */
    b0 = b[j];                                                                   /* :19 */
    float den = xx[j] + L[j];
    b1 = draw_b1(xj, e, n, xx[j], b0, den, sqrtf(Ve / den), orng_normal(&g, mk, iter, ORNG_Z1, 0));                /* :20 */
    b2 = draw_norm(0.0f, sqrtf(Ve / den), orng_normal(&g, mk, iter, ORNG_Z2, 0)); /* :21 */
    v_axpy_to(e1, e, xj, b1 - b0, n);                                            /* :22 */
    if (pi > 0) {                                                                /* :23 */
      v_axpy_to(e2, e, xj, b2 - b0, n);                                          /* :24 */
      if (stable) {
#ifdef ACC_WIDE
        float diff = (float)(e_sqnorm_acc(e2, n) - e_sqnorm_acc(e1, n));
#else
        float diff = e_sqnorm(e2, n) - e_sqnorm(e1, n);
#endif
        float LR = (pi / (1.0f - pi)) * f_exp(C * diff);
        pj = 1.0f / (1.0f + LR);
      } else {
        cj = (1 - pi) * f_exp(C * e_sqnorm(e1, n));                              /* :25 */
        dj = (pi)*f_exp(C * e_sqnorm(e2, n));                                    /* :26 */
        pj = cj / (cj + dj);                                                     /* :27 */
      }
      /* R::rbinom(1,pj)==1 ; NaN pj compares false, like rbinom's NaN return */
      if (orng_bernoulli(&g, (double)pj, mk, iter)) {                  /* :28 */
        b[j] = b1; d[j] = 1; memcpy(e, e1, sizeof(E_T) * n);                     /* :29 */
      } else {
        b[j] = b2; d[j] = 0; memcpy(e, e2, sizeof(E_T) * n);                     /* :31 */
      }
    } else {
      d[j] = 1; b[j] = b1; memcpy(e, e1, sizeof(E_T) * n);                       /* :34 */
    }
  }
  for (int64_t i = 0; i < n; i++) e_io[i] = (float)e[i];                          /* :37 */
  free(e); free(e1); free(e2);
  return 0;
}


/* ---- wgr's row resampling, R/wgr.R:68: Use = sort(sample(n, n*bag, rp)) - 1 ---------------------------------
 * R's sample() runs on its serial stream; here the subset is a pure function of (seed, iteration):
 * without replacement = the rows holding the k smallest of n uniform keys (a uniformly random k-subset);
 * with replacement = floor(u_t * n) for k uniforms.  Returned sorted ascending.  k = (int)(n*bag). */
static int cmp_keyidx(const void *a, const void *b) {
  const double *x = (const double *)a, *y = (const double *)b;
  return (x[0] < y[0]) ? -1 : (x[0] > y[0]) ? 1 : ((x[1] < y[1]) ? -1 : (x[1] > y[1]));
}
static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }
int FN(oracle_bag_rows)(uint64_t seed, uint32_t iter, int64_t n, int64_t k, int rp, int *use) {
  orng_t g = { seed, ORNG_PHILOX };
  if (rp) {
    for (int64_t t = 0; t < k; t++) { int r = (int)(orng_uniform(&g, (uint32_t)t, iter, ORNG_BAG, 1) * (double)n); use[t] = r >= n ? (int)n - 1 : r; }
  } else {
    double *kv = (double *)malloc(sizeof(double) * 2 * n);
    if (!kv) return 1;
    for (int64_t i = 0; i < n; i++) { kv[2 * i] = orng_uniform(&g, (uint32_t)i, iter, ORNG_BAG, 0); kv[2 * i + 1] = (double)i; }
    qsort(kv, n, 2 * sizeof(double), cmp_keyidx);
    for (int64_t t = 0; t < k; t++) use[t] = (int)kv[2 * t + 1];
    free(kv);
  }
  qsort(use, k, sizeof(int), cmp_int);
  return 0;
}

/* ---- KMUP2: the sweep on a row subsample, /root/reference/src/Rcpp20260726ai.cpp:41-77 ------------------------
 * E has n0 entries, Use (0-based, n entries) selects the rows; the returned e has n entries (the gathered
 * subsample, :76).  Quirks kept: the conditional mean's numerator adds b0, not xx*b0 (:59); the denominator is
 * xx*bg + L with bg = n0/n (:47,59), where xx is whatever the caller passes (wgr passes colSums(X^2)*bag). */
int FN(oracle_kmup2)(const float *X, int64_t n0, int64_t p, int64_t ldx, const int *Use, int64_t n, float *b, float *d,
                     const float *xx, const float *E, float *e_out, const float *L, float Ve, float pi, uint64_t seed,
                     uint32_t iter, int rng_mode, int stable, uint32_t marker0) {
  orng_t g = { seed, rng_mode };
  E_T *e0 = (E_T *)malloc(sizeof(E_T) * n), *e1 = (E_T *)malloc(sizeof(E_T) * n), *e2 = (E_T *)malloc(sizeof(E_T) * n);
  float *H = (float *)malloc(sizeof(float) * n);
  if (!e0 || !e1 || !e2 || !H) return 1;
  float b0, b1, b2, cj, dj, pj;
  float C = -0.5f / sqrtf(Ve);                                                   /* :46 */
  float bg = (float)n0 / (float)n;                                               /* :47 */
  for (int64_t k = 0; k < n; k++) e0[k] = e1[k] = e2[k] = (E_T)E[Use[k]];        /* :49-53 */
  for (int64_t j = 0; j < p; j++) {                                              /* :54 */
    for (int64_t x = 0; x < n; x++) H[x] = X[j * ldx + Use[x]];                  /* :55-57 */
    uint32_t mk = marker0 + (uint32_t)j;
    b0 = b[j];                                                                   /* :58 */
    float den = xx[j] * bg + L[j];
    b1 = draw_b1_add(H, e0, n, b0, den, sqrtf(Ve / den), orng_normal(&g, mk, iter, ORNG_Z1, 0));   /* :59 (sic: + b0) */
    b2 = draw_norm(0.0f, sqrtf(Ve / den), orng_normal(&g, mk, iter, ORNG_Z2, 0));                 /* :60 */
    v_axpy_to(e1, e0, H, b1 - b0, n);                                            /* :61 */
    if (pi > 0) {                                                                /* :62 */
      v_axpy_to(e2, e0, H, b2 - b0, n);                                          /* :63 */
      if (stable) {
#ifdef ACC_WIDE
        float diff = (float)(e_sqnorm_acc(e2, n) - e_sqnorm_acc(e1, n));
#else
        float diff = e_sqnorm(e2, n) - e_sqnorm(e1, n);
#endif
        pj = 1.0f / (1.0f + (pi / (1.0f - pi)) * f_exp(C * diff));
      } else {
        cj = (1 - pi) * f_exp(C * e_sqnorm(e1, n));                              /* :64 */
        dj = (pi)*f_exp(C * e_sqnorm(e2, n));                                    /* :65 */
        pj = cj / (cj + dj);                                                     /* :66 */
      }
      if (orng_bernoulli(&g, (double)pj, mk, iter)) { b[j] = b1; d[j] = 1; memcpy(e0, e1, sizeof(E_T) * n); }   /* :67-68 */
      else { b[j] = b2; d[j] = 0; memcpy(e0, e2, sizeof(E_T) * n); }             /* :70 */
    } else { d[j] = 1; b[j] = b1; memcpy(e0, e1, sizeof(E_T) * n); }             /* :73 */
  }
  for (int64_t k = 0; k < n; k++) e_out[k] = (float)e0[k];                       /* :76 */
  free(e0); free(e1); free(e2); free(H);
  return 0;
}

/* ---- the seven fused samplers, /root/reference/src/Rcpp20260726ai.cpp:589-987 -----------------
 * Outputs follow the reference's return lists.  VB has p entries for A/B/L/Dpi and 1 for
 * C/RR/Cpi.  last_* (optional, may be NULL) receive the chain state after the final iteration:
 * b, d, e, vb (p entries; the common variance is replicated), scal = {mu, ve, vb_common, pi}. */
int FN(oracle_bayes)(int model, const float *y, const float *X, int64_t n, int64_t p, int64_t ldx,
                     float it, float bi, float pi_in, float df, float R2, uint64_t seed, int rng_mode,
                     float *o_mu, float *o_B, float *o_D, float *o_hat, float *o_VB, float *o_ve, float *o_h2,
                     float *o_MSx, float *o_Pi, float *o_PVAL,
                     float *last_b, float *last_d, float *last_e, float *last_vb, float *last_scal) {
  orng_t g = { seed, rng_mode };
  const uint32_t GM = ORNG_GLOBAL_MARKER;
  int iit = (int)it, ibi = (int)bi;                                   /* e.g. :611, :642 */
  float *xx = (float *)malloc(sizeof(float) * p), *vx = (float *)malloc(sizeof(float) * p);
  float *b = (float *)calloc(p, sizeof(float)), *d = (float *)calloc(p, sizeof(float));
  float *B = (float *)calloc(p, sizeof(float)), *D = (float *)calloc(p, sizeof(float));
  float *VBv = (float *)calloc(p, sizeof(float)), *vbv = (float *)malloc(sizeof(float) * p);
  float *Lmbv = (float *)malloc(sizeof(float) * p);
  E_T *e = (E_T *)malloc(sizeof(E_T) * n), *e1 = (E_T *)malloc(sizeof(E_T) * n), *e2 = (E_T *)malloc(sizeof(E_T) * n);
  if (!xx || !vx || !b || !d || !B || !D || !VBv || !vbv || !Lmbv || !e || !e1 || !e2) return 1;

  float MSx;
  FN(oracle_stats)(X, n, p, ldx, xx, vx, &MSx);                       /* :593-598 */
  float vy = v_fvar(y, n);                                            /* :599 */
  float pi = pi_in;
  if (model == M_BAYESCPI || model == M_BAYESDPI) pi = 0.5f;          /* :869, :935 */
  float Sb;
  if (model == M_BAYESC || model == M_BAYESCPI) Sb = df * (R2)*vy / MSx / (1 - pi);   /* :714, :871 */
  else Sb = (R2)*df * vy / MSx;                                       /* :600, :651, :775, :824, :937 */
  float Se = (1 - R2) * df * vy;                                      /* :601 */
  float Phi = MSx * (1 - R2) / R2;                                    /* :773 (BayesL) */
  float mu = v_mean(y, n);                                            /* :602 */
  float b0, b1, b2, eM, h2, C = 0, MU = 0, VE = 0, VBs = 0, Pi = 0, LR, pj, vg, ve = vy, vb = Sb;
  float Lmb = ve / vb;                                                /* :725, :826, :882 */
  for (int64_t j = 0; j < p; j++) { vbv[j] = Sb; Lmbv[j] = ve * (1.0f / Sb); }   /* :608-609 cwiseInverse */
  for (int64_t i = 0; i < n; i++) { float t = y[i] - mu; e[i] = t; } /* :610 */
  float Pi0 = pi / (1.0f - pi);                                       /* :665, :726, :883 */
  const int per_marker_vb = (model == M_BAYESA || model == M_BAYESB || model == M_BAYESL || model == M_BAYESDPI);

  for (int i = 0; i < iit; i++) {
    uint32_t itx = (uint32_t)i;
    C = -0.5f / sqrtf(ve);                                            /* :667 (unused by A/L/RR) */
    ACC_T dsum = 0;
    for (int64_t j = 0; j < p; j++) {
      const float *xj = X + j * ldx;
      uint32_t mk = (uint32_t)j;
      float lam = per_marker_vb ? Lmbv[j] : Lmb;
      float den = xx[j] + lam;
      b0 = b[j];
      b1 = draw_b1(xj, e, n, xx[j], b0, den, sqrtf(ve / den), orng_normal(&g, mk, itx, ORNG_Z1, 0));
      switch (model) {
      case M_BAYESA: case M_BAYESL:                                   /* :613-618, :787-792 */
        b[j] = b1;
        vbv[j] = (float)((double)(Sb + b1 * b1) / orng_chisq(&g, (double)(df + 1), mk, itx, ORNG_CHI));
        v_axpy(e, xj, b1 - b0, n);
        break;
      case M_BAYESRR:                                                 /* :834-837 */
        v_axpy(e, xj, b1 - b0, n);
        b[j] = b1;
        break;
      case M_BAYESB: case M_BAYESC:                                   /* :669-681, :729-741 */
        v_axpy_to(e1, e, xj, b1 - b0, n);
        v_axpy_to(e2, e, xj, 0 - b0, n);
        {
#ifdef ACC_WIDE
          float diff = (float)(e_sqnorm_acc(e2, n) - e_sqnorm_acc(e1, n));
#else
          float diff = e_sqnorm(e2, n) - e_sqnorm(e1, n);
#endif
          LR = Pi0 * f_exp(C * diff);
        }
        pj = 1.0f / (1.0f + LR);
        if (orng_bernoulli(&g, (double)pj, mk, itx)) {
          b[j] = b1; d[j] = 1;
        } else {
          b[j] = draw_norm(0.0f, sqrtf(ve / den), orng_normal(&g, mk, itx, ORNG_Z2, 0)); d[j] = 0;
        }
        if (model == M_BAYESB)
          vbv[j] = (float)((double)(Sb + b[j] * b[j]) / orng_chisq(&g, (double)(df + 1), mk, itx, ORNG_CHI));
        v_axpy(e, xj, b[j] - b0, n);
        break;
      case M_BAYESCPI:                                                /* :886-900 */
        b2 = draw_norm(0.0f, sqrtf(ve / den), orng_normal(&g, mk, itx, ORNG_Z2, 0));
        v_axpy_to(e1, e, xj, b1 - b0, n);
        v_axpy_to(e2, e, xj, 0 - b0, n);
        {
#ifdef ACC_WIDE
          float diff = (float)(e_sqnorm_acc(e2, n) - e_sqnorm_acc(e1, n));
#else
          float diff = e_sqnorm(e2, n) - e_sqnorm(e1, n);
#endif
          LR = Pi0 * f_exp(C * diff);
        }
        pj = 1.0f / (1.0f + LR);
        if (orng_bernoulli(&g, (double)pj, mk, itx)) { b[j] = b1; d[j] = 1; }
        else { b[j] = b2; d[j] = 0; }
        v_axpy(e, xj, b[j] - b0, n);
        break;
      case M_BAYESDPI:                                                /* :954-968 */
        b2 = draw_norm(0.0f, sqrtf(ve / den), orng_normal(&g, mk, itx, ORNG_Z2, 0));
        v_axpy_to(e1, e, xj, b1 - b0, n);
        v_axpy_to(e2, e, xj, b2 - b0, n);
        {
#ifdef ACC_WIDE
          float diff = (float)(e_sqnorm_acc(e1, n) - e_sqnorm_acc(e2, n));
#else
          float diff = e_sqnorm(e1, n) - e_sqnorm(e2, n);
#endif
          pj = (1 - pi) * f_exp(C * diff);
        }
        if (pj > 1) pj = 1;
        if (orng_bernoulli(&g, (double)pj, mk, itx)) { b[j] = b1; d[j] = 1; }
        else { b[j] = b2; d[j] = 0; }
        vbv[j] = (float)((double)(Sb + b[j] * b[j]) / orng_chisq(&g, (double)(df + 1), mk, itx, ORNG_CHI));
        v_axpy(e, xj, b[j] - b0, n);
        break;
      }
      dsum += (ACC_T)d[j];
    }
    /* intercept, :620-621 */
    eM = draw_norm(e_mean(e, n), sqrtf(ve / n), orng_normal(&g, GM, itx, ORNG_G_MU, 0));
    mu += eM;
    for (int64_t k = 0; k < n; k++) e[k] = e[k] - eM;
    switch (model) {
    case M_BAYESA: case M_BAYESB: case M_BAYESDPI:                    /* :622-623, :685-686, :971-972 */
      ve = (float)((double)(e_sqnorm(e, n) + Se) / orng_chisq(&g, (double)(n + df), GM, itx, ORNG_G_VE));
      for (int64_t j = 0; j < p; j++) Lmbv[j] = ve * (1.0f / vbv[j]);
      break;
    case M_BAYESL:                                                    /* :796-797 */
      ve = (float)((double)(e_sqnorm(e, n) + Se) / orng_chisq(&g, (double)(n + df), GM, itx, ORNG_G_VE));
      for (int64_t j = 0; j < p; j++) Lmbv[j] = sqrtf(Phi * ve / vbv[j]);
      break;
    case M_BAYESRR:                                                   /* :841-843: ve first */
      ve = (float)((double)(e_sqnorm(e, n) + Se) / orng_chisq(&g, (double)(n + df), GM, itx, ORNG_G_VE));
      vb = (float)((double)(v_sqnorm(b, p) + Sb) / orng_chisq(&g, (double)(p + df), GM, itx, ORNG_G_VB));
      Lmb = ve / vb;
      break;
    case M_BAYESC: case M_BAYESCPI:                                   /* :745-747, :903-905: vb first */
      vb = (float)((double)(v_sqnorm(b, p) + Sb) / orng_chisq(&g, (double)(model == M_BAYESC ? df + p : p + df), GM, itx, ORNG_G_VB));
      ve = (float)((double)(e_sqnorm(e, n) + Se) / orng_chisq(&g, (double)(n + df), GM, itx, ORNG_G_VE));
      Lmb = ve / vb;
      break;
    }
    if (model == M_BAYESCPI) {                                        /* :906-907 */
      pi = (float)(dsum / (ACC_T)p);
      Sb = df * (R2)*vy / MSx / (1 - pi);
      /* note: Pi0 is NOT refreshed inside the loop in the reference (:883 is outside) */
    }
    if (model == M_BAYESDPI) pi = (float)(dsum / (ACC_T)p);           /* :973 */
    if (i > ibi) {                                                    /* :624, :687 ... */
      MU += mu; VE += ve;
      for (int64_t j = 0; j < p; j++) B[j] += b[j];
      if (per_marker_vb) for (int64_t j = 0; j < p; j++) VBv[j] += vbv[j];
      else VBs += vb;
      if (model != M_BAYESA && model != M_BAYESL && model != M_BAYESRR) for (int64_t j = 0; j < p; j++) D[j] += d[j];
      if (model == M_BAYESCPI || model == M_BAYESDPI) Pi += pi;
    }
  }
  float MCMC = it - bi;                                               /* :626 */
  MU /= MCMC; VE /= MCMC;
  for (int64_t j = 0; j < p; j++) { B[j] /= MCMC; D[j] /= MCMC; }
  if (per_marker_vb) { for (int64_t j = 0; j < p; j++) VBv[j] /= MCMC; vg = (float)v_sum_acc(VBv, p); }
  else { VBs /= MCMC; vg = VBs * MSx; }
  if (model == M_BAYESCPI || model == M_BAYESDPI) Pi = 1 - Pi / MCMC; /* :911, :977 */
  if (model == M_BAYESCPI) vg = VBs * MSx / Pi;                       /* :913 */
  h2 = vg / (vg + VE);
  /* fit = X*B + MU, :629-630 */
  for (int64_t k = 0; k < n; k++) o_hat[k] = 0;
  {
    ACC_T *acc = (ACC_T *)calloc(n, sizeof(ACC_T));
    for (int64_t j = 0; j < p; j++) { const float *xj = X + j * ldx; ACC_T Bj = (ACC_T)B[j]; for (int64_t k = 0; k < n; k++) acc[k] += (ACC_T)xj[k] * Bj; }
    for (int64_t k = 0; k < n; k++) { float f = (float)acc[k]; o_hat[k] = f + MU; }
    free(acc);
  }
  *o_mu = MU; *o_ve = VE; *o_h2 = h2; *o_MSx = MSx; *o_Pi = Pi;
  memcpy(o_B, B, sizeof(float) * p);
  memcpy(o_D, D, sizeof(float) * p);
  if (per_marker_vb) memcpy(o_VB, VBv, sizeof(float) * p); else o_VB[0] = VBs;
  if (o_PVAL) for (int64_t j = 0; j < p; j++) o_PVAL[j] = -1.0f * logf(1.0f - D[j]);   /* :912 */
  if (last_b) memcpy(last_b, b, sizeof(float) * p);
  if (last_d) memcpy(last_d, d, sizeof(float) * p);
  if (last_e) for (int64_t k = 0; k < n; k++) last_e[k] = (float)e[k];
  if (last_vb) for (int64_t j = 0; j < p; j++) last_vb[j] = per_marker_vb ? vbv[j] : vb;
  if (last_scal) { last_scal[0] = mu; last_scal[1] = ve; last_scal[2] = vb; last_scal[3] = pi; }
  free(xx); free(vx); free(b); free(d); free(B); free(D); free(VBv); free(vbv); free(Lmbv); free(e); free(e1); free(e2);
  return 0;
}

/* ---- two-effect samplers BayesA2 / BayesB2 / BayesRR2, /root/reference/src/Rcpp20260726ai.cpp:990-1218 --------------
 * One residual, two marker panels X1 (p1) and X2 (p2) swept one after the other in every iteration, each with its own
 * prior scale (Sb_k from MSx_k) and variance(s); one intercept and one residual variance.  model2: 0 = A2, 1 = B2, 2 = RR2.
 * Variates: panel-2 markers carry the global ids p1 .. p1+p2-1; RR2's second common-variance chi-square is drawn with the
 * marker word ORNG_GLOBAL_MARKER - 1.  BayesB2's inclusion probability cj/(cj+dj) (:1111-1113) is evaluated in the
 * stable form 1/(1 + pi/(1-pi) exp(C(|e2|^2-|e1|^2))), like BayesB's own (:673-674) -- the literal form is 0/0 once
 * 0.5|e|^2/sqrt(ve) exceeds ~103 (see oracle_kmup's `stable`).  Note the alternative state of B2 is the drawn b_t2 in
 * BOTH the likelihood comparison and the update (:1108-1109), unlike BayesB whose comparison uses 0 (:672).
 * VB1 / VB2: p_k entries for A2 and B2, one entry for RR2.  last_* (optional): b1, b2, e, {mu, ve} after the final iteration. */
#define ORNG_GLOBAL_MARKER2 (ORNG_GLOBAL_MARKER - 1u)
int FN(oracle_bayes2)(int model2, const float *y, const float *X1, int64_t p1, const float *X2, int64_t p2, int64_t n,
                      float it, float bi, float pi, float df, float R2, uint64_t seed, int rng_mode,
                      float *o_mu, float *o_B1, float *o_D1, float *o_VB1, float *o_B2, float *o_D2, float *o_VB2,
                      float *o_ve, float *o_hat, float *o_h2, float *last_b1, float *last_b2, float *last_e, float *last_scal) {
  orng_t g = { seed, rng_mode };
  const uint32_t GM = ORNG_GLOBAL_MARKER;
  const int iit = (int)it, ibi = (int)bi;
  const float *Xs[2] = { X1, X2 };
  const int64_t ps[2] = { p1, p2 };
  float *xx[2], *vx[2], *b[2], *d[2], *B[2], *D[2], *VBv[2], *vbv[2], *Lmbv[2];
  float MSx[2], Sb[2], vbc[2], VBs[2] = { 0, 0 }, Lmb[2];
  for (int k = 0; k < 2; k++) {
    const int64_t p = ps[k];
    xx[k] = (float *)malloc(sizeof(float) * p); vx[k] = (float *)malloc(sizeof(float) * p);
    b[k] = (float *)calloc(p, sizeof(float)); d[k] = (float *)calloc(p, sizeof(float));
    B[k] = (float *)calloc(p, sizeof(float)); D[k] = (float *)calloc(p, sizeof(float));
    VBv[k] = (float *)calloc(p, sizeof(float)); vbv[k] = (float *)malloc(sizeof(float) * p); Lmbv[k] = (float *)malloc(sizeof(float) * p);
    if (!xx[k] || !vx[k] || !b[k] || !d[k] || !B[k] || !D[k] || !VBv[k] || !vbv[k] || !Lmbv[k]) return 1;
    FN(oracle_stats)(Xs[k], n, p, n, xx[k], vx[k], &MSx[k]);            /* :997-1008 */
  }
  E_T *e = (E_T *)malloc(sizeof(E_T) * n), *e1 = (E_T *)malloc(sizeof(E_T) * n), *e2 = (E_T *)malloc(sizeof(E_T) * n);
  if (!e || !e1 || !e2) return 1;
  const float vy = v_fvar(y, n);                                        /* :1009 */
  for (int k = 0; k < 2; k++) Sb[k] = (R2)*df * vy / MSx[k];            /* :1010-1011 */
  const float Se = (1 - R2) * df * vy;                                  /* :1012 */
  float mu = v_mean(y, n), ve = vy, MU = 0, VE = 0, eM, C = 0;
  for (int k = 0; k < 2; k++) {
    for (int64_t j = 0; j < ps[k]; j++) { vbv[k][j] = Sb[k]; Lmbv[k][j] = ve * (1.0f / Sb[k]); }   /* :1021-1024 */
    Lmb[k] = MSx[k]; vbc[k] = 0;                                        /* RR2 starts with Lmb = MSx, :1190 */
  }
  for (int64_t i = 0; i < n; i++) { float t = y[i] - mu; e[i] = t; }    /* :1025 */
  const float Pi0 = pi / (1.0f - pi);
  for (int i = 0; i < iit; i++) {
    const uint32_t itx = (uint32_t)i;
    C = -0.5f / sqrtf(ve);                                              /* :1103 */
    for (int k = 0; k < 2; k++) {
      const uint32_t mk0 = (uint32_t)(k ? p1 : 0);
      for (int64_t j = 0; j < ps[k]; j++) {
        const float *xj = Xs[k] + j * n;
        const uint32_t mk = mk0 + (uint32_t)j;
        const float lam = (model2 == 2) ? Lmb[k] : Lmbv[k][j];
        const float den = xx[k][j] + lam;
        const float b0 = b[k][j];
        const float b1 = draw_b1(xj, e, n, xx[k][j], b0, den, sqrtf(ve / den), orng_normal(&g, mk, itx, ORNG_Z1, 0));
        if (model2 == 1) {                                              /* :1104-1121 */
          const float b2 = draw_norm(0.0f, sqrtf(ve / den), orng_normal(&g, mk, itx, ORNG_Z2, 0));
          v_axpy_to(e1, e, xj, b1 - b0, n);
          v_axpy_to(e2, e, xj, b2 - b0, n);
#ifdef ACC_WIDE
          const float diff = (float)(e_sqnorm_acc(e2, n) - e_sqnorm_acc(e1, n));
#else
          const float diff = e_sqnorm(e2, n) - e_sqnorm(e1, n);
#endif
          const float pj = 1.0f / (1.0f + Pi0 * f_exp(C * diff));
          if (orng_bernoulli(&g, (double)pj, mk, itx)) { b[k][j] = b1; d[k][j] = 1; }
          else { b[k][j] = b2; d[k][j] = 0; }
        } else {
          b[k][j] = b1;                                                 /* :1031, :1196 */
        }
        if (model2 != 2)                                                /* :1032, :1119 */
          vbv[k][j] = (float)((double)(Sb[k] + b[k][j] * b[k][j]) / orng_chisq(&g, (double)(df + 1), mk, itx, ORNG_CHI));
        v_axpy(e, xj, b[k][j] - b0, n);
      }
    }
    eM = draw_norm(e_mean(e, n), sqrtf(ve / n), orng_normal(&g, GM, itx, ORNG_G_MU, 0));   /* :1042 */
    mu += eM;
    for (int64_t q = 0; q < n; q++) e[q] = e[q] - eM;
    ve = (float)((double)(e_sqnorm(e, n) + Se) / orng_chisq(&g, (double)(n + df), GM, itx, ORNG_G_VE));   /* :1044 */
    if (model2 == 2) {                                                  /* :1207-1209 */
      vbc[0] = (float)((double)(Sb[0] + v_sqnorm(b[0], p1)) / orng_chisq(&g, (double)(df + p1), GM, itx, ORNG_G_VB));
      vbc[1] = (float)((double)(Sb[1] + v_sqnorm(b[1], p2)) / orng_chisq(&g, (double)(df + p2), ORNG_GLOBAL_MARKER2, itx, ORNG_G_VB));
      Lmb[0] = ve / vbc[0]; Lmb[1] = ve / vbc[1];
    } else {
      for (int k = 0; k < 2; k++) for (int64_t j = 0; j < ps[k]; j++) Lmbv[k][j] = ve * (1.0f / vbv[k][j]);   /* :1045-1046 */
    }
    if (i > ibi) {                                                      /* :1047 */
      MU += mu; VE += ve;
      for (int k = 0; k < 2; k++) {
        for (int64_t j = 0; j < ps[k]; j++) { B[k][j] += b[k][j]; D[k][j] += d[k][j]; }
        if (model2 == 2) VBs[k] += vbc[k]; else for (int64_t j = 0; j < ps[k]; j++) VBv[k][j] += vbv[k][j];
      }
    }
  }
  const float MCMC = it - bi;                                           /* :1049 */
  MU /= MCMC; VE /= MCMC;
  float vg;
  for (int k = 0; k < 2; k++) {
    for (int64_t j = 0; j < ps[k]; j++) { B[k][j] /= MCMC; D[k][j] /= MCMC; VBv[k][j] /= MCMC; }
    VBs[k] /= MCMC;
  }
  if (model2 == 2) vg = VBs[0] * MSx[0] + VBs[1] * MSx[1];              /* :1213 */
  else vg = (float)v_sum_acc(VBv[0], p1) + (float)v_sum_acc(VBv[1], p2);   /* :1051 */
  /* fit = X1*B1 + X2*B2; fit += MU  (:1052-1053): two float products added, then the intercept */
  {
    ACC_T *acc = (ACC_T *)malloc(sizeof(ACC_T) * n);
    float *f1 = (float *)malloc(sizeof(float) * n);
    for (int k = 0; k < 2; k++) {
      for (int64_t q = 0; q < n; q++) acc[q] = 0;
      for (int64_t j = 0; j < ps[k]; j++) { const float *xj = Xs[k] + j * n; const ACC_T Bj = (ACC_T)B[k][j]; for (int64_t q = 0; q < n; q++) acc[q] += (ACC_T)xj[q] * Bj; }
      if (k == 0) for (int64_t q = 0; q < n; q++) f1[q] = (float)acc[q];
      else for (int64_t q = 0; q < n; q++) { const float f = f1[q] + (float)acc[q]; o_hat[q] = f + MU; }
    }
    free(acc); free(f1);
  }
  *o_mu = MU; *o_ve = VE; *o_h2 = vg / (vg + VE);
  memcpy(o_B1, B[0], sizeof(float) * p1); memcpy(o_B2, B[1], sizeof(float) * p2);
  memcpy(o_D1, D[0], sizeof(float) * p1); memcpy(o_D2, D[1], sizeof(float) * p2);
  if (model2 == 2) { o_VB1[0] = VBs[0]; o_VB2[0] = VBs[1]; }
  else { memcpy(o_VB1, VBv[0], sizeof(float) * p1); memcpy(o_VB2, VBv[1], sizeof(float) * p2); }
  if (last_b1) memcpy(last_b1, b[0], sizeof(float) * p1);
  if (last_b2) memcpy(last_b2, b[1], sizeof(float) * p2);
  if (last_e) for (int64_t q = 0; q < n; q++) last_e[q] = (float)e[q];
  if (last_scal) { last_scal[0] = mu; last_scal[1] = ve; }
  for (int k = 0; k < 2; k++) { free(xx[k]); free(vx[k]); free(b[k]); free(d[k]); free(B[k]); free(D[k]); free(VBv[k]); free(vbv[k]); free(Lmbv[k]); }
  free(e); free(e1); free(e2);
  return 0;
}

/* ---- wgr(): the split shape, /root/reference/R/wgr.R:2-169 (bag = 1) ---------------------------------
 * R arithmetic is double; KMUP is entered through the Rcpp glue that narrows every argument to
 * float (/root/reference/src/RcppExports.cpp:20-27) and widens the returned b, d, e back to double
 * (/root/reference/src/Rcpp20260726ai.cpp:37).  X arrives as an R numeric (double) matrix.
 * o_Vb has p entries when iv (or de) is set, else 1.  Iteration i (1-based in R) uses RNG
 * iteration word i-1.
 * Polygenic term (eigK, wgr.R:23-32,70-78,116-119,148-150): U = the first pk eigenvectors (n x pk, column-major,
 * rows already restricted to non-missing y), V = their eigenvalues; pass U = NULL for eigK = NULL.  The kernel
 * sweep KMUP(U,h,dh,xxK,e,Lk,Ve,0) draws with marker ids 0x80000000 + k so that its variates are independent of the
 * marker sweep's.  o_u (n) receives U %*% H, o_Vk the posterior mean of Vp. */
#define ORNG_KERNEL_MARKER0 0x80000000u
int FN(oracle_wgr)(const double *y, const double *X, int64_t n, int64_t p, int64_t ldx, int it, int bi, int th,
                   int iv, int de, double pi, double df, double R2, uint64_t seed, int rng_mode, int stable,
                   const double *U, const double *V, int64_t pk, double bag, int rp,
                   double *o_mu, double *o_b, double *o_Vb, double *o_d, double *o_Ve, double *o_hat, double *o_cxx,
                   double *o_u, double *o_Vk) {
  orng_t g = { seed, rng_mode };
  const uint32_t GM = ORNG_GLOBAL_MARKER;
  if (de) iv = 1;                                                             /* wgr.R:9 */
  if (!U) pk = 0;
  if (bag != 1.0 && pk > 0) return 2;   /* the reference indexes a subsampled e with full-length row ids here (wgr.R:73-79): undefined */
  if (bag != 1.0) df = df / (bag * bag);                                      /* wgr.R:20 */
  const int64_t nbag = (bag != 1.0) ? (int64_t)((double)n * bag) : n;
  int *use = (int *)malloc(sizeof(int) * (nbag + 1));
  float *ebag = (float *)malloc(sizeof(float) * (nbag + 1));
  if (!use || !ebag) return 1;
  float *Xf = (float *)malloc(sizeof(float) * n * p);
  float *bf = (float *)malloc(sizeof(float) * p), *dfl = (float *)malloc(sizeof(float) * p), *xxf = (float *)malloc(sizeof(float) * p);
  float *Lf = (float *)malloc(sizeof(float) * p), *ef = (float *)malloc(sizeof(float) * n);
  double *xx = (double *)malloc(sizeof(double) * p), *b = (double *)calloc(p, sizeof(double)), *d = (double *)malloc(sizeof(double) * p);
  double *Vb = (double *)malloc(sizeof(double) * p), *L = (double *)malloc(sizeof(double) * p), *e = (double *)malloc(sizeof(double) * n);
  double *B = (double *)calloc(p, sizeof(double)), *D = (double *)calloc(p, sizeof(double)), *VB = (double *)calloc(p, sizeof(double));
  float *Uf = (float *)malloc(sizeof(float) * (n * pk + 1)), *hf = (float *)malloc(sizeof(float) * (pk + 1)), *dhf = (float *)malloc(sizeof(float) * (pk + 1));
  float *xxKf = (float *)malloc(sizeof(float) * (pk + 1)), *Lkf = (float *)malloc(sizeof(float) * (pk + 1));
  double *h = (double *)calloc(pk + 1, sizeof(double)), *H = (double *)calloc(pk + 1, sizeof(double));
  if (!Xf || !bf || !dfl || !xxf || !Lf || !ef || !xx || !b || !d || !Vb || !L || !e || !B || !D || !VB || !Uf || !hf || !dhf || !xxKf || !Lkf || !h || !H) return 1;
  for (int64_t j = 0; j < p; j++) for (int64_t i = 0; i < n; i++) Xf[j * n + i] = (float)X[j * ldx + i];
  for (int64_t k = 0; k < pk; k++) for (int64_t i = 0; i < n; i++) Uf[k * n + i] = (float)U[k * n + i];
  /* post = seq(bi,it,th); mc = length(post)                                   wgr.R:41-42 */
  int mc = 0; for (int q = bi; q <= it; q += th) mc++;
  double MSx = 0, sy = 0;
  for (int64_t j = 0; j < p; j++) {                                           /* wgr.R:46,51 */
    const double *xj = X + j * ldx; double s2 = 0, s1 = 0;
    for (int64_t i = 0; i < n; i++) { s2 += xj[i] * xj[i]; s1 += xj[i]; }
    xx[j] = s2 * bag; double m = s1 / (double)n, v = 0;                         /* wgr.R:46: crossprod * bag */
    for (int64_t i = 0; i < n; i++) v += (xj[i] - m) * (xj[i] - m);
    MSx += v / (double)(n - 1);
    d[j] = 1;                                                                 /* wgr.R:48 */
  }
  for (int64_t i = 0; i < n; i++) sy += y[i];
  double mu = sy / (double)n;                                                 /* wgr.R:49 */
  double vy = 0; for (int64_t i = 0; i < n; i++) { e[i] = y[i] - mu; vy += e[i] * e[i]; }  /* wgr.R:50,56 */
  vy /= (double)(n - 1);
  double Va = MSx, Ve = 1;                                                    /* wgr.R:52-54 */
  for (int64_t j = 0; j < p; j++) { Vb[j] = Va; L[j] = Vb[j] / Ve; }          /* wgr.R:53,55 (sic) */
  double Sb = (R2)*df * vy / MSx, Se = (1 - R2) * df * vy;                    /* wgr.R:58-59 */
  double Sk = R2 * vy * (df + 2), Vp = 1, VP = 0;                             /* wgr.R:60; Vk = rep(1,pk) wgr.R:30 */
  double B0 = 0, VA = 0, VE = 0;
  for (int i = 1; i <= it; i++) {                                             /* wgr.R:66 */
    uint32_t itx = (uint32_t)(i - 1);
    if (pk > 0) {                                                             /* wgr.R:70-78 */
      for (int64_t k = 0; k < pk; k++) { hf[k] = (float)h[k]; dhf[k] = 0.0f; xxKf[k] = 1.0f; Lkf[k] = (float)(Ve / (V[k] * Vp)); }
      for (int64_t k = 0; k < n; k++) ef[k] = (float)e[k];
      int rc = FN(oracle_kmup)(Uf, n, pk, n, hf, dhf, xxKf, ef, Lkf, (float)Ve, 0.0f, seed, itx, rng_mode, stable, ORNG_KERNEL_MARKER0);
      if (rc) return rc;
      for (int64_t k = 0; k < pk; k++) h[k] = (double)hf[k];
      for (int64_t k = 0; k < n; k++) e[k] = (double)ef[k];
    }
    for (int64_t j = 0; j < p; j++) { bf[j] = (float)b[j]; dfl[j] = (float)d[j]; xxf[j] = (float)xx[j]; Lf[j] = (float)L[j]; }
    for (int64_t k = 0; k < n; k++) ef[k] = (float)e[k];
    int rc;
    if (bag != 1.0) {                                                         /* wgr.R:68,85: KMUP2 on the resampled rows */
      rc = FN(oracle_bag_rows)(seed, itx, n, nbag, rp, use);
      if (rc) return rc;
      rc = FN(oracle_kmup2)(Xf, n, p, n, use, nbag, bf, dfl, xxf, ef, ebag, Lf, (float)Ve, (float)pi, seed, itx, rng_mode, stable, 0u);
    } else {
      rc = FN(oracle_kmup)(Xf, n, p, n, bf, dfl, xxf, ef, Lf, (float)Ve, (float)pi, seed, itx, rng_mode, stable, 0u);   /* wgr.R:85 */
    }
    if (rc) return rc;
    if (pi > 0) for (int64_t j = 0; j < p; j++) d[j] = (double)dfl[j];        /* wgr.R:86 */
    for (int64_t j = 0; j < p; j++) b[j] = (double)bf[j];                     /* wgr.R:87 */
    const int64_t ne = (bag != 1.0) ? nbag : n;                               /* e is the subsample after KMUP2 */
    for (int64_t k = 0; k < ne; k++) e[k] = (double)((bag != 1.0) ? ebag[k] : ef[k]);   /* wgr.R:88 */
    if (iv) {                                                                 /* wgr.R:91-111 */
      if (de) for (int64_t j = 0; j < p; j++) Vb[j] = sqrt(b[j] * b[j] * Ve / MSx);
      else for (int64_t j = 0; j < p; j++) Vb[j] = (Sb + b[j] * b[j]) / orng_chisq(&g, df + 1, (uint32_t)j, itx, ORNG_CHI);
    } else {                                                                  /* wgr.R:112-115 */
      double bb = 0; for (int64_t j = 0; j < p; j++) bb += b[j] * b[j];
      Va = (bb + Sb) / orng_chisq(&g, df + (double)p, GM, itx, ORNG_G_VB);
      for (int64_t j = 0; j < p; j++) Vb[j] = Va;
    }
    if (pk > 0) {                                                             /* wgr.R:116-119 */
      double hv = 0; for (int64_t k = 0; k < pk; k++) hv += h[k] * h[k] / V[k];
      Vp = (hv + Sk) / orng_chisq(&g, df + (double)pk, GM, itx, ORNG_G_VK);
    }
    double ee = 0; for (int64_t k = 0; k < ne; k++) ee += e[k] * e[k];
    Ve = (ee + Se) / orng_chisq(&g, (double)n * bag + df, GM, itx, ORNG_G_VE);   /* wgr.R:121: n*bag+df */
    for (int64_t j = 0; j < p; j++) L[j] = Ve / Vb[j];                        /* wgr.R:122 */
    for (int64_t k = 0; k < n; k++) e[k] = y[k] - mu;                         /* wgr.R:124 */
    for (int64_t j = 0; j < p; j++) { const double *xj = X + j * ldx; double bj = b[j]; if (bj != 0) for (int64_t k = 0; k < n; k++) e[k] -= xj[k] * bj; }
    for (int64_t q = 0; q < pk; q++) { const double *uq = U + q * n; double hq = h[q]; for (int64_t k = 0; k < n; k++) e[k] -= uq[k] * hq; }
    double me = 0; for (int64_t k = 0; k < n; k++) me += e[k]; me /= (double)n;
    double mu0 = me + (Ve / (double)n) * orng_normal(&g, GM, itx, ORNG_G_MU, 0);   /* wgr.R:125 sd = Ve/n (sic) */
    mu += mu0;                                                                /* wgr.R:126 */
    for (int64_t k = 0; k < n; k++) e[k] -= mu0;                              /* wgr.R:127 */
    if (i >= bi && ((i - bi) % th) == 0) {                                    /* wgr.R:129 i %in% post */
      B0 += mu; VE += Ve;
      for (int64_t j = 0; j < p; j++) { B[j] += b[j]; D[j] += d[j]; }
      if (iv) for (int64_t j = 0; j < p; j++) VB[j] += Vb[j]; else VA += Va;
      for (int64_t k = 0; k < pk; k++) H[k] += h[k];
      if (pk > 0) VP += Vp;
    }
  }
  B0 /= mc; VE /= mc;                                                         /* wgr.R:141-145 */
  double mD = 0; for (int64_t j = 0; j < p; j++) { D[j] /= mc; mD += D[j]; } mD /= (double)p;
  for (int64_t j = 0; j < p; j++) B[j] = B[j] / mc / mD;                      /* wgr.R:143 */
  if (iv) { for (int64_t j = 0; j < p; j++) o_Vb[j] = VB[j] / mc; } else o_Vb[0] = VA / mc;
  for (int64_t k = 0; k < n; k++) o_hat[k] = B0;                              /* wgr.R:152 */
  for (int64_t j = 0; j < p; j++) { const double *xj = X + j * ldx; double Bj = B[j]; for (int64_t k = 0; k < n; k++) o_hat[k] += xj[k] * Bj; }
  if (pk > 0) {                                                               /* wgr.R:146-150 */
    for (int64_t k = 0; k < n; k++) o_u[k] = 0;
    for (int64_t q = 0; q < pk; q++) { const double *uq = U + q * n; double Hq = H[q] / mc; for (int64_t k = 0; k < n; k++) o_u[k] += uq[k] * Hq; }
    for (int64_t k = 0; k < n; k++) o_hat[k] += o_u[k];
    *o_Vk = VP / mc;
  }
  double cxx = 0; for (int64_t j = 0; j < p; j++) cxx += xx[j]; cxx /= (double)p;
  *o_mu = B0; *o_Ve = VE; *o_cxx = cxx;
  memcpy(o_b, B, sizeof(double) * p); memcpy(o_d, D, sizeof(double) * p);
  free(Xf); free(bf); free(dfl); free(xxf); free(Lf); free(ef); free(xx); free(b); free(d); free(Vb); free(L); free(e); free(B); free(D); free(VB);
  free(Uf); free(hf); free(dhf); free(xxKf); free(Lkf); free(h); free(H); free(use); free(ebag);
  return 0;
}

/* ---- EM / Gauss-Seidel family, /root/reference/src/Rcpp20260726ai.cpp:80-128 (emBA), :250-305 (emDE), :308-354 (emRR),
 * :463-521 (emML) -------------------------------------------------------------------------------------------------------
 * Deterministic coordinate updates b_j = (X_j.e + xx_j b_j)/(xx_j + lambda_j) in a marker order that is re-shuffled in
 * place before every sweep: std::shuffle(order.begin(), order.end(), std::mt19937(i)) (:103, :277, :331, :491).
 *
 * THIRD-PARTY ALGORITHM (not under /root/reference): the C++ standard library's std::mt19937, std::shuffle and
 * std::uniform_int_distribution.  The reference pins no toolchain (DESCRIPTION:13-15), and std::shuffle's draw sequence is
 * implementation-defined; restated here is GNU libstdc++ as of GCC 11 (bits/stl_algo.h shuffle + __gen_two_uniform_ints,
 * bits/uniform_int_dist.h Lemire "nearly divisionless" downscaling for 32-bit generators), i.e. what an R package built
 * with g++ >= 11 on Linux runs.  tests/test_em_order.py pins this restatement against the std::shuffle of the libstdc++
 * installed in this image (through the product's bwgr_em_order, which calls the library itself). */
#ifndef BWGR_ORACLE_MT_DEFINED
#define BWGR_ORACLE_MT_DEFINED
typedef struct { uint32_t s[624]; int idx; } omt_t;
static void omt_seed(omt_t *g, uint32_t seed) {                       /* std::mt19937(seed) */
  g->s[0] = seed;
  for (int k = 1; k < 624; k++) g->s[k] = 1812433253u * (g->s[k - 1] ^ (g->s[k - 1] >> 30)) + (uint32_t)k;
  g->idx = 624;
}
static uint32_t omt_next(omt_t *g) {
  if (g->idx >= 624) {
    for (int k = 0; k < 624; k++) {
      uint32_t yv = (g->s[k] & 0x80000000u) | (g->s[(k + 1) % 624] & 0x7FFFFFFFu);
      uint32_t v = g->s[(k + 397) % 624] ^ (yv >> 1);
      if (yv & 1u) v ^= 0x9908B0DFu;
      g->s[k] = v;
    }
    g->idx = 0;
  }
  uint32_t yv = g->s[g->idx++];
  yv ^= yv >> 11; yv ^= (yv << 7) & 0x9D2C5680u; yv ^= (yv << 15) & 0xEFC60000u; yv ^= yv >> 18;
  return yv;
}
/* uniform_int_distribution<unsigned long>{0, range-1}(g) for a 32-bit generator: _S_nd<uint64_t>(g, (uint32_t)range) */
static uint32_t omt_below(omt_t *g, uint32_t range) {
  uint64_t product = (uint64_t)omt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (uint32_t)(0u - range) % range;
    while (low < threshold) { product = (uint64_t)omt_next(g) * (uint64_t)range; low = (uint32_t)product; }
  }
  return (uint32_t)(product >> 32);
}
static void oem_shuffle(int *order, int64_t p, uint32_t seed) {        /* std::shuffle(first, last, std::mt19937(seed)) */
  if (p <= 0) return;
  omt_t g; omt_seed(&g, seed);
  const uint64_t urngrange = 0xFFFFFFFFull, urange = (uint64_t)p;
#define OEM_SWAP(a_, b_) do { int t_ = order[a_]; order[a_] = order[b_]; order[b_] = t_; } while (0)
  if (urngrange / urange >= urange) {                                  /* two swap positions per draw */
    int64_t i = 1;
    if ((urange % 2) == 0) { uint32_t r = omt_below(&g, 2u); OEM_SWAP(i, (int64_t)r); i++; }
    while (i != p) {
      const uint64_t swap_range = (uint64_t)i + 1;
      const uint64_t b1 = swap_range + 1;
      const uint32_t x = omt_below(&g, (uint32_t)(swap_range * b1));   /* {0, b0*b1 - 1} */
      const uint64_t pos0 = x / b1, pos1 = x % b1;
      OEM_SWAP(i, (int64_t)pos0); i++;
      OEM_SWAP(i, (int64_t)pos1); i++;
    }
    return;
  }
  for (int64_t i = 1; i != p; ++i) { uint32_t r = omt_below(&g, (uint32_t)(i + 1)); OEM_SWAP(i, (int64_t)r); }
#undef OEM_SWAP
}
#endif

/* the marker order of sweep `upto` (0-based): identity shuffled with seeds 0, 1, ..., upto */
int FN(oracle_em_order)(int64_t p, int upto, int *order) {
  for (int64_t j = 0; j < p; j++) order[j] = (int)j;
  for (int i = 0; i <= upto; i++) oem_shuffle(order, p, (uint32_t)i);
  return 0;
}

enum { EM_RR = 0, EM_BA = 1, EM_DE = 2, EM_ML = 3, EM_BB = 4, EM_BC = 5, EM_BCPI = 6, EM_BL = 7, EM_EN = 8, EM_LASSO = 9 };

/* dot of two n-vectors, one of them the residual */
static ACC_T oem_dot_ey(const E_T *e, const float *y, int64_t n) {
  ACC_T s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t i = 0;
  for (; i + 8 <= n; i += 8)
    for (int l = 0; l < 8; l++) s[l] += (ACC_T)e[i + l] * (ACC_T)y[i + l];
  for (int l = 0; i < n; i++, l++) s[l] += (ACC_T)e[i] * (ACC_T)y[i];
  return red8(s);
}

/* Members and reference lines: emRR :308-354, emBA :80-128, emDE :250-305, emML :463-521, emBB :131-187, emBC :190-247,
 * emBCpi :1502-1545 (natural marker order: no shuffle), emBL :357-397, emEN :400-460, lasso :1463-1500 (natural order).
 * par: Pi for emBB / emBC / emBCpi (reference default 0.75), alpha for emBL / emEN (0.02); ignored otherwise.
 * o_d: p entries (emBB / emBC / emBCpi), o_vbvec: p entries (emBA / emBB / emDE Vb); untouched otherwise.
 * o_scal[6]: emRR {Va, Ve, h2}; emBA / emBB / emDE {0, Ve, h2}; emML {Vb, Ve, h2, Va}; emBC {Va, Ve, h2, Vg};
 * emBCpi {Va, Ve, h2, Vg, pi}; emBL {0, 0, h2}; emEN {Va*cxx, Ve, h2}; lasso {Lmb, 0, h2}.
 * maxit = 0: the reference's count (200 sweeps; 300 with the convergence test for emDE / emML / emEN).
 * D: emML's optional per-marker weights (NULL = none). */
int FN(oracle_em)(int model, const float *y, const float *X, int64_t n, int64_t p, int64_t ldx, float df, float R2, float par,
                  const float *D, int maxit_in, float *o_mu, float *o_b, float *o_d, float *o_hat, float *o_vbvec,
                  float *o_scal, int *o_iters) {
  float *xx = (float *)malloc(sizeof(float) * p), *vx = (float *)malloc(sizeof(float) * p);
  float *b = (float *)calloc(p, sizeof(float)), *bc = (float *)calloc(p, sizeof(float)), *d = (float *)calloc(p, sizeof(float));
  float *vbv = (float *)malloc(sizeof(float) * p), *Lmbv = (float *)malloc(sizeof(float) * p);
  int *order = (int *)malloc(sizeof(int) * p);
  E_T *e = (E_T *)malloc(sizeof(E_T) * n), *e1 = (E_T *)malloc(sizeof(E_T) * n), *e2 = (E_T *)malloc(sizeof(E_T) * n);
  if (!xx || !vx || !b || !bc || !d || !vbv || !Lmbv || !order || !e || !e1 || !e2) return 1;
  float sumvx;
  FN(oracle_stats)(X, n, p, ldx, xx, vx, &sumvx);                       /* xx, vx, vx.sum() */
  float MSx = sumvx;
  const float vy = v_fvar(y, n);
  float mu = v_mean(y, n);                                              /* :98, :256, :326, :478 */
  for (int64_t k = 0; k < n; k++) e[k] = (E_T)(y[k] - mu);              /* e = y.array()-mu (float) */
  for (int64_t j = 0; j < p; j++) order[j] = (int)j;
  float ve = 0, vb = 0, va = 0, Lmb = 0, Sb = 0, Se = 0, Sa = 0, Rho = 0, cxx = 0, h2 = 0;
  float Pi = par, Pi0 = 0, PriorPi = 0, alpha = par, Lmb1 = 0, Lmb2 = 0, Sy = 0, trAC22 = 0;
  const int conv = (model == EM_DE || model == EM_ML || model == EM_EN || model == EM_LASSO);
  const int shuffled = (model != EM_BCPI && model != EM_LASSO);
  float *yx = (model == EM_LASSO) ? (float *)calloc(p, sizeof(float)) : NULL;
  const int maxit = maxit_in > 0 ? maxit_in : (conv ? 300 : 200);        /* :81, :251, :309, :465, :401 */
  const float tol = (model == EM_DE) ? 10e-6f : (model == EM_EN) ? 10e-11f : 10e-8f;   /* :252, :402, :466 */
  if (model == EM_BA || model == EM_BB) {
    ve = 1;                                                               /* :84, :135 */
    for (int64_t j = 0; j < p; j++) { vbv[j] = 1.0f; Lmbv[j] = ve * (1.0f / vbv[j]); }   /* :87-88 */
    if (model == EM_BB) {
      if (Pi > 0.5f) Pi = 1 - Pi;                                         /* :141 */
      MSx = sumvx * Pi;                                                   /* :147 */
      Pi0 = (1 - Pi) / Pi;                                                /* :154 */
    }
    Sb = R2 * (df + 2) * vy / MSx;                                        /* :96, :148 */
    Se = (1 - R2) * (df + 2) * vy;                                        /* :97, :149 */
  } else if (model == EM_RR) {
    Lmb = MSx;                                                            /* :319 */
    Rho = MSx * (1 - R2) / R2;                                            /* :320 */
    ve = 0.5f * vy;                                                       /* :322 */
    vb = ve / MSx;                                                        /* :323 */
    Se = (1 - R2) * (df + 2) * vy;                                        /* :324 */
    Sb = R2 * (df + 2) * vy / MSx;                                        /* :325 */
  } else if (model == EM_DE) {
    for (int64_t j = 0; j < p; j++) if (xx[j] == 0) xx[j] = 0.1f;         /* :261 */
    cxx = MSx * (1 - R2) / R2;                                            /* :265 */
    for (int64_t j = 0; j < p; j++) Lmbv[j] = (float)p + cxx;             /* :269 */
  } else if (model == EM_ML) {
    Lmb = MSx;                                                            /* :486 */
  } else if (model == EM_BC || model == EM_BCPI) {
    if (Pi > 0.5f) Pi = 1 - Pi;                                           /* :197, :1508 */
    PriorPi = Pi;                                                         /* :1511 */
    MSx = sumvx * Pi * (1 - Pi);                                          /* :203, :1512 */
    Sa = R2 * (df + 2) * vy / MSx;                                        /* :204 */
    Se = (1 - R2) * (df + 2) * vy;                                        /* :205 */
    ve = Sa; va = Se; Lmb = ve / va;                                      /* :209-211 (sic) */
    Pi0 = (1 - Pi) / Pi;                                                  /* :213 */
  } else if (model == EM_BL) {
    { ACC_T sx = 0; for (int64_t j = 0; j < p; j++) sx += (ACC_T)xx[j]; cxx = (float)(sx / (ACC_T)p); }   /* xx.mean(), :368 */
    const float hh = R2;                                                  /* h2 = R2, :359 */
    Lmb1 = cxx * ((1 - hh) / hh) * alpha * 0.5f;                          /* :369 */
    Lmb2 = cxx * ((1 - hh) / hh) * (1 - alpha);                           /* :370 */
  } else if (model == EM_EN) {
    cxx = sumvx * (1 - R2) / R2;                                          /* :412 */
    Sy = sqrtf(vy);                                                       /* :414 */
    Lmb = cxx;                                                            /* :415 */
    Lmb1 = 0.5f * Lmb * alpha * Sy;                                       /* :416 */
    Lmb2 = Lmb * (1 - alpha);                                             /* :417 */
    for (int64_t k = 0; k < p; k++) trAC22 += 1.0f / (xx[k] + Lmb);       /* a plain float loop in the reference, :418-419 */
  } else if (model == EM_LASSO) {
    ACC_T sx = 0; for (int64_t j = 0; j < p; j++) sx += (ACC_T)xx[j];
    Lmb = (float)(sx / (ACC_T)p) / (float)p;                              /* Lmb = xx.mean()/p, :1472 */
  }
  int numit = 0;
  for (int i = 0; i < maxit; i++) {
    if (conv) memcpy(bc, b, sizeof(float) * p);                           /* bc = b, :276, :490, :424 */
    if (shuffled) oem_shuffle(order, p, (uint32_t)i);                     /* :103, :158, :217, :277, :331, :374, :427, :491 */
    const float Cc = -0.5f / sqrtf(ve);                                   /* :157, :216, :1522 (used by the soft-selection members) */
    for (int64_t jj = 0; jj < p; jj++) {
      const int64_t j = order[jj];
      const float *xj = X + j * ldx;
      const float b0 = b[j];
      float bnew;
      if (model == EM_LASSO) {
        v_axpy(e, xj, 0.0f - b0, n);                                      /* e += gen.col(j)*b[j], :1477 */
        const ACC_T yxa = v_dot_acc(xj, e, n);                            /* yx[j] = e.dot(gen.col(j)), :1478 */
        yx[j] = (float)yxa;
        if (yxa > 0) { bnew = (float)((yxa - (ACC_T)Lmb) / (ACC_T)xx[j]); if (bnew < 0) bnew = 0; }   /* :1480-1481 */
        else { bnew = (float)((yxa + (ACC_T)Lmb) / (ACC_T)xx[j]); if (bnew > 0) bnew = 0; }           /* :1483-1484 */
        v_axpy(e, xj, bnew, n);                                           /* e -= gen.col(j)*b[j], :1485 */
        b[j] = bnew;
        continue;
      }
      if (model == EM_BB || model == EM_BC || model == EM_BCPI) {
        const float den = xx[j] + ((model == EM_BB) ? Lmbv[j] : Lmb);
        const float b1 = draw_b1(xj, e, n, xx[j], b0, den, 0.0f, 0.0);    /* :161, :220, :1525 */
        v_axpy_to(e1, e, xj, b1 - b0, n);                                 /* e1 = e - gen.col(j)*(b1-b0) */
        v_axpy_to(e2, e, xj, 0.0f - b0, n);                               /* e2 = e - gen.col(j)*(0-b0) */
#ifdef ACC_WIDE
        const float dif = (float)(e_sqnorm_acc(e2, n) - e_sqnorm_acc(e1, n));
#else
        const float dif = e_sqnorm(e2, n) - e_sqnorm(e1, n);
#endif
        const float LR = Pi0 * f_exp(Cc * dif);                           /* :164, :223, :1528 */
        d[j] = 1.0f / (1.0f + LR);
        bnew = b1 * d[j];
        if (model == EM_BB) vbv[j] = (Sb + bnew * bnew) / (df + 1);       /* :167 */
      } else if (model == EM_EN || model == EM_BL) {
        /* OLS = gen.col(j).dot(e) + xx[j]*b0: kept in the accumulator type, like the conditional mean of draw_b1 */
        const ACC_T ols = v_dot_acc(xj, e, n) + (ACC_T)(xx[j] * b0);
        if (model == EM_EN) {
          const float den = Lmb2 + xx[j];
          if (ols > 0) { bnew = (float)((ols - (ACC_T)Lmb1) / (ACC_T)den); if (bnew < 0) bnew = 0; }    /* :434 */
          else { bnew = (float)((ols + (ACC_T)Lmb1) / (ACC_T)den); if (bnew > 0) bnew = 0; }            /* :436 */
        } else {
          const ACC_T half = (ACC_T)0.5f * ols / (ACC_T)(xx[j] + cxx);    /* Half_L2, :380 */
          const float den = Lmb2 + xx[j];
          if (ols > 0) { const ACC_T G = (ACC_T)0.5f * (ols - (ACC_T)Lmb1) / (ACC_T)den; bnew = (float)(G > 0 ? G + half : half); }   /* :382-383 */
          else { const ACC_T G = (ACC_T)0.5f * (ols + (ACC_T)Lmb1) / (ACC_T)den; bnew = (float)(G < 0 ? G + half : half); }           /* :385-386 */
        }
      } else {
        float den;
        if (model == EM_BA || model == EM_DE) den = xx[j] + Lmbv[j];      /* :107, :282 */
        else if (model == EM_ML && D) den = xx[j] + Lmb / D[j];           /* :496 */
        else den = xx[j] + Lmb;                                           /* :335, :498 */
        bnew = draw_b1(xj, e, n, xx[j], b0, den, 0.0f, 0.0);
      }
      const float db = bnew - b0;
      v_axpy(e, xj, db, n);                                               /* :108, :168, :227, :284, :336, :388, :440, :500 */
      b[j] = bnew;
      if (model == EM_BA) {
        vbv[j] = (Sb + b[j] * b[j]) / (df + 1);                           /* :110 */
        v_axpy(e, xj, db, n);                                             /* :111 -- the second, identical update is the reference's */
      }
    }
    ACC_T sb2 = 0, sd = 0;
    for (int64_t j = 0; j < p; j++) { sb2 += (ACC_T)b[j] * (ACC_T)b[j]; sd += (ACC_T)d[j]; }
    const float b2n = (float)sb2, dmean = (float)(sd / (ACC_T)p);         /* b.squaredNorm(), d.mean() */
    if (model == EM_BA || model == EM_BB) {
      ve = (e_sqnorm(e, n) + Se) / ((float)n + df);                       /* :113, :170 */
      for (int64_t j = 0; j < p; j++) Lmbv[j] = ve * (1.0f / vbv[j]);     /* :114, :171 */
    } else if (model == EM_RR) {
      vb = (b2n + Sb) / ((float)p + df);                                  /* :338 */
      ve = (e_sqnorm(e, n) + Se) / ((float)n + df);                       /* :339 */
      Lmb = sqrtf(Rho * ve / vb);                                         /* :340 */
    } else if (model == EM_BC) {
      ve = (e_sqnorm(e, n) + Se) / ((float)n + df);                       /* :229 */
      va = (b2n + Sa) / ((float)p + df) / (dmean - Pi);                   /* :230 */
      Lmb = ve / va;                                                      /* :231 */
    } else if (model == EM_BCPI) {
      Pi = ((1.0f - dmean) * (float)p + PriorPi * df) / ((float)p + df);  /* :1533 */
      Pi0 = (1.0f - Pi) / Pi;                                             /* :1534 */
      MSx = sumvx * Pi * (1.0f - Pi);                                     /* :1535 */
      Sa = R2 * (df + 2) * vy / MSx;                                      /* :1536 */
      ve = (e_sqnorm(e, n) + Se) / ((float)n + df);                       /* :1538 */
      va = (b2n + Sa) / ((float)p + df) / (dmean - Pi);                   /* :1539 */
      Lmb = ve / va;                                                      /* :1540 */
    }
    if (model == EM_LASSO) {
      float tmp = 0.0f;
      for (int64_t j = 0; j < p; j++) tmp += fabsf(yx[j]) - fabsf(b[j] * xx[j]);   /* a plain float loop, :1487-1488 */
      Lmb = 2.0f * tmp / (float)p;                                        /* :1489 */
      Lmb = 2.0f * sqrtf(fabsf(Lmb));                                     /* :1490 */
    }
    const float eM = e_mean(e, n);                                        /* :115-117, :286-288, :341-343, :502-504 ... */
    mu += eM;
    for (int64_t k = 0; k < n; k++) e[k] = e[k] - (E_T)eM;
    if (model == EM_DE) {
      ve = (float)oem_dot_ey(e, y, n) / (float)(n - 1);                   /* Ve = e.dot(y)/(n-1), :289 */
      for (int64_t j = 0; j < p; j++) {
        vbv[j] = b[j] * b[j] + ve / (xx[j] + Lmbv[j] + 0.0001f);          /* :290 */
        Lmbv[j] = sqrtf(cxx * ve / vbv[j]);                               /* :292 */
      }
    } else if (model == EM_ML) {
      ACC_T s1 = 0, s2 = 0;
      for (int64_t k = 0; k < n; k++) {
        const float ym = y[k] - mu;                                       /* (y.array()-mu) */
        s1 += (ACC_T)ym * (ACC_T)e[k];                                    /* :505 */
        const E_T dif = (E_T)ym - e[k];                                   /* (y-mu) - e */
        s2 += (ACC_T)ym * (ACC_T)dif;                                     /* :506 */
      }
      ve = (float)s1 / (float)n;
      vb = (float)s2 / (float)((float)n * MSx);
      Lmb = ve / vb;                                                      /* :507 */
    } else if (model == EM_EN) {
      ve = (float)oem_dot_ey(e, y, n) / (float)(n - 1);                   /* :445 */
      va = (b2n + trAC22 * ve) / (float)p;                                /* :446 */
      Lmb = ve / va;                                                      /* :447 */
      Lmb1 = 0.5f * Lmb * alpha * Sy;                                     /* :448 */
      Lmb2 = Lmb * (1 - alpha);                                           /* :449 */
    }
    ++numit;
    if (conv) {
      ACC_T c = 0; for (int64_t j = 0; j < p; j++) c += (ACC_T)fabsf(bc[j] - b[j]);
      if ((float)c < tol) break;                                          /* :295-296, :451-452, :509-510 */
    }
  }
  /* fit */
  if (model == EM_ML) {
    for (int64_t k = 0; k < n; k++) o_hat[k] = (float)((E_T)y[k] - e[k]); /* fit = y - e, :512 */
    h2 = vb * MSx / (vb * MSx + ve);                                      /* :513 */
  } else if (model == EM_LASSO) {
    for (int64_t k = 0; k < n; k++) o_hat[k] = (float)((E_T)y[k] - e[k]); /* fit = y - e, :1493 */
    h2 = 1.0f - ((float)oem_dot_ey(e, y, n) / (float)(n - 1)) / vy;       /* :1494 */
  } else {
    ACC_T *acc = (ACC_T *)calloc(n, sizeof(ACC_T));
    for (int64_t j = 0; j < p; j++) { const float *xj = X + j * ldx; ACC_T Bj = (ACC_T)b[j]; for (int64_t k = 0; k < n; k++) acc[k] += (ACC_T)xj[k] * Bj; }
    for (int64_t k = 0; k < n; k++) { float f = (float)acc[k]; o_hat[k] = f + mu; }   /* :120-121 */
    free(acc);
    if (model == EM_DE) {
      ACC_T sv = 0; for (int64_t j = 0; j < p; j++) sv += (ACC_T)vbv[j];
      h2 = (float)sv / ((float)sv + ve);                                  /* :304 */
    } else if (model == EM_BL) {
      float *ef = (float *)malloc(sizeof(float) * n);
      for (int64_t k = 0; k < n; k++) ef[k] = (float)e[k];
      h2 = 1 - v_fvar(ef, n) / vy;                                        /* :396 */
      free(ef);
    } else if (model == EM_EN) h2 = va * cxx / (va * cxx + ve);           /* :459 */
    else h2 = 1 - ve / vy;                                                /* :119, :178, :237, :344, :1542 */
  }
  *o_mu = mu; memcpy(o_b, b, sizeof(float) * p);
  if (o_d && (model == EM_BB || model == EM_BC || model == EM_BCPI)) memcpy(o_d, d, sizeof(float) * p);
  if (o_vbvec && (model == EM_BA || model == EM_DE || model == EM_BB)) memcpy(o_vbvec, vbv, sizeof(float) * p);
  for (int k = 0; k < 6; k++) o_scal[k] = 0;
  o_scal[1] = ve; o_scal[2] = h2;
  if (model == EM_RR || model == EM_ML) o_scal[0] = vb;
  if (model == EM_ML) o_scal[3] = vb * MSx;                               /* Va = vb*MSx, :519 */
  if (model == EM_BC || model == EM_BCPI) { o_scal[0] = va; o_scal[3] = va * MSx; }   /* Va, Vg = va*MSx, :243, :1547 */
  if (model == EM_BCPI) o_scal[4] = Pi;
  if (model == EM_EN) o_scal[0] = va * cxx;                               /* :457 */
  if (model == EM_BL) o_scal[1] = 0;
  if (model == EM_LASSO) { o_scal[0] = Lmb; o_scal[1] = 0; }
  free(yx);
  *o_iters = numit;
  free(xx); free(vx); free(b); free(bc); free(d); free(vbv); free(Lmbv); free(order); free(e); free(e1); free(e2);
  return 0;
}
