/*
 * oracle/bwgr_rng.h -- TEST INFRASTRUCTURE ONLY (CPU oracle), never linked into the product.
 *
 * Counter-based variate generator of the build (the "RNG contract" of DESIGN.md section 5).
 * It replaces the reference's serial R nmath stream
 *   R::rnorm / R::rchisq / R::rbinom   (/root/reference/src/Rcpp20260726ai.cpp:20-21,28,
 *                                       615,617,620,622,670,675,678,680,683,685 ...)
 *   rchisq / rnorm in R                (/root/reference/R/wgr.R:100,109,113,117,121,125)
 * whose source (R nmath) is not under /root/reference and whose stream cannot be reproduced by
 * per-marker parallel draws.  Same *distributions*, different *stream*: every variate is a pure
 * function of (seed, iteration, marker, purpose, attempt).
 *
 * Philox4x32-10 is restated from the published algorithm (Salmon, Moraes, Dror, Shaw, SC'11,
 * "Parallel random numbers: as easy as 1, 2, 3"; Random123 v1.x) and pinned by its three
 * published known-answer vectors in tests/test_oracle_rng.py.
 *
 * The HIP product carries its own, separately written implementation of this contract
 * (bwgr_amd/csrc/rng.hip.h); tests compare the two variate by variate.
 */
#ifndef BWGR_ORACLE_RNG_H
#define BWGR_ORACLE_RNG_H
#include <stdint.h>
#include <math.h>
#include "bwgr_rstream.h"   /* ORNG_RSTREAM: an UNVERIFIED restatement of R's serial stream, for users who have R (see that header) */

/* purposes (third counter word) */
enum {
  ORNG_Z1 = 0,      /* N(0,1) behind the "in-model" effect b1          */
  ORNG_Z2 = 1,      /* N(0,1) behind the alternative effect b2         */
  ORNG_U  = 2,      /* U(0,1) behind the Bernoulli inclusion draw      */
  ORNG_CHI = 3,     /* chi-square behind the per-marker variance vb_j  */
  ORNG_G_MU = 16,   /* N(0,1) behind the intercept draw                */
  ORNG_G_VE = 17,   /* chi-square behind the residual variance         */
  ORNG_G_VB = 18,   /* chi-square behind the common marker variance    */
  ORNG_G_VK = 19,   /* chi-square behind the polygenic variance (wgr)  */
  ORNG_BAG = 20     /* uniforms behind wgr's row resampling sample(n, n*bag, rp) */
};
#define ORNG_GLOBAL_MARKER 0xFFFFFFFFu  /* first counter word of per-iteration scalars */

/* rng_mode */
#define ORNG_PHILOX 0
#define ORNG_DEGENERATE 1  /* z = 0, chi-square = its mean, u = 0.5: turns the sampler into
                              deterministic Gauss-Seidel for the analytic invariants */
#define ORNG_RSTREAM 2     /* R's own serial stream (Mersenne-Twister, inversion normals, nmath's rgamma / rbinom), drawn in the order the
                              oracle's code -- and so the reference's -- asks for variates; counters are ignored; seed it with
                              oracle_rstream_seed() = set.seed() before the call.  UNVERIFIED: no R in the build image (bwgr_rstream.h) */

typedef struct { uint64_t seed; int mode; } orng_t;

static inline void orng_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform strictly inside (0,1) from two 32-bit words */
static inline double orng_u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) / 9007199254740992.0;
}

static inline void orng_block(const orng_t *g, uint32_t marker, uint32_t iter, uint32_t purpose,
                              uint32_t k, uint32_t out[4]) {
  uint32_t ctr[4] = { marker, iter, purpose, k };
  uint32_t key[2] = { (uint32_t)g->seed, (uint32_t)(g->seed >> 32) };
  orng_philox4x32_10(ctr, key, out);
}

static inline double orng_uniform(const orng_t *g, uint32_t marker, uint32_t iter, uint32_t purpose, uint32_t k) {
  if (g->mode == ORNG_DEGENERATE) return 0.5;
  if (g->mode == ORNG_RSTREAM) return ors_unif_rand();
  uint32_t x[4]; orng_block(g, marker, iter, purpose, k, x);
  return orng_u53(x[0], x[1]);
}

/* Box-Muller (cosine branch), one normal per Philox block */
static inline double orng_normal(const orng_t *g, uint32_t marker, uint32_t iter, uint32_t purpose, uint32_t k) {
  if (g->mode == ORNG_DEGENERATE) return 0.0;
  if (g->mode == ORNG_RSTREAM) return ors_norm_rand();
  uint32_t x[4]; orng_block(g, marker, iter, purpose, k, x);
  double u1 = orng_u53(x[0], x[1]);
  double u2 = orng_u53(x[2], x[3]);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

/* Gamma(a, scale 1), Marsaglia & Tsang (2000) without the squeeze step; a < 1 by the
 * Gamma(a+1) * U^(1/a) boost.  Attempt t uses blocks k = 2t (normal) and 2t+1 (uniform);
 * the boost uniform is block k = 0xFFFFFFFF. */
static inline double orng_gamma(const orng_t *g, double a, uint32_t marker, uint32_t iter, uint32_t purpose) {
  if (g->mode == ORNG_DEGENERATE) return a;
  double boost = 1.0;
  if (a < 1.0) {
    double u = orng_uniform(g, marker, iter, purpose, 0xFFFFFFFFu);
    boost = pow(u, 1.0 / a);
    a += 1.0;
  }
  const double d = a - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  for (uint32_t t = 0; t < 0x7FFFFFFFu; t++) {
    double x = orng_normal(g, marker, iter, purpose, 2u * t);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = orng_uniform(g, marker, iter, purpose, 2u * t + 1u);
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return boost * d * v;
  }
  return boost * d;
}

static inline double orng_chisq(const orng_t *g, double nu, uint32_t marker, uint32_t iter, uint32_t purpose) {
  if (g->mode == ORNG_RSTREAM) return ors_rchisq(nu);   /* R::rchisq(nu) = rgamma(nu / 2, 2) */
  return 2.0 * orng_gamma(g, 0.5 * nu, marker, iter, purpose);
}

/* the Bernoulli inclusion draw: `R::rbinom(1, pj) == 1` (/root/reference/src/Rcpp20260726ai.cpp:28, :675 ...).  Contract: u < pj on the marker's
 * ORNG_U uniform (a NaN pj compares false, as rbinom's NaN == 1 does); R stream: nmath's inversion rbinom, one uniform */
static inline int orng_bernoulli(const orng_t *g, double pj, uint32_t marker, uint32_t iter) {
  if (g->mode == ORNG_RSTREAM) return ors_rbinom1(pj);
  return orng_uniform(g, marker, iter, ORNG_U, 0) < pj;
}

#endif
