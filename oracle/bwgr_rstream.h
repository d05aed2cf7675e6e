/*
 * oracle/bwgr_rstream.h -- TEST INFRASTRUCTURE ONLY (CPU oracle), never linked into the product.
 *
 * UNVERIFIED restatement of R's default random stream, so that someone who HAS R can compare the oracle with real bWGR
 * (tools/make_r_fixtures.R writes the fixtures, tests/test_r_fixtures.py reads them).  R and its nmath are not in the build
 * image, nothing here could be run against them: every function below is written from the published algorithms as R
 * implements them, and is labelled so.  With rng_mode = ORNG_RSTREAM the oracle's draws no longer depend on (marker,
 * iteration, purpose): they are taken one after the other from this serial stream, in the order the reference's code
 * consumes it under Rcpp::RNGScope (/root/reference/src/RcppExports.cpp:19):
 *   R::rnorm(mu, sd)   /root/reference/src/Rcpp20260726ai.cpp:20-21, 615, 670, 678, 683 ...  = mu + sd * norm_rand()
 *   R::rchisq(df)      :617, :680, :685 ...                                                  = rgamma(df / 2, 2)
 *   R::rbinom(1, p)    :28, :675 ...                                                         (one uniform, inversion)
 *   rnorm / rchisq in R  /root/reference/R/wgr.R:100,109,113,117,121,125                      (vectorised: element by element)
 * wgr's bagging (sample(), R/wgr.R:68) is NOT covered: R's sample() draws by rejection from unif_rand bits.
 *
 * What is restated (defaults of R >= 3.6: RNGkind("Mersenne-Twister", "Inversion", "Rejection")):
 *   set.seed(s)     RNG_Init: s <- 69069 s + 1 fifty times, then 625 more for the seed table; dummy[0] = mti = 624
 *   unif_rand()     MT19937 genrand (Matsumoto & Nishimura 1998), * 2^-32, fixed up into (0, 1)
 *   norm_rand()     INVERSION: u = unif_rand(); u = (int)(2^27 u) + unif_rand(); qnorm(u / 2^27)
 *   qnorm           Wichura's AS 241 (PPND16)
 *   exp_rand()      Ahrens & Dieter 1972 (algorithm SA)
 *   rgamma(a, s)    a >= 1: Ahrens & Dieter 1982 (GD);  a < 1: Ahrens & Dieter 1974 (GS)
 *   rbinom(1, p)    the inversion branch (n p < 30) of Kachitvichyanukul & Schmeiser's BTPE driver as R calls it
 */
#ifndef BWGR_ORACLE_RSTREAM_H
#define BWGR_ORACLE_RSTREAM_H
#include <stdint.h>
#include <math.h>

#define ORS_N 624
#define ORS_M 397
typedef struct {
  uint32_t mt[ORS_N];
  int mti;
  /* rgamma keeps the quantities of its last shape parameter (static variables in nmath) */
  double g_aa, g_aaa, g_s, g_s2, g_d, g_q0, g_b, g_si, g_c;
} ors_state;

static ors_state ors_g;   /* one global stream, like R's (the oracle is single-threaded) */

static void ors_set_seed(uint32_t seed) {
  for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
  uint32_t dummy0 = 0;
  for (int j = 0; j < ORS_N + 1; j++) {
    seed = 69069u * seed + 1u;
    if (j == 0) dummy0 = seed; else ors_g.mt[j - 1] = seed;
  }
  (void)dummy0;
  ors_g.mti = ORS_N;          /* FixupSeeds: dummy[0] = 624, so the first draw regenerates the table */
  ors_g.g_aa = 0.0; ors_g.g_aaa = 0.0;
}

static double ors_unif_rand(void) {
  static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
  uint32_t y;
  uint32_t *mt = ors_g.mt;
  if (ors_g.mti >= ORS_N) {
    int kk;
    for (kk = 0; kk < ORS_N - ORS_M; kk++) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + ORS_M] ^ (y >> 1) ^ mag01[y & 0x1u];
    }
    for (; kk < ORS_N - 1; kk++) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (ORS_M - ORS_N)] ^ (y >> 1) ^ mag01[y & 0x1u];
    }
    y = (mt[ORS_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[ORS_N - 1] = mt[ORS_M - 1] ^ (y >> 1) ^ mag01[y & 0x1u];
    ors_g.mti = 0;
  }
  y = mt[ors_g.mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  double v = (double)y * 2.3283064365386963e-10;   /* [0, 1) */
  /* fixup: strictly inside (0, 1) */
  const double i2_32m1 = 2.328306437080797e-10;
  if (v <= 0.0) return 0.5 * i2_32m1;
  if (1.0 - v <= 0.0) return 1.0 - 0.5 * i2_32m1;
  return v;
}

/* AS 241 (PPND16), lower tail, not log */
static double ors_qnorm(double p) {
  double q = p - 0.5, r, val;
  if (fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r + 45921.953931549871457) * r +
                   13731.693765509461125) * r + 1971.5909503065514427) * r + 133.14166789178437745) * r + 3.387132872796366608) /
          (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r + 21213.794301586595867) * r +
              5394.1960214247511077) * r + 687.1870074920579083) * r + 42.313330701600911252) * r + 1.0);
    return val;
  }
  r = (q < 0) ? p : 1.0 - p;
  r = sqrt(-log(r));
  if (r <= 5.0) {
    r -= 1.6;
    val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r + 0.24178072517745061177) * r + 1.27045825245236838258) * r +
              3.64784832476320460504) * r + 5.7694972214606914055) * r + 4.6303378461565452959) * r + 1.42343711074968357734) /
          (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r + 0.14810397642748007459) * r +
              0.68976733498510000455) * r + 1.6763848301838038494) * r + 2.05319162663775882187) * r + 1.0);
  } else {
    r -= 5.0;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r + 0.026532189526576123093) * r +
              0.29656057182850489123) * r + 1.7848265399172913358) * r + 5.4637849111641143699) * r + 6.6579046435011037772) /
          (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r +
              0.0148753612908506148525) * r + 0.13692988092273580531) * r + 0.59983224644127559443) * r + 1.0);
  }
  return (q < 0.0) ? -val : val;
}

static double ors_norm_rand(void) {
  const double BIG = 134217728.0;   /* 2^27 */
  double u = ors_unif_rand();
  u = (double)(int)(BIG * u) + ors_unif_rand();
  return ors_qnorm(u / BIG);
}

static double ors_exp_rand(void) {
  /* q[k-1] = sum_{i=1..k} log(2)^i / i! */
  static const double q[] = {0.6931471805599453, 0.9333736875190459, 0.9888777961838675, 0.9984589039328338, 0.9998292811061389,
                             0.9999833164100727, 0.9999985691438767, 0.9999998906925558, 0.9999999924734159, 0.9999999995283275,
                             0.9999999999728814, 0.9999999999985598, 0.9999999999999289, 0.9999999999999968, 0.9999999999999999,
                             1.0000000000000000};
  double a = 0.0;
  double u = ors_unif_rand();
  while (u <= 0.0 || u >= 1.0) u = ors_unif_rand();
  for (;;) {
    u += u;
    if (u > 1.0) break;
    a += q[0];
  }
  u -= 1.0;
  if (u <= q[0]) return a + u;
  int i = 0;
  double ustar = ors_unif_rand(), umin = ustar;
  do {
    ustar = ors_unif_rand();
    if (umin > ustar) umin = ustar;
    i++;
  } while (u > q[i]);
  return a + umin * q[0];
}

static double ors_rgamma(double a, double scale) {
  const double sqrt32 = 5.656854, exp_m1 = 0.36787944117144233;
  const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191, q4 = 0.00144121, q5 = -7.388e-5, q6 = 2.4511e-4, q7 = 2.424e-4;
  const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062, a4 = -0.1662921, a5 = 0.1423657, a6 = -0.1367177, a7 = 0.1233795;
  ors_state *S = &ors_g;
  double e, p, q, r, t, u, v, w, x, ret_val;
  if (!(a > 0.0) || !(scale > 0.0)) return (a == 0.0 || scale == 0.0) ? 0.0 : NAN;
  if (a < 1.0) {   /* GS */
    e = 1.0 + exp_m1 * a;
    for (;;) {
      p = e * ors_unif_rand();
      if (p >= 1.0) {
        x = -log((e - p) / a);
        if (ors_exp_rand() >= (1.0 - a) * log(x)) break;
      } else {
        x = exp(log(p) / a);
        if (ors_exp_rand() >= x) break;
      }
    }
    return scale * x;
  }
  /* GD */
  if (a != S->g_aa) { S->g_aa = a; S->g_s2 = a - 0.5; S->g_s = sqrt(S->g_s2); S->g_d = sqrt32 - S->g_s * 12.0; }
  t = ors_norm_rand();
  x = S->g_s + 0.5 * t;
  ret_val = x * x;
  if (t >= 0.0) return scale * ret_val;
  u = ors_unif_rand();
  if (S->g_d * u <= t * t * t) return scale * ret_val;
  if (a != S->g_aaa) {
    S->g_aaa = a;
    r = 1.0 / a;
    S->g_q0 = ((((((q7 * r + q6) * r + q5) * r + q4) * r + q3) * r + q2) * r + q1) * r;
    if (a <= 3.686) { S->g_b = 0.463 + S->g_s + 0.178 * S->g_s2; S->g_si = 1.235; S->g_c = 0.195 / S->g_s - 0.079 + 0.16 * S->g_s; }
    else if (a <= 13.022) { S->g_b = 1.654 + 0.0076 * S->g_s2; S->g_si = 1.68 / S->g_s + 0.275; S->g_c = 0.062 / S->g_s + 0.024; }
    else { S->g_b = 1.77; S->g_si = 0.75; S->g_c = 0.1515 / S->g_s; }
  }
  if (x > 0.0) {
    v = t / (S->g_s + S->g_s);
    if (fabs(v) <= 0.25) q = S->g_q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
    else q = S->g_q0 - S->g_s * t + 0.25 * t * t + (S->g_s2 + S->g_s2) * log(1.0 + v);
    if (log(1.0 - u) <= q) return scale * ret_val;
  }
  for (;;) {
    e = ors_exp_rand();
    u = ors_unif_rand();
    u = u + u - 1.0;
    t = (u < 0.0) ? S->g_b - S->g_si * e : S->g_b + S->g_si * e;
    if (t >= -0.71874483771719) {
      v = t / (S->g_s + S->g_s);
      if (fabs(v) <= 0.25) q = S->g_q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
      else q = S->g_q0 - S->g_s * t + 0.25 * t * t + (S->g_s2 + S->g_s2) * log(1.0 + v);
      if (q > 0.0) {
        w = expm1(q);
        if (S->g_c * fabs(u) <= w * exp(e - 0.5 * t * t)) break;
      }
    }
  }
  x = S->g_s + 0.5 * t;
  return scale * x * x;
}

static double ors_rchisq(double df) { return (!(df >= 0.0) || isinf(df)) ? NAN : ors_rgamma(df / 2.0, 2.0); }

/* rbinom(1, pp) == 1 ?   (NaN, pp < 0, pp > 1: R returns NaN with a warning and the reference's `== 1` is false) */
static int ors_rbinom1(double pp) {
  if (!(pp >= 0.0) || !(pp <= 1.0)) return 0;
  if (pp == 0.0) return 0;
  if (pp == 1.0) return 1;
  const double p = (pp < 1.0 - pp) ? pp : 1.0 - pp, qq = 1.0 - p, r = p / qq, g = r * 2.0;   /* n = 1: g = r (n + 1) */
  int ix;
  for (;;) {
    ix = 0;
    double f = qq;              /* qn = q^n */
    double u = ors_unif_rand();
    int done = 0;
    for (;;) {
      if (u < f) { done = 1; break; }
      if (ix > 110) break;
      u -= f;
      ix++;
      f *= (g / ix - r);
    }
    if (done) break;
  }
  if (pp > 0.5) ix = 1 - ix;
  return ix == 1;
}

#endif
