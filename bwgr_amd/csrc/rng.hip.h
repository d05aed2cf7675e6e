// bwgr_amd/csrc/rng.hip.h -- device side of the RNG contract.
//
// Replaces the reference's serial R nmath draws (R::rnorm / R::rchisq / R::rbinom,
// src/Rcpp20260726ai.cpp:20-21,28,615,617,670,675,678,680 ...) by variates that are pure functions of
// (seed, iteration, marker, purpose, attempt), so that any lane can produce the draw of any marker.
// Philox4x32-10 (Salmon et al., SC'11); all transforms in fp64 so that the float the sampler finally
// rounds to does not depend on which libm evaluated log/cos.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bwgr {

enum : uint32_t {
  RNG_Z1 = 0, RNG_Z2 = 1, RNG_U = 2, RNG_CHI = 3,
  RNG_G_MU = 16, RNG_G_VE = 17, RNG_G_VB = 18, RNG_G_VK = 19, RNG_BAG = 20
};
static constexpr uint32_t RNG_GLOBAL_MARKER = 0xFFFFFFFFu;

struct Rng {
  uint32_t k0, k1;
  int degenerate;
};

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                             uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
  const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
  c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double rng_uniform(const Rng &g, uint32_t marker, uint32_t iter, uint32_t purpose, uint32_t k) {
  if (g.degenerate) return 0.5;
  const uint4 x = philox4x32_10(marker, iter, purpose, k, g.k0, g.k1);
  return u53(x.x, x.y);
}

__device__ __forceinline__ double rng_normal(const Rng &g, uint32_t marker, uint32_t iter, uint32_t purpose, uint32_t k) {
  if (g.degenerate) return 0.0;
  const uint4 x = philox4x32_10(marker, iter, purpose, k, g.k0, g.k1);
  const double u1 = u53(x.x, x.y), u2 = u53(x.z, x.w);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

// Gamma(a,1): Marsaglia-Tsang without squeeze; attempt t draws blocks 2t (normal), 2t+1 (uniform)
__device__ inline double rng_gamma(const Rng &g, double a, uint32_t marker, uint32_t iter, uint32_t purpose) {
  if (g.degenerate) return a;
  double boost = 1.0;
  if (a < 1.0) {
    const double u = rng_uniform(g, marker, iter, purpose, 0xFFFFFFFFu);
    boost = pow(u, 1.0 / a);
    a += 1.0;
  }
  const double d = a - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  for (uint32_t t = 0; t < 0x7FFFFFFFu; ++t) {
    const double x = rng_normal(g, marker, iter, purpose, 2u * t);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double u = rng_uniform(g, marker, iter, purpose, 2u * t + 1u);
    if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return boost * d * v;
  }
  return boost * d;
}

__device__ __forceinline__ double rng_chisq(const Rng &g, double nu, uint32_t marker, uint32_t iter, uint32_t purpose) {
  return 2.0 * rng_gamma(g, 0.5 * nu, marker, iter, purpose);
}

}  // namespace bwgr
