// bwgr_amd/csrc/sweep.hip.h -- the exact blocked Gibbs sweep for gfx950.
//
// What it computes: one pass of the per-marker recurrence of the reference
//   KMUP                      src/Rcpp20260726ai.cpp:18-36
//   BayesA/B/C/L/RR/Cpi/Dpi   src/Rcpp20260726ai.cpp:613-619, 668-682, 728-742, 786-793, 833-838,
//                             885-901, 953-969
// over markers [j_begin, j_end), in marker order, as the same Markov chain.
//
// How: markers are taken in blocks of m <= 128.  For a block B the reference's
//   r_j = X_j . e_current                      (e_current already holds the updates of markers < j)
// is evaluated as  r_j = X_j . e_blockstart - sum_{k<j, k in B} G_jk * delta_k  with G = X_B' X_B
// precomputed once per panel (exact integers for int8 genotypes), and |e - x d|^2 differences as
// 2 r (d1-d2) + G_jj (d2^2 - d1^2).  The residual is carried in fp64 (the reference's is float): with the
// n-vector round-off gone the blocked form and the marker-by-marker form are the same numbers, which is
// what lets the CPU oracle's wide flavour pin this kernel to ~1e-7 over whole chains.
//
// Work split: K workgroups, workgroup w owns rows [w*R, (w+1)*R) of every column (its slab of e
// lives in LDS for the whole launch; X is read once from HBM into an LDS tile).  Per block each
// workgroup forms its slab's partial dots (fp64), the K partial vectors are all-gathered through
// HBM with write-through stores + per-workgroup epoch flags, and every workgroup then replays the
// identical in-block recurrence (one wave, lane = marker, r in registers, Gram block in LDS), so
// no second exchange is needed before the slab update.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.hip.h"

namespace bwgr {

static constexpr int SW_THREADS = 512;
static constexpr int SW_MAXM = 128;            // markers per block (2 lane-groups of the recurrence wave)
static constexpr int SW_FLAG_STRIDE = 32;      // uint32 words between flags (128 B)
static constexpr uint64_t SW_TIMEOUT_TICKS = 400000000ull;  // 4 s of the 100 MHz wall clock

// per-launch flags
enum : int {
  SWF_SELECT = 1,     // Bernoulli inclusion step (KMUP pi>0, B, C, Cpi, Dpi)
  SWF_ALT_B2 = 2,     // the alternative residual uses b2 (KMUP, Dpi); otherwise 0 (B, C, Cpi)
  SWF_MH = 4,         // BayesDpi acceptance  min(1,(1-pi) exp(C(|e1|^2-|e2|^2)))
  SWF_LAM_VEC = 8,    // per-marker lambda array (else the common scalar)
  SWF_VB_VEC = 16     // per-marker variance draw vb_j = (Sb + b_j^2)/chisq(df+1)
};

// scalars produced on the device by the per-iteration tail kernel (or filled by the host for KMUP)
struct ChainScalars {
  float ve, vb, lam, pi;        // current residual / common marker variance, common lambda, pi
  float Sb, Se, C, odds;        // priors; C = -0.5/sqrt(ve); odds = pi/(1-pi) at chain start
  float mu, dfp1, vy, MSx;
  float MU, VE, VBs, Pi;        // posterior sums
  double sum_d, sum_b2;         // written by the sweep (marker order, fp64)
  uint32_t error;               // non-zero: an exchange gave up
  uint32_t pad;
};

struct SweepArgs {
  const void *X; int64_t ld;    // column-major, ld = K*R rows (zero padded)
  const void *gram;             // [nblocks][m][m]
  int n, p, m, K, R;
  int blk_begin, blk_end;
  int flags;
  double *e;                    // residual, ld entries (padding rows stay 0)
  float *b, *d, *vb;
  const float *xx, *lam;
  ChainScalars *sc;
  uint32_t iter;
  Rng rng;
  double *xpart;                // [2][K][SW_MAXM]
  uint32_t *xflags;             // [K*SW_FLAG_STRIDE] epochs, then the abort word
};

template <typename XT> struct XTraits;
template <> struct XTraits<int8_t> { using GT = int32_t; static constexpr int PER16 = 16; };
template <> struct XTraits<float> { using GT = double; static constexpr int PER16 = 4; };

// padded LDS row length (elements) so that consecutive markers start on odd multiples of 16 B
template <typename XT> __host__ __device__ inline int tile_rp(int R) { return R + XTraits<XT>::PER16; }

struct StageBuf {               // per-block per-marker constants, lane = marker
  float b0[SW_MAXM], xxb0[SW_MAXM], den[SW_MAXM], b2[SW_MAXM];
  double sdz1[SW_MAXM], u[SW_MAXM], chi[SW_MAXM];
};

template <typename XT> __host__ __device__ inline size_t sweep_lds_bytes(int m, int R) {
  size_t s = 0;
  s += (size_t)2 * m * tile_rp<XT>(R) * sizeof(XT);            // tiles
  s = (s + 15) & ~(size_t)15;
  s += (size_t)m * m * sizeof(typename XTraits<XT>::GT);       // gram block
  s = (s + 15) & ~(size_t)15;
  s += (size_t)R * sizeof(double);                             // e slab
  s += 2 * sizeof(StageBuf);
  s += (size_t)(SW_THREADS / 64) * SW_MAXM * sizeof(double);   // row-group partials (up to 8 groups)
  s += SW_MAXM * sizeof(double);                               // r0
  s += 3 * SW_MAXM * sizeof(float);                            // delta, bnew, dnew
  s += 64;                                                     // control words
  return s;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ void st_agent_u64(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent_f64(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// cooperative copy of one block's slab tile X[rows of wg, markers j0..j0+mB) into LDS
template <typename XT>
__device__ inline void load_tile(XT *tile, const XT *X, int64_t ld, int row0, int R, int Rp, int j0, int mB,
                                 int tid0, int nthreads) {
  constexpr int PER = XTraits<XT>::PER16;
  const int cpr = R / PER;  // 16-byte chunks per marker
  const int total = mB * cpr;
  for (int c = tid0; c < total; c += nthreads) {
    const int jj = c / cpr, ii = c - jj * cpr;
    const uint4 v = *reinterpret_cast<const uint4 *>(X + (int64_t)(j0 + jj) * ld + row0 + ii * PER);
    *reinterpret_cast<uint4 *>(tile + (size_t)jj * Rp + ii * PER) = v;
  }
}

// per-marker constants of one block (thread = marker)
__device__ inline void stage_marker(StageBuf &st, int t, int j, const SweepArgs &a, float ve, float lam_common,
                                    float dfp1) {
  const float b0 = a.b[j];
  const float xxj = a.xx[j];
  const float lamj = (a.flags & SWF_LAM_VEC) ? a.lam[j] : lam_common;
  const float den = xxj + lamj;
  const float sd = sqrtf(ve / den);
  const uint32_t mk = (uint32_t)j;
  st.b0[t] = b0;
  st.xxb0[t] = xxj * b0;
  st.den[t] = den;
  st.sdz1[t] = (double)sd * rng_normal(a.rng, mk, a.iter, RNG_Z1, 0);
  const bool need_b2 = (a.flags & SWF_SELECT) != 0;
  st.b2[t] = need_b2 ? (float)((double)0.0f + (double)sd * rng_normal(a.rng, mk, a.iter, RNG_Z2, 0)) : 0.0f;
  st.u[t] = need_b2 ? rng_uniform(a.rng, mk, a.iter, RNG_U, 0) : 0.0;
  st.chi[t] = (a.flags & SWF_VB_VEC) ? rng_chisq(a.rng, (double)dfp1, mk, a.iter, RNG_CHI) : 1.0;
}

template <typename XT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep(const SweepArgs a) {
  using GT = typename XTraits<XT>::GT;
  constexpr int PER = XTraits<XT>::PER16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int m = a.m, R = a.R, K = a.K;
  const int Rp = tile_rp<XT>(R);
  const int row0 = wg * R;

  // ---- LDS carve (every offset a multiple of 16 B) ----
  size_t off = 0;
  XT *tile0 = reinterpret_cast<XT *>(smem);
  XT *tile1 = tile0 + (size_t)m * Rp;
  off = ((size_t)2 * m * Rp * sizeof(XT) + 15) & ~(size_t)15;
  GT *gram_s = reinterpret_cast<GT *>(smem + off);
  off = (off + (size_t)m * m * sizeof(GT) + 15) & ~(size_t)15;
  double *e_s = reinterpret_cast<double *>(smem + off); off += (size_t)R * sizeof(double);
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 2 * sizeof(StageBuf);
  double *part_s = reinterpret_cast<double *>(smem + off); off += (size_t)(SW_THREADS / 64) * SW_MAXM * sizeof(double);
  double *r0_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  float *delta_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  float *bnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  float *dnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  volatile int *ctrl_s = reinterpret_cast<volatile int *>(smem + off);

  const XT *X = reinterpret_cast<const XT *>(a.X);
  const GT *gram = reinterpret_cast<const GT *>(a.gram);

  // scalars of this iteration
  const float ve = a.sc->ve, lam_common = a.sc->lam, Cc = a.sc->C, odds = a.sc->odds;
  const float one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb, dfp1 = a.sc->dfp1;

  const int nb = a.blk_end - a.blk_begin;
  const int mpad = (m <= 64) ? 64 : 128;
  const int ngroups = SW_THREADS / mpad;     // row groups of the dot phase
  const int rows_per_group = R / ngroups;    // R is a multiple of 128, so a multiple of 16

  for (int i = tid; i < R; i += SW_THREADS) e_s[i] = a.e[row0 + i];
  if (tid == 0) ctrl_s[0] = 1;

  // prologue: tile, constants and Gram block of the first block
  {
    const int j0 = a.blk_begin * m;
    const int mB = min(m, a.p - j0);
    load_tile<XT>(tile0, X, a.ld, row0, R, Rp, j0, mB, tid, SW_THREADS);
    if (tid < mB) stage_marker(stage[0], tid, j0 + tid, a, ve, lam_common, dfp1);
    const GT *gsrc = gram + (size_t)a.blk_begin * m * m;
    for (int c = tid; c < m * m; c += SW_THREADS) gram_s[c] = gsrc[c];
  }
  double sum_d = 0.0, sum_b2 = 0.0;  // meaningful in wave 0 only

  for (int s = 0; s < nb; ++s) {
    const int blk = a.blk_begin + s;
    const int j0 = blk * m;
    const int mB = min(m, a.p - j0);
    const int buf = s & 1;
    XT *tile = buf ? tile1 : tile0;
    XT *tile_next = buf ? tile0 : tile1;
    StageBuf &st = stage[buf];
    __syncthreads();  // tile, stage[buf], gram_s, e_s of this block are in place

    // ---- slab partial dots: lane = marker, row groups across the workgroup ----
    {
      const int g = tid / mpad, t = tid - g * mpad;
      if (t < mB) {
        const int r_lo = g * rows_per_group;
        const XT *tp = tile + (size_t)t * Rp + r_lo;
        const double *ep = e_s + r_lo;
        double acc = 0.0;
        for (int c = 0; c < rows_per_group; c += 16) {
          if constexpr (PER == 16) {
            const uint4 xv = *reinterpret_cast<const uint4 *>(tp + c);
            const uint32_t w[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double2 ea = *reinterpret_cast<const double2 *>(ep + c + 4 * q);
              const double2 eb = *reinterpret_cast<const double2 *>(ep + c + 4 * q + 2);
              const int x0 = (int)(int8_t)(w[q] & 0xFF), x1 = (int)(int8_t)((w[q] >> 8) & 0xFF);
              const int x2 = (int)(int8_t)((w[q] >> 16) & 0xFF), x3 = (int)(int8_t)(w[q] >> 24);
              acc = fma((double)x0, ea.x, acc);
              acc = fma((double)x1, ea.y, acc);
              acc = fma((double)x2, eb.x, acc);
              acc = fma((double)x3, eb.y, acc);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 xv = *reinterpret_cast<const float4 *>(tp + c + 4 * q);
              const double2 ea = *reinterpret_cast<const double2 *>(ep + c + 4 * q);
              const double2 eb = *reinterpret_cast<const double2 *>(ep + c + 4 * q + 2);
              acc = fma((double)xv.x, ea.x, acc);
              acc = fma((double)xv.y, ea.y, acc);
              acc = fma((double)xv.z, eb.x, acc);
              acc = fma((double)xv.w, eb.y, acc);
            }
          }
        }
        part_s[g * SW_MAXM + t] = acc;
      }
    }
    __syncthreads();

    // ---- combine row groups, all-gather the K slab partials ----
    double mine = 0.0;
    if (tid < mB) {
      for (int g = 0; g < ngroups; ++g) mine += part_s[g * SW_MAXM + tid];
    }
    if (K > 1) {
      const uint32_t epoch = (uint32_t)(s + 1);
      double *slot = a.xpart + ((size_t)(s & 1) * K) * SW_MAXM;
      if (tid < mB) st_agent_u64(slot + (size_t)wg * SW_MAXM + tid, mine);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(a.xflags + (size_t)wg * SW_FLAG_STRIDE, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wave == 0) {
        uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
        const uint64_t t0 = wall_clock64();
        int ok = 0;
        for (;;) {
          bool all_here = true;
          for (int w = lane; w < K; w += 64) {
            const uint32_t f = __hip_atomic_load(a.xflags + (size_t)w * SW_FLAG_STRIDE, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
            all_here = all_here && (f >= epoch);
          }
          if (__all(all_here)) { ok = 1; break; }
          const uint32_t ab = __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__any(ab != 0)) break;
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) {
            if (lane == 0) __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) ctrl_s[0] = ok;
      }
      __syncthreads();
      if (ctrl_s[0] == 0) {  // uniform: some workgroup never arrived; give up, report
        if (tid == 0) a.sc->error = 1u;
        return;
      }
      if (tid < mB) {
        double r = 0.0;
        for (int w = 0; w < K; ++w) r += ld_agent_f64(slot + (size_t)w * SW_MAXM + tid);
        r0_s[tid] = r;
      }
    } else {
      if (tid < mB) r0_s[tid] = mine;
    }
    __syncthreads();

    // ---- wave 0: the in-block recurrence; waves 1..7: stream in the next block ----
    if (wave == 0) {
      const int ngrp = (mB + 63) >> 6;
      double r[2];
      r[0] = (lane < mB) ? r0_s[lane] : 0.0;
      r[1] = (64 + lane < mB) ? r0_s[64 + lane] : 0.0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < ngrp) {
          const int base = 64 * q;
          const int t = base + lane;
          const bool live = t < mB;
          const float c_b0 = live ? st.b0[t] : 0.0f, c_xxb0 = live ? st.xxb0[t] : 0.0f;
          const float c_den = live ? st.den[t] : 1.0f, c_b2 = live ? st.b2[t] : 0.0f;
          const double c_sdz1 = live ? st.sdz1[t] : 0.0, c_u = live ? st.u[t] : 0.0;
          const double c_gjj = live ? (double)gram_s[(size_t)t * m + t] : 0.0;
          float o_b = 0.0f, o_d = 0.0f, o_delta = 0.0f;
          const int cnt = min(64, mB - base);
          for (int l = 0; l < cnt; ++l) {
            const double rj = readlane_f64(r[q], l);
            const float b0 = readlane_f32(c_b0, l);
            const float xxb0 = readlane_f32(c_xxb0, l);
            const float den = readlane_f32(c_den, l);
            const double sdz1 = readlane_f64(c_sdz1, l);
            const float mean = ((float)rj + xxb0) / den;
            const float b1 = (float)((double)mean + sdz1);
            float bn = b1, dn = 1.0f;
            if (a.flags & SWF_SELECT) {
              const float b2 = readlane_f32(c_b2, l);
              const double u = readlane_f64(c_u, l);
              const double gjj = readlane_f64(c_gjj, l);
              const float d1f = b1 - b0;
              const float d2f = (a.flags & SWF_ALT_B2) ? (b2 - b0) : (0.0f - b0);
              const double D1 = (double)d1f, D2 = (double)d2f;
              // |e2|^2 - |e1|^2 with e_k = e - x d_k
              const double diffd = 2.0 * rj * (D1 - D2) + gjj * (D2 * D2 - D1 * D1);
              float pj;
              if (a.flags & SWF_MH) {
                const float diff = (float)(-diffd);
                pj = one_minus_pi * expf(Cc * diff);
                if (pj > 1.0f) pj = 1.0f;
              } else {
                const float diff = (float)diffd;
                const float LR = odds * expf(Cc * diff);
                pj = 1.0f / (1.0f + LR);
              }
              const bool acc = u < (double)pj;
              bn = acc ? b1 : b2;
              dn = acc ? 1.0f : 0.0f;
            }
            const float delta = bn - b0;
            const double dd = (double)delta;
            const GT *grow = gram_s + (size_t)(base + l) * m;
            if (q == 0) {
              r[0] = fma(-(double)grow[lane], dd, r[0]);
              if (ngrp > 1) r[1] = fma(-(double)((64 + lane < m) ? grow[64 + lane] : (GT)0), dd, r[1]);
            } else {
              r[1] = fma(-(double)((64 + lane < m) ? grow[64 + lane] : (GT)0), dd, r[1]);
            }
            if (lane == l) { o_b = bn; o_d = dn; o_delta = delta; }
            sum_d += (double)dn;
            sum_b2 = fma((double)bn, (double)bn, sum_b2);
          }
          if (live) { delta_s[t] = o_delta; bnew_s[t] = o_b; dnew_s[t] = o_d; }
        }
      }
    } else if (s + 1 < nb) {
      const int j1 = j0 + m;
      const int mB1 = min(m, a.p - j1);
      load_tile<XT>(tile_next, X, a.ld, row0, R, Rp, j1, mB1, tid - 64, SW_THREADS - 64);
      const int t = tid - 64;
      if (t < mB1) stage_marker(stage[buf ^ 1], t, j1 + t, a, ve, lam_common, dfp1);
    }
    __syncthreads();

    // ---- outputs of the block (workgroup 0 is the only writer of marker state) ----
    if (wg == 0 && tid < mB) {
      const float bn = bnew_s[tid];
      a.b[j0 + tid] = bn;
      a.d[j0 + tid] = dnew_s[tid];
      if (a.flags & SWF_VB_VEC) a.vb[j0 + tid] = (float)((double)(Sb + bn * bn) / st.chi[tid]);
    }
    // ---- slab update e_i <- e_i - x_ij * delta_j, j ascending (x*delta is exact in fp64) ----
    for (int i = tid; i < R; i += SW_THREADS) {
      double ev = e_s[i];
      const XT *tp = tile + i;
      for (int jj = 0; jj < mB; ++jj) {
        const double xv = (double)tp[(size_t)jj * Rp];
        ev = fma(-xv, (double)delta_s[jj], ev);
      }
      e_s[i] = ev;
    }
    // ---- Gram block of the next block (gram_s is free once the recurrence has finished) ----
    if (s + 1 < nb) {
      const GT *gsrc = gram + (size_t)(blk + 1) * m * m;
      for (int c = tid; c < m * m; c += SW_THREADS) gram_s[c] = gsrc[c];
    }
  }
  __syncthreads();
  for (int i = tid; i < R; i += SW_THREADS) a.e[row0 + i] = e_s[i];
  if (wg == 0 && tid == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; }
}

}  // namespace bwgr
