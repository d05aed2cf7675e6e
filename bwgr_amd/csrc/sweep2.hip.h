// bwgr_amd/csrc/sweep2.hip.h -- the pipelined exact sweep: K "streamer" workgroups + 1 "sequencer" workgroup.
//
// Same Markov chain and the same blocked algebra as sweep.hip.h (k_sweep), but the in-block recurrence is no longer
// replicated in every workgroup behind an all-gather.  Roles:
//
//   streamer w (blockIdx < K)   owns rows [wR,(w+1)R) of every marker: its slab of e (fp64) and three consecutive
//                               tiles of X stay in LDS.  For block i it waits for delta_i, applies it to its slab and
//                               then forms the slab dots of block i+2 -- i.e. q_b = X_b' e^(b-2), one block of lag.
//   sequencer (blockIdx == K)   r0_b = sum_w q_b^(w) - Gx_b' delta_{b-1}   (Gx_b = X_{b-1}' X_b, precomputed, exact),
//                               then the in-block recurrence of sweep.hip.h on G_bb, and publishes delta_b.
//
// With the lag the two hand-offs (q: write-through payload + epoch flag per streamer; delta: 8-byte {epoch, float}
// granules, MI355X_MICROARCH "R2") overlap with the other role's work instead of adding up per block.  Waits:
// sequencer(b) needs q_b, which streamers publish after delta_{b-2}; streamer(i) needs delta_i.  No cycle.
// Every spin is bounded by the 100 MHz wall clock and a shared abort word.
#pragma once
#include "sweep.hip.h"

namespace bwgr {

#ifdef BWGR_STAMPS
#define S2STAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph2[k] += t_ - tl2; tl2 = t_; } } while (0)
#define S2STAMP_DECL unsigned long long ph2[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, tl2 = __builtin_amdgcn_s_memtime()
#define S2STAMP_FLUSH(base, n) do { if (tid == 0 && a.stamps) for (int k_ = 0; k_ < (n); ++k_) a.stamps[(base) + k_] += ph2[k_]; } while (0)
#else
#define S2STAMP(k) do { } while (0)
#define S2STAMP_DECL do { } while (0)
#define S2STAMP_FLUSH(base, n) do { } while (0)
#endif

static constexpr int S2_NSLOT = 4;   // ring depth of the q / delta buffers (2 would do; 4 keeps lines apart)

template <typename XT> __host__ __device__ inline size_t s2_seq_lds_bytes(int m);
template <typename XT> __host__ __device__ inline size_t sweep2_lds_bytes(int m, int R) {
  size_t streamer = (size_t)3 * m * tile_rp<XT>(R) * sizeof(XT);
  streamer = (streamer + 15) & ~(size_t)15;
  streamer += (size_t)R * sizeof(double) + SW_MAXM * sizeof(double) + (size_t)(SW_THREADS / 64) * SW_MAXM * sizeof(double) + 64;
  const size_t seq = s2_seq_lds_bytes<XT>(m);
  return streamer > seq ? streamer : seq;
}

// tile macros with an explicit thread numbering (streamers use all 512 threads)
#define S2_ISSUE1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) name = src_[c_]; }
#define S2_TILE_ISSUE(j0_, mB_) do { const int tot_ = (mB_) * (R / PER); \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(X + (size_t)(j0_) * R); BWGR_TILE_EACH(S2_ISSUE1) } while (0)
#define S2_COMMIT1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) { \
    const int jj_ = (int)(((float)c_ + 0.5f) * rcpr_), ii_ = c_ - jj_ * cpr_; \
    *reinterpret_cast<uint4 *>((dst_) + (size_t)jj_ * Rp + ii_ * PER) = name; } }
#define S2_TILE_COMMIT(dstp, mB_) do { XT *dst_ = (dstp); const int cpr_ = R / PER; const int tot_ = (mB_) * cpr_; \
    const float rcpr_ = 1.0f / (float)cpr_; BWGR_TILE_EACH(S2_COMMIT1) } while (0)

// ------------------------------------------------------------------------------------------------------------------
// streamer
// ------------------------------------------------------------------------------------------------------------------
template <typename XT>
__device__ __forceinline__ void s2_slab_dots(const XT *tile, const double *e_s, double *part_s, int mB, int Rp, int mpad,
                                             int rows_per_group, int tid) {
  constexpr int PER = XTraits<XT>::PER16;
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
          acc = fma((double)x0, ea.x, acc); acc = fma((double)x1, ea.y, acc);
          acc = fma((double)x2, eb.x, acc); acc = fma((double)x3, eb.y, acc);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 xv = *reinterpret_cast<const float4 *>(tp + c + 4 * q);
          const double2 ea = *reinterpret_cast<const double2 *>(ep + c + 4 * q);
          const double2 eb = *reinterpret_cast<const double2 *>(ep + c + 4 * q + 2);
          acc = fma((double)xv.x, ea.x, acc); acc = fma((double)xv.y, ea.y, acc);
          acc = fma((double)xv.z, eb.x, acc); acc = fma((double)xv.w, eb.y, acc);
        }
      }
    }
    part_s[g * SW_MAXM + t] = acc;
  }
}

template <typename XT>
__device__ __forceinline__ void s2_streamer(const SweepArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // declared here so that LDS accesses stay ds_*
  constexpr int PER = XTraits<XT>::PER16;
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int m = a.m, R = a.R, K = a.K;
  const int Rp = tile_rp<XT>(R);
  const int row0 = wg * R;
  const int nb = a.blk_end - a.blk_begin;
  XT *tile_base = reinterpret_cast<XT *>(smem);
  const int tile_elems = m * Rp;
#define S2_TILE(i_) (tile_base + (size_t)((i_) % 3) * tile_elems)
  size_t off = ((size_t)3 * m * Rp * sizeof(XT) + 15) & ~(size_t)15;
  double *e_s = reinterpret_cast<double *>(smem + off); off += (size_t)R * sizeof(double);
  double *delta_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  double *part_s = reinterpret_cast<double *>(smem + off); off += (size_t)(SW_THREADS / 64) * SW_MAXM * sizeof(double);
  volatile int *fail_s = reinterpret_cast<volatile int *>(smem + off);   // set once, on a failed wait
  const XT *X = reinterpret_cast<const XT *>(a.X) + (size_t)wg * a.p * R;
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  const int mpad = (m <= 64) ? 64 : 128;
  const int ngroups = SW_THREADS / mpad, rows_per_group = R / ngroups;
  auto blk_j0 = [&](int b) { return (a.blk_begin + b) * m; };
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };

  uint4 tp0 = make_uint4(0, 0, 0, 0), tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0;
  for (int i = tid; i < R; i += SW_THREADS) e_s[i] = a.e[row0 + i];
  if (tid == 0) fail_s[0] = 0;
  S2_TILE_ISSUE(blk_j0(0), blk_m(0)); S2_TILE_COMMIT(S2_TILE(0), blk_m(0));
  if (nb > 1) { S2_TILE_ISSUE(blk_j0(1), blk_m(1)); S2_TILE_COMMIT(S2_TILE(1), blk_m(1)); }
  if (nb > 2) S2_TILE_ISSUE(blk_j0(2), blk_m(2));
  __syncthreads();

  // publish the slab dots of block b (computed against the current e_s): payload write-through, then the epoch flag
  auto publish = [&](int b, const XT *tile) {
    const int mB = blk_m(b);
    s2_slab_dots<XT>(tile, e_s, part_s, mB, Rp, mpad, rows_per_group, tid);
    __syncthreads();
    if (tid < mB) {
      double mine = 0.0;
      for (int g = 0; g < ngroups; ++g) mine += part_s[g * SW_MAXM + tid];
      st_agent_u64(a.qpart + ((size_t)(b % S2_NSLOT) * K + wg) * SW_MAXM + tid, mine);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      st_agent_u32(a.xflags + (size_t)wg * SW_FLAG_STRIDE, (uint32_t)(b + 1));
  };
  publish(0, S2_TILE(0));
  if (nb > 1) publish(1, S2_TILE(1));
  S2STAMP_DECL;

  for (int i = 0; i < nb; ++i) {
    const int mB = blk_m(i);
    S2STAMP(5);
    // tile(i+2) lands in the buffer tile(i-1) used; its loads were issued one iteration ago
    if (i + 2 < nb) S2_TILE_COMMIT(S2_TILE(i + 2), blk_m(i + 2));
    if (i + 3 < nb) S2_TILE_ISSUE(blk_j0(i + 3), blk_m(i + 3));
    S2STAMP(0);
    // delta_i: one 8-byte {epoch, float} granule per marker, polled by the thread that needs it
    int bad = 0;
    if (tid < mB) {
      const unsigned long long *g = a.dgran + (size_t)(i % S2_NSLOT) * SW_MAXM + tid;
      const uint32_t epoch = (uint32_t)(i + 1);
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        const unsigned long long v = ld_agent_raw64(g);
        if ((uint32_t)(v >> 32) == epoch) { delta_s[tid] = (double)__uint_as_float((uint32_t)v); break; }
        if ((++spins & 63u) == 0u) {
          if (ld_agent_u32(abortw) != 0u) { bad = 1; break; }
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) {
            st_agent_u32(abortw, 1u); bad = 1; break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (bad) fail_s[0] = 1;
    __syncthreads();
    if (fail_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
    S2STAMP(1);
    // slab update with tile(i) (fp64, x*delta exact)
    {
      const XT *tile = S2_TILE(i);
      const int nparts = (R >= SW_THREADS) ? 1 : SW_THREADS / R;
      const int per = (mB + nparts - 1) / nparts;
      for (int i0 = 0; i0 < R; i0 += SW_THREADS) {
        const int part = (R >= SW_THREADS) ? 0 : tid / R;
        const int r = (R >= SW_THREADS) ? i0 + tid : tid - part * R;
        const bool active = (r < R) && (part < nparts);
        double acc = 0.0;
        if (active) {
          const int ja = part * per, jb = min(mB, ja + per);
          const XT *tp = tile + r;
          int jj = ja;
          for (; jj + 8 <= jb; jj += 8) {
            double xv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = (double)tp[(size_t)(jj + u) * Rp]; dv[u] = delta_s[jj + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(xv[u], dv[u], acc);
          }
          for (; jj < jb; ++jj) acc = fma((double)tp[(size_t)jj * Rp], delta_s[jj], acc);
        }
        if (nparts == 1) { if (active) e_s[r] -= acc; }
        else {
          if (active) part_s[part * R + r] = acc;
          __syncthreads();
          if (tid < R) { double t = 0.0; for (int q = 0; q < nparts; ++q) t += part_s[q * R + tid]; e_s[tid] -= t; }
        }
      }
    }
    __syncthreads();
    S2STAMP(2);
    if (i + 2 < nb) publish(i + 2, S2_TILE(i + 2));
    S2STAMP(4);
  }
  if (wg == 0) S2STAMP_FLUSH(0, 6);
  __syncthreads();
  for (int i = tid; i < R; i += SW_THREADS) a.e[row0 + i] = e_s[i];
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer
// ------------------------------------------------------------------------------------------------------------------
// LDS: the strict upper triangle of G_bb, row-packed and DOUBLE-buffered (the next block's copy lands while wave 0
// runs the recurrence), the off-diagonal block Gx_{b+1} (single: written during recurrence(b), read right after it),
// two StageBuf + two SpecBuf, and small vectors.
template <typename XT> __host__ __device__ inline size_t s2_seq_lds_bytes(int m) {
  using GT = typename XTraits<XT>::GT;
  const int pstride = ((m * (m - 1) / 2 + 3) / 4) * 4;
  size_t s = ((size_t)2 * ((pstride < 4 ? 4 : pstride) + 256) * sizeof(GT) + 15) & ~(size_t)15;   // 64 + 192 entries of slack per buffer
  s += ((size_t)m * m * sizeof(GT) + 15) & ~(size_t)15;
  s += 2 * sizeof(StageBuf) + 2 * sizeof(SpecBuf);
  s += 4 * SW_MAXM * sizeof(double);      // r0, r0n, delta, acc_corr
  s += 2 * SW_MAXM * sizeof(float);       // bnew, dnew
  s += SW_MAXM * sizeof(int);             // acc_k
  s += 4 * SW_MAXM * sizeof(double);      // part
  s += 64;
  return s;
}

template <typename XT, bool SELECT>
__device__ __forceinline__ void s2_sequencer(const SweepArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using GT = typename XTraits<XT>::GT;
  constexpr int GPT = 16 / sizeof(GT);
  constexpr int MAXMX = XTraits<XT>::MAXM;
  constexpr int NHELP = SW_THREADS - 64;
  constexpr int XCH = (MAXMX * MAXMX / GPT + NHELP - 1) / NHELP;                    // chunks of Gx per helper thread
  constexpr int PCH = ((MAXMX * (MAXMX - 1) / 2 + 3) / 4 * 4 / GPT + NHELP - 1) / NHELP;   // chunks of packed G
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = a.m, K = a.K;
  const int nb = a.blk_end - a.blk_begin;
  const int pstride = a.pstride;
  size_t off = 0;
  // each packed buffer has 64 entries of slack in front and 192 behind, so the recurrence's row loads need no clamp
  GT *gp_base = reinterpret_cast<GT *>(smem) + 64;
  const int gp_elems = (pstride < 4 ? 4 : pstride) + 256;
#define S2_GP(i_) (gp_base + (size_t)((i_) & 1) * gp_elems)
  off = ((size_t)2 * gp_elems * sizeof(GT) + 15) & ~(size_t)15;
  GT *gx_s = reinterpret_cast<GT *>(smem + off); off += ((size_t)m * m * sizeof(GT) + 15) & ~(size_t)15;
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 2 * sizeof(StageBuf);
  SpecBuf *specb = reinterpret_cast<SpecBuf *>(smem + off); off += 2 * sizeof(SpecBuf);
  double *r0_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  off += SW_MAXM * sizeof(double);   // (spare)
  double *delta_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  double *acc_corr = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  float *bnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  float *dnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  int *acc_k = reinterpret_cast<int *>(smem + off); off += SW_MAXM * sizeof(int);
  double *part_s = reinterpret_cast<double *>(smem + off); off += 4 * SW_MAXM * sizeof(double);
  volatile int *ctrl_s = reinterpret_cast<volatile int *>(smem + off);   // [0] ok flag, [1] number of accepted markers
  const GT *gramp = reinterpret_cast<const GT *>(a.gramp);
  const GT *gramx = reinterpret_cast<const GT *>(a.gramx);
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  const float Cc = a.sc->C, odds = a.sc->odds, one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;
  const int pchunks = pstride / GPT, xchunks = m * m / GPT;
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };
  auto prow = [&](int k) { return k * (m - 1) - k * (k - 1) / 2; };   // offset of packed row k

  auto wait_q = [&](int b) -> int {
    const uint32_t epoch = (uint32_t)(b + 1);
    const uint64_t t0 = wall_clock64();
    for (;;) {
      bool all_here = true;
      for (int w = lane; w < K; w += 64)
        all_here = all_here && (ld_agent_u32(a.xflags + (size_t)w * SW_FLAG_STRIDE) >= epoch);
      if (__all(all_here)) return 1;
      if (__any(ld_agent_u32(abortw) != 0u)) return 0;
      if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) {
        if (lane == 0) st_agent_u32(abortw, 1u);
        return 0;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  // sum over a third of the streamers of q_b[t]; fixed order; result in part_s[part][t]
  auto gather_q = [&](int b, int part, int t, int mB) {
    const double *slot = a.qpart + (size_t)(b % S2_NSLOT) * K * SW_MAXM;
    const int wq = (K + 2) / 3;
    double r = 0.0;
    for (int wbase = 0; wbase < wq; wbase += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int w = part * wq + wbase + u;
        v[u] = (t < mB && wbase + u < wq && w < K) ? ld_agent_f64(slot + (size_t)w * SW_MAXM + t) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) r += v[u];
    }
    part_s[part * SW_MAXM + t] = r;
  };
  auto copy16 = [&](void *dst, const void *src, int nchunks, int t0, int nth) {
    for (int c = t0; c < nchunks; c += nth) reinterpret_cast<uint4 *>(dst)[c] = reinterpret_cast<const uint4 *>(src)[c];
  };

  // ---- prologue: block 0 ----
  {
    const int mB = blk_m(0);
    copy16(S2_GP(0), gramp + (size_t)a.blk_begin * pstride, pchunks, tid, SW_THREADS);
    copy16(&stage[0], a.ps.blocks + a.blk_begin, (int)(sizeof(StageBuf) / 16), tid, SW_THREADS);
    copy16(&specb[0], a.ps.spec + a.blk_begin, (int)(sizeof(SpecBuf) / 16), tid, SW_THREADS);
    if (tid == 0) { ctrl_s[0] = 1; ctrl_s[1] = 0; }
    __syncthreads();
    if (wave >= 1 && wave <= 6) {
      const int ok = wait_q(0);
      if (!ok && lane == 0) ctrl_s[0] = 0;
      if (ok) gather_q(0, (tid - 64) >> 7, (tid - 64) & 127, mB);
    }
    __syncthreads();
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }
    if (tid < mB) r0_s[tid] = (part_s[tid] + part_s[SW_MAXM + tid]) + part_s[2 * SW_MAXM + tid];
  }
  double sum_d = 0.0, sum_b2 = 0.0;
  S2STAMP_DECL;

  for (int b = 0; b < nb; ++b) {
    const int blk = a.blk_begin + b;
    const int j0 = blk * m;
    const int mB = blk_m(b);
    const bool have_next = (b + 1 < nb);
    const int mBn = have_next ? blk_m(b + 1) : 0;
    const StageBuf &st = stage[b & 1];
    const SpecBuf &sb = specb[b & 1];
    const GT *gp = S2_GP(b);
    __syncthreads();   // r0_s, gp_s[b&1], stage[b&1], specb[b&1] of this block are in place
    S2STAMP(0);

    if (wave == 0) {
      // ---- the in-block recurrence ----
      const int ngrp = (mB + 63) >> 6;
      double r[2];
      LaneConst lc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        const bool live = t < mB;
        r[q] = live ? (SELECT ? (r0_s[t] - sb.spec[t]) : r0_s[t]) : 0.0;
        lc[q].b0 = live ? st.b0[t] : 0.0f; lc[q].xxb0 = live ? st.xxb0[t] : 0.0f;
        lc[q].b2 = live ? st.b2[t] : 0.0f; lc[q].drej = live ? st.drej[t] : 0.0f;
        lc[q].rden = live ? st.rden[t] : 1.0; lc[q].sdz1 = live ? st.sdz1[t] : 0.0;
        lc[q].u = live ? st.u[t] : 2.0; lc[q].gjj = live ? sb.gjj[t] : 0.0;
      }
      // packed row k: entry for marker j (> k) sits at prow(k) + j - k - 1
      auto gat = [&](int k, int j) -> GT { return (j > k && j < m) ? gp[prow(k) + j - k - 1] : (GT)0; };
      unsigned long long accmask[2] = {0ull, 0ull};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < ngrp) {
          const int base = 64 * q;
          const int cnt = min(64, mB - base);
          if (!SELECT) {
            // packed row k = base+l: own-group entry of lane at oA + lane (needed only for lane > l), other-group entry
            // (q == 0: marker 64+lane) at oB + lane.  The buffers carry slack, so loads are unconditional; the own-group
            // value is masked, the other-group value of dead lanes (marker >= m) only feeds registers nobody reads.
            int oA = prow(base) - 1, oB = prow(base) + 63;   // + lane, at l = 0
            if (q == 0 && ngrp > 1) {
              GT gn0 = gp[oA + lane], gn1 = gp[oB + lane];
              for (int l = 0; l < cnt; ++l) {
                const GT g0 = (lane > l) ? gn0 : (GT)0, g1 = gn1;
                oA += m - 2 - (base + l); oB += m - 2 - (base + l);
                gn0 = gp[oA + lane]; gn1 = gp[oB + lane];          // next row (slack makes the last one harmless)
                const float dl = lane_b1(r[0], lc[0]) - lc[0].b0;
                const double dd = (double)readlane_f32(dl, l);
                r[0] = fma(-(double)g0, dd, r[0]);
                r[1] = fma(-(double)g1, dd, r[1]);
              }
            } else {
              GT gn0 = gp[oA + lane];
              for (int l = 0; l < cnt; ++l) {
                const GT g0 = (lane > l) ? gn0 : (GT)0;
                oA += m - 2 - (base + l);
                gn0 = gp[oA + lane];
                const float dl = lane_b1(r[q], lc[q]) - lc[q].b0;
                const double dd = (double)readlane_f32(dl, l);
                r[q] = fma(-(double)g0, dd, r[q]);
              }
            }
          } else {
            int front = 0;
            while (front < cnt) {
              const float b1 = lane_b1(r[q], lc[q]);
              const bool acc = lane_accept(r[q], b1, lc[q], a.flags, Cc, odds, one_minus_pi);
              const unsigned long long bal = __ballot(acc && lane >= front && lane < cnt);
              if (bal == 0ull) break;
              const int js = __ffsll((long long)bal) - 1;
              const float corr_f1 = b1 - lc[q].b0;
              const double corr = (double)readlane_f32(corr_f1, js) - (double)readlane_f32(lc[q].drej, js);
              r[q] = fma(-(double)gat(base + js, base + lane), corr, r[q]);
              if (q == 0 && ngrp > 1) r[1] = fma(-(double)gat(base + js, 64 + lane), corr, r[1]);
              accmask[q] |= (1ull << js);
              front = js + 1;
            }
          }
        }
      }
      // outputs; delta_b goes out at once as {epoch, float} granules (one 8-byte write-through store per marker)
      unsigned long long *gslot = a.dgran + (size_t)(b % S2_NSLOT) * SW_MAXM;
      const int nacc0 = __popcll(accmask[0]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        if (t < mB) {
          const float b1 = lane_b1(r[q], lc[q]);
          const bool inc = SELECT ? (((accmask[q] >> lane) & 1ull) != 0ull) : true;
          const float bn = inc ? b1 : lc[q].b2;
          const float dn = inc ? 1.0f : 0.0f;
          const float dl = bn - lc[q].b0;
          st_agent_raw64(gslot + t, ((unsigned long long)(uint32_t)(b + 1) << 32) | (unsigned long long)__float_as_uint(dl));
          delta_s[t] = (double)dl; bnew_s[t] = bn; dnew_s[t] = dn;
          if (SELECT && inc) {   // what this marker changed relative to the speculated step
            const int idx = (q ? nacc0 : 0) + __popcll(accmask[q] & ((1ull << lane) - 1ull));
            acc_k[idx] = t; acc_corr[idx] = (double)(b1 - lc[q].b0) - (double)lc[q].drej;
          }
          sum_d += (double)dn;
          sum_b2 = fma((double)bn, (double)bn, sum_b2);
        }
      }
      if (SELECT && lane == 0) ctrl_s[1] = nacc0 + __popcll(accmask[1]);
      S2STAMP(1);
    } else if (have_next) {
      // ---- helpers: bring in block b+1 (packed G, Gx, constants), then the streamers' q_{b+1} ----
      static_assert(PCH <= 5 && XCH <= 10, "named prefetch registers cover 5 + 10 chunks per helper thread");
      const uint4 z4 = make_uint4(0, 0, 0, 0);
      uint4 gq0 = z4, gq1 = z4, gq2 = z4, gq3 = z4, gq4 = z4;
      uint4 xq0 = z4, xq1 = z4, xq2 = z4, xq3 = z4, xq4 = z4, xq5 = z4, xq6 = z4, xq7 = z4, xq8 = z4, xq9 = z4;
      const uint4 *gsrc = reinterpret_cast<const uint4 *>(gramp + (size_t)(blk + 1) * pstride);
      const uint4 *xsrc = reinterpret_cast<const uint4 *>(gramx + (size_t)(blk + 1) * m * m);
#define S2_G_EACH(X) X(0, gq0) X(1, gq1) X(2, gq2) X(3, gq3) X(4, gq4)
#define S2_X_EACH(X) X(0, xq0) X(1, xq1) X(2, xq2) X(3, xq3) X(4, xq4) X(5, xq5) X(6, xq6) X(7, xq7) X(8, xq8) X(9, xq9)
#define S2_GLD(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < pchunks) name = gsrc[c_]; }
#define S2_XLD(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < xchunks) name = xsrc[c_]; }
      S2_G_EACH(S2_GLD)
      S2_X_EACH(S2_XLD)
      constexpr int NCH = (int)(sizeof(StageBuf) / 16), NSP = (int)(sizeof(SpecBuf) / 16);
      static_assert(NCH <= NHELP && NSP <= NHELP, "one chunk per helper thread");
      uint4 spre = make_uint4(0, 0, 0, 0), cpre = spre;
      if (tid - 64 < NCH) spre = reinterpret_cast<const uint4 *>(a.ps.blocks + blk + 1)[tid - 64];
      if (tid - 64 < NSP) cpre = reinterpret_cast<const uint4 *>(a.ps.spec + blk + 1)[tid - 64];
      if (wave >= 1 && wave <= 6) {
        const int ok = wait_q(b + 1);
        if (!ok && lane == 0) ctrl_s[0] = 0;
        if (ok) gather_q(b + 1, (tid - 64) >> 7, (tid - 64) & 127, mBn);
      }
      uint4 *gdst = reinterpret_cast<uint4 *>(S2_GP(b + 1));
      uint4 *xdst = reinterpret_cast<uint4 *>(gx_s);
#define S2_GST(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < pchunks) gdst[c_] = name; }
#define S2_XST(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < xchunks) xdst[c_] = name; }
      S2_G_EACH(S2_GST)
      S2_X_EACH(S2_XST)
#undef S2_GLD
#undef S2_XLD
#undef S2_GST
#undef S2_XST
      if (tid - 64 < NCH) reinterpret_cast<uint4 *>(&stage[(b + 1) & 1])[tid - 64] = spre;
      if (tid - 64 < NSP) reinterpret_cast<uint4 *>(&specb[(b + 1) & 1])[tid - 64] = cpre;
    }
    __syncthreads();   // A: recurrence done; block b+1's Gram/constants are in LDS; q_{b+1} partial sums in part_s
    S2STAMP(2);
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }

    if (tid < mB) {   // marker state of this block
      const float bn = bnew_s[tid];
      a.b[j0 + tid] = bn;
      a.d[j0 + tid] = dnew_s[tid];
      if (a.flags & SWF_VB_VEC) a.vb[j0 + tid] = (float)((double)(Sb + bn * bn) / st.chi[tid]);
    }
    if (have_next) {
      if (SELECT) {
        // r0_{b+1} = sum_w q - Gx' drej_b (precomputed) - sum_{accepted k} Gx[k][.] * (what k changed beyond drej)
        if (tid < mBn) {
          double r = ((part_s[tid] + part_s[SW_MAXM + tid]) + part_s[2 * SW_MAXM + tid]) - specb[(b + 1) & 1].xspec[tid];
          const int nacc = ctrl_s[1];
          for (int i = 0; i < nacc; ++i) r = fma(-(double)gx_s[(size_t)acc_k[i] * m + tid], acc_corr[i], r);
          r0_s[tid] = r;
        }
        S2STAMP(3);
      } else {
        // r0_{b+1} = sum_w q - Gx' delta_b  (dense; 4 k-ranges per marker, LDS reads batched)
        const int sp = tid >> 7, j = tid & 127;
        double xa = 0.0;
        if (j < mBn) {
          const int k0 = sp * 32, kx1 = min(k0 + 32, mB);
          int k = k0;
          for (; k + 8 <= kx1; k += 8) {
            double gv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gv[u] = (double)gx_s[(size_t)(k + u) * m + j]; dv[u] = delta_s[k + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) xa = fma(gv[u], dv[u], xa);
          }
          for (; k < kx1; ++k) xa = fma((double)gx_s[(size_t)k * m + j], delta_s[k], xa);
        }
        const double q3 = (tid < mBn) ? ((part_s[tid] + part_s[SW_MAXM + tid]) + part_s[2 * SW_MAXM + tid]) : 0.0;
        __syncthreads();   // the q partial sums have been read; part_s is free
        S2STAMP(3);
        if (j < mBn) part_s[sp * SW_MAXM + j] = xa;
        __syncthreads();
        if (tid < mBn) r0_s[tid] = q3 - ((part_s[tid] + part_s[SW_MAXM + tid]) + (part_s[2 * SW_MAXM + tid] + part_s[3 * SW_MAXM + tid]));
      }
      S2STAMP(4);
    }
  }
  S2STAMP_FLUSH(6, 5);
  if (wave == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
    if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; }
  }
}

template <typename XT, bool SELECT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep2(const SweepArgs a) {
  if ((int)blockIdx.x == a.K) s2_sequencer<XT, SELECT>(a);
  else s2_streamer<XT>(a);
}

}  // namespace bwgr
