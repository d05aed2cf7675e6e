// bwgr_amd: the affine models' in-block solve as ONE triangular matrix-vector product per 128-marker block.
//
// The samplers without a Bernoulli draw (BayesRR, BayesA, BayesL, wgr / KMUP at pi = 0, the affine EM members) update marker
// j of a block from  t_j = (r_j + xx_j b0_j) rden_j + sd_j z_j,  r_j = r0_j - sum_{k<j} G_jk delta_k,  delta_k = dscale (t_k - b0_k):
// a lane-ordered recurrence of 128 dependent steps per block in k_sweep2's sequencer (sweep2.hip.h, ~106 cycles a step).
// With  M = diag(dscale rden) G_L  (G_L the strict lower triangle of the block's Gram matrix) the un-rounded steps
// d = t - b0 solve  (I + M) d = c - b0,  c_j = (r0_j + xx_j b0_j) rden_j + sd_j z_j,  so  d = W (c - b0)  with
// W = (I + M)^-1  lower triangular.  rden is fixed during a sweep (the variances move between sweeps), hence:
//
//   k_affine_inv   one workgroup per block, the whole GPU, before the sweep: W by blocked forward substitution on 16 x 16
//                  tiles in fp64, written in the order the sequencer's lanes read it (73.7 KB per block);
//   s2_sequencer_winv   the sequencer of k_sweep2w: per block a dense cross term (r0 = q - Gx' delta_prev), the 128 x 128
//                  triangular product out of registers (waves 0-3: nine tiles each, loaded a block ahead), outputs.
//                  Waves 4-7 gather the streamers' slab dots of the next block and write the block's state.
//
// The one departure from the serial chain: the serial chain feeds the float-ROUNDED draw of marker k into the markers
// behind it, the product feeds the un-rounded one (relative 2^-24 per term, random signs, no accept / reject behind it):
// the chains agree to ~1e-7 relative per sweep and stay there (the sweep is a contraction), inside the 1e-6 the parity tests
// allow; tests/test_gpu_parity2.py::test_affine_winv_* bound it over 200 iterations.  int8 panels with 16-bit Gram staging
// only; everything else keeps the serial sequencer.
// Reference for the recurrence being solved: /root/reference/src/Rcpp20260726ai.cpp:612-619 (BayesA), :833-838 (BayesRR),
// :20-31 (KMUP, pi = 0).
#pragma once
#include "sweep2.hip.h"
#include "sweep3.hip.h"

namespace bwgr {

static constexpr int S2W_WDOUBLES = 4 * 9 * 256;   // per block: four waves x nine 16 x 16 tiles
static constexpr size_t S2W_INV_LDS = (size_t)(36 * 256 + 8 * 16 * 18 + 8 * 256 + SW_MAXM) * sizeof(double) + (size_t)(SW_MAXM * (SW_MAXM - 1) / 2) * sizeof(int32_t);
// wave w holds block rows 7 - w (8 - w tiles) and w (w + 1 tiles): nine tiles each
__host__ __device__ inline void s2w_tile_of(int w, int q, int &I, int &J) { if (q < 8 - w) { I = 7 - w; J = q; } else { I = w; J = q - (8 - w); } }

// ------------------------------------------------------------------------------------------------------------------
// W = (I + diag(scl) G_L)^-1 of every block
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_affine_inv(const SweepArgs a, double *winv, double dscale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *Wt = reinterpret_cast<double *>(smem);   // [36 * 256]: tile (I, J), J <= I, at I (I + 1) / 2 + J; element [r][c]
  double *Gt = Wt + 36 * 256;                      // [8 * 16 * 18]: G tiles of one block row, [J][r][t], rows padded to 18
  double *St = Gt + 8 * 16 * 18;                   // [8 * 256]
  double *scl = St + 8 * 256;                      // [SW_MAXM]
  int32_t *gp_s = reinterpret_cast<int32_t *>(scl + SW_MAXM);   // the block's packed Gram rows (one coalesced pass over global memory)
  const int tid = threadIdx.x;
  const int blk = a.blk_begin + (int)blockIdx.x;
  const int m = a.m, mB = min(m, a.p - blk * m);
  const int32_t *gp = reinterpret_cast<const int32_t *>(a.gramp) + (size_t)blk * a.pstride;
  auto prow = [&](int k) { return k * (m - 1) - k * (k - 1) / 2; };
  for (int i = tid; i < m * (m - 1) / 2; i += 512) gp_s[i] = gp[i];
  auto gat = [&](int row, int col) -> double {   // G[row][col], col < row, both inside the block
    return (col < row && row < mB) ? (double)gp_s[prow(col) + row - col - 1] : 0.0;
  };
  if (tid < SW_MAXM) scl[tid] = (tid < mB) ? dscale * a.ps.blocks[blk].rden[tid] : 0.0;
  __syncthreads();
  if (tid < 256) {   // the eight diagonal tiles
    const int rr = tid & 15, tt = tid >> 4;
    for (int I = 0; I < 8; ++I) Gt[(I * 16 + rr) * 18 + tt] = gat(16 * I + rr, 16 * I + tt);
  }
  __syncthreads();
  if (tid < 128) {   // their inverses: one column per thread, forward substitution in registers
    const int I = tid >> 4, c = tid & 15;
    const double *g = Gt + (size_t)I * 16 * 18;
    double w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int t = 0; t + 1 < r; t += 2) { acc0 = fma(g[r * 18 + t], w[t], acc0); acc1 = fma(g[r * 18 + t + 1], w[t + 1], acc1); }
      if (r & 1) acc0 = fma(g[r * 18 + r - 1], w[r - 1], acc0);
      w[r] = (r > c) ? -scl[16 * I + r] * (acc0 + acc1) : ((r == c) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Wt[(I * (I + 1) / 2 + I) * 256 + r * 16 + c] = w[r];
  }
  __syncthreads();
  // block rows 1..7 on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15] of a
  // k-step of four; result register v is D[(l >> 4) + 4 v][l & 15]): wave J forms the target tile (I, J), J < I
  typedef double s2w_v4d __attribute__((ext_vector_type(4)));
  const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  for (int I = 1; I < 8; ++I) {
    for (int J = tid >> 8; J < I; J += 2) {   // G tiles (I, 0..I-1)
      const int rr = tid & 15, tt = (tid >> 4) & 15;
      Gt[(J * 16 + rr) * 18 + tt] = gat(16 * I + rr, 16 * J + tt);
    }
    __syncthreads();
    if (wv < I) {
      const int J = wv;
      // S_IJ = diag(scl_I) sum_{K=J..I-1} G_IK W_KJ
      s2w_v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int K = J; K < I; ++K) {
        const double *g = Gt + (size_t)(K * 16 + l15) * 18 + l4;
        const double *wk = Wt + (size_t)(K * (K + 1) / 2 + J) * 256 + l4 * 16 + l15;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(g[4 * s4], wk[64 * s4], acc, 0, 0, 0);
      }
      double *stj = St + J * 256;
#pragma unroll
      for (int v = 0; v < 4; ++v) stj[(l4 + 4 * v) * 16 + l15] = scl[16 * I + l4 + 4 * v] * acc[v];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave wrote the tile, this wave reads it)
      // W_IJ = -W_II S_IJ
      s2w_v4d acc2 = {0.0, 0.0, 0.0, 0.0};
      const double *wd = Wt + (size_t)(I * (I + 1) / 2 + I) * 256 + l15 * 16 + l4;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(wd[4 * s4], stj[(4 * s4 + l4) * 16 + l15], acc2, 0, 0, 0);
      double *wo = Wt + (size_t)(I * (I + 1) / 2 + J) * 256;
#pragma unroll
      for (int v = 0; v < 4; ++v) wo[(l4 + 4 * v) * 16 + l15] = -acc2[v];
    }
    __syncthreads();
  }
  // out, in the sequencer's order: wave w, tile q, half h, lane l holds W[16 I + (l & 15)][16 J + 4 (l >> 4) + 2 h + {0, 1}]
  double2 *dst = reinterpret_cast<double2 *>(winv + (size_t)blk * S2W_WDOUBLES);
  for (int ch = tid; ch < S2W_WDOUBLES / 2; ch += 512) {
    const int w = ch / 1152, chunk = ch - w * 1152;
    const int qh = chunk >> 6, l = chunk & 63, q = qh >> 1, h = qh & 1;
    int I, J; s2w_tile_of(w, q, I, J);
    const double *src = Wt + (size_t)(I * (I + 1) / 2 + J) * 256 + (l & 15) * 16 + 4 * (l >> 4) + 2 * h;
    dst[ch] = make_double2(src[0], src[1]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// cross Gram blocks as the sequencer's MFMA operand, in the order its lanes load them: per block 2048 pieces of 16 bytes,
// piece ((w * 2 + plane) * 2 + ch) * 64 + lane = bytes k = 64 ch + 16 (lane >> 4) .. + 15 of row j = 16 w + (lane & 15) of
// plane `plane`; byte k of row j of plane P = byte P of G[k][j] minus 128 (markers k of block b-dist against markers j of
// block b; pad markers hold G = 0)
// ------------------------------------------------------------------------------------------------------------------
static constexpr int S2W_PBYTES = 2 * SW_MAXM * SW_MAXM;            // 32 768 per block and distance
static constexpr int S2W_MAXDIST = 5;   // distances 1-3 as register operands of waves 0-3, distances 4-5 (pipelines five and six blocks deep) through LDS
static constexpr int S2W_NEARD = 3;
static constexpr int S2W_DSLOTS = 8;   // ring of the blocks' step digits (a block's digits are read up to five blocks later, while the current block's are written)

__global__ void k_gx_planes(const int32_t *src, unsigned char *dst, int m, int64_t nblocks, int dist, int *bad) {
  const int64_t total = nblocks * (S2W_PBYTES / 16);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(t & 63), ch = (int)((t >> 6) & 1), plane = (int)((t >> 7) & 1), w = (int)((t >> 8) & 7);
    const int64_t blk = t >> 11;
    const int j = 16 * w + (l & 15), k0 = 64 * ch + 16 * (l >> 4);
    uint32_t wd[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    if (blk >= dist && j < m) {
      const int32_t *g = src + (size_t)blk * m * m + j;
      int nbad = 0;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = k0 + u;
        const int32_t v = (k < m) ? g[(size_t)k * m] : 0;
        if (v < 0 || v > 65535) ++nbad;
        const uint32_t byte = (((uint32_t)v >> (8 * plane)) & 0xFFu) ^ 0x80u;
        wd[u >> 2] = (wd[u >> 2] & ~(0xFFu << (8 * (u & 3)))) | (byte << (8 * (u & 3)));
      }
      if (nbad && plane == 0) atomicAdd(bad, nbad);
    }
    reinterpret_cast<uint4 *>(dst)[t] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
  }
}

static constexpr int S2W_THREADS = 768;   // k_sweep2w's workgroups: twelve waves in the sequencer (the streamers use the first eight)
static constexpr int S2W_DROW = 144;   // bytes per digit row in LDS (128 + 16: the four rows on different banks)
struct S2WArgs {
  const double *winv;                        // [nblocks][S2W_WDOUBLES]
  const unsigned char *gxt[S2W_MAXDIST];     // gxt[d-1]: planes of the distance-d cross Gram blocks (missing distances repeat gxt[0])
  int nd;                                    // distances in use = lag - 1
  unsigned long long *qsum;                  // fixed-point streamers: [nblocks][SW_MAXM][2] slab-dot sums (integer atomics, low byte = count)
  int fx;                                    // 1: the streamers are s2w_streamer_fx (K3 workgroups of 128 rows), 0: k_sweep2's (K of R rows)
  int K3, sub;                               // streamer workgroups, streamers per slab
  int nq;                                    // copies of the sums (1, 2 or 4; streamer w adds into copy w mod nq: fewer atomics queue on one word)
  int ahead;                                 // blocks a prefetcher may run ahead of the sequencer
  int npf;                                   // L2 prefetch workgroups (blockIdx K + 8, K + 16, ...: the sequencer's XCD)
  int dbg;                                   // experiments (BWGR_DBGW), timing only: 1 = every step published as zero (the chain stands still), 2 = the sequencer does
                                             // not wait for the slab dots, 4 = the streamers do not wait for the steps, 8 = no sequencer, 16 = no streamers,
                                             // 32 = W loaded for the first block only, 64 = the Gram planes loaded for the first block only, 128 = b, d, vb not written
};

__host__ __device__ inline size_t s2w_fx_lds(int L);
__host__ __device__ inline size_t s2w_lds_bytes(int m, int R, int maxlag = 4) {
  size_t s = 3 * sizeof(StageBuf);
  s += (size_t)(2 * 4 + 1 + 1 + 1 + 1) * SW_MAXM * sizeof(double);   // q parts [parity][4], cross, rhs, d, delta
  s += (size_t)(S2W_DSLOTS * 4 + 1) * S2W_DROW;                       // delta digits of the last blocks, a row of zeros
  s += 256;
  if (maxlag > S2W_NEARD + 1) s += (size_t)(maxlag - 1 - S2W_NEARD) * S2W_PBYTES;   // the far distances' planes of one block
  size_t streamer = s2i_lds_bytes(m, R, 4);
  if (s2w_fx_lds(maxlag) > streamer) streamer = s2w_fx_lds(maxlag);
  return s > streamer ? s : streamer;
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer
// ------------------------------------------------------------------------------------------------------------------
#define S2W_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// the experiment switches (BWGR_DBGW) are compiled in only with -DBWGR_EXPERIMENTS (tools/ab3_probe.py builds that library)
#ifdef BWGR_EXPERIMENTS
#define S2W_DBG(A_) ((A_).dbg)
#else
#define S2W_DBG(A_) 0
#endif
#ifdef BWGR_STAMPS
#define S2WSTAMP(k) do { if (tid == 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph2[k] += t_ - tl2; tl2 = t_; } } while (0)
#define S2WSTAMP_FLUSH() do { if (tid == 512 && a.stamps) for (int k_ = 0; k_ < 8; ++k_) a.stamps[16 + k_] += ph2[k_]; } while (0)
#define S2WSTAMP0(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph2[k] += t_ - tl2; tl2 = t_; } } while (0)
#define S2WSTAMP0_FLUSH() do { if (tid == 0 && a.stamps) for (int k_ = 0; k_ < 8; ++k_) a.stamps[24 + k_] += ph2[k_]; } while (0)
#else
#define S2WSTAMP(k) do { } while (0)
#define S2WSTAMP_FLUSH() do { } while (0)
#define S2WSTAMP0(k) do { } while (0)
#define S2WSTAMP0_FLUSH() do { } while (0)
#endif
// 16 bytes per lane from global memory straight into LDS at lds_base + 16 * lane.  Inline asm, not the builtin: hipcc tracks
// the builtin's LDS writes and puts a vmcnt(0) in front of the next LDS access it cannot prove disjoint.  (M0 is not live in
// compiled code around these: nothing else in the function uses it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void s2w_dma16(const void *gsrc, unsigned char *lds_base) {
  const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(la) : "memory", "m0");
}
#pragma clang diagnostic pop

// The K streamers' slab dots of one block, two halves of the streamers on two threads per marker.  The words are REQUESTED a
// block before they are needed (with lag >= 3 the streamers are ahead and the words complete: no memory round trip in the
// block period) and polled only if a tag is missing.
static constexpr int S2W_QW = 16;   // words a polling thread requests early (K <= 32 streamers: all of them; 20 words spill registers); the words beyond are polled when due
struct S2WPoll {
  unsigned long long v[S2W_QW];
};
__device__ __forceinline__ void s2w_q_request(const SweepArgs &a, int b, int part, int t, S2WPoll &P) {
  const int K = a.K, wq = (K + 1) / 2;
  const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.qpart + (size_t)(b % S2_NSLOT) * K * SW_MAXM) + (size_t)min(part * wq, K - 1) * SW_MAXM + t;
  const int nw = min(S2W_QW, min(wq, K - part * wq));   // (uniform)
#pragma unroll
  for (int u = 0; u < S2W_QW; ++u) P.v[u] = ld_agent_raw64(slot + (size_t)min(u, max(nw - 1, 0)) * SW_MAXM);   // (unconditional: repeats of the last word are L2 hits)
}
__device__ __forceinline__ int s2w_q_collect(const SweepArgs &a, int b, int part, int t, int mB, S2WPoll &P, double *dst) {
  const int K = a.K, wq = (K + 1) / 2;
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.qpart + (size_t)(b % S2_NSLOT) * K * SW_MAXM) + (size_t)min(part * wq, K - 1) * SW_MAXM + t;
  const unsigned long long tag = s2_qtag(b);
  const int nw = (t < mB) ? min(S2W_QW, min(wq, K - part * wq)) : 0;   // (the words beyond S2W_QW are the product waves': S2WPollX)
  const uint64_t t0 = wall_clock64();
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < S2W_QW; ++u) ok = ok && (u >= nw || (P.v[u] & 0xFFull) == tag);
    if (__ballot(!ok) == 0ull) break;
    if ((++spins & 63u) == 0u) {
      if (ld_agent_u32(abortw) != 0u) return 0;
      if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
    }
    __builtin_amdgcn_s_sleep(1);
    if (!ok) {
#pragma unroll
      for (int u = 0; u < S2W_QW; ++u) P.v[u] = ld_agent_raw64(slot + (size_t)min(u, max(nw - 1, 0)) * SW_MAXM);
    }
  }
  double r = 0.0;
#pragma unroll
  for (int u = 0; u < S2W_QW; ++u) r += (u < nw) ? __longlong_as_double((long long)(P.v[u] & ~0xFFull)) : 0.0;
  dst[t] = r;
  return 1;
}

// With more than 2 * S2W_QW streamers (K = 40 at n = 10 000) the words S2W_QW .. of each half are the product waves' (waves 4-7,
// 256 threads, the same two threads per marker): up to S2W_QX more words per thread, same early request.
static constexpr int S2W_QX = 8;   // K <= 2 * (S2W_QW + S2W_QX) = 48 streamers
struct S2WPollX {
  unsigned long long v[S2W_QX];
};
__device__ __forceinline__ void s2w_qx_request(const SweepArgs &a, int b, int part, int t, S2WPollX &P) {
  const int K = a.K, wq = (K + 1) / 2;
  const int nx = min(wq, K - part * wq) - S2W_QW;   // (uniform)
  if (nx <= 0) return;
  const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.qpart + (size_t)(b % S2_NSLOT) * K * SW_MAXM) + (size_t)(part * wq + S2W_QW) * SW_MAXM + t;
#pragma unroll
  for (int u = 0; u < S2W_QX; ++u) P.v[u] = ld_agent_raw64(slot + (size_t)min(u, nx - 1) * SW_MAXM);
}
__device__ __forceinline__ int s2w_qx_collect(const SweepArgs &a, int b, int part, int t, int mB, S2WPollX &P, double *dst) {
  const int K = a.K, wq = (K + 1) / 2;
  const int nxu = min(wq, K - part * wq) - S2W_QW;   // (uniform)
  if (nxu <= 0) { dst[t] = 0.0; return 1; }
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.qpart + (size_t)(b % S2_NSLOT) * K * SW_MAXM) + (size_t)(part * wq + S2W_QW) * SW_MAXM + t;
  const unsigned long long tag = s2_qtag(b);
  const int nx = (t < mB) ? nxu : 0;
  const uint64_t t0 = wall_clock64();
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int u = 0; u < S2W_QX; ++u) ok = ok && (u >= nx || (P.v[u] & 0xFFull) == tag);
    if (__ballot(!ok) == 0ull) break;
    if ((++spins & 63u) == 0u) {
      if (ld_agent_u32(abortw) != 0u) return 0;
      if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
    }
    __builtin_amdgcn_s_sleep(1);
    if (!ok) {
#pragma unroll
      for (int u = 0; u < S2W_QX; ++u) P.v[u] = ld_agent_raw64(slot + (size_t)min(u, nxu - 1) * SW_MAXM);
    }
  }
  double r = 0.0;
#pragma unroll
  for (int u = 0; u < S2W_QX; ++u) r += (u < nx) ? __longlong_as_double((long long)(P.v[u] & ~0xFFull)) : 0.0;
  dst[t] = r;
  return 1;
}

// fixed-point streamers (S2WArgs::fx): one self-counting word per marker and half (digits 0-2 / 3-6), whatever the number of streamers
__device__ __forceinline__ unsigned long long s2w_qf_load(const S2WArgs &A, const unsigned long long *g) {   // the sum over the copies (counts add up in the low byte)
  unsigned long long v = ld_agent_raw64(g);
  if (A.nq > 1) v += ld_agent_raw64(g + 2 * SW_MAXM);
  if (A.nq > 2) { v += ld_agent_raw64(g + 4 * SW_MAXM); v += ld_agent_raw64(g + 6 * SW_MAXM); }
  return v;
}
__device__ __forceinline__ unsigned long long s2w_qf_request(const SweepArgs &a, const S2WArgs &A, int b, int part, int t) {
  return s2w_qf_load(A, A.qsum + ((size_t)(a.blk_begin + b) * A.nq * SW_MAXM + t) * 2 + part);
}
__device__ __forceinline__ int s2w_qf_collect(const SweepArgs &a, const S2WArgs &A, int b, int part, int t, int mB, unsigned long long v, double invS, double *dst) {
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const unsigned long long *g = A.qsum + ((size_t)(a.blk_begin + b) * A.nq * SW_MAXM + t) * 2 + part;
  const unsigned long long need = (unsigned long long)A.K3;
  const uint64_t t0 = wall_clock64();
  unsigned spins = 0;
  for (;;) {
    const bool ok = (t >= mB) || ((v & 0xFFull) == need) || (S2W_DBG(A) & 2);
    if (__ballot(!ok) == 0ull) break;
    if ((++spins & 63u) == 0u) {
      if (ld_agent_u32(abortw) != 0u) return 0;
      if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
    }
    __builtin_amdgcn_s_sleep(1);
    if (!ok) v = s2w_qf_load(A, g);
  }
  dst[t] = (t < mB) ? (double)((long long)v >> 8) * (part ? 16777216.0 * invS : invS) : 0.0;
  return 1;
}

// x[l] + x[l ^ 16] + (the same of l ^ 32): the sum over the wave's four rows of 16 lanes, in every lane -- gfx950's row swaps
// (v_permlane16_swap / v_permlane32_swap: VALU, no trip through the LDS crossbar); the additions are those of the xor butterfly
__device__ __forceinline__ double s2w_rows_sum(double v) {
  {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  return v;
}

// Block c, in four phases between raw barriers (no memory-counter drain at a barrier):
//   X  all eight waves: the cross terms  sum_{d=1..nd} G_d' delta_{c-d}  as int8 MFMA products -- two byte planes of the 16-bit
//      Gram entries (LDS, by DMA) against the four balanced base-256 digits of the steps in block-common fixed point; wave w
//      owns markers 16 w .. 16 w + 15
//   R  threads 0-127:   r0 = q - cross, rhs = (r0 + xx b0) rden + sd z - b0
//   M  waves 0-3:       d = W rhs out of registers; then the requests for block c+1: W (the same registers), the Gram planes
//                       (single LDS copy: X has read block c's), and the constants of block c+2
//      waves 4-7:       collect q_{c+1} (requested a block ago), request q_{c+2}
//   O  wave 4:          b, d, vb, the delta granules for the streamers, the digits of this block's steps
// Block c+1 begins with vmcnt(0) in waves 0-3: everything requested in M; the planes had the outputs' time only, so part of
// their latency shows in the period (~2k cycles) -- the streamers' period is longer.
template <bool FX>
__device__ __forceinline__ void s2_sequencer_winv(const SweepArgs &a, const S2WArgs &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = a.m, nb = a.blk_end - a.blk_begin, nd = A.nd;
  size_t off = 0;
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 3 * sizeof(StageBuf);
  double *q_s = reinterpret_cast<double *>(smem + off); off += (size_t)8 * SW_MAXM * sizeof(double);      // [parity][part 0..3][marker]
    double *rhs_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  double *d_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);                  // the un-rounded steps
  off += SW_MAXM * sizeof(double);   // (unused)
  int8_t *ddig_s = reinterpret_cast<int8_t *>(smem + off); off += (size_t)(S2W_DSLOTS * 4 + 1) * S2W_DROW;   // [block & 7][digit 0..3][k]; then the zero row
  int8_t *zrow_s = ddig_s + (size_t)S2W_DSLOTS * 4 * S2W_DROW;
  double *bias_s = reinterpret_cast<double *>(smem + off); off += S2W_DSLOTS * sizeof(double);            // [block & 7]: 32896 * the sum of the block's fixed-point steps
  double *scd_s = reinterpret_cast<double *>(smem + off); off += S2W_DSLOTS * sizeof(double);             // [block & 7]: 2^-(their binary point)
  int *ctrl_s = reinterpret_cast<int *>(smem + off); off += 256;
  unsigned char *far_s = smem + off;   // [distance - 4][S2W_PBYTES]: block c+1's far planes, each wave its own 8 KB (DMA after B2 of block c, read in xr_early)
  const float Sb = a.sc->Sb;
  const double dscale = (a.flags & SWF_DELTA2) ? 2.0 : 1.0;
  constexpr int NCH = (int)(sizeof(StageBuf) / 16), SPIECES = (NCH + 63) >> 6;
  static_assert(sizeof(StageBuf) % 1024 == 0, "whole 1 KiB pieces");
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };

  // ---- the requests of waves 0-3 ----
  // W: ordinary loads, one defining site (inside the block loop): the compiler's own wait covers them
  double2 w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15, w16, w17;
  auto issue_w = [&](int c) {   // 18 loads
    const unsigned char *src = reinterpret_cast<const unsigned char *>(A.winv + (size_t)(a.blk_begin + min(c, nb - 1)) * S2W_WDOUBLES) + (size_t)(wave - 4) * 2304 * 8 + (size_t)lane * 16;
#define S2W_W1(i, name) name = *reinterpret_cast<const double2 *>(src + (size_t)(i) * 1024);
    S2W_W1(0, w0) S2W_W1(1, w1) S2W_W1(2, w2) S2W_W1(3, w3) S2W_W1(4, w4) S2W_W1(5, w5) S2W_W1(6, w6) S2W_W1(7, w7) S2W_W1(8, w8)
    S2W_W1(9, w9) S2W_W1(10, w10) S2W_W1(11, w11) S2W_W1(12, w12) S2W_W1(13, w13) S2W_W1(14, w14) S2W_W1(15, w15) S2W_W1(16, w16) S2W_W1(17, w17)
#undef S2W_W1
  };
  // waves 0-3: the planes of the wave's 32 markers (two 16-marker tiles), block c, straight into the MFMA's operand registers
  // (ordinary loads, requested right after block c-1's cross terms have read the registers: a block ahead).  Always all 24
  // loads: the wait counts rely on it.
  s2_v4i a1p0, a1p1, a1p2, a1p3, a2p0, a2p1, a2p2, a2p3, a3p0, a3p1, a3p2, a3p3;   // tile 0, distance d: {plane 0, plane 1} x {k < 64, k >= 64}
  s2_v4i b1p0, b1p1, b1p2, b1p3, b2p0, b2p1, b2p2, b2p3, b3p0, b3p1, b3p2, b3p3;   // tile 1
  // (Requested together after the block's right-hand sides, the distance-2 / 3 planes first: xr_early waits for those only.  Requesting
  // them a whole block ahead -- right after xr_early has read the previous ones -- made the sequencer alone faster, 2.53 -> 2.34 us per
  // block, and the sweep slower, 2.77 -> 2.83: sixteen load instructions in front of B0 wait in the memory pipeline's queue behind the
  // q polls, and B0 is on the block's critical path.)
#define S2W_A1(d_, o_, r0, r1, r2, r3) { const unsigned char *p_ = A.gxt[(d_) - 1] + boff + (o_); \
      r0 = *reinterpret_cast<const s2_v4i *>(p_); r1 = *reinterpret_cast<const s2_v4i *>(p_ + 1024); \
      r2 = *reinterpret_cast<const s2_v4i *>(p_ + 2048); r3 = *reinterpret_cast<const s2_v4i *>(p_ + 3072); }
  auto issue_a1 = [&](int c) {
    const size_t boff = (size_t)(a.blk_begin + min(c, nb - 1)) * S2W_PBYTES + (size_t)wave * 8192 + (size_t)lane * 16;
    S2W_A1(1, 0, a1p0, a1p1, a1p2, a1p3) S2W_A1(1, 4096, b1p0, b1p1, b1p2, b1p3)
  };
  auto issue_a23 = [&](int c) {
    const size_t boff = (size_t)(a.blk_begin + min(c, nb - 1)) * S2W_PBYTES + (size_t)wave * 8192 + (size_t)lane * 16;
    S2W_A1(2, 0, a2p0, a2p1, a2p2, a2p3) S2W_A1(2, 4096, b2p0, b2p1, b2p2, b2p3)
    S2W_A1(3, 0, a3p0, a3p1, a3p2, a3p3) S2W_A1(3, 4096, b3p0, b3p1, b3p2, b3p3)
  };
#undef S2W_A1
  auto issue_stage = [&](int c) {   // the constants of block c: six pieces
    const int cc = min(c, nb - 1);
    const unsigned char *ssrc = reinterpret_cast<const unsigned char *>(a.ps.blocks + a.blk_begin + cc) + (size_t)lane * 16;
    unsigned char *sl = reinterpret_cast<unsigned char *>(&stage[cc % 3]);
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int pc = min(wave + 4 * i, SPIECES - 1); s2w_dma16(ssrc + (size_t)pc * 1024, sl + (size_t)pc * 1024); }
  };

  // ---- prologue ----
  if (tid == 0) ctrl_s[0] = 1;
  for (int i = tid; i < (int)((S2W_DSLOTS * 4 + 1) * S2W_DROW / 4); i += S2W_THREADS) reinterpret_cast<uint32_t *>(ddig_s)[i] = 0u;
  if (tid < S2W_DSLOTS) { bias_s[tid] = 0.0; scd_s[tid] = 1.0; }
  if (wave < 4) { issue_stage(0); issue_stage(1); }
  __syncthreads();
  S2WPoll QP;
  constexpr bool fx = FX;
  const double invSq = fx ? s3_pow2(-a.sc->e3_sh) : 1.0;
  unsigned long long qf = 0ull;
  if (wave >= 8) {
    const int gpart = (tid - 512) >> 7, gt = tid & 127;
    if constexpr (fx) {
      qf = s2w_qf_request(a, A, 0, gpart, gt);
      if (!s2w_qf_collect(a, A, 0, gpart, gt, blk_m(0), qf, invSq, q_s + (size_t)gpart * SW_MAXM)) ctrl_s[0] = 0;
    } else {
      s2w_q_request(a, 0, gpart, gt, QP);
      if (!s2w_q_collect(a, 0, gpart, gt, blk_m(0), QP, q_s + (size_t)gpart * SW_MAXM)) ctrl_s[0] = 0;
    }
  } else if (wave >= 4 && fx) {   // (parts 2 and 3 stay zero)
    q_s[(size_t)(2 + ((tid - 256) >> 7)) * SW_MAXM + (tid & 127)] = 0.0; q_s[(size_t)(6 + ((tid - 256) >> 7)) * SW_MAXM + (tid & 127)] = 0.0;
  } else if (wave >= 4) {   // block 0's words beyond the pollers' (their own registers: the loop's have one defining site)
    const int xpart = (tid - 256) >> 7, xt = tid & 127;
    S2WPollX Q0;
    s2w_qx_request(a, 0, xpart, xt, Q0);
    if (!s2w_qx_collect(a, 0, xpart, xt, blk_m(0), Q0, q_s + (size_t)(2 + xpart) * SW_MAXM)) ctrl_s[0] = 0;
  }
  double sum_d = 0.0, sum_b2 = 0.0;

  // the phases every thread takes part in
  // X + R fused (waves 0-3): wave w forms the cross terms of markers 32 w .. 32 w + 31 (two tiles of 16) and, on its lanes
  // 0-15, their right-hand sides.  MFMA roles: A = the four digit rows of the steps (rows 4-15 zero), B = a byte plane of a
  // tile's 16 Gram columns, so lane j < 16 ends up with the four digit sums of ITS marker in its four accumulator registers
  // (no cross-lane step).
  // Two parts: what does not need block c-1's steps -- the distance-2 / 3 terms and the slab dots, folded into a partial r0 right after
  // the previous block's product (xr_early, while wave 8 writes that block's steps) -- and the distance-1 term and the right-hand sides
  // once those steps' digits exist (xr_late, between B0 and B2: the only part on the block's critical path).
  double rp0 = 0.0, rp1 = 0.0;   // lanes 0-15: q - (distance-2 / 3 cross terms) of the wave's two markers of the NEXT block
#define S2W_X1(CR, r0, r1, r2, r3) { \
      s2_v4i acc0 = {0, 0, 0, 0}, acc1 = acc0; \
      acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(dv0, r0, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(dv0, r2, acc1, 0, 0, 0); \
      acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(dv1, r1, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(dv1, r3, acc1, 0, 0, 0); \
      /* lane j < 16: acc_P[n] = sum_k digit_n[k] plane_P[marker j of the tile][k] */ \
      const double t3 = fma(256.0, (double)acc1[3], (double)acc0[3]), t2 = fma(256.0, (double)acc1[2], (double)acc0[2]); \
      const double t1 = fma(256.0, (double)acc1[1], (double)acc0[1]), t0 = fma(256.0, (double)acc1[0], (double)acc0[0]); \
      const double val = fma(256.0, fma(256.0, fma(256.0, t3, t2), t1), t0); \
      CR = fma(val + bias_, scd_, CR); }   /* (the planes are biased by 128: bias = (128 + 256 * 128) sum_k q_k; scd = 2^-sh) */
#define S2W_XD(d_, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3) if (nd >= (d_) && c - (d_) >= 0) { \
      const int slot = (c - (d_)) & (S2W_DSLOTS - 1); \
      const int8_t *dg = (i16 < 4 ? ddig_s + (size_t)(slot * 4 + i16) * S2W_DROW : zrow_s) + 16 * kg; \
      const s2_v4i dv0 = *reinterpret_cast<const s2_v4i *>(dg), dv1 = *reinterpret_cast<const s2_v4i *>(dg + 64); \
      const double bias_ = bias_s[slot], scd_ = scd_s[slot]; \
      S2W_X1(cross0, ra0, ra1, ra2, ra3) S2W_X1(cross1, rb0, rb1, rb2, rb3) }
#define S2W_XDR(d_, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3) if (c - (d_) >= 0) { \
      const int slot = (c - (d_)) & (S2W_DSLOTS - 1); \
      const int8_t *dg = (i16 < 4 ? ddig_s + (size_t)(slot * 4 + i16) * S2W_DROW : zrow_s) + 16 * kg; \
      const s2_v4i dv0 = *reinterpret_cast<const s2_v4i *>(dg), dv1 = *reinterpret_cast<const s2_v4i *>(dg + 64); \
      const double bias_ = bias_s[slot], scd_ = scd_s[slot]; \
      S2W_X1(cross0, ra0, ra1, ra2, ra3) S2W_X1(cross1, rb0, rb1, rb2, rb3) }
  auto xr_early = [&](int c, int mB) {   // (block c >= 1; q_c was collected before the barrier that precedes this)
    const int i16 = lane & 15, kg = lane >> 4;
    double cross0 = 0.0, cross1 = 0.0;
    S2W_XD(2, a2p0, a2p1, a2p2, a2p3, b2p0, b2p1, b2p2, b2p3)
    S2W_XD(3, a3p0, a3p1, a3p2, a3p3, b3p0, b3p1, b3p2, b3p3)
    for (int d = S2W_NEARD + 1; d <= nd; ++d) {   // the far distances: operands from this wave's own 8 KB of the LDS copy
      const unsigned char *fp = far_s + (size_t)(d - S2W_NEARD - 1) * S2W_PBYTES + (size_t)wave * 8192 + (size_t)lane * 16;
      const s2_v4i f0 = *reinterpret_cast<const s2_v4i *>(fp), f1 = *reinterpret_cast<const s2_v4i *>(fp + 1024);
      const s2_v4i f2 = *reinterpret_cast<const s2_v4i *>(fp + 2048), f3 = *reinterpret_cast<const s2_v4i *>(fp + 3072);
      const s2_v4i g0 = *reinterpret_cast<const s2_v4i *>(fp + 4096), g1 = *reinterpret_cast<const s2_v4i *>(fp + 5120);
      const s2_v4i g2 = *reinterpret_cast<const s2_v4i *>(fp + 6144), g3 = *reinterpret_cast<const s2_v4i *>(fp + 7168);
      S2W_XDR(d, f0, f1, f2, f3, g0, g1, g2, g3)
    }
    if (lane < 16) {
      const double *qq = q_s + (size_t)(c & 1) * 4 * SW_MAXM;
      const int t0 = 32 * wave + lane, t1 = t0 + 16;
      rp0 = (t0 < mB) ? ((qq[t0] + qq[SW_MAXM + t0]) + (qq[2 * SW_MAXM + t0] + qq[3 * SW_MAXM + t0])) - cross0 : 0.0;
      rp1 = (t1 < mB) ? ((qq[t1] + qq[SW_MAXM + t1]) + (qq[2 * SW_MAXM + t1] + qq[3 * SW_MAXM + t1])) - cross1 : 0.0;
    }
  };
  auto xr_late = [&](int c, int mB) {
    const int i16 = lane & 15, kg = lane >> 4;
    double cross0 = 0.0, cross1 = 0.0;
    S2W_XD(1, a1p0, a1p1, a1p2, a1p3, b1p0, b1p1, b1p2, b1p3)
    if (lane < 16) {
      const StageBuf &st = stage[c % 3];
      if (c == 0) {   // block 0 has no earlier blocks; its slab dots arrive with the first barrier
        const double *qq = q_s;
        const int t0 = 32 * wave + lane, t1 = t0 + 16;
        rp0 = (t0 < mB) ? ((qq[t0] + qq[SW_MAXM + t0]) + (qq[2 * SW_MAXM + t0] + qq[3 * SW_MAXM + t0])) : 0.0;
        rp1 = (t1 < mB) ? ((qq[t1] + qq[SW_MAXM + t1]) + (qq[2 * SW_MAXM + t1] + qq[3 * SW_MAXM + t1])) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 32 * wave + 16 * u + lane;
        double rhs = 0.0;
        if (t < mB) {
          const double r0 = (u ? rp1 : rp0) - (u ? cross1 : cross0);
          rhs = fma(r0 + (double)st.xxb0[t], st.rden[t], st.sdz1[t]) - (double)st.b0[t];
        }
        rhs_s[t] = rhs;
      }
    }
  };
#undef S2W_XD
#undef S2W_XDR
#undef S2W_X1
  auto issue_far = [&](int c) {   // eight 1 KiB pieces per far distance: this wave's 32 markers of block c
    const size_t boff = (size_t)(a.blk_begin + min(c, nb - 1)) * S2W_PBYTES + (size_t)wave * 8192 + (size_t)lane * 16;
    for (int d = S2W_NEARD + 1; d <= nd; ++d) {
      const unsigned char *src = A.gxt[d - 1] + boff;
      unsigned char *dst = far_s + (size_t)(d - S2W_NEARD - 1) * S2W_PBYTES + (size_t)wave * 8192;
#pragma unroll
      for (int k = 0; k < 8; ++k) s2w_dma16(src + (size_t)k * 1024, dst + (size_t)k * 1024);
    }
  };

  if (wave < 4) {
    // ================= waves 0-3: cross terms and right-hand sides =================
    S2STAMP_DECL;
    // (the loop starts at "block -1", which only issues requests: the plane registers then have ONE defining site, inside the
    // loop, and the compiler needs no second register set to carry a prologue's values into it)
#pragma clang loop unroll(disable)
    for (int c = -1; c < nb; ++c) {
      if (c >= 0) {
        S2WSTAMP0(0);
        S2W_BAR();                                         // B0: + the digits of block c-1 (q_c came with B3 of block c-1)
        S2WSTAMP0(1);
        if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }
        xr_late(c, blk_m(c));
        S2WSTAMP0(2);
        S2W_BAR();                                         // B2: rhs
      }
      issue_stage(c + 2);   // (two DMA instructions -- and eight per far distance --, then twenty-four loads: these waves sit in the memory pipeline's queue while waves 4-7 form the product)
      issue_far(c + 1);
      issue_a23(c + 1);
      issue_a1(c + 1);
      if (c >= 0) {
        S2WSTAMP0(3);
        S2W_BAR();                                         // B3: d
      }
      S2WSTAMP0(4);
      // in flight on this wave's in-order memory counter, oldest first: the two DMA instructions (constants of block c+2; block c+1's are
      // a block older), the far planes' DMA, the sixteen distance-2 / 3 loads of block c+1, the eight distance-1 loads.  All but the last eight: the constants
      // (no compiler-visible result to wait for) and the planes xr_early(c+1) reads; the distance-1 planes are xr_late's, a phase later.
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      S2WSTAMP0(5);
      if (c >= 0 && c + 1 < nb) xr_early(c + 1, blk_m(c + 1));   // (beside wave 8's outputs of block c)
      S2WSTAMP0(7);
    }
    S2WSTAMP0_FLUSH();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave < 8) {
    // ================= waves 4-7: the product =================
    const int wv = wave - 4, r16 = lane & 15, cq = lane >> 4;
    const int xpart = (tid - 256) >> 7, xt = tid & 127;
    S2WPollX QX;
#pragma clang loop unroll(disable)
    for (int c = -1; c < nb; ++c) {
      double acc_hi = 0.0, acc_lo = 0.0;
      if (c >= 0) {
        S2W_BAR();                                         // B0
        if (ctrl_s[0] == 0) return;
        S2W_BAR();                                         // B2: rhs
#define S2W_TILE(q, A_, B_) { const bool hi_ = (q) < 8 - wv; const int J_ = hi_ ? (q) : (q) - (8 - wv); \
          const double2 x0_ = *reinterpret_cast<const double2 *>(rhs_s + 16 * J_ + 4 * cq), x1_ = *reinterpret_cast<const double2 *>(rhs_s + 16 * J_ + 4 * cq + 2); \
          double t_ = A_.x * x0_.x; t_ = fma(A_.y, x0_.y, t_); t_ = fma(B_.x, x1_.x, t_); t_ = fma(B_.y, x1_.y, t_); \
          if (hi_) acc_hi += t_; else acc_lo += t_; }
        S2W_TILE(0, w0, w1) S2W_TILE(1, w2, w3) S2W_TILE(2, w4, w5) S2W_TILE(3, w6, w7) S2W_TILE(4, w8, w9)
        S2W_TILE(5, w10, w11) S2W_TILE(6, w12, w13) S2W_TILE(7, w14, w15) S2W_TILE(8, w16, w17)
#undef S2W_TILE
        acc_hi = s2w_rows_sum(acc_hi); acc_lo = s2w_rows_sum(acc_lo);
        if (cq == 0) d_s[16 * (7 - wv) + r16] = acc_hi;
        if (cq == 1) d_s[16 * wv + r16] = acc_lo;
        if constexpr (!fx) if (c + 1 < nb) { if (!s2w_qx_collect(a, c + 1, xpart, xt, blk_m(c + 1), QX, q_s + (size_t)(((c + 1) & 1) * 4 + 2 + xpart) * SW_MAXM)) ctrl_s[0] = 0; }
        S2W_BAR();                                         // B3: d
      }
      if (!(S2W_DBG(A) & 32) || c < 0) issue_w(c + 1);     // (into the registers just read; the wave waits in the memory pipeline's queue while wave 8 writes the outputs)
      if constexpr (!fx) s2w_qx_request(a, min(c + 2, nb - 1), xpart, xt, QX);
    }
  } else {
    // ================= waves 8-11: the dots of the next block; wave 8: the outputs of this one =================
    const int gpart = (tid - 512) >> 7, gt = tid & 127;
    S2STAMP_DECL;
    if constexpr (fx) qf = s2w_qf_request(a, A, min(1, nb - 1), gpart, gt); else s2w_q_request(a, min(1, nb - 1), gpart, gt, QP);
    for (int c = 0; c < nb; ++c) {
      const int blk = a.blk_begin + c, j0 = blk * m;
      const int mB = blk_m(c);
      const bool have_next = c + 1 < nb;
      S2WSTAMP(5);
      S2W_BAR();                                          // B0
      S2WSTAMP(0);
      if (ctrl_s[0] == 0) return;
      S2W_BAR();                                          // B2
      S2WSTAMP(2);
      if constexpr (fx) {
        if (have_next) { if (!s2w_qf_collect(a, A, c + 1, gpart, gt, blk_m(c + 1), qf, invSq, q_s + (size_t)(((c + 1) & 1) * 4 + gpart) * SW_MAXM)) ctrl_s[0] = 0; }
        qf = s2w_qf_request(a, A, min(c + 2, nb - 1), gpart, gt);
      } else {
        if (have_next) { if (!s2w_q_collect(a, c + 1, gpart, gt, blk_m(c + 1), QP, q_s + (size_t)(((c + 1) & 1) * 4 + gpart) * SW_MAXM)) ctrl_s[0] = 0; }
        s2w_q_request(a, min(c + 2, nb - 1), gpart, gt, QP);   // (every pass, the last ones too: the compiler's wait counts merge over all paths)
      }
      S2WSTAMP(3);
      S2W_BAR();                                          // B3: d
      S2WSTAMP(4);
      if (wave == 8) {
        // the part the next block's cross terms wait for: the steps, their granules for the streamers, their digits
        const StageBuf &st = stage[c % 3];
        unsigned long long *gslot = a.dgran + (size_t)(c % S2_NSLOT) * SW_MAXM;
        float dl_own[2] = {0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = 64 * q + lane;
          float dl = 0.0f;
          if (t < mB) {
            const float b0 = st.b0[t];
            const float bn = (float)(d_s[t] + (double)b0);
            dl = (S2W_DBG(A) & 1) ? 0.0f : (bn - b0) * (float)dscale;
          }
          dl_own[q] = dl;
        }
        const uint32_t exmax = wave_max_u32(max((__float_as_uint(dl_own[0]) >> 23) & 0xFFu, (__float_as_uint(dl_own[1]) >> 23) & 0xFFu));
        if (lane < mB) st_agent_raw64(gslot + lane, s2_dgranule(c, exmax, dl_own[0]));
        if (64 + lane < mB) st_agent_raw64(gslot + 64 + lane, s2_dgranule(c, exmax, dl_own[1]));
        // the steps in block-common fixed point, |q| < 2^30, as four balanced base-256 digits: the next blocks' cross terms
        const int sh = 156 - (int)exmax;
        const double scq = __hiloint2double((1023 + sh) << 20, 0);
        int sq2 = 0;   // this lane's two fixed-point steps, |.| < 2^30 each
        int8_t *dg = ddig_s + (size_t)((c & (S2W_DSLOTS - 1)) * 4) * S2W_DROW;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = 64 * q + lane;
          const int qi = (int)rint((double)dl_own[q] * scq);
          sq2 += qi;
          const uint32_t u = ((uint32_t)qi + 0x00808080u) ^ 0x00808080u;
          dg[t] = (int8_t)(u & 0xFFu); dg[S2W_DROW + t] = (int8_t)((u >> 8) & 0xFFu); dg[2 * S2W_DROW + t] = (int8_t)((u >> 16) & 0xFFu); dg[3 * S2W_DROW + t] = (int8_t)(u >> 24);
        }
        // their sum over the block, as two DPP reductions of 16-bit halves (a 64-bit butterfly through the LDS crossbar cost ~ 1k cycles here)
        const long long sq = ((long long)wave_sum_i32(sq2 >> 16) << 16) + (long long)wave_sum_i32(sq2 & 0xFFFF);
        if (lane == 0) { bias_s[c & (S2W_DSLOTS - 1)] = 32896.0 * (double)sq; scd_s[c & (S2W_DSLOTS - 1)] = __hiloint2double((1023 - sh) << 20, 0); }
        if (lane == 0 && A.npf > 0) st_agent_u32(a.xflags + (size_t)a.K * SW_FLAG_STRIDE + 1, (uint32_t)(c + 1));   // progress, for the prefetchers
      } else if (!(S2W_DBG(A) & 128)) {
        // ... and what nobody in the sweep waits for, on the other three waves of the group: wave 9 the effects (and the sums), wave 10
        // the indicators, wave 11 the variances.  (stage[c % 3] and d_s are read here until the next B0: the constants of block c+3 land
        // in this buffer after B2 of block c+1, the next product writes d_s after that barrier too.)
        const StageBuf &st = stage[c % 3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = 64 * q + lane;
          if (t < mB) {
            const float b0 = st.b0[t];
            const float bn = (S2W_DBG(A) & 1) ? b0 : (float)(d_s[t] + (double)b0);
            if (wave == 9) { a.b[j0 + t] = bn; sum_d += 1.0; sum_b2 = fma((double)bn, (double)bn, sum_b2); }
            else if (wave == 10) a.d[j0 + t] = 1.0f;
            else if (a.flags & SWF_VB_VEC) a.vb[j0 + t] = (float)((double)(Sb + bn * bn) / st.chi[t]);
          }
        }
      }
    }
    S2WSTAMP_FLUSH();
    if (wave == 9) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
      if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; }
    }
  }
}
#undef S2W_BAR

// ------------------------------------------------------------------------------------------------------------------
// The affine sweeps' own streamers: k_sweep3's arithmetic (sweep3.hip.h: the residual rows as 55-bit fixed point in registers
// on one per-sweep scale, the update's int32 MFMA accumulators added straight into them, the dots as exact integers summed by
// self-counting 64-bit atomics -- one pair of words per marker for the sequencer instead of K) on k_sweep2's data flow: block j's
// steps come from the sequencer (delta granules), the dots of block j + L are formed once they have been applied.
//   waves 0-1 (update): poll delta_j, its digits, the update with tile j (a two-wave sync through an LDS counter in between),
//                       the residual's digits; waves 2-7 (dots): the dots of block j + L against those digits, the atomics.
// One workgroup barrier per block.  128 rows per workgroup (twice k_sweep2's workgroups): L + 1 tiles of 18 KB stay in LDS.
// ------------------------------------------------------------------------------------------------------------------
static constexpr int S2W_FXR = 128;
__host__ __device__ inline size_t s2w_fx_lds(int L) {
  const size_t Rp = S2W_FXR + 16;
  return (size_t)(L + 1) * SW_MAXM * Rp + 2 * 16 * Rp + 2 * 16 * (size_t)S2_DP + (size_t)2 * 64 * S3_OS * 4 + (size_t)8 * 32 * S3_OS * 4 + 64;
}
__device__ __forceinline__ void s2w_streamer_fx(const SweepArgs &a, const S2WArgs &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, grp = lane >> 4;
  const int w = (int)blockIdx.x;
  constexpr int R3 = S2W_FXR, Rp = R3 + 16, NU = 2, ND = 6, cprs = 3;
  const int m = a.m, R = a.R, L = a.lag, NR = L + 1;
  const int slab = w / A.sub, hsub = w - slab * A.sub;
  const int nb = a.blk_end - a.blk_begin;
  const int8_t *Xs = reinterpret_cast<const int8_t *>(a.X) + (size_t)slab * a.p * R + (size_t)hsub * R3;   // marker j: Xs + j * R
  const int64_t row0 = (int64_t)slab * R + (int64_t)hsub * R3;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const size_t tile_b = (size_t)SW_MAXM * Rp;
  int8_t *tile0 = reinterpret_cast<int8_t *>(smem);
  size_t off = (size_t)NR * tile_b;
  int8_t *edig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * Rp;     // [parity][n][row]
  int8_t *ddig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * S2_DP;  // [parity][n][marker]
  int *outu = reinterpret_cast<int *>(smem + off); off += (size_t)2 * 64 * S3_OS * 4;     // [update wave][row 64][n]
  int *outd = reinterpret_cast<int *>(smem + off); off += (size_t)8 * 32 * S3_OS * 4;     // [wave][marker 32][n]
  uint32_t *ctl_s = reinterpret_cast<uint32_t *>(smem + off);                            // [0] failure, [1] overflow, [2] update waves' counter
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  auto blk_j0 = [&](int b) { return (a.blk_begin + b) * m; };
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };

  for (int i = tid; i < (int)((2 * 16 * Rp + 2 * 16 * S2_DP) / 4); i += SW_THREADS) reinterpret_cast<uint32_t *>(edig0)[i] = 0u;   // (adjacent)
  if (tid < 16) ctl_s[tid] = 0u;
  long long e_own = 0;
  const bool upd = wave < NU;
  if (upd) e_own = __double2ll_rn(a.e[row0 + 64 * wave + lane] * S);

  // tile moves: two 16-byte chunks per thread (1024 chunks per tile), loads unconditional (clamped), stores guarded; two tiles in
  // flight in two register sets (tile t travels in set t & 1)
  s3_u4 ta0 = {0, 0, 0, 0}, ta1 = ta0, tb0 = ta0, tb1 = ta0;
#define S2F_ISSUE1(u, name) { const int cc_ = min(tid + (u) * SW_THREADS, tot_ - 1); const int jj_ = min(cc_ >> cprs, mBt_ - 1), ii_ = cc_ & ((1 << cprs) - 1); \
    name = __builtin_nontemporal_load(reinterpret_cast<const s3_u4 *>(Xs + (size_t)(j0t_ + jj_) * R + ii_ * 16)); }
#define S2F_TILE_ISSUE(b_, T0, T1) do { const int bb_ = min((b_), nb - 1); const int j0t_ = blk_j0(bb_), mBt_ = blk_m(bb_), tot_ = m << cprs; S2F_ISSUE1(0, T0) S2F_ISSUE1(1, T1) } while (0)
#define S2F_COMMIT1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) { const int jj_ = c_ >> cprs, ii_ = c_ & ((1 << cprs) - 1); \
    *reinterpret_cast<s3_u4 *>(dst_ + (size_t)jj_ * Rp + ii_ * 16) = name; } }
#define S2F_TILE_COMMIT(b_, T0, T1) do { int8_t *dst_ = tile0 + (size_t)((b_) % NR) * tile_b; const int tot_ = m << cprs; S2F_COMMIT1(0, T0) S2F_COMMIT1(1, T1) } while (0)

  // the slab dots of block b against the residual digits in edig: markers in groups of 16, groups gm and gm + ND together
  auto dots = [&](int b, const int8_t *edig) {
    const int8_t *tile = tile0 + (size_t)(b % NR) * tile_b;
    for (int gm = wave - NU; 16 * gm < m; gm += 2 * ND) {
      const int gm2 = gm + ND;
      const bool two = 16 * gm2 < m;
      const int8_t *bp = edig + (size_t)m16 * Rp + 16 * grp;
      const int8_t *ap = tile + (size_t)(16 * gm + m16) * Rp + 16 * grp;
      const int8_t *ap2 = tile + (size_t)(16 * (two ? gm2 : gm) + m16) * Rp + 16 * grp;
      s2_v4i acc = {0, 0, 0, 0}, acc2 = acc;
#pragma unroll
      for (int r = 0; r < R3; r += 64) {
        const s2_v4i bv = *reinterpret_cast<const s2_v4i *>(bp + r);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap + r), bv, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap2 + r), bv, acc2, 0, 0, 0);
      }
      int *od = outd + (size_t)wave * 32 * S3_OS;
      if (m16 < 8) {      // lane: digit n = m16 of markers 16 gm + 4 grp + reg (rows 0..15 of the scratch) and of group gm2 (rows 16..31)
        int *op = od + (size_t)(4 * grp) * S3_OS + m16;
        op[0] = acc[0]; op[S3_OS] = acc[1]; op[2 * S3_OS] = acc[2]; op[3 * S3_OS] = acc[3];
        op[16 * S3_OS] = acc2[0]; op[17 * S3_OS] = acc2[1]; op[18 * S3_OS] = acc2[2]; op[19 * S3_OS] = acc2[3];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane < (two ? 32 : 16)) {
        const int4 o0 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3_OS);
        const int4 o1 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3_OS + 4);
        const long long lo = (long long)o0.x + ((long long)o0.y << 8) + ((long long)o0.z << 16);
        const long long hi = (long long)o0.w + ((long long)o1.x << 8) + ((long long)o1.y << 16) + ((long long)o1.z << 24);
        const int mk = 16 * ((lane < 16) ? gm : gm2) + (lane & 15);
        unsigned long long *qs = A.qsum + ((((size_t)(a.blk_begin + b) * A.nq + (w & (A.nq - 1))) * SW_MAXM + mk) * 2);
        __hip_atomic_fetch_add((gu64_t *)qs, (unsigned long long)((lo << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add((gu64_t *)(qs + 1), (unsigned long long)((hi << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is read before the next pass overwrites it
    }
  };

  // ---- prologue: tiles 0 .. L into LDS, tiles L+1 and L+2 in flight; the dots of blocks 0 .. L-1 against the starting residual ----
  for (int b = 0; b <= L && b < nb; ++b) { S2F_TILE_ISSUE(b, ta0, ta1); S2F_TILE_COMMIT(b, ta0, ta1); }
  // (set of tile t is t & 1)
  if ((L + 1) & 1) { S2F_TILE_ISSUE(L + 1, tb0, tb1); S2F_TILE_ISSUE(L + 2, ta0, ta1); }
  else { S2F_TILE_ISSUE(L + 1, ta0, ta1); S2F_TILE_ISSUE(L + 2, tb0, tb1); }
  if (upd) {
    if ((unsigned long long)(e_own + (1ll << 54)) >> 55) ctl_s[1] = 1u;
    s3_put_digits7(e_own, edig0 + (size_t)16 * Rp + 64 * wave + lane, Rp);      // parity 1: block 0 writes parity 0
  }
  unsigned long long pre = 0ull;   // early-requested granule of the next block's step (update waves: thread = marker)
  if (upd) pre = ld_agent_raw64(a.dgran + tid);
  __syncthreads();
  if (!upd) for (int b = 0; b < L && b < nb; ++b) dots(b, edig0 + (size_t)16 * Rp);

  // one block; T0, T1: the register set of tile j+L+1 (committed here) and then of tile j+L+3 (requested here)
  auto step = [&](int j, s3_u4 &T0, s3_u4 &T1) -> bool {
    const int mB = blk_m(j), par = j & 1;
    int8_t *edig = edig0 + (size_t)par * 16 * Rp;
    int8_t *ddig = ddig0 + (size_t)par * 16 * S2_DP;
    if (upd) {
      // the steps of block j: one granule per marker, polled by the thread that needs it
      int bad = 0;
      double qd = 0.0;
      if (tid < mB) {
        const unsigned long long *g = a.dgran + (size_t)(j % S2_NSLOT) * SW_MAXM + tid;
        const uint64_t t0 = wall_clock64();
        unsigned spins = 0;
        unsigned long long v = pre;
        for (;;) {
          if (s2_dgranule_is(v, j) || (S2W_DBG(A) & 4)) { if (S2W_DBG(A) & 4) v = 0ull; break; }
          v = ld_agent_raw64(g);
          if (s2_dgranule_is(v, j)) break;
          if ((++spins & 63u) == 0u) {
            if (ld_agent_u32(abortw) != 0u) { bad = 1; break; }
            if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); bad = 1; break; }
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!bad) qd = rint((double)__uint_as_float((uint32_t)v) * S);
      }
      if (bad) ctl_s[0] = 1u;
      if (!(fabs(qd) < 18014398509481984.0)) { ctl_s[1] = 1u; qd = 0.0; }       // 2^54
      s3_put_digits7((long long)qd, ddig + tid, S2_DP);
      // the two update waves need each other's digits: an LDS counter (no workgroup barrier: the dots waves are busy)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(&ctl_s[2], 1u);
      while (__hip_atomic_load(&ctl_s[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u * (uint32_t)(j + 1)) __builtin_amdgcn_s_sleep(0);
      // ---- slab update: out[row][n] = sum_markers x[row][marker] * digit_n(step[marker]) ----
      const int8_t *tile = tile0 + (size_t)(j % NR) * tile_b;
      const int rowoff = 4 * (16 * wave + m16);
      s2_v4i acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      for (int s0 = 0; s0 < mB; s0 += 64) {
        const int8_t *tp = tile + __mul24(s0 + 4 * grp, Rp) + rowoff;
        uint32_t c[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            c[u][q] = *reinterpret_cast<const uint32_t *>(tp + (16 * u + q) * Rp);
        const int8_t *bp = ddig + (size_t)m16 * S2_DP + s0 + 4 * grp;
        const s2_v4i bv = {*reinterpret_cast<const int *>(bp), *reinterpret_cast<const int *>(bp + 16),
                           *reinterpret_cast<const int *>(bp + 32), *reinterpret_cast<const int *>(bp + 48)};
        uint32_t rw[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t t0 = __builtin_amdgcn_perm(c[u][1], c[u][0], 0x05010400u), t1 = __builtin_amdgcn_perm(c[u][1], c[u][0], 0x07030602u);
          const uint32_t t2 = __builtin_amdgcn_perm(c[u][3], c[u][2], 0x05010400u), t3 = __builtin_amdgcn_perm(c[u][3], c[u][2], 0x07030602u);
          rw[0][u] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); rw[1][u] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
          rw[2][u] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); rw[3][u] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
        }
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{(int)rw[0][0], (int)rw[0][1], (int)rw[0][2], (int)rw[0][3]}, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{(int)rw[1][0], (int)rw[1][1], (int)rw[1][2], (int)rw[1][3]}, bv, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{(int)rw[2][0], (int)rw[2][1], (int)rw[2][2], (int)rw[2][3]}, bv, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{(int)rw[3][0], (int)rw[3][1], (int)rw[3][2], (int)rw[3][3]}, bv, acc3, 0, 0, 0);
      }
      int *ou = outu + (size_t)wave * 64 * S3_OS;
      if (m16 < 8) {      // lane: digit n = m16; acc_k[reg] belongs to local row 4 (4 grp + reg) + k
        int *op = ou + (size_t)(4 * (4 * grp)) * S3_OS + m16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          op[(4 * reg + 0) * S3_OS] = acc0[reg]; op[(4 * reg + 1) * S3_OS] = acc1[reg];
          op[(4 * reg + 2) * S3_OS] = acc2[reg]; op[(4 * reg + 3) * S3_OS] = acc3[reg];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes (in order; no other wave reads this scratch)
      {
        const int4 o0 = *reinterpret_cast<const int4 *>(ou + (size_t)lane * S3_OS);
        const int4 o1 = *reinterpret_cast<const int4 *>(ou + (size_t)lane * S3_OS + 4);
        long long v = (long long)o0.x + ((long long)o0.y << 8) + ((long long)o0.z << 16) + ((long long)o0.w << 24);
        v += ((long long)o1.x << 32) + ((long long)o1.y << 40) + ((long long)o1.z << 48);
        e_own -= v;
      }
      if ((unsigned long long)(e_own + (1ll << 54)) >> 55) ctl_s[1] = 1u;       // left the 55-bit range
      s3_put_digits7(e_own, edig + 64 * wave + lane, Rp);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return false; }
    // tile j+L+1 (in registers for two iterations) lands in the slot of tile j, spent; the first request for the steps of block
    // j+1 (older than the tile loads on the in-order memory counter); the loads of tile j+L+3 into the registers just freed
    if (j + L + 1 < nb) S2F_TILE_COMMIT(j + L + 1, T0, T1);
    if (upd) pre = ld_agent_raw64(a.dgran + (size_t)((j + 1) % S2_NSLOT) * SW_MAXM + tid);
    S2F_TILE_ISSUE(j + L + 3, T0, T1);
    if (!upd && j + L < nb) dots(j + L, edig);
    return true;
  };
  // (tile j+L+1 travels in set (j+L+1) & 1)
  for (int j = 0; j < nb; j += 2) {
    if ((L + 1) & 1) { if (!step(j, tb0, tb1)) return; if (j + 1 < nb && !step(j + 1, ta0, ta1)) return; }
    else { if (!step(j, ta0, ta1)) return; if (j + 1 < nb && !step(j + 1, tb0, tb1)) return; }
  }
  __syncthreads();
  if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
  if (ctl_s[1] && tid == 0) a.sc->error = 2u;
  if (upd) a.e[row0 + 64 * wave + lane] = (double)e_own * invS;
#undef S2F_ISSUE1
#undef S2F_TILE_ISSUE
#undef S2F_COMMIT1
#undef S2F_TILE_COMMIT
}

// A prefetcher: a workgroup on the sequencer's XCD (workgroups go round-robin over the eight XCDs) that reads W and the Gram
// planes of the blocks a little ahead of the sequencer, so that the sequencer's own loads -- 172 KB per block through ONE CU,
// whose outstanding-miss capacity bounds it at ~40 GB/s from HBM -- find their lines in the XCD's L2.
__device__ __forceinline__ void s2w_prefetcher(const SweepArgs &a, const S2WArgs &A, int i) {
  const int tid = threadIdx.x, nb = a.blk_end - a.blk_begin;
  const uint32_t *progw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE + 1;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const int AHEAD = A.ahead;
  uint32_t sink = 0u;
  for (int c = i; c < nb; c += A.npf) {
    const uint64_t t0 = wall_clock64();
    while ((int)ld_agent_u32(progw) + AHEAD < c) {
      if (ld_agent_u32(abortw) != 0u || wall_clock64() - t0 > SW_TIMEOUT_TICKS) return;
      __builtin_amdgcn_s_sleep(8);
    }
    const size_t blk = (size_t)(a.blk_begin + c);
    const uint4 *w = reinterpret_cast<const uint4 *>(A.winv + blk * S2W_WDOUBLES);
    for (int k = tid; k < S2W_WDOUBLES / 2; k += S2W_THREADS) sink ^= w[k].x;
    for (int d = 0; d < A.nd; ++d) {
      const uint4 *g = reinterpret_cast<const uint4 *>(A.gxt[d] + blk * S2W_PBYTES);
      for (int k = tid; k < S2W_PBYTES / 16; k += S2W_THREADS) sink ^= g[k].x;
    }
  }
  if (sink == 0x9E3779B9u && a.stamps) a.stamps[255] = sink;   // (keeps the loads)
}

template <bool FX>
__global__ __launch_bounds__(S2W_THREADS) void k_sweep2w(const SweepArgs a, const S2WArgs A) {
  if (a.redo_only && a.sc->redo == 0u) return;   // (the fp64 fallback of a fixed-point sweep that stayed in range)
  const int KS = FX ? A.K3 : a.K;   // streamer workgroups; then the sequencer; then every eighth workgroup a prefetcher
  if ((int)blockIdx.x > KS) { const int r = (int)blockIdx.x - KS; if ((r & 7) == 0 && !(S2W_DBG(A) & 8)) s2w_prefetcher(a, A, (r >> 3) - 1); return; }
  if ((int)blockIdx.x == KS) { if (S2W_DBG(A) & 8) return; s2_sequencer_winv<FX>(a, A); }
  else if (S2W_DBG(A) & 16) return;
  else if (threadIdx.x >= SW_THREADS) return;                            // the streamers are eight waves (a wave that has ended leaves the barriers' count)
  else if ((a.flags & SWF_DEBUG_WITHHOLD) && blockIdx.x == 0) return;   // test hook: a streamer that never shows up
  else if constexpr (FX) s2w_streamer_fx(a, A);
  else s2_streamer_i8(a);
}

}  // namespace bwgr
