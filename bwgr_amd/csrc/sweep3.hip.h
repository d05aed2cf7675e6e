// bwgr_amd/csrc/sweep3.hip.h -- the trajectory engine: the exact sweep of the selection models (KMUP pi > 0, BayesB / C /
// Cpi / Dpi) on int8 panels with the streamers taken OUT of the per-block hand-off loop.
//
// Same Markov chain and the same blocked algebra as sweep.hip.h / sweep2.hip.h.  What changes is who waits for whom.
// In a selection model the step a marker takes when it is NOT included, drej_j = b2_j - b0_j, does not depend on the
// residual (k_prestage knows it before the sweep starts), and included markers are sparse.  So:
//
//   streamer w (blockIdx 1..K3)   owns R3 rows of every marker and walks the blocks on the ALL-REJECTED trajectory: for
//       block b it forms the slab dots q_b = X_b' e against its current slab of e and at once applies e -= X_b drej_b from
//       the same tile (no ring of tiles, no wait for the sequencer).  What the included markers changed beyond drej --
//       a short list per block -- is folded in D blocks later from single re-read columns: e -= x_k * corr_k.
//       The residual slab is held in 55-bit FIXED POINT (one power-of-two scale per sweep, k_escale): the update's int32
//       MFMA accumulators add straight into it, its bytes are the dots' digits, there is no exponent plumbing, and the
//       slab dots are exact integers, so the K3 partial sums are combined by 64-bit integer atomics in any order with the
//       same bits.  Each 8-byte sum carries its own arrival count in its low byte: the datum is the flag.
//   sequencer (blockIdx 0)        r_j = sum_w q_b[j] - spec_j - sum over the included markers k of the last D-1 blocks of
//       G_kj corr_k, then the exact speculative rounds of sweep.hip.h on wave 0.  No Gram block is staged through LDS: the
//       rows of included markers (about 1 % of them) are read on demand from the resident cross-Gram arrays, which now
//       reach D-1 blocks back.  Wave 0 publishes each block's list {k, corr} for the streamers.
//
// Dependencies: sequencer(b) needs q_b; streamer(b) needs the list of block b-D.  D blocks of slack (default 12) cover
// both hand-off latencies, so in steady state nobody waits.  Every spin is bounded by the wall clock + the abort word.
#pragma once
#include "sweep2.hip.h"

namespace bwgr {

static constexpr int S3_MAXD = 16;                    // deepest fold-in lag, blocks
static constexpr int S3_LSTRIDE = 2 * SW_MAXM + 2;    // 8-byte words of one block's list: header, two per entry, one spare
static constexpr int S3_ND = 7;                       // signed base-256 digits of the fixed-point residual and steps (|q| < 2^55)
static constexpr int S3_OS = 12;                      // dwords per row of the int32 recombination scratch (8 used; b128 reads conflict-free)

struct Sweep3Args {
  SweepArgs a;
  const void *gx[S3_MAXD];       // gx[d-1], d = 1..D-1: [nblocks][m][m] cross Gram blocks X_{b-d}' X_b (uint16 or int32)
  const void *gp;                // packed strict upper triangles of the diagonal blocks (uint16 or int32)
  int D;                         // a block's list is folded into the streamers' residual before the dots of block b + D
  int K3, R3, sub;               // streamer workgroups, rows of each, streamers per panel slab (R = sub * R3)
  int g16;                       // Gram element type: 1 uint16, 0 int32
  unsigned long long *qsum;      // [nblocks][SW_MAXM][2] {low digits, high digits} << 8 | arrivals; zero before the launch
  unsigned long long *lists;     // [nblocks][S3_LSTRIDE] epoch-tagged words
  uint32_t epoch;                // this launch's tag (24 bits, never 0)
};

// ---- fixed-point scale of one sweep.  With 2^k above both the largest |e_i| at the start of the sweep and the largest step
// |x| * |drej_j| a marker can apply to a row, e_fixed = e * 2^sh with sh = 44 - k: eleven bits of headroom below the 55-bit
// digit range (within a sweep the residual random-walks over p steps: |e| may pass its starting maximum several times over,
// and where p > n the steps themselves are larger than the residual), and a grid 2^-44 relative to that scale, far below the
// last bit of a float step.  k_prestage leaves the largest exponent field of drej in sc->e3_dex (reset here after use). ----
__global__ void k_escale_reset(ChainScalars *sc) { sc->e3_dex = 0u; }
__global__ __launch_bounds__(1024) void k_escale(const double *e, int64_t ld, ChainScalars *sc, int xbits) {
  __shared__ uint32_t mx;
  if (threadIdx.x == 0) mx = 0u;
  __syncthreads();
  uint32_t ex = 0u;
  for (int64_t i = threadIdx.x; i < ld; i += blockDim.x) ex = max(ex, (uint32_t)((__double2hiint(e[i]) & 0x7FFFFFFF) >> 20));
  atomicMax(&mx, ex);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ke = (int)mx - 1022;                       // |e| < 2^ke
    const int kd = (int)sc->e3_dex - 126 + xbits;        // |x * drej| < 2^kd
    int k = max(ke, kd);
    k = max(-200, min(200, k));                          // (all zero / inf: any scale; the sweep raises the range flag if need be)
    sc->e3_sh = 44 - k;
    sc->e3_dex = 0u;
  }
}
__device__ __forceinline__ double s3_pow2(int k) { return __hiloint2double((1023 + k) << 20, 0); }   // 2^k, |k| < 1022

// k_spec3: spec_j = sum_{k<j, same block} G_kj * drej_k with drej on the sweep's fixed-point grid (what the streamers apply),
// and the Gram diagonal.  One workgroup of 128 threads per block, thread = marker j, four partial sums.
__global__ __launch_bounds__(128) void k_spec3(const SweepArgs a, int blk_begin) {
  const int blk = blk_begin + blockIdx.x, j = threadIdx.x, m = a.m;
  const int mB = min(m, a.p - blk * m);
  const int32_t *G = reinterpret_cast<const int32_t *>(a.gram) + (size_t)blk * m * m;
  SpecBuf &sp = a.ps.spec[blk];
  __shared__ double dr[128];
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  dr[j] = (j < mB) ? rint((double)a.ps.blocks[blk].drej[j] * S) * invS : 0.0;
  __syncthreads();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, gjj = 0.0;
  if (j < mB) {
    gjj = (double)G[(size_t)j * m + j];
    int k = 0;
    for (; k + 4 <= j; k += 4) {
      s0 = fma((double)G[(size_t)k * m + j], dr[k], s0);
      s1 = fma((double)G[(size_t)(k + 1) * m + j], dr[k + 1], s1);
      s2 = fma((double)G[(size_t)(k + 2) * m + j], dr[k + 2], s2);
      s3 = fma((double)G[(size_t)(k + 3) * m + j], dr[k + 3], s3);
    }
    for (; k < j; ++k) s0 = fma((double)G[(size_t)k * m + j], dr[k], s0);
  }
  sp.spec[j] = (s0 + s1) + (s2 + s3); sp.xspec[j] = 0.0; sp.gjj[j] = gjj;
}

// list words
__device__ __forceinline__ unsigned long long s3_hdr(uint32_t epoch, int count) { return ((unsigned long long)epoch << 40) | (0xFFull << 32) | (unsigned long long)(uint32_t)count; }
__device__ __forceinline__ bool s3_epoch_is(unsigned long long w, uint32_t epoch) { return (uint32_t)(w >> 40) == epoch; }

typedef unsigned int s3_u4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t s3_streamer_lds(int R3) {
  const size_t Rp = (size_t)R3 + 16;
  return 2 * (size_t)SW_MAXM * Rp + 2 * 16 * Rp + 2 * 16 * (size_t)S2_DP + (size_t)64 * S3_OS * 4 * 4 + (size_t)8 * 16 * S3_OS * 4 + 64;
}
__host__ __device__ inline size_t s3_seq_lds(int D) {
  size_t s = 2 * sizeof(StageBuf) + 2 * 2 * SW_MAXM * sizeof(double);            // constants, spec + gjj
  s += 2 * SW_MAXM * sizeof(double) + 2 * 3 * SW_MAXM * sizeof(double);          // q sums, far-field partial sums (three waves)
  s += 2 * 3 * SW_MAXM * sizeof(float);                                          // state of a block
  s += (size_t)D * SW_MAXM * (sizeof(double) + sizeof(long long) + sizeof(int)); // lists of the last D blocks
  return s + 256;
}

// ------------------------------------------------------------------------------------------------------------------
// streamer
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s3_streamer(const Sweep3Args &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, grp = lane >> 4;
  const int w = (int)blockIdx.x - 1;
  const int m = a.m, R = a.R, R3 = A.R3, Rp = R3 + 16, D = A.D;
  const int slab = w / A.sub, hsub = w - slab * A.sub;
  const int nb = a.blk_end - a.blk_begin;
  const int NU = R3 >> 6;                      // update waves (64 rows each); the other 8 - NU waves form the dots
  const int ND = 8 - NU;
  const int cprs = (R3 == 256) ? 4 : (R3 == 128 ? 3 : 2);   // log2 of the 16-byte chunks per marker
  const int8_t *Xs = reinterpret_cast<const int8_t *>(a.X) + (size_t)slab * a.p * R + (size_t)hsub * R3;   // marker j: Xs + j * R
  const int64_t row0 = (int64_t)slab * R + (int64_t)hsub * R3;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const size_t tile_b = (size_t)SW_MAXM * Rp;
  int8_t *tile0 = reinterpret_cast<int8_t *>(smem);
  size_t off = 2 * tile_b;
  int8_t *edig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * Rp;     // [parity][n][row]
  int8_t *ddig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * S2_DP;  // [parity][n][marker]
  int *outu = reinterpret_cast<int *>(smem + off); off += (size_t)64 * S3_OS * 4 * 4;     // [update wave][row 64][n]
  int *outd = reinterpret_cast<int *>(smem + off); off += (size_t)8 * 16 * S3_OS * 4;     // [wave][marker 16][n]
  uint32_t *ctl_s = reinterpret_cast<uint32_t *>(smem + off);                            // [0] failure, [1] overflow
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  auto blk_j0 = [&](int b) { return (a.blk_begin + b) * m; };
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };

  for (int i = tid; i < (int)((2 * 16 * Rp + 2 * 16 * S2_DP) / 4); i += SW_THREADS) reinterpret_cast<uint32_t *>(edig0)[i] = 0u;   // edig0 and ddig0 are adjacent
  if (tid < 16) ctl_s[tid] = 0u;
  // the residual rows of update wave u: lane = row 64 u + lane
  long long e_own = 0;
  const bool upd = wave < NU;
  if (upd) e_own = __double2ll_rn(a.e[row0 + 64 * wave + lane] * S);

  // tile moves: four 16-byte chunks per thread, all loads unconditional (clamped), stores guarded
  s3_u4 tp0 = {0, 0, 0, 0}, tp1 = tp0, tp2 = tp0, tp3 = tp0;
#define S3_TILE_EACH(X) X(0, tp0) X(1, tp1) X(2, tp2) X(3, tp3)
#define S3_ISSUE1(u, name) { const int cc_ = min(tid + (u) * SW_THREADS, tot_ - 1); const int jj_ = min(cc_ >> cprs, mBt_ - 1), ii_ = cc_ & ((1 << cprs) - 1); \
    name = __builtin_nontemporal_load(reinterpret_cast<const s3_u4 *>(Xs + (size_t)(j0t_ + jj_) * R + ii_ * 16)); }
#define S3_TILE_ISSUE(b_) do { const int j0t_ = blk_j0(b_), mBt_ = blk_m(b_), tot_ = m << cprs; S3_TILE_EACH(S3_ISSUE1) } while (0)
#define S3_COMMIT1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) { const int jj_ = c_ >> cprs, ii_ = c_ & ((1 << cprs) - 1); \
    *reinterpret_cast<s3_u4 *>(dst_ + (size_t)jj_ * Rp + ii_ * 16) = name; } }
#define S3_TILE_COMMIT(b_) do { int8_t *dst_ = tile0 + (size_t)((b_) & 1) * tile_b; const int tot_ = m << cprs; S3_TILE_EACH(S3_COMMIT1) } while (0)

  // the included markers of block bs (relative index): e -= x_k * corr_k for this wave's rows.  `pre` holds words 0..63 of
  // the list as requested one iteration ago (a list of up to 31 entries arrives with that single load)
  auto fold_list = [&](int bs, unsigned long long pre) -> int {
    const int Bs = a.blk_begin + bs;
    const unsigned long long *L = A.lists + (size_t)Bs * S3_LSTRIDE;
    const int8_t *col = Xs + (size_t)(Bs * m) * R + 64 * wave + lane;
    const uint64_t t0 = wall_clock64();
    unsigned spins = 0;
    unsigned long long hv = __builtin_amdgcn_readfirstlane((uint32_t)pre) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(pre >> 32)) << 32);
    while (!s3_epoch_is(hv, A.epoch)) {
      hv = ld_agent_raw64(L);
      hv = __builtin_amdgcn_readfirstlane((uint32_t)hv) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(hv >> 32)) << 32);
      if (s3_epoch_is(hv, A.epoch)) break;
      if ((++spins & 63u) == 0u) {
        if (ld_agent_u32(abortw) != 0u) return 0;
        if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    const int cnt = (int)(uint32_t)hv;
    for (int c0 = 0; c0 < cnt; c0 += 31) {                 // 31 entries per pass: words 1 + 2 c0 .. of the list, one per lane
      const int nw = min(62, 2 * (cnt - c0));
      unsigned long long wv = pre;                          // pass 0: lane i holds word i (header in lane 0)
      bool have = (c0 == 0);
      for (;;) {
        const bool mine = lane >= 1 && lane <= nw;
        if (!have) wv = mine ? ld_agent_raw64(L + 2 * c0 + lane) : 0ull;
        if (__ballot(mine && !s3_epoch_is(wv, A.epoch)) == 0ull) break;
        have = false;
        if ((++spins & 63u) == 0u) {
          if (ld_agent_u32(abortw) != 0u) return 0;
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const uint32_t wlo = (uint32_t)wv, whi = (uint32_t)(wv >> 32);
      for (int e0 = 0; e0 < nw / 2; e0 += 8) {              // eight columns in flight
        int xb[8]; long long cq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ee = min(e0 + u, nw / 2 - 1);
          const uint32_t a0 = __builtin_amdgcn_readlane(wlo, 1 + 2 * ee), a1 = __builtin_amdgcn_readlane(whi, 1 + 2 * ee);
          const uint32_t b0 = __builtin_amdgcn_readlane(wlo, 2 + 2 * ee);
          const int k = (int)(a1 & 0xFFu);
          cq[u] = (e0 + u < nw / 2) ? (long long)(((unsigned long long)b0 << 32) | (unsigned long long)a0) : 0ll;
          xb[u] = upd ? (int)col[(size_t)k * R] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) e_own -= (long long)xb[u] * cq[u];
      }
    }
    return 1;
  };

  // ---- prologue ----
  S3_TILE_ISSUE(0);
  S3_TILE_COMMIT(0);
  if (nb > 1) S3_TILE_ISSUE(1);
  float drej_pre = (tid < SW_MAXM && tid < blk_m(0)) ? a.ps.blocks[a.blk_begin].drej[tid] : 0.0f;
  unsigned long long lpre = 0ull;
  __syncthreads();

  for (int b = 0; b < nb; ++b) {
    const int mB = blk_m(b), par = b & 1;
    int8_t *tile = tile0 + (size_t)par * tile_b;
    int8_t *edig = edig0 + (size_t)par * 16 * Rp;
    int8_t *ddig = ddig0 + (size_t)par * 16 * S2_DP;
    // A: what the included markers of block b - D changed
    if (b >= D && upd) { if (!fold_list(b - D, lpre)) ctl_s[0] = 1u; }
    // B: digits of the residual rows and of this block's rejected steps
    if (upd) {
      if ((unsigned long long)(e_own + (1ll << 54)) >> 55) ctl_s[1] = 1u;       // left the 55-bit range
      put_digits<S3_ND>(e_own, edig + 64 * wave + lane, Rp);
    }
    if (tid < SW_MAXM) {
      const double qd = rint((double)drej_pre * S);
      if (!(fabs(qd) < 18014398509481984.0)) ctl_s[1] = 1u;                      // 2^54
      put_digits<S3_ND>((tid < mB) ? (long long)qd : 0ll, ddig + tid, S2_DP);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
    // C: tile b+1 (in registers since the last iteration) lands in the other buffer, whose last reader was block b-1; the loads
    // of tile b+2 go out; the list of block b+1-D and the rejected steps of block b+1 are requested
    if (b + 1 < nb) S3_TILE_COMMIT(b + 1);
    if (b + 2 < nb) S3_TILE_ISSUE(b + 2);
    if (b + 1 < nb) {
      drej_pre = (tid < SW_MAXM && tid < blk_m(b + 1)) ? a.ps.blocks[a.blk_begin + b + 1].drej[tid] : 0.0f;
      if (b + 1 >= D) lpre = ld_agent_raw64(A.lists + (size_t)(a.blk_begin + b + 1 - D) * S3_LSTRIDE + lane);
    }
    if (upd) {
      // ---- slab update with the rejected steps: out[row][n] = sum_markers x[row][marker] * digit_n(drej[marker]) ----
      // lane (m16, grp): row quad 16 wave + m16 (rows 4 * that + k for accumulator k); k slots (dword u, byte q) of step s0 are the
      // markers s0 + 16 u + 4 grp + q (the interleave keeps the four lane groups on different LDS banks)
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
    } else {
      // ---- slab dots of block b against the digits of e: markers in groups of 16, group gm on wave NU + gm % ND ----
      for (int gm = wave - NU; 16 * gm < m; gm += ND) {
        const int8_t *ap = tile + (size_t)(16 * gm + m16) * Rp + 16 * grp;
        const int8_t *bp = edig + (size_t)m16 * Rp + 16 * grp;
        s2_v4i acc = {0, 0, 0, 0};
        for (int r = 0; r < R3; r += 64)
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap + r), *reinterpret_cast<const s2_v4i *>(bp + r), acc, 0, 0, 0);
        int *od = outd + (size_t)wave * 16 * S3_OS;
        if (m16 < 8) {      // lane: digit n = m16 of markers 16 gm + 4 grp + reg
          int *op = od + (size_t)(4 * grp) * S3_OS + m16;
          op[0] = acc[0]; op[S3_OS] = acc[1]; op[2 * S3_OS] = acc[2]; op[3 * S3_OS] = acc[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16) {
          const int4 o0 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3_OS);
          const int4 o1 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3_OS + 4);
          const long long lo = (long long)o0.x + ((long long)o0.y << 8) + ((long long)o0.z << 16);
          const long long hi = (long long)o0.w + ((long long)o1.x << 8) + ((long long)o1.y << 16) + ((long long)o1.z << 24);
          unsigned long long *qs = A.qsum + ((size_t)(a.blk_begin + b) * SW_MAXM + 16 * gm + lane) * 2;
          __hip_atomic_fetch_add((gu64_t *)qs, (unsigned long long)((lo << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gu64_t *)(qs + 1), (unsigned long long)((hi << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is read before the next group overwrites it
      }
    }
  }
  // the lists of the last D blocks
  if (upd) for (int bs = max(0, nb - D); bs < nb; ++bs) {
    if (!fold_list(bs, 0ull)) { ctl_s[0] = 1u; break; }
  }
  if (upd && ((unsigned long long)(e_own + (1ll << 54)) >> 55)) ctl_s[1] = 1u;
  __syncthreads();
  if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
  if (ctl_s[1] && tid == 0) a.sc->error = 2u;
  if (upd) a.e[row0 + 64 * wave + lane] = (double)e_own * invS;
#undef S3_TILE_EACH
#undef S3_ISSUE1
#undef S3_TILE_ISSUE
#undef S3_COMMIT1
#undef S3_TILE_COMMIT
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer
// ------------------------------------------------------------------------------------------------------------------
template <typename GT>
__device__ __forceinline__ void s3_sequencer(const Sweep3Args &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = a.m, nb = a.blk_end - a.blk_begin, D = A.D, pstride = a.pstride;
  size_t off = 0;
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 2 * sizeof(StageBuf);
  double *spec_s = reinterpret_cast<double *>(smem + off); off += (size_t)2 * 2 * SW_MAXM * sizeof(double);   // [parity][spec | gjj][marker]
  double *q_s = reinterpret_cast<double *>(smem + off); off += (size_t)2 * SW_MAXM * sizeof(double);
  double *far_s = reinterpret_cast<double *>(smem + off); off += (size_t)2 * 3 * SW_MAXM * sizeof(double);    // [parity][far wave][marker]
  float *state_s = reinterpret_cast<float *>(smem + off); off += (size_t)2 * 3 * SW_MAXM * sizeof(float);    // [parity][b | d | vb][marker]
  double *accC = reinterpret_cast<double *>(smem + off); off += (size_t)D * SW_MAXM * sizeof(double);         // lists of the last D blocks: what marker k changed beyond drej
  long long *accQ = reinterpret_cast<long long *>(smem + off); off += (size_t)D * SW_MAXM * sizeof(long long);   // ... on the fixed-point grid
  int *accK = reinterpret_cast<int *>(smem + off); off += (size_t)D * SW_MAXM * sizeof(int);
  int *ctrl_s = reinterpret_cast<int *>(smem + off);   // [0] ok flag, [8 + b % D] number of included markers of block b
  const GT *gp_all = reinterpret_cast<const GT *>(A.gp);
  const float Cc = a.sc->C, odds = a.sc->odds, one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };
  auto prow = [&](int k) { return k * (m - 1) - k * (k - 1) / 2; };   // offset of packed row k
  constexpr int NCH = (int)(sizeof(StageBuf) / 16);                   // 384 chunks of constants, then 64 of spec, 64 of the Gram diagonal
  static_assert(NCH + 128 == 512, "four 16-byte chunks for each lane of the two staging waves");

  // ---- the helpers' work for block c (relative): everything that does not depend on block c-1's rounds ----
  uint4 sg0 = make_uint4(0, 0, 0, 0), sg1 = sg0, sg2 = sg0, sg3 = sg0;   // staging waves: block c's chunks, requested one phase earlier
  auto stage_src = [&](int c, int ch) -> const uint4 * {
    const int blk = a.blk_begin + c;
    if (ch < NCH) return reinterpret_cast<const uint4 *>(a.ps.blocks + blk) + ch;
    if (ch < NCH + 64) return reinterpret_cast<const uint4 *>(a.ps.spec[blk].spec) + (ch - NCH);
    return reinterpret_cast<const uint4 *>(a.ps.spec[blk].gjj) + (ch - NCH - 64);
  };
  auto stage_dst = [&](int c, int ch) -> uint4 * {
    if (ch < NCH) return reinterpret_cast<uint4 *>(&stage[c & 1]) + ch;
    return reinterpret_cast<uint4 *>(spec_s + (size_t)(c & 1) * 2 * SW_MAXM) + (ch - NCH);   // spec (64 chunks) then gjj (64 chunks), contiguous
  };
  auto stage_request = [&](int c) {   // waves 2-3
    const int t = tid - 128;
    sg0 = *stage_src(c, t); sg1 = *stage_src(c, t + 128); sg2 = *stage_src(c, t + 256); sg3 = *stage_src(c, t + 384);
  };
  auto stage_commit = [&](int c) {
    const int t = tid - 128;
    *stage_dst(c, t) = sg0; *stage_dst(c, t + 128) = sg1; *stage_dst(c, t + 256) = sg2; *stage_dst(c, t + 384) = sg3;
  };
  auto poll_q = [&](int c) -> int {   // wave 1: the K3 slab dots of block c, summed by the streamers' atomics; two markers per lane
    const int mBc = blk_m(c);
    const unsigned long long *g = A.qsum + (size_t)(a.blk_begin + c) * SW_MAXM * 2;
    const unsigned long long need = (unsigned long long)A.K3;
    const bool n0 = lane < mBc, n1 = 64 + lane < mBc;
    const uint64_t t0 = wall_clock64();
    unsigned spins = 0;
    unsigned long long l0 = need, h0 = need, l1 = need, h1 = need;
    for (;;) {
      if (n0) { l0 = ld_agent_raw64(g + 2 * lane); h0 = ld_agent_raw64(g + 2 * lane + 1); }
      if (n1) { l1 = ld_agent_raw64(g + 2 * (64 + lane)); h1 = ld_agent_raw64(g + 2 * (64 + lane) + 1); }
      if ((l0 & 0xFFull) == need && (h0 & 0xFFull) == need && (l1 & 0xFFull) == need && (h1 & 0xFFull) == need) break;
      if ((++spins & 63u) == 0u) {
        if (ld_agent_u32(abortw) != 0u) return 0;
        if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    double *qd = q_s + (size_t)(c & 1) * SW_MAXM;
    qd[lane] = n0 ? fma((double)((long long)h0 >> 8), 16777216.0, (double)((long long)l0 >> 8)) * invS : 0.0;
    qd[64 + lane] = n1 ? fma((double)((long long)h1 >> 8), 16777216.0, (double)((long long)l1 >> 8)) * invS : 0.0;
    return 1;
  };
  auto far_field = [&](int c, int hw) {   // waves 4-6: sum over the included markers of blocks c-D+1 .. c-2 of G_kj corr_k, lane = markers 2 lane, 2 lane + 1
    const int blk = a.blk_begin + c;
    const int t0 = min(2 * lane, m - 2);             // m is even
    double s0 = 0.0, s1 = 0.0;
    int idx = 0;
    for (int d = 2; d < D && d <= c; ++d) {
      const int sb = (c - d) % D;
      const int cnt = ctrl_s[8 + sb];
      const GT *gd = reinterpret_cast<const GT *>(A.gx[d - 1]) + (size_t)blk * m * m + t0;
      const int *kk = accK + (size_t)sb * SW_MAXM;
      const double *cc = accC + (size_t)sb * SW_MAXM;
      int i = (3 + hw - idx % 3) % 3;                // this wave takes every third entry of the running list
      idx += cnt;
      for (; i < cnt; i += 24) {                     // eight rows in flight
        GT g0[8], g1[8]; double cf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ii = min(i + 3 * u, cnt - 1);
          const GT *row = gd + (size_t)kk[ii] * m;
          g0[u] = row[0]; g1[u] = row[1];
          cf[u] = (i + 3 * u < cnt) ? cc[ii] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s0 = fma((double)g0[u], cf[u], s0); s1 = fma((double)g1[u], cf[u], s1); }
      }
    }
    double *fd = far_s + ((size_t)(c & 1) * 3 + hw) * SW_MAXM;
    if (2 * lane < m) { fd[2 * lane] = s0; fd[2 * lane + 1] = s1; }
  };
  auto store_state = [&](int c) {   // wave 7
    const int j0c = (a.blk_begin + c) * m, mBc = blk_m(c);
    const float *sp = state_s + (size_t)(c & 1) * 3 * SW_MAXM;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = lane + 64 * h;
      if (t < mBc) {
        a.b[j0c + t] = sp[t];
        a.d[j0c + t] = sp[SW_MAXM + t];
        if (a.flags & SWF_VB_VEC) a.vb[j0c + t] = sp[2 * SW_MAXM + t];
      }
    }
  };
  auto helper_phase = [&](int c) {   // block c >= 1, while wave 0 runs block c-1
    if (wave == 1) { if (!poll_q(c)) ctrl_s[0] = 0; }
    else if (wave <= 3) { stage_commit(c); if (c + 1 < nb) stage_request(c + 1); }
    else if (wave <= 6) far_field(c, wave - 4);
    else if (c >= 2) store_state(c - 2);
  };

  // ---- prologue: block 0 ----
  if (tid < 64) ctrl_s[tid] = (tid == 0) ? 1 : 0;
  __syncthreads();
  if (wave == 1) { if (!poll_q(0)) ctrl_s[0] = 0; }
  else if (wave == 2 || wave == 3) { stage_request(0); stage_commit(0); if (nb > 1) stage_request(1); }
  else if (wave >= 4 && wave <= 6) far_field(0, wave - 4);
  __syncthreads();
  if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }

  double sum_d = 0.0, sum_b2 = 0.0;
  double rnext[2] = {0.0, 0.0};     // wave 0: what block b's included markers change in block b+1 (distance 1), accumulated as they appear
  for (int b = 0; b < nb; ++b) {
    const int mB = blk_m(b), blk = a.blk_begin + b;
    const bool have_next = (b + 1 < nb);
    if (wave == 0) {
      const StageBuf &st = stage[b & 1];
      const double *sps = spec_s + (size_t)(b & 1) * 2 * SW_MAXM;
      const double *qd = q_s + (size_t)(b & 1) * SW_MAXM;
      const double *fd = far_s + (size_t)(b & 1) * 3 * SW_MAXM;
      const int sbk = b % D;
      int *lk = accK + (size_t)sbk * SW_MAXM;
      double *lc_ = accC + (size_t)sbk * SW_MAXM;
      long long *lq = accQ + (size_t)sbk * SW_MAXM;
      const GT *gp = gp_all + (size_t)blk * pstride;
      const bool use1 = have_next && D >= 2;   // (D = 1: the streamers fold block b's list in before the dots of block b+1)
      const GT *g1 = use1 ? reinterpret_cast<const GT *>(A.gx[0]) + (size_t)(blk + 1) * m * m : nullptr;
      const int mBn = have_next ? blk_m(b + 1) : 0;
      double r[2], chi[2];
      LaneConst lc[2];
      {
        double spc[2], rd[2], sz[2], gj[2], ch[2], qq[2], f0[2], f1[2], f2[2];
        float fb0[2], fxx[2], fb2[2], fdr[2], fta[2], ftr[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = 64 * q + lane;
          spc[q] = sps[t]; gj[q] = sps[SW_MAXM + t]; qq[q] = qd[t]; f0[q] = fd[t]; f1[q] = fd[SW_MAXM + t]; f2[q] = fd[2 * SW_MAXM + t];
          fb0[q] = st.b0[t]; fxx[q] = st.xxb0[t]; fb2[q] = st.b2[t]; fdr[q] = st.drej[t];
          rd[q] = st.rden[t]; sz[q] = st.sdz1[t]; ch[q] = st.chi[t]; fta[q] = st.tacc[t]; ftr[q] = st.trej[t];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bool live = 64 * q + lane < mB;
          r[q] = live ? (((qq[q] - spc[q]) - ((f0[q] + f1[q]) + f2[q])) + rnext[q]) : 0.0;
          lc[q].b0 = live ? fb0[q] : 0.0f; lc[q].xxb0 = live ? fxx[q] : 0.0f;
          lc[q].b2 = live ? fb2[q] : 0.0f; lc[q].drej = live ? fdr[q] : 0.0f;
          lc[q].rden = live ? rd[q] : 1.0; lc[q].sdz1 = live ? sz[q] : 0.0;
          lc[q].gjj = live ? gj[q] : 0.0; lc[q].mk = a.marker0 + (uint32_t)(blk * m + 64 * q + lane);
          lc[q].tacc = live ? fta[q] : -INFINITY; lc[q].trej = live ? ftr[q] : -INFINITY;   // dead lanes: certain reject
          chi[q] = live ? ch[q] : 1.0;
        }
      }
      rnext[0] = 0.0; rnext[1] = 0.0;
      unsigned long long accmask[2] = {0ull, 0ull};
      int nacc = 0;
      const int cnt0 = min(64, mB), cnt1 = max(0, mB - 64);
      int front0 = 0, front1 = 0;
      for (;;) {   // exact speculative rounds over both lane groups: every lane assumes "nobody before me is included"
        const float b1a = lane_b1(r[0], lc[0]), b1b = lane_b1(r[1], lc[1]);
        const bool aa = lane_accept(r[0], b1a, lc[0], a.flags, Cc, odds, one_minus_pi, a.rng, a.iter);
        const bool ab = lane_accept(r[1], b1b, lc[1], a.flags, Cc, odds, one_minus_pi, a.rng, a.iter);
        const unsigned long long bal0 = __ballot(aa && lane >= front0 && lane < cnt0);
        const unsigned long long bal1 = __ballot(ab && lane >= front1 && lane < cnt1);
        int g, js;
        if (bal0) { g = 0; js = __ffsll((long long)bal0) - 1; front0 = js + 1; }
        else if (bal1) { g = 1; js = __ffsll((long long)bal1) - 1; front0 = 64; front1 = js + 1; }   // group 0 is final: group 1's votes were valid
        else break;
        const int k = 64 * g + js;
        const float dacc = g ? readlane_f32(b1b - lc[1].b0, js) : readlane_f32(b1a - lc[0].b0, js);
        const float drj = g ? readlane_f32(lc[1].drej, js) : readlane_f32(lc[0].drej, js);
        const long long cq = (long long)rint((double)dacc * S) - (long long)rint((double)drj * S);   // what the streamers will fold in
        const double corr = (double)cq * invS;
        // rows of marker k, on demand: packed diagonal block (entries for markers t > k) and the distance-1 cross block
        const int pr = prow(k);
        const int ta = lane, tb = min(64 + lane, m - 1);
        const GT ga = gp[min(pr + max(ta - k - 1, 0), pstride - 1)], gb = gp[min(pr + max(tb - k - 1, 0), pstride - 1)];   // (the last row is empty)
        GT xa = (GT)0, xb = (GT)0;
        if (use1) { const GT *row = g1 + (size_t)k * m; xa = row[min(lane, m - 1)]; xb = row[tb]; }
        r[0] = fma(-(double)((ta > k) ? ga : (GT)0), corr, r[0]);
        r[1] = fma(-(double)((64 + lane > k && 64 + lane < m) ? gb : (GT)0), corr, r[1]);
        rnext[0] = fma(-(double)xa, corr, rnext[0]);
        rnext[1] = fma(-(double)xb, corr, rnext[1]);
        accmask[g] |= (1ull << js);
        if (lane == 0) { lk[nacc] = k; lc_[nacc] = corr; lq[nacc] = cq; }
        ++nacc;
      }
      // outputs of the block
      float *sp = state_s + (size_t)(b & 1) * 3 * SW_MAXM;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        if (t < mB) {
          const float b1 = lane_b1(r[q], lc[q]);
          const bool inc = ((accmask[q] >> lane) & 1ull) != 0ull;
          const float bn = inc ? b1 : lc[q].b2;
          const float dn = inc ? 1.0f : 0.0f;
          sp[t] = bn; sp[SW_MAXM + t] = dn;
          if (a.flags & SWF_VB_VEC) sp[2 * SW_MAXM + t] = (float)((double)(Sb + bn * bn) / chi[q]);
          sum_d += (double)dn;
          sum_b2 = fma((double)bn, (double)bn, sum_b2);
        }
      }
      if (!use1) { rnext[0] = 0.0; rnext[1] = 0.0; }
      else { if (!(lane < mBn)) rnext[0] = 0.0; if (!(64 + lane < mBn)) rnext[1] = 0.0; }
      // the block's list for the streamers: header + two words per entry, one word per lane and pass
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // lane 0's list writes
      {
        unsigned long long *L = A.lists + (size_t)blk * S3_LSTRIDE;
        for (int w0 = 0; w0 < 1 + 2 * nacc; w0 += 64) {
          const int wi = w0 + lane;
          if (wi < 1 + 2 * nacc) {
            unsigned long long v;
            if (wi == 0) v = s3_hdr(A.epoch, nacc);
            else {
              const int e = (wi - 1) >> 1;
              const unsigned long long cqv = (unsigned long long)lq[e];
              v = ((wi - 1) & 1) ? (((unsigned long long)A.epoch << 40) | (0xEEull << 32) | (cqv >> 32))
                                 : (((unsigned long long)A.epoch << 40) | ((unsigned long long)(uint32_t)lk[e] << 32) | (cqv & 0xFFFFFFFFull));
            }
            st_agent_raw64(L + wi, v);
          }
        }
      }
      if (lane == 0) ctrl_s[8 + sbk] = nacc;
    } else if (have_next) {
      helper_phase(b + 1);
    }
    __syncthreads();   // block b's rounds are done, its list is in LDS; everything block b+1 needs from the helpers is in LDS
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }
  }
  if (wave == 7) {   // the state of the last two blocks is still in LDS
    if (nb >= 2) store_state(nb - 2);
    store_state(nb - 1);
  }
  if (wave == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
    if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; }
  }
}

template <typename GT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep3(const Sweep3Args A) {
  if (blockIdx.x == 0) s3_sequencer<GT>(A);
  else if ((A.a.flags & SWF_DEBUG_WITHHOLD) && blockIdx.x == 1) return;   // test hook: a streamer that never shows up
  else s3_streamer(A);
}

}  // namespace bwgr
