// bwgr_amd: k_sweep3p -- TWO chains on one pass over the genotypes.
//
// A chain of k_sweep3 (sweep3.hip.h) keeps K3 streamer workgroups busy for about half of each block period and reads every
// genotype byte for itself; chains side by side are limited by compute units (K3 + 1 each), not by HBM.  Here one set of streamers
// serves two chains of the same panel: the int8 MFMA products have 16 digit columns and a chain uses seven, so chain 1's digits ride
// in columns 8..14 of the same instructions -- one tile load, one LDS tile, the same MFMA count and one barrier per block for both --
// and every per-chain piece (residual rows, list fold-in, digit split, recombination, atomic slab-dot sums, lists) exists twice.
// Each chain has its own sequencer workgroup (blockIdx 0 and 1; s3_sequencer unchanged) and its own scales, epochs and scratch:
// a chain run in a pair is bit-identical to the same chain run alone (tests/test_gpu_parity2.py::test_chain_pairs_*).
#pragma once
#include "sweep3.hip.h"

namespace bwgr {

static constexpr int S3P_OS = 20;   // dwords per row of the int32 recombination scratch: 16 digit columns, padded (b128 reads spread over the banks)

__host__ __device__ inline size_t s3p_streamer_lds(int R3) {
  const size_t Rp = (size_t)R3 + 16;
  return 2 * (size_t)SW_MAXM * Rp + 2 * 16 * Rp + 2 * 16 * (size_t)S2_DP + (size_t)64 * S3P_OS * 4 * 4 + (size_t)8 * 32 * S3P_OS * 4 + 64;
}

__device__ __forceinline__ void s3p_streamer(const Sweep3Args &A0, const Sweep3Args &A1, int w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a0 = A0.a, &a1 = A1.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, grp = lane >> 4;
  const int m = a0.m, R = a0.R, R3 = A0.R3, Rp = R3 + 16, D = A0.D;
  const int slab = w / A0.sub, hsub = w - slab * A0.sub;
  const int nb = a0.blk_end - a0.blk_begin;
  const int NU = R3 >> 6;                      // update waves (64 rows each); the other 8 - NU waves form the dots
  const int ND = 8 - NU;
  const int cprs = (R3 == 256) ? 4 : (R3 == 128 ? 3 : 2);   // log2 of the 16-byte chunks per marker
  const int8_t *Xs = reinterpret_cast<const int8_t *>(a0.X) + (size_t)slab * a0.p * R + (size_t)hsub * R3;   // marker j: Xs + j * R
  const int64_t row0 = (int64_t)slab * R + (int64_t)hsub * R3;
  uint32_t *abortw0 = a0.xflags + (size_t)a0.K * SW_FLAG_STRIDE, *abortw1 = a1.xflags + (size_t)a1.K * SW_FLAG_STRIDE;
  const size_t tile_b = (size_t)SW_MAXM * Rp;
  int8_t *tile0 = reinterpret_cast<int8_t *>(smem);
  size_t off = 2 * tile_b;
  int8_t *edig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * Rp;     // [parity][n][row]: rows 0-6 chain 0, 8-14 chain 1
  int8_t *ddig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * S2_DP;  // [parity][n][marker]: likewise
  int *outu = reinterpret_cast<int *>(smem + off); off += (size_t)64 * S3P_OS * 4 * 4;    // [update wave][row 64][n]
  int *outd = reinterpret_cast<int *>(smem + off); off += (size_t)8 * 32 * S3P_OS * 4;    // [wave][marker 32][n]
  uint32_t *ctl_s = reinterpret_cast<uint32_t *>(smem + off);                            // [0] failure, [1] / [2] overflow of chain 0 / 1
  const double S0 = s3_pow2(a0.sc->e3_sh), invS0 = s3_pow2(-a0.sc->e3_sh);
  const double S1 = s3_pow2(a1.sc->e3_sh), invS1 = s3_pow2(-a1.sc->e3_sh);
  auto blk_j0 = [&](int b) { return (a0.blk_begin + b) * m; };
  auto blk_m = [&](int b) { return min(m, a0.p - (a0.blk_begin + b) * m); };

  for (int i = tid; i < (int)((2 * 16 * Rp + 2 * 16 * S2_DP) / 4); i += SW_THREADS) reinterpret_cast<uint32_t *>(edig0)[i] = 0u;   // (adjacent)
  if (tid < 16) ctl_s[tid] = 0u;
  long long e_own0 = 0, e_own1 = 0;
  const bool upd = wave < NU;
  if (upd) { e_own0 = __double2ll_rn(a0.e[row0 + 64 * wave + lane] * S0); e_own1 = __double2ll_rn(a1.e[row0 + 64 * wave + lane] * S1); }

  s3_u4 ta0 = {0, 0, 0, 0}, ta1 = ta0, ta2 = ta0, ta3 = ta0, tb0 = ta0, tb1 = ta0, tb2 = ta0, tb3 = ta0;
#define S3P_TILE_EACH(X) X(0, tp0) X(1, tp1) X(2, tp2) X(3, tp3)
#define S3P_ISSUE1(u, name) { const int cc_ = min(tid + (u) * SW_THREADS, tot_ - 1); const int jj_ = min(cc_ >> cprs, mBt_ - 1), ii_ = cc_ & ((1 << cprs) - 1); \
    name = __builtin_nontemporal_load(reinterpret_cast<const s3_u4 *>(Xs + (size_t)(j0t_ + jj_) * R + ii_ * 16)); }
#define S3P_TILE_ISSUE(b_) do { const int j0t_ = blk_j0(b_), mBt_ = blk_m(b_), tot_ = m << cprs; S3P_TILE_EACH(S3P_ISSUE1) } while (0)
#define S3P_COMMIT1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) { const int jj_ = c_ >> cprs, ii_ = c_ & ((1 << cprs) - 1); \
    *reinterpret_cast<s3_u4 *>(dst_ + (size_t)jj_ * Rp + ii_ * 16) = name; } }
#define S3P_TILE_COMMIT(b_) do { int8_t *dst_ = tile0 + (size_t)((b_) & 1) * tile_b; const int tot_ = m << cprs; S3P_TILE_EACH(S3P_COMMIT1) } while (0)

  // the included markers of block bs of one chain: e -= x_k * corr_k for this wave's rows (s3_streamer's fold_list)
  auto fold_list = [&](const Sweep3Args &AX, uint32_t *abw, long long &e_own, int bs, unsigned long long pre) -> int {
    const int Bs = AX.a.blk_begin + bs;
    const unsigned long long *L = AX.lists + (size_t)Bs * S3_LSTRIDE;
    const int8_t *col = Xs + (size_t)(Bs * m) * R + 64 * wave + lane;
    const uint64_t t0 = wall_clock64();
    unsigned spins = 0;
    unsigned long long hv = __builtin_amdgcn_readfirstlane((uint32_t)pre) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(pre >> 32)) << 32);
    while (!s3_epoch_is(hv, AX.epoch)) {
      hv = ld_agent_raw64(L);
      hv = __builtin_amdgcn_readfirstlane((uint32_t)hv) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(hv >> 32)) << 32);
      if (s3_epoch_is(hv, AX.epoch)) break;
      if ((++spins & 63u) == 0u) {
        if (ld_agent_u32(abw) != 0u) return 0;
        if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abw, 1u); return 0; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    const int cnt = (int)(uint32_t)hv;
    for (int c0 = 0; c0 < cnt; c0 += 31) {
      const int nw = min(62, 2 * (cnt - c0));
      unsigned long long wv = pre;
      bool have = (c0 == 0);
      for (;;) {
        const bool mine = lane >= 1 && lane <= nw;
        if (!have) wv = mine ? ld_agent_raw64(L + 2 * c0 + lane) : 0ull;
        if (__ballot(mine && !s3_epoch_is(wv, AX.epoch)) == 0ull) break;
        have = false;
        if ((++spins & 63u) == 0u) {
          if (ld_agent_u32(abw) != 0u) return 0;
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abw, 1u); return 0; }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const uint32_t wlo = (uint32_t)wv, whi = (uint32_t)(wv >> 32);
      for (int e0 = 0; e0 < nw / 2; e0 += 8) {
        int xb[8]; long long cq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ee = min(e0 + u, nw / 2 - 1);
          const uint32_t w0 = __builtin_amdgcn_readlane(wlo, 1 + 2 * ee), w1 = __builtin_amdgcn_readlane(whi, 1 + 2 * ee);
          const uint32_t b0 = __builtin_amdgcn_readlane(wlo, 2 + 2 * ee);
          const int k = (int)(w1 & 0xFFu);
          cq[u] = (e0 + u < nw / 2) ? (long long)(((unsigned long long)b0 << 32) | (unsigned long long)w0) : 0ll;
          xb[u] = upd ? (int)col[(size_t)k * R] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) e_own -= (long long)xb[u] * cq[u];
      }
    }
    return 1;
  };

  // ---- prologue: tile 0 into LDS, tiles 1 and 2 in flight ----
  {
    s3_u4 &tp0 = ta0, &tp1 = ta1, &tp2 = ta2, &tp3 = ta3;
    S3P_TILE_ISSUE(0);
    S3P_TILE_COMMIT(0);
    if (nb > 2) S3P_TILE_ISSUE(2);
  }
  if (nb > 1) { s3_u4 &tp0 = tb0, &tp1 = tb1, &tp2 = tb2, &tp3 = tb3; S3P_TILE_ISSUE(1); }
  float drej_pre0 = a0.ps.blocks[a0.blk_begin].drej[tid & (SW_MAXM - 1)];   // (used by waves 6-7)
  float drej_pre1 = a1.ps.blocks[a1.blk_begin].drej[tid & (SW_MAXM - 1)];   // (used by waves 4-5)
  unsigned long long lpre0 = 0ull, lpre1 = 0ull;
  __syncthreads();

  auto step = [&](int b, s3_u4 &tp0, s3_u4 &tp1, s3_u4 &tp2, s3_u4 &tp3) -> bool {
    const int mB = blk_m(b), par = b & 1;
    int8_t *tile = tile0 + (size_t)par * tile_b;
    int8_t *edig = edig0 + (size_t)par * 16 * Rp;
    int8_t *ddig = ddig0 + (size_t)par * 16 * S2_DP;
    // A: what the included markers of block b - D changed, chain by chain
    // (a joint fold -- both lists' column bytes requested together, one round trip -- measured slower, 27-29 against 22.5 ms per
    // pair sweep at C4: sixteen byte loads per lane and block, unconditional or under uniform branches, cost more than the second wait)
    if (b >= D && upd) {
      if (!fold_list(A0, abortw0, e_own0, b - D, lpre0)) ctl_s[0] = 1u;
      if (!fold_list(A1, abortw1, e_own1, b - D, lpre1)) ctl_s[0] = 1u;
    }
    // B: digits of the residual rows (chain 0 in digit rows 0-6, chain 1 in rows 8-14) and of this block's rejected steps
    if (upd) {
      if ((unsigned long long)(e_own0 + (1ll << 54)) >> 55) ctl_s[1] = 1u;
      if ((unsigned long long)(e_own1 + (1ll << 54)) >> 55) ctl_s[2] = 1u;
      s3_put_digits7(e_own0, edig + 64 * wave + lane, Rp);
      s3_put_digits7(e_own1, edig + (size_t)8 * Rp + 64 * wave + lane, Rp);
    } else if (tid >= SW_THREADS - SW_MAXM) {                                    // waves 6-7: chain 0's steps
      const int t = tid - (SW_THREADS - SW_MAXM);
      const double qd = (t < mB) ? rint((double)drej_pre0 * S0) : 0.0;
      if (!(fabs(qd) < 18014398509481984.0)) ctl_s[1] = 1u;                      // 2^54
      s3_put_digits7((long long)qd, ddig + t, S2_DP);
    } else if (tid >= SW_THREADS - 2 * SW_MAXM) {                                // waves 4-5: chain 1's steps
      const int t = tid - (SW_THREADS - 2 * SW_MAXM);
      const double qd = (t < mB) ? rint((double)drej_pre1 * S1) : 0.0;
      if (!(fabs(qd) < 18014398509481984.0)) ctl_s[2] = 1u;
      s3_put_digits7((long long)qd, ddig + (size_t)8 * S2_DP + t, S2_DP);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ctl_s[0]) { if (tid == 0) { a0.sc->error = 1u; a1.sc->error = 1u; st_agent_u32(abortw0, 1u); st_agent_u32(abortw1, 1u); } return false; }
    // C: tile b+1 lands in the other buffer; the loads of tile b+3 go out; the lists of block b+1-D and the steps of block b+1
    if (b + 1 < nb) S3P_TILE_COMMIT(b + 1);
    {
      const int bn1 = min(b + 1, nb - 1);
      drej_pre0 = a0.ps.blocks[a0.blk_begin + bn1].drej[tid & (SW_MAXM - 1)];
      drej_pre1 = a1.ps.blocks[a1.blk_begin + bn1].drej[tid & (SW_MAXM - 1)];
      lpre0 = ld_agent_raw64(A0.lists + (size_t)(a0.blk_begin + max(bn1 - D, 0)) * S3_LSTRIDE + lane);
      lpre1 = ld_agent_raw64(A1.lists + (size_t)(a1.blk_begin + max(bn1 - D, 0)) * S3_LSTRIDE + lane);
    }
    if (b + 3 < nb) S3P_TILE_ISSUE(b + 3);
    if (upd) {
      // ---- slab update with both chains' rejected steps: out[row][n] = sum_markers x[row][marker] * digit_n(drej[marker]) ----
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
      int *ou = outu + (size_t)wave * 64 * S3P_OS;
      {                   // lane: digit column n = m16 (0-6 chain 0, 8-14 chain 1); acc_k[reg] belongs to local row 4 (4 grp + reg) + k
        int *op = ou + (size_t)(4 * (4 * grp)) * S3P_OS + m16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          op[(4 * reg + 0) * S3P_OS] = acc0[reg]; op[(4 * reg + 1) * S3P_OS] = acc1[reg];
          op[(4 * reg + 2) * S3P_OS] = acc2[reg]; op[(4 * reg + 3) * S3P_OS] = acc3[reg];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes (in order; no other wave reads this scratch)
      {
        const int4 o0 = *reinterpret_cast<const int4 *>(ou + (size_t)lane * S3P_OS), o1 = *reinterpret_cast<const int4 *>(ou + (size_t)lane * S3P_OS + 4);
        const int4 o2 = *reinterpret_cast<const int4 *>(ou + (size_t)lane * S3P_OS + 8), o3 = *reinterpret_cast<const int4 *>(ou + (size_t)lane * S3P_OS + 12);
        long long v = (long long)o0.x + ((long long)o0.y << 8) + ((long long)o0.z << 16) + ((long long)o0.w << 24);
        v += ((long long)o1.x << 32) + ((long long)o1.y << 40) + ((long long)o1.z << 48);
        e_own0 -= v;
        long long u = (long long)o2.x + ((long long)o2.y << 8) + ((long long)o2.z << 16) + ((long long)o2.w << 24);
        u += ((long long)o3.x << 32) + ((long long)o3.y << 40) + ((long long)o3.z << 48);
        e_own1 -= u;
      }
    } else {
      // ---- slab dots of block b against both chains' residual digits ----
      for (int gm = wave - NU; 16 * gm < m; gm += 2 * ND) {
        const int gm2 = gm + ND;
        const bool two = 16 * gm2 < m;
        const int8_t *bp = edig + (size_t)m16 * Rp + 16 * grp;
        const int8_t *ap = tile + (size_t)(16 * gm + m16) * Rp + 16 * grp;
        const int8_t *ap2 = tile + (size_t)(16 * (two ? gm2 : gm) + m16) * Rp + 16 * grp;
        s2_v4i acc = {0, 0, 0, 0}, acc2 = acc;
        for (int r = 0; r < R3; r += 64) {
          const s2_v4i bv = *reinterpret_cast<const s2_v4i *>(bp + r);
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap + r), bv, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap2 + r), bv, acc2, 0, 0, 0);
        }
        int *od = outd + (size_t)wave * 32 * S3P_OS;
        {                   // lane: digit column n = m16 of markers 16 gm + 4 grp + reg (rows 0..15 of the scratch) and of group gm2 (rows 16..31)
          int *op = od + (size_t)(4 * grp) * S3P_OS + m16;
          op[0] = acc[0]; op[S3P_OS] = acc[1]; op[2 * S3P_OS] = acc[2]; op[3 * S3P_OS] = acc[3];
          op[16 * S3P_OS] = acc2[0]; op[17 * S3P_OS] = acc2[1]; op[18 * S3P_OS] = acc2[2]; op[19 * S3P_OS] = acc2[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < (two ? 32 : 16)) {
          const int4 o0 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3P_OS), o1 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3P_OS + 4);
          const int4 o2 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3P_OS + 8), o3 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3P_OS + 12);
          const long long lo0 = (long long)o0.x + ((long long)o0.y << 8) + ((long long)o0.z << 16);
          const long long hi0 = (long long)o0.w + ((long long)o1.x << 8) + ((long long)o1.y << 16) + ((long long)o1.z << 24);
          const long long lo1 = (long long)o2.x + ((long long)o2.y << 8) + ((long long)o2.z << 16);
          const long long hi1 = (long long)o2.w + ((long long)o3.x << 8) + ((long long)o3.y << 16) + ((long long)o3.z << 24);
          const int mk = 16 * ((lane < 16) ? gm : gm2) + (lane & 15);
          unsigned long long *q0 = A0.qsum + ((size_t)(a0.blk_begin + b) * SW_MAXM + mk) * 2;
          unsigned long long *q1 = A1.qsum + ((size_t)(a1.blk_begin + b) * SW_MAXM + mk) * 2;
          __hip_atomic_fetch_add((gu64_t *)q0, (unsigned long long)((lo0 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gu64_t *)(q0 + 1), (unsigned long long)((hi0 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gu64_t *)q1, (unsigned long long)((lo1 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add((gu64_t *)(q1 + 1), (unsigned long long)((hi1 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is read before the next pass overwrites it
      }
    }
    return true;
  };
  for (int b = 0; b < nb; b += 2) {
    if (!step(b, tb0, tb1, tb2, tb3)) return;                     // b even: tile b+1 travels in set 1
    if (b + 1 < nb && !step(b + 1, ta0, ta1, ta2, ta3)) return;
  }
  // the lists of the last D blocks
  if (upd) for (int bs = max(0, nb - D); bs < nb; ++bs) {
    if (!fold_list(A0, abortw0, e_own0, bs, 0ull)) { ctl_s[0] = 1u; break; }
    if (!fold_list(A1, abortw1, e_own1, bs, 0ull)) { ctl_s[0] = 1u; break; }
  }
  if (upd && ((unsigned long long)(e_own0 + (1ll << 54)) >> 55)) ctl_s[1] = 1u;
  if (upd && ((unsigned long long)(e_own1 + (1ll << 54)) >> 55)) ctl_s[2] = 1u;
  __syncthreads();
  if (ctl_s[0]) { if (tid == 0) { a0.sc->error = 1u; a1.sc->error = 1u; } return; }
  if (ctl_s[1] && tid == 0) a0.sc->error = 2u;
  if (ctl_s[2] && tid == 0) a1.sc->error = 2u;
  if (upd) { a0.e[row0 + 64 * wave + lane] = (double)e_own0 * invS0; a1.e[row0 + 64 * wave + lane] = (double)e_own1 * invS1; }
#undef S3P_TILE_EACH
#undef S3P_ISSUE1
#undef S3P_TILE_ISSUE
#undef S3P_COMMIT1
#undef S3P_TILE_COMMIT
}

// blockIdx 0 / 1: the sequencers of chain 0 / 1; 2 ..: the shared streamers
template <typename GT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep3p(const Sweep3Args A0, const Sweep3Args A1) {
  if (blockIdx.x == 0) s3_sequencer<GT>(A0);        // (two call sites: a reference chosen at run time would copy the arguments to scratch memory)
  else if (blockIdx.x == 1) s3_sequencer<GT>(A1);
  else if ((A0.a.flags & SWF_DEBUG_WITHHOLD) && blockIdx.x == 2) return;   // test hook: a streamer that never shows up
  else s3p_streamer(A0, A1, (int)blockIdx.x - 2);
}

}  // namespace bwgr
