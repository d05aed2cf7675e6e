// bwgr_amd/csrc/sweep4.hip.h -- the trajectory engine, fourth generation: the exact sweep of the selection models (KMUP pi > 0,
// BayesB / C / Cpi / Dpi; /root/reference/src/Rcpp20260726ai.cpp:18-36, 666-688, 728-742, 884-909, 950-975) on int8 panels with
// 128-marker blocks and 16-bit Gram entries, while few markers are in the model.
//
// Same Markov chain, same blocked algebra and the same fixed-point residual as sweep3.hip.h (k_sweep3), whose per-block phases
// were bound by per-phase latencies (one barrier, one memory round trip and one digit split per 128 markers in the streamers; one
// wave on one compute unit for everything serial in the sequencer).  What changes:
//
//   streamer (blockIdx 1..K3)   takes SS blocks (a SUPER-BLOCK, SS * 128 markers) per barrier phase: the slab dots of all of them
//       against the same slab of e, then all their rejected steps at once.  The dots of block i of a super-block therefore miss
//       the rejected steps of blocks 0..i-1 of the same super-block; those terms, sum_d Gx(d)' drej_{b-d}, do not depend on the
//       residual and are added chip-wide by k_spec4 before the sweep.  Tiles arrive by LDS-DMA (global_load_lds_dwordx4 with a
//       source-address swizzle, so that both MFMA operand reads are conflict-free without padding), a whole phase ahead, with
//       hand-counted s_waitcnt: no load result ever sits in a register, so the compiler has nothing to drain.
//   sequencer (blockIdx 0)      works on QUADS (four blocks, 512 markers): each of its eight waves owns 64 markers of the quad --
//       loads their constants and slab-dot sums itself, applies the Gram rows of the included markers of the last DQ quads to
//       its own lanes, evaluates them -- and the chain is a TOKEN that walks the waves through LDS: a wave holding it runs the
//       exact speculative rounds of sweep.hip.h on its 64 lanes (two compares per lane, lane_quick's radii; lane_accept in the
//       sliver), publishes each included marker in an LDS ring that the waves behind it consume while they wait, and hands on.
//       Everything that is not the rounds (constants, q, far-field rows, outputs, list words) runs on all eight waves in parallel,
//       off the chain.
//
// Lists (what the included markers of a quad changed beyond their rejected steps) are folded into the streamers' residual before
// the dots of quad Q + DQ; the cross Gram arrays reach 4 DQ - 1 blocks back.
#pragma once
#include "sweep3.hip.h"

namespace bwgr {

#ifdef BWGR_STAMPS
// diagnostic build: s_memtime sums per section, lane 0 of every sequencer wave (stamps[64 + 8 wave + k]) and of streamer 0's first
// update wave (stamps[0..7]) and first dots wave (stamps[8..15]); flushed once at the end
#define S4ST_DECL unsigned long long ph4[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl4 = __builtin_amdgcn_s_memtime()
#define S4ST(k, cond) do { if (cond) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph4[k] += t_ - tl4; tl4 = t_; } } while (0)
#define S4ST_ADD(k, v, cond) do { if (cond) ph4[k] += (unsigned long long)(v); } while (0)
#define S4ST_FLUSH(base, cond) do { if ((cond) && a.stamps) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&a.stamps[(base) + k_], ph4[k_]); } while (0)
#else
#define S4ST_DECL do { } while (0)
#define S4ST(k, cond) do { } while (0)
#define S4ST_ADD(k, v, cond) do { } while (0)
#define S4ST_FLUSH(base, cond) do { } while (0)
#endif

static constexpr int S4_QB = 4;                          // blocks per quad
static constexpr int S4_QM = S4_QB * SW_MAXM;            // markers per quad
static constexpr int S4_LSTRIDE = 2 * S4_QM + 8;         // 8-byte words of a quad's list: header, two per entry
static constexpr int S4_RING = 2048;                     // included markers of the last DQ quads (DQ <= 4)
static constexpr int S4_MAXDQ = 4;
static constexpr int S4_DPS = S4_QM + 16;                // bytes per digit row of the rejected steps' digits

struct Sweep4Args {
  SweepArgs a;
  const void *gx[S3_MAXD];       // gx[d-1], d = 1..4 DQ - 1: [nblocks][128][128] uint16 cross Gram blocks X_{b-d}' X_b
  const void *gp;                // packed strict upper triangles of the diagonal blocks, uint16
  const void *gd;                // [nblocks][128][128] the diagonal blocks in full, uint16 (whole aligned rows for the sequencer's DMA)
  int DQ;                        // a quad's list is folded into the streamers' residual before the dots of quad Q + DQ
  int SS;                        // blocks per streamer phase (2 or 4; divides S4_QB)
  int K3, R3, sub;               // streamer workgroups, rows of each (128), streamers per panel slab
  unsigned long long *qsum;      // [nblocks][SW_MAXM][2] {low digits, high digits} << 8 | arrivals; zero before the launch
  unsigned long long *lists;     // [nquads][S4_LSTRIDE] epoch-tagged words
  uint32_t epoch;
  int dbg;                       // experiment switches (BWGR_DBG4)
  int seq;                       // which sequencer: 1 the token walk over eight waves, 2 the chain wave with helpers
  int npf;                       // L2 prefetcher workgroups: blockIdx 8, 16, .. 8 npf (the sequencer's XCD under round-robin placement)
};
// which role a workgroup of k_sweep4's grid plays: 0 the sequencer; 8 i (i = 1..npf) prefetcher i - 1; every other one a streamer
__device__ __forceinline__ int s4_streamer_index(int b, int npf) { return b - 1 - min(npf, (b - 1) / 8); }

// k_spec4: everything about a block that does not depend on the residual.  spec_j = sum_{k<j, same block} G_kj dr_k + the rejected
// steps of the blocks before it in its streamer super-block, sum_{d=1..pos} sum_k Gx(d)_kj dr_{b-d,k}, with dr on the sweep's
// fixed-point grid (what the streamers apply); the Gram diagonal; lane_quick's centre and radii.  One workgroup of 128 threads per
// block, thread = marker j, four partial sums per term, fixed order.
__global__ __launch_bounds__(128) void k_spec4(const Sweep4Args A, int blk_begin) {
  const SweepArgs &a = A.a;
  if (!(a.sc->inc_rate < a.gate3)) return;   // this sweep is k_sweep2's
  const int blk = blk_begin + blockIdx.x, j = threadIdx.x, m = a.m;
  const int mB = min(m, a.p - blk * m);
  const int pos = blk % A.SS;                 // position in the streamers' super-block (laid out from block 0 of the panel; launches start on quad boundaries)
  const int32_t *G = reinterpret_cast<const int32_t *>(a.gram) + (size_t)blk * m * m;
  SpecBuf &sp = a.ps.spec[blk];
  const StageBuf &st = a.ps.blocks[blk];
  __shared__ double dr[4][128];
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  dr[0][j] = (j < mB) ? rint((double)st.drej[j] * S) * invS : 0.0;
  for (int d = 1; d <= pos; ++d) dr[d][j] = rint((double)a.ps.blocks[blk - d].drej[j] * S) * invS;   // (earlier blocks are full blocks)
  __syncthreads();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, gjj = 0.0;
  double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
  if (j < mB) {
    gjj = (double)G[(size_t)j * m + j];
    int k = 0;
    for (; k + 4 <= j; k += 4) {
      s0 = fma((double)G[(size_t)k * m + j], dr[0][k], s0);
      s1 = fma((double)G[(size_t)(k + 1) * m + j], dr[0][k + 1], s1);
      s2 = fma((double)G[(size_t)(k + 2) * m + j], dr[0][k + 2], s2);
      s3 = fma((double)G[(size_t)(k + 3) * m + j], dr[0][k + 3], s3);
    }
    for (; k < j; ++k) s0 = fma((double)G[(size_t)k * m + j], dr[0][k], s0);
    for (int d = 1; d <= pos; ++d) {
      const uint16_t *Gx = reinterpret_cast<const uint16_t *>(A.gx[d - 1]) + (size_t)blk * m * m;
      for (k = 0; k < m; k += 4) {
        x0 = fma((double)Gx[(size_t)k * m + j], dr[d][k], x0);
        x1 = fma((double)Gx[(size_t)(k + 1) * m + j], dr[d][k + 1], x1);
        x2 = fma((double)Gx[(size_t)(k + 2) * m + j], dr[d][k + 2], x2);
        x3 = fma((double)Gx[(size_t)(k + 3) * m + j], dr[d][k + 3], x3);
      }
    }
  }
  // spec: what k_spec3 writes (k_sweep3 may run the launch instead); xspec: the super-block terms, k_sweep4's alone
  // (xspec: everything k_sweep4 subtracts from q -- the in-block terms plus the super-block ones)
  sp.spec[j] = (s0 + s1) + (s2 + s3); sp.xspec[j] = ((s0 + s1) + (s2 + s3)) + ((x0 + x1) + (x2 + x3)); sp.gjj[j] = gjj;
  double zc = 0.0, ha = INFINITY, hr = INFINITY;   // unused lanes: a certain reject
  if (j < mB) {
    LaneConst c;
    c.b0 = st.b0[j]; c.xxb0 = st.xxb0[j]; c.b2 = st.b2[j]; c.drej = st.drej[j];
    c.rden = st.rden[j]; c.sdz1 = st.sdz1[j]; c.gjj = gjj; c.tacc = st.tacc[j]; c.trej = st.trej[j]; c.mk = 0u;
    lane_quick(c, a.flags, a.sc->C, zc, ha, hr);
  }
  QuickBuf &qb = a.ps.quick[blk];
  qb.zc[j] = zc; qb.ha[j] = ha; qb.hr[j] = hr;
}

// ---- LDS words shared by waves of one workgroup: explicit ds_* accesses the compiler may neither cache nor reorder ----
typedef __attribute__((address_space(3))) unsigned long long lu64_t;
typedef __attribute__((address_space(3))) uint32_t lu32_t;
__device__ __forceinline__ unsigned long long lds_ld64(const unsigned long long *p) {
  return __hip_atomic_load((const lu64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st64(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store((lu64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_ld32(const uint32_t *p) {
  return __hip_atomic_load((const lu32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st32(uint32_t *p, uint32_t v) {
  __hip_atomic_store((lu32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// LDS-DMA of 16 bytes per lane from a wave-uniform base (SGPR pair) plus a per-lane byte offset to lds_addr + 16 * lane
__device__ __forceinline__ void s4_dma16s(const void *gbase, uint32_t voff, uint32_t la) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbase), "s"(la) : "memory", "m0");
}
// ... 4 bytes per lane; the LDS address is made scalar here (callers pass values that are wave-uniform but not provably so)
__device__ __forceinline__ void s4_dma4u(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  const uint32_t las = (uint32_t)__builtin_amdgcn_readfirstlane((int)la);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(gbase), "s"(las) : "memory", "m0");
}
#pragma clang diagnostic pop

__host__ __device__ inline size_t s4_streamer_lds(int R3, int SS) {
  const size_t Rp = (size_t)R3 + 16;
  return (size_t)2 * SS * SW_MAXM * R3        // two tiles
       + (size_t)2 * 8 * Rp                    // digits of e [parity][8][Rp] (row 7 stays zero)
       + (size_t)2 * 8 * S4_DPS                // digits of the rejected steps [parity][8][S4_DPS]
       + (size_t)2 * S4_QM * 4                 // the rejected steps themselves, landed by DMA [parity][512 floats]
       + (size_t)(R3 / 64) * 64 * S3_OS * 4    // update waves' recombination scratch
       + (size_t)(8 - R3 / 64) * 32 * S3_OS * 4   // dots waves' recombination scratch
       + 64;
}

// ------------------------------------------------------------------------------------------------------------------
// streamer: R3 = 128 rows; waves 0-1 own 64 rows of e each (fold, digits, update), waves 2-7 form the slab dots
// ------------------------------------------------------------------------------------------------------------------
template <int SS>
__device__ __forceinline__ void s4_streamer(const Sweep4Args &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  constexpr int R3 = 128, Rp = R3 + 16, NU = 2, ND = 6, MS = SS * SW_MAXM;   // markers per phase
  constexpr int NG = MS / 16;                                                    // marker groups per phase
  constexpr int NPIECE = MS / 8;                                                 // 1 KiB DMA pieces per tile (8 markers each)
  constexpr int NPASS = (NG + 2 * 6 - 1) / (2 * 6);                              // passes of a dots wave over the phase's marker groups
  constexpr int NATOM = 2 * NPASS;                                               // atomic instructions a dots wave issues per phase
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wave);
  const int m16 = lane & 15, grp = lane >> 4;
  const int w = s4_streamer_index((int)blockIdx.x, A.npf);
  const int m = SW_MAXM, R = a.R, DQ = A.DQ;
  const int slab = w / A.sub, hsub = w - slab * A.sub;
  const int nb = a.blk_end - a.blk_begin;
  const int nph = (nb + SS - 1) / SS;
  const int j_lo = a.blk_begin * m;                                  // first marker of the launch
  const int j_hi = min(a.p, a.blk_end * m);                          // one past its last marker
  const int8_t *Xs = reinterpret_cast<const int8_t *>(a.X) + (size_t)slab * a.p * R + (size_t)hsub * R3;   // marker j: Xs + j * R
  const int64_t row0 = (int64_t)slab * R + (int64_t)hsub * R3;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  constexpr size_t tile_b = (size_t)MS * R3;
  unsigned char *tile0 = smem;
  size_t off = 2 * tile_b;
  int8_t *edig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 8 * Rp;
  int8_t *ddig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 8 * S4_DPS;
  float *drej0 = reinterpret_cast<float *>(smem + off); off += (size_t)2 * S4_QM * 4;
  int *outu = reinterpret_cast<int *>(smem + off); off += (size_t)NU * 64 * S3_OS * 4;
  int *outd = reinterpret_cast<int *>(smem + off); off += (size_t)ND * 32 * S3_OS * 4;
  uint32_t *ctl_s = reinterpret_cast<uint32_t *>(smem + off);   // [0] failure, [1] overflow
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  const uint32_t tile_la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)tile0;
  const uint32_t drej_la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)reinterpret_cast<unsigned char *>(drej0);

  for (int i = tid; i < (int)((2 * 8 * Rp + 2 * 8 * S4_DPS) / 4); i += SW_THREADS) reinterpret_cast<uint32_t *>(edig0)[i] = 0u;   // adjacent
  if (tid < 16) ctl_s[tid] = 0u;
  long long e_own = 0;
  const bool upd = wvu < NU;
  if (upd) e_own = __double2ll_rn(a.e[row0 + 64 * wave + lane] * S);

  // ---- tile DMA: piece pc (0..NPIECE-1) = markers 8 pc .. 8 pc + 7 of the phase; lane l fills LDS slot (marker 8 pc + (l >> 3),
  // position l & 7) with the marker's 16-byte chunk (l & 7) ^ ((l >> 3) & 7): chunk c of marker jj sits at position c ^ (jj & 7).
  // Pieces pc = wave, wave + 8, ...  Markers past the launch's last one are clamped (their digits are zero, their dots unused). ----
  const uint32_t lane_mk = (uint32_t)(lane >> 3), lane_ch = (uint32_t)((lane & 7) ^ ((lane >> 3) & 7));
  auto tile_issue = [&](int ph) {
    const int jb = j_lo + ph * MS;
    const uint32_t la0 = tile_la + (uint32_t)((ph & 1) * tile_b);
#pragma unroll
    for (int u = 0; u < NPIECE / 8; ++u) {
      const int pc = wvu + 8 * u;
      const int jj = min(jb + 8 * pc + (int)lane_mk, j_hi - 1);
      const uint32_t voff = (uint32_t)(jj - j_lo) * (uint32_t)R + lane_ch * 16u;      // (a launch's slab of the panel stays below 4 GiB: checked on the host)
      s4_dma16s(Xs + (size_t)j_lo * R, voff, la0 + (uint32_t)pc * 1024u);
    }
  };
  // the rejected steps of phase ph: 2 KiB (SS = 4) as 1 KiB pieces by waves 2 and 3; lane l of piece u holds markers 256 u + 4 l .. + 3
  auto drej_issue = [&](int ph) {
    constexpr int NP = MS / 256;
    if (wvu >= 2 && wvu < 2 + NP) {
      const int u = wvu - 2;
      const int mk0 = ph * MS + 256 * u + 4 * lane;                    // marker index within the launch
      const int blk = min(a.blk_begin + (mk0 >> 7), a.blk_end - 1);    // (phases past the end: clamped, zeroed when digitised)
      const unsigned char *src = reinterpret_cast<const unsigned char *>(a.ps.blocks[blk].drej + (mk0 & 127));
      const unsigned char *base = reinterpret_cast<const unsigned char *>(a.ps.blocks + a.blk_begin);
      s4_dma16s(base, (uint32_t)(src - base), drej_la + (uint32_t)(((ph & 1) * S4_QM + 256 * u) * 4));
    }
  };
  // digits of the rejected steps of phase ph (waves 2, 3: each lane its own four markers, from the bytes its own DMA landed)
  auto drej_digits = [&](int ph) {
    constexpr int NP = MS / 256;
    if (wvu >= 2 && wvu < 2 + NP) {
      const int u = wvu - 2;
      const int mk0 = 256 * u + 4 * lane;
      const float4 dv = *reinterpret_cast<const float4 *>(drej0 + (size_t)(ph & 1) * S4_QM + mk0);
      const float dvv[4] = {dv.x, dv.y, dv.z, dv.w};
      unsigned long long ub[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool live = j_lo + ph * MS + mk0 + q < j_hi;
        const double qd = live ? rint((double)dvv[q] * S) : 0.0;
        if (!(fabs(qd) < 18014398509481984.0)) ctl_s[1] = 1u;                      // 2^54
        ub[q] = ((unsigned long long)(long long)qd + 0x0080808080808080ull) ^ 0x0080808080808080ull;
      }
      int8_t *dd = ddig0 + (size_t)(ph & 1) * 8 * S4_DPS + mk0;
#pragma unroll
      for (int n = 0; n < S3_ND; ++n) {
        const uint32_t wd = (uint32_t)((ub[0] >> (8 * n)) & 0xFFu) | ((uint32_t)((ub[1] >> (8 * n)) & 0xFFu) << 8) |
                            ((uint32_t)((ub[2] >> (8 * n)) & 0xFFu) << 16) | ((uint32_t)((ub[3] >> (8 * n)) & 0xFFu) << 24);
        *reinterpret_cast<uint32_t *>(dd + (size_t)n * S4_DPS) = wd;
      }
    }
  };

  // ---- lists: quad Q's included markers, e -= x_k * corr_k for this wave's rows (update waves) ----
  const int nq = (nb + S4_QB - 1) / S4_QB;
  auto list_of = [&](int Q) { return A.lists + (size_t)Q * S4_LSTRIDE; };
  // up to eight entries' column bytes requested a phase ahead (pre_n >= 0: the list was complete then and had pre_n <= 8 entries)
  int pre_n = -1, pre_q = -1;
  int pxb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long pcq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long lpre = 0ull;     // lane i: word i of the list of quad lpre_q, requested a phase ago
  int lpre_q = -1;
  auto fold_slow = [&](int Q) -> int {
    const unsigned long long *L = list_of(Q);
    const int8_t *col = Xs + (size_t)(j_lo + Q * S4_QM) * R + 64 * wave + lane;
    const uint64_t t0 = wall_clock64();
    unsigned spins = 0;
    unsigned long long hv;
    for (;;) {
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
      unsigned long long wv = 0ull;
      for (;;) {
        const bool mine = lane >= 1 && lane <= nw;
        wv = mine ? ld_agent_raw64(L + 2 * c0 + lane) : 0ull;
        if (__ballot(mine && !s3_epoch_is(wv, A.epoch)) == 0ull) break;
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
          const uint32_t b0 = __builtin_amdgcn_readlane(wlo, 2 + 2 * ee), b1 = __builtin_amdgcn_readlane(whi, 2 + 2 * ee);
          const int k = (int)(a1 & 0xFFu) | (int)((b1 & 0x1u) << 8);
          // (the entry carries the marker's two float steps {included, rejected}: what it changed beyond its rejected step, on the grid)
          cq[u] = (e0 + u < nw / 2) ? ((long long)rint((double)__uint_as_float(a0) * S) - (long long)rint((double)__uint_as_float(b0) * S)) : 0ll;
          xb[u] = (int)col[(size_t)k * R];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) e_own -= (long long)xb[u] * cq[u];
      }
    }
    return 1;
  };
  // the first half of the prefetched path: the list words of quad Q as requested a phase ago; if they are all there and few, request the
  // column bytes now (consumed at the top of the next phase)
  auto fold_prefetch = [&](int Q) {
    pre_n = -1; pre_q = Q;
    if (Q < 0 || Q >= nq || lpre_q != Q) return;
    const unsigned long long hv = __builtin_amdgcn_readfirstlane((uint32_t)lpre) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(lpre >> 32)) << 32);
    if (!s3_epoch_is(hv, A.epoch)) return;
    const int cnt = (int)(uint32_t)hv;
    if (cnt > 8) return;
    const bool mine = lane >= 1 && lane <= 2 * cnt;
    if (__ballot(mine && !s3_epoch_is(lpre, A.epoch)) != 0ull) return;
    const int8_t *col = Xs + (size_t)(j_lo + Q * S4_QM) * R + 64 * wave + lane;
    const uint32_t wlo = (uint32_t)lpre, whi = (uint32_t)(lpre >> 32);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ee = min(u, max(cnt - 1, 0));
      const uint32_t a0 = __builtin_amdgcn_readlane(wlo, 1 + 2 * ee), a1 = __builtin_amdgcn_readlane(whi, 1 + 2 * ee);
      const uint32_t b0 = __builtin_amdgcn_readlane(wlo, 2 + 2 * ee), b1 = __builtin_amdgcn_readlane(whi, 2 + 2 * ee);
      const int k = (cnt > 0) ? ((int)(a1 & 0xFFu) | (int)((b1 & 0x1u) << 8)) : 0;
      pcq[u] = (u < cnt) ? ((long long)rint((double)__uint_as_float(a0) * S) - (long long)rint((double)__uint_as_float(b0) * S)) : 0ll;
      pxb[u] = (int)col[(size_t)k * R];
    }
    pre_n = cnt;
  };

  // ---- prologue: tile 0 and the rejected steps of phase 0 ----
  tile_issue(0);
  drej_issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  drej_digits(0);
  __syncthreads();
  int folded = 0;   // lists of quads < folded are inside e_own
  S4ST_DECL;
  const bool sts = (w == 0 && (tid == 0 || tid == 64 * NU));

  for (int ph = 0; ph < nph; ++ph) {
    const int par = ph & 1;
    const unsigned char *tile = tile0 + (size_t)par * tile_b;
    int8_t *edig = edig0 + (size_t)par * 8 * Rp;
    const int8_t *ddig = ddig0 + (size_t)par * 8 * S4_DPS;
    const int b0p = ph * SS;                               // first block of the phase (relative)
    const int nbp = min(SS, nb - b0p);                     // blocks of the phase
    S4ST(0, sts);
    // ---- top: lists due before this phase's dots, digits of e; the loads issued in the previous phase are waited for ----
    if (upd) {
      const int due = (b0p / S4_QB) - DQ;                  // lists of quads <= due must be inside e
      if (!(A.dbg & 512)) while (folded <= due) {
        if (pre_n >= 0 && pre_q == folded) {
#pragma unroll
          for (int u = 0; u < 8; ++u) e_own -= (long long)pxb[u] * pcq[u];
        } else if (!fold_slow(folded)) { ctl_s[0] = 1u; break; }
        pre_n = -1;
        ++folded;
      }
      if ((unsigned long long)(e_own + (1ll << 54)) >> 55) ctl_s[1] = 1u;       // left the 55-bit range
      S4ST(1, sts);
      s3_put_digits7(e_own, edig + 64 * wave + lane, Rp);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's pieces of the tile
      S4ST(2, sts);
    } else {
      // the dots waves' only older memory operations are the previous phase's atomics (younger than the DMAs): leave them in flight
      if (ph == 0 || (A.dbg & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NATOM) : "memory");
      S4ST(1, sts);
      if (ph > 0) drej_digits(ph);
      S4ST(2, sts);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    S4ST(3, sts);
    if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
    // ---- the next phase's tile and rejected steps (into the buffers last read in phase ph - 1) ----
    if (ph + 1 < nph && !(A.dbg & 16)) { tile_issue(ph + 1); drej_issue(ph + 1); }
    S4ST(4, sts);
    if (upd) {
      // the list that is due at the top of the next phase: its words were requested a phase ago; its columns are requested now
      {
        const int nxt_due = ((b0p + SS) / S4_QB) - DQ;
        if (!(A.dbg & 512) && folded <= nxt_due) fold_prefetch(folded);
        const int lq = max(0, min(nq - 1, (((b0p + 2 * SS) / S4_QB) - DQ)));   // the list after that: its words are requested now
        lpre = ld_agent_raw64(list_of(lq) + lane);
        lpre_q = lq;
      }
      S4ST(5, sts);
      // ---- slab update with the rejected steps of all SS blocks: out[row][n] = sum_markers x[row][marker] * digit_n(drej[marker]) ----
      // lane (m16, grp): row quad 16 wave + m16; k slots (dword u, byte q) of step s0 are the markers s0 + 16 u + 4 grp + q
      const int chunk = 4 * wave + (m16 >> 2), within = 4 * (m16 & 3);
      s2_v4i acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      const int8_t *bpd = ddig + (size_t)min(m16, 7) * S4_DPS + 4 * grp;
      for (int s0 = 0; s0 < ((A.dbg & 256) ? 0 : nbp * SW_MAXM); s0 += 64) {
        uint32_t c[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int jj = s0 + 16 * u + 4 * grp + q;
            c[u][q] = *reinterpret_cast<const uint32_t *>(tile + (size_t)jj * R3 + (((chunk ^ (4 * (grp & 1) + q)) & 7) << 4) + within);
          }
        const int8_t *bp = bpd + s0;
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
      // ---- slab dots of the phase's markers against the digits of e: marker groups of 16, two groups per pass; every dots wave issues
      // exactly three pairs of atomics per full phase (six vector-memory operations: the count the top of the next phase waits by) ----
      const int x7 = m16 & 7;
      const int offa0 = ((grp ^ x7) & 7) << 4, offa1 = (((4 + grp) ^ x7) & 7) << 4;
      const int8_t *bp = edig + (size_t)min(m16, 7) * Rp + 16 * grp;
      const s2_v4i bv0 = *reinterpret_cast<const s2_v4i *>(bp), bv1 = *reinterpret_cast<const s2_v4i *>(bp + 64);
      const int ng = nbp * 8;
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int gm = (wvu - NU) + 2 * ND * pass, gm2 = gm + ND;
        const bool one = gm < ng, two = gm2 < ng;
        const unsigned char *ap = tile + (size_t)(16 * (one ? gm : 0) + m16) * R3;
        const unsigned char *ap2 = tile + (size_t)(16 * (two ? gm2 : 0) + m16) * R3;
        s2_v4i acc = {0, 0, 0, 0}, acc2 = acc;
        if (!(A.dbg & 256)) {
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap + offa0), bv0, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap2 + offa0), bv0, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap + offa1), bv1, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap2 + offa1), bv1, acc2, 0, 0, 0);
        }
        int *od = outd + (size_t)(wave - NU) * 32 * S3_OS;
        if (m16 < 8) {      // lane: digit n = m16 of markers 16 gm + 4 grp + reg (rows 0..15 of the scratch) and of group gm2 (rows 16..31)
          int *op = od + (size_t)(4 * grp) * S3_OS + m16;
          op[0] = acc[0]; op[S3_OS] = acc[1]; op[2 * S3_OS] = acc[2]; op[3 * S3_OS] = acc[3];
          op[16 * S3_OS] = acc2[0]; op[17 * S3_OS] = acc2[1]; op[18 * S3_OS] = acc2[2]; op[19 * S3_OS] = acc2[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 32) {
          const int4 o0 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3_OS);
          const int4 o1 = *reinterpret_cast<const int4 *>(od + (size_t)lane * S3_OS + 4);
          const long long lo = (long long)o0.x + ((long long)o0.y << 8) + ((long long)o0.z << 16);
          const long long hi = (long long)o0.w + ((long long)o1.x << 8) + ((long long)o1.y << 16) + ((long long)o1.z << 24);
          const int g = (lane < 16) ? gm : gm2;
          const bool live = (lane < 16) ? one : two;
          const int mkq = 16 * (live ? g : 0) + (lane & 15);                          // marker within the phase
          unsigned long long *qs = A.qsum + ((size_t)(a.blk_begin + b0p) * SW_MAXM + mkq) * 2;
          // (a group past the launch's last block adds zero to a live word of the phase's first block: the count of operations stays)
          const unsigned long long vlo = live ? (unsigned long long)((lo << 8) + 1) : 0ull, vhi = live ? (unsigned long long)((hi << 8) + 1) : 0ull;
          if (!(A.dbg & 8)) {
            __hip_atomic_fetch_add((gu64_t *)qs, vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add((gu64_t *)(qs + 1), vhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the scratch is read before the next pass overwrites it
      }
    }
  }
  S4ST(6, sts);
  S4ST_FLUSH(tid == 0 ? 0 : 8, sts);
  // the lists of the last DQ quads
  if (upd && !(A.dbg & 512)) for (; folded < nq; ++folded) {
    if (!fold_slow(folded)) { ctl_s[0] = 1u; break; }
  }
  if (upd && ((unsigned long long)(e_own + (1ll << 54)) >> 55)) ctl_s[1] = 1u;
  __syncthreads();
  if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
  if (ctl_s[1] && tid == 0) a.sc->error = 2u;
  if (upd) a.e[row0 + 64 * wave + lane] = (double)e_own * invS;
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer
//
// Eight waves, wave w owns markers 64 w .. 64 w + 63 of the quad in flight (block 4 Q + (w >> 1), half w & 1), one marker per lane.
// Per quad a wave: (1) requests, in one batch, the Gram rows of the FAR included markers (those of quads Q - DQ + 1 .. Q - 2: known
// since before its last step) for its lanes; (2) completes q, forms r; (3) applies the far rows and then the NEAR entries (quads
// Q - 1 and Q) published so far, whose rows sit in LDS; (4) evaluates its lanes and ANNOUNCES its first two candidates: their rows
// towards every block of this quad and the next are requested by LDS-DMA -- a row is there long before the marker is published;
// (5) requests the next quad's constants into a second register set; (6) waits for the token, applying entries as they appear
// (LDS reads only); (7) runs its rounds, publishing each included marker; (8) hands the token on; (9) writes its outputs.
// An included marker that was not announced (a third one of a wave, or one that became a candidate late) is published without a
// row slot: every wave then reads its row straight from global memory.
// ------------------------------------------------------------------------------------------------------------------
static constexpr int S4_NSLOT = 16;                        // announce slots per quad: two per wave
static constexpr int S4_NTGT = 8;                          // target blocks of an announced row set: the blocks of its quad and of the next
static constexpr int S4_NPAR = 3;                          // quads whose row sets are alive (Q - 1, Q, and the one being overwritten for Q + 1 ...)
static constexpr int S4_FARB = 16;                         // far entries per batch of loads
__host__ __device__ inline size_t s4_seq_lds() {
  return (size_t)S4_RING * (sizeof(double) + sizeof(float) * 2 + sizeof(int) + sizeof(int)) + 32 + 16 * sizeof(int) + 16 * sizeof(void *) + 8 * 2 * sizeof(double)
       + (size_t)S4_NPAR * S4_NSLOT * S4_NTGT * SW_MAXM * 2 + (size_t)8 * SW_MAXM * 2 + 64;
}
__device__ __forceinline__ void s4_sequencer(const Sweep4Args &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  typedef uint16_t GT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wave);
  const int m = SW_MAXM, nb = a.blk_end - a.blk_begin, DQ = A.DQ;
  const int nq = (nb + S4_QB - 1) / S4_QB;
  size_t off = 0;
  double *accC = reinterpret_cast<double *>(smem + off); off += (size_t)S4_RING * sizeof(double);     // what marker k changed beyond drej
  float2 *accS = reinterpret_cast<float2 *>(smem + off); off += (size_t)S4_RING * sizeof(float2);     // ... as the two float steps {included, rejected}
  int *accK = reinterpret_cast<int *>(smem + off); off += (size_t)S4_RING * sizeof(int);               // marker index within the launch
  int *accT = reinterpret_cast<int *>(smem + off); off += (size_t)S4_RING * sizeof(int);               // its announce slot, or -1
  unsigned long long *ctl = reinterpret_cast<unsigned long long *>(smem + off); off += 32;             // [0] {token, inclusions so far}, [1] failure
  uint32_t *posq = reinterpret_cast<uint32_t *>(smem + off); off += 16 * sizeof(uint32_t);             // [Q & 15]: inclusions before quad Q
  const unsigned char **tab = reinterpret_cast<const unsigned char **>(smem + off); off += 16 * sizeof(void *);   // [0] diagonal blocks in full, [d] distance-d cross blocks
  double *red = reinterpret_cast<double *>(smem + off); off += 8 * 2 * sizeof(double);                 // [wave][2] posterior sums
  GT *nrow = reinterpret_cast<GT *>(smem + off); off += (size_t)S4_NPAR * S4_NSLOT * S4_NTGT * SW_MAXM * sizeof(GT);   // [quad % 3][slot][target block 0..7][128]
  GT *srow = reinterpret_cast<GT *>(smem + off) + (size_t)wave * SW_MAXM;                               // this wave's scratch row (unannounced markers)
  const uint32_t srow_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)reinterpret_cast<unsigned char *>(srow));
  const uint32_t nrow_la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)reinterpret_cast<unsigned char *>(nrow);
  const float Cc = a.sc->C, odds = a.sc->odds, one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const bool vbv = (a.flags & SWF_VB_VEC) != 0;
  const int hw = wvu & 1;                          // which half of its block this wave owns
  const int t = 64 * hw + lane;                    // this lane's marker within its block
  constexpr int ring = S4_RING;

  if (tid == 0) { lds_st64(ctl, 0ull); lds_st64(ctl + 1, 0ull); }
  if (tid < 16) posq[tid] = 0u;
  if (tid >= 64 && tid < 64 + 16) tab[tid - 64] = (tid == 64) ? reinterpret_cast<const unsigned char *>(A.gd) : reinterpret_cast<const unsigned char *>(A.gx[tid - 65]);
  __syncthreads();

  // ---- this lane's marker: constants of the quad in flight (c_*) and of the next one (n_*, requested while the wave waits for the token) ----
  float c_b0 = 0.f, c_b2 = 0.f, c_drej = 0.f, c_xxb0 = 0.f, c_tacc = 0.f, c_trej = 0.f;
  double c_rden = 0.0, c_sdz1 = 0.0, c_chi = 1.0, c_spec = 0.0, c_xspec = 0.0, c_gjj = 0.0, c_zc = 0.0, c_ha = INFINITY, c_hr = INFINITY;
  unsigned long long c_qlo = 0ull, c_qhi = 0ull;
  float n_b0 = 0.f, n_b2 = 0.f, n_drej = 0.f, n_xxb0 = 0.f, n_tacc = 0.f, n_trej = 0.f;
  double n_rden = 0.0, n_sdz1 = 0.0, n_chi = 1.0, n_spec = 0.0, n_xspec = 0.0, n_gjj = 0.0, n_zc = 0.0, n_ha = INFINITY, n_hr = INFINITY;
  unsigned long long n_qlo = 0ull, n_qhi = 0ull;
  auto blk_of = [&](int Q) { return S4_QB * Q + (wvu >> 1); };                 // relative block of this wave in quad Q
#define S4_LOAD_CONSTS(Q_, P_) do {                                            /* unconditional loads (clamped block) */ \
    const int cl_ = min(blk_of(Q_), nb - 1), bk_ = a.blk_begin + cl_; \
    const StageBuf &st_ = a.ps.blocks[bk_]; const SpecBuf &sp_ = a.ps.spec[bk_]; const QuickBuf &qb_ = a.ps.quick[bk_]; \
    P_##b0 = st_.b0[t]; P_##b2 = st_.b2[t]; P_##drej = st_.drej[t]; P_##xxb0 = st_.xxb0[t]; P_##tacc = st_.tacc[t]; P_##trej = st_.trej[t]; \
    P_##rden = st_.rden[t]; P_##sdz1 = st_.sdz1[t]; P_##chi = st_.chi[t]; \
    P_##spec = sp_.spec[t]; P_##xspec = sp_.xspec[t]; P_##gjj = sp_.gjj[t]; \
    P_##zc = qb_.zc[t]; P_##ha = qb_.ha[t]; P_##hr = qb_.hr[t]; \
    const unsigned long long *g_ = A.qsum + ((size_t)bk_ * SW_MAXM + t) * 2; \
    P_##qlo = ld_agent_raw64(g_); P_##qhi = ld_agent_raw64(g_ + 1); } while (0)
  bool failed = false;
  // the table's base address for distance d, back in scalar registers (the same word in every lane)
  auto tab_base = [&](int d) -> const unsigned char * {
    const unsigned long long bu = (unsigned long long)(uintptr_t)tab[d];
    return reinterpret_cast<const unsigned char *>((uintptr_t)(
        (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bu) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bu >> 32)) << 32)));
  };
  // Gram entry (included marker kk of the launch, this lane's marker) for a wave whose lanes sit in relative block c, straight from
  // global memory: row kk of the diagonal block in full when kk is in the same block, else of the cross block of their distance
  auto row_entry = [&](int kk, int c) -> GT {
    const int ck = kk >> 7, kl = kk & 127, d = c - ck;
    // (a GLOBAL load: a pointer that went through LDS comes back generic, and a flat load waits on both memory counters)
    return ((const __attribute__((address_space(1))) GT *)(uintptr_t)tab_base(d))[((size_t)(a.blk_begin + c) * m + (size_t)kl) * m + (size_t)t];
  };
  // ---- ring entries in batches: lane u forms the descriptor of entry e0 + u (every lane in parallel: two LDS round trips per batch
  // instead of two per entry), the wave then takes them one by one through v_readlane ----
  unsigned long long d_gb = 0ull;   // lane u: global address of the entry's Gram row towards this wave's block
  double d_cf = 0.0;                // lane u: its coefficient (0 beyond the batch)
  int d_loff = -1, d_klm = -1;      // lane u: byte offset of its announced row in the LDS row sets, or -1; its marker's index if it sits in this wave's block, else -1
  int par = 0, parp = 2;            // Q % 3 and (Q - 1) % 3
  auto describe = [&](uint32_t e0, int n, int c, int Qc, bool near) {
    const int sl = (int)((e0 + (uint32_t)min(lane, n - 1)) & (ring - 1));
    const int kk = accK[sl];
    const int slot = near ? accT[sl] : -1;
    d_cf = (lane < n) ? accC[sl] : 0.0;
    const int ck = kk >> 7, kl = kk & 127, Qk = kk >> 9;
    d_gb = (unsigned long long)(uintptr_t)tab[c - ck] + (((unsigned long long)(a.blk_begin + c) * m + (unsigned long long)kl) * m) * sizeof(GT);
    d_klm = (ck == c) ? kl : -1;
    d_loff = (slot >= 0) ? ((((Qk == Qc ? par : parp) * S4_NSLOT + slot) * S4_NTGT + (c - S4_QB * Qk)) * SW_MAXM) * (int)sizeof(GT) : -1;
  };
  // (rows are kept as 32-bit values: the compiler packs 16-bit ones in pairs and waits for each pair of loads to do so)
  auto desc_row_global = [&](int u) -> uint32_t {  // entry u of the batch: this lane's element of its row, from global memory
    const unsigned long long gb = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d_gb, u) |
                                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d_gb >> 32), u) << 32);
    return (uint32_t)((const __attribute__((address_space(1))) GT *)(uintptr_t)gb)[t];
  };
  // the row of an UNANNOUNCED marker (rare): landed in this wave's scratch row by DMA and waited for on the spot.  Written as inline asm on
  // purpose: with no load the compiler knows of inside the wait loop and the rounds, it puts no s_waitcnt vmcnt there, and the requests
  // that are meant to stay in flight across them (the next quad's constants, the announced rows) do
#ifdef BWGR_STAMPS
  unsigned long long slow_ticks = 0, slow_n = 0;
#endif
  auto slow_row = [&](unsigned long long gb) -> uint32_t {   // gb: wave-uniform global address of the 256-byte row
    const unsigned char *rb = reinterpret_cast<const unsigned char *>((uintptr_t)(
        (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb >> 32)) << 32)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (earlier reads of the scratch row are done)
#ifdef BWGR_STAMPS
    const unsigned long long ts_ = __builtin_amdgcn_s_memtime();
#endif
    s3_dma4s(rb, (uint32_t)lane * 4u, srow_la);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef BWGR_STAMPS
    slow_ticks += __builtin_amdgcn_s_memtime() - ts_; slow_n += 1;
#endif
    return (uint32_t)srow[t];
  };
  double sum_d = 0.0, sum_b2 = 0.0;
  S4ST_DECL;
  const bool stq = (lane == 0);
#ifdef BWGR_STAMPS
  unsigned long long xtr[4] = {0, 0, 0, 0}, xtr2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // the far entries of a quad (quads Q - DQ + 1 .. Q - 2): the first batch of rows is requested a step ahead (right after the wave's
  // step in the quad before) and applied after q; fcf: lane u holds the coefficient of the batch's entry u
  uint32_t fg[S4_FARB];
  double fcf = 0.0;
  uint32_t f0 = 0u, f1 = 0u;
  int nf0 = 0;
#define S4_FAR_REQUEST(Q_) do { \
    const int cn_ = blk_of(Q_); \
    f0 = ((Q_) - DQ + 1 > 0) ? lds_ld32(posq + (((Q_) - DQ + 1) & 15)) : 0u; \
    f1 = ((Q_) - 1 > 0) ? lds_ld32(posq + (((Q_) - 1) & 15)) : 0u;                 /* entries before quad Q_ - 1 */ \
    nf0 = (cn_ < nb) ? (int)min((uint32_t)S4_FARB, f1 - f0) : 0; \
    fcf = 0.0; \
    if (nf0 > 0) { \
      describe(f0, nf0, cn_, (Q_), false); \
      fcf = d_cf; \
      _Pragma("unroll") for (int u_ = 0; u_ < S4_FARB; ++u_) fg[u_] = desc_row_global(min(u_, nf0 - 1)); \
    } else { _Pragma("unroll") for (int u_ = 0; u_ < S4_FARB; ++u_) fg[u_] = 0; } } while (0)

  S4_LOAD_CONSTS(0, c_);
  S4_FAR_REQUEST(0);
  for (int Q = 0; Q < nq; ++Q) {
    S4ST(0, stq);
    const int c = blk_of(Q);
    const bool live_w = c < nb;                                   // this wave has a block in this quad
    const int mBc = live_w ? min(m, a.p - (a.blk_begin + c) * m) : 0;
    const bool valid = live_w && t < mBc;
    const uint32_t mystep = (uint32_t)(8 * Q + wvu);
    // ---- (1) the far entries' first batch of rows: requested a step ago (S4_FAR_REQUEST), applied after q ----
    // ---- (5) the next quad's constants and slab-dot words into the second register set: a whole quad period to land ----
    if (Q + 1 < nq) S4_LOAD_CONSTS(Q + 1, n_);
    S4ST(1, stq);
    // ---- (2) the slab dots of this lane's marker: complete when both words carry K3 arrivals ----
    if (live_w) {
      const unsigned long long need = (unsigned long long)A.K3;
      const unsigned long long *g = A.qsum + ((size_t)(a.blk_begin + c) * SW_MAXM + t) * 2;
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        if (__ballot(valid && ((c_qlo & 0xFFull) != need || (c_qhi & 0xFFull) != need)) == 0ull) break;
        if (A.dbg & 8) break;   // (timing experiment only: the streamers publish nothing)
        if ((++spins & 63u) == 0u) {
          if (ld_agent_u32(abortw) != 0u || (uint32_t)lds_ld64(ctl + 1) != 0u) { failed = true; break; }
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); failed = true; break; }
        }
        __builtin_amdgcn_s_sleep(1);
        c_qlo = ld_agent_raw64(g); c_qhi = ld_agent_raw64(g + 1);
      }
    }
    if (failed) { lds_st64(ctl + 1, 1ull); break; }
    S4ST(2, stq);
    double r = valid ? fma((double)((long long)c_qhi >> 8), 16777216.0, (double)((long long)c_qlo >> 8)) * invS - c_xspec : 0.0;
    const double hr_l = valid ? c_hr : INFINITY, ha_l = valid ? c_ha : INFINITY;   // dead lanes: a certain reject
    // ---- (3) far rows, in ring order; then the near entries published so far ----
    if (live_w) {
#pragma unroll
      for (int u = 0; u < S4_FARB; ++u) r = fma(-(double)fg[u], readlane_f64(fcf, u), r);
      for (uint32_t fb = f0 + (uint32_t)nf0; fb != f1; ) {          // more far entries than one batch (dense chains): blocking batches
        const int n = (int)min((uint32_t)S4_FARB, f1 - fb);
        describe(fb, n, c, Q, false);
        uint32_t g2[S4_FARB];
#pragma unroll
        for (int u = 0; u < S4_FARB; ++u) g2[u] = desc_row_global(min(u, n - 1));
#pragma unroll
        for (int u = 0; u < S4_FARB; ++u) r = fma(-(double)g2[u], readlane_f64(d_cf, u), r);
        fb += (uint32_t)n;
      }
    }
    // near entries: ring [napp, to); the row of an announced marker is in LDS (set (its quad) % 3, its slot, target block c - 4 * (its
    // quad)); an unannounced one's row comes from global memory
    uint32_t napp = f1;
    auto apply_near = [&](uint32_t to) {
      while (napp != to) {
        if (to - napp == 1u) {   // one new entry (the usual case while a wave waits): every lane reads its words itself, nothing goes through scalar registers
          const int sl = (int)(napp & (ring - 1));
          const int kk = accK[sl], slot = accT[sl];
          const double cf = accC[sl];
          const int ck = kk >> 7, kl = kk & 127, Qk = kk >> 9;
          uint32_t g1;
          if (__builtin_expect(__builtin_amdgcn_readfirstlane(slot) >= 0, 1))
            g1 = (uint32_t)nrow[(size_t)((((Qk == Q ? par : parp) * S4_NSLOT + slot) * S4_NTGT + (c - S4_QB * Qk)) * SW_MAXM) + t];
          else
            g1 = slow_row((unsigned long long)(uintptr_t)tab[c - ck] + (((unsigned long long)(a.blk_begin + c) * m + (unsigned long long)kl) * m) * sizeof(GT));
          r = fma(-(double)g1, (ck != c || t > kl) ? cf : 0.0, r);
          napp = to;
          break;
        }
        const int n = (int)min(8u, to - napp);
        describe(napp, n, c, Q, true);
        uint32_t g[8];
        const unsigned long long slow = __ballot(lane < n && d_loff < 0);   // entries without a row slot
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int loff = __builtin_amdgcn_readlane(d_loff, min(u, n - 1));
          g[u] = (uint32_t)*reinterpret_cast<const GT *>(reinterpret_cast<const unsigned char *>(nrow) + max(loff, 0) + 2 * t);
        }
        if (__builtin_expect(slow != 0ull, 0)) {
#pragma unroll
          for (int u = 0; u < 8; ++u) if ((slow >> u) & 1ull)
            g[u] = slow_row((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d_gb, u) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d_gb >> 32), u) << 32));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double cf = readlane_f64(d_cf, u);
          const int klm = __builtin_amdgcn_readlane(d_klm, min(u, n - 1));
          r = fma(-(double)g[u], (t > klm) ? cf : 0.0, r);
        }
        napp += (uint32_t)n;
      }
    };
    unsigned long long cw = lds_ld64(ctl);
    asm volatile("" ::: "memory");   // (the ring entries are read after the word that announces them)
    if (live_w) apply_near((uint32_t)(cw >> 32));
    S4ST(3, stq);
    // ---- (4) the first two candidates under the state so far are announced: their rows towards this quad's and the next quad's
    // blocks are requested by DMA (slots 2 w, 2 w + 1 of this quad's set) ----
    int jsA = -1, jsB = -1;
#ifdef BWGR_STAMPS
    const double zpre = fabs(r - c_zc) / hr_l;   // how close to a candidate this lane is when the wave announces
    if (live_w) {
      xtr2[0] += __builtin_popcountll(__ballot(valid && zpre >= 1.0));
      xtr2[1] += __builtin_popcountll(__ballot(valid && zpre >= 0.9));
      xtr2[2] += __builtin_popcountll(__ballot(valid && zpre >= 0.8));
      xtr2[3] += __builtin_popcountll(__ballot(valid && zpre >= 0.6));
    }
#endif
    if (live_w) {
      unsigned long long cand = __ballot(valid && !(fabs(r - c_zc) < hr_l));
      if (cand) { jsA = (int)__builtin_ctzll(cand); cand &= cand - 1ull; }
      if (cand) jsB = (int)__builtin_ctzll(cand);
      const int tl = min(S4_NTGT, nb - S4_QB * Q);               // target blocks that exist
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int js = i ? jsB : jsA;
        if (js >= 0) {
          const int kl = 64 * hw + js, slot = 2 * wvu + i;
          for (int ti = c - S4_QB * Q; ti < tl; ++ti) {
            const int d = ti - (c - S4_QB * Q);
            const unsigned char *rb = tab_base(d) + (((size_t)(a.blk_begin + S4_QB * Q + ti) * m + (size_t)kl) * m) * sizeof(GT);
            s3_dma4s(rb, (uint32_t)lane * 4u, nrow_la + (uint32_t)((((par * S4_NSLOT + slot) * S4_NTGT + ti) * SW_MAXM) * (int)sizeof(GT)));
          }
        }
      }
    }
    S4ST(4, stq);
    // ---- (6) wait for the token; the markers included meanwhile are applied as they appear ----
    {
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        cw = lds_ld64(ctl);
        asm volatile("" ::: "memory");
#ifdef BWGR_STAMPS
        xtr[2] += 1;
        if (live_w && napp != (uint32_t)(cw >> 32)) {
          const unsigned long long ta_ = __builtin_amdgcn_s_memtime();
          xtr[1] += (uint32_t)(cw >> 32) - napp;
          apply_near((uint32_t)(cw >> 32));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          xtr[0] += __builtin_amdgcn_s_memtime() - ta_;
        }
#else
        if (live_w && (!(A.dbg & 4) || (uint32_t)cw == mystep)) apply_near((uint32_t)(cw >> 32));
#endif
        if ((uint32_t)cw == mystep) break;
        if ((++spins & 255u) == 0u) {
          if ((uint32_t)lds_ld64(ctl + 1) != 0u) { failed = true; break; }
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); failed = true; break; }
        }
        if (mystep - (uint32_t)cw > 1u && !(A.dbg & 2)) { if (A.dbg & 32) __builtin_amdgcn_s_sleep(8); else if (A.dbg & 64) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(1); }   // only the next wave in line polls at full rate
      }
    }
    if (!(A.dbg & 1)) __builtin_amdgcn_s_setprio(3);   // the chain's wave goes first wherever it shares an issue port
    if (failed) { lds_st64(ctl + 1, 1ull); break; }
#ifdef BWGR_STAMPS
    if (stq && mystep > 0u) ph4[1] += __builtin_amdgcn_s_memtime() - lds_ld64(ctl + 2);   // hand-off: from the store of the token to the exit of this wave's wait
#endif
    S4ST(5, stq);
    uint32_t ninc = (uint32_t)(cw >> 32);
    const uint32_t n0 = ninc;
    unsigned long long am = 0ull;
    // ---- (7) the exact speculative rounds on this wave's 64 lanes (sweep.hip.h, quick_rounds): a lane that is not a certain reject
    // is the next candidate; certain accepts are taken, the sliver asks lane_accept ----
    if (live_w && !(A.dbg & 128)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's announced rows have landed (published markers point at them)
      LaneConst lc;
      lc.b0 = c_b0; lc.xxb0 = c_xxb0; lc.b2 = c_b2; lc.drej = c_drej; lc.rden = c_rden; lc.sdz1 = c_sdz1; lc.gjj = c_gjj;
      lc.tacc = c_tacc; lc.trej = c_trej; lc.mk = a.marker0 + (uint32_t)((a.blk_begin + c) * m + t);
      unsigned long long livem = (mBc - 64 * hw >= 64) ? ~0ull : ((mBc - 64 * hw > 0) ? ((1ull << (mBc - 64 * hw)) - 1ull) : 0ull);
      double z = r - c_zc;
      const double drejd = (double)c_drej;
      const GT *ownA = nrow + (size_t)(((par * S4_NSLOT + 2 * wvu) * S4_NTGT + (c - S4_QB * Q)) * SW_MAXM) + t;
      const GT *ownB = ownA + (size_t)S4_NTGT * SW_MAXM;
      for (;;) {
        const unsigned long long cand = livem & ~__ballot(fabs(z) < hr_l);
        if (cand == 0ull) break;
        const int js = (int)__builtin_ctzll(cand);
        const int slot = (js == jsA) ? 2 * wvu : ((js == jsB) ? 2 * wvu + 1 : -1);
        GT g0;
        if (js == jsA) g0 = *ownA;
        else if (js == jsB) g0 = *ownB;
        else g0 = (GT)slow_row((unsigned long long)(uintptr_t)tab[0] + (((unsigned long long)(a.blk_begin + c) * m + (unsigned long long)(64 * hw + js)) * m) * sizeof(GT));
        const unsigned long long accm = __ballot(fabs(z) > ha_l);
        const float b1 = lane_b1(r, lc);
        const float d1f = b1 - c_b0;
        const double cd = (double)d1f - drejd;                    // per lane: what its accepted step changes beyond the speculated one
        livem &= (~1ull << js);
        if (__builtin_expect(!((accm >> js) & 1ull), 0)) {        // between the radii: the full test decides; a reject leaves its speculated step standing
          if (!((__ballot(lane_accept(r, b1, lc, a.flags, Cc, odds, one_minus_pi, a.rng, a.iter)) >> js) & 1ull)) continue;
        }
        const double corr = readlane_f64(cd, js);
        am |= (1ull << js);
#ifdef BWGR_STAMPS
        if (slot < 0) {
          xtr[3] += 1;
          const double zp = readlane_f64(zpre, js);
          if (zp >= 1.0) xtr2[4] += 1; else if (zp >= 0.9) xtr2[5] += 1; else if (zp >= 0.8) xtr2[6] += 1; else if (zp >= 0.6) xtr2[7] += 1;
        }
#endif
        {   // publish: the waves behind this one apply it to their lanes while they wait
          const int sl = (int)(ninc & (ring - 1));
          if (lane == 0) { accK[sl] = 128 * c + 64 * hw + js; accT[sl] = slot; accC[sl] = corr; accS[sl] = make_float2(readlane_f32(d1f, js), readlane_f32(c_drej, js)); }
          ++ninc;
          asm volatile("" ::: "memory");   // (the entry before the word that announces it; one wave's LDS operations complete in order)
          lds_st64(ctl, ((unsigned long long)ninc << 32) | (unsigned long long)mystep);
        }
        const double gm0 = (lane > js) ? (double)g0 : 0.0;
        r = fma(-gm0, corr, r);
        z = fma(-gm0, corr, z);
      }
    }
    S4ST(6, stq);
    S4ST_ADD(7, ninc - n0, stq);
    // ---- (8) hand the token on ----
    if (!(A.dbg & 1)) __builtin_amdgcn_s_setprio(0);
    if (wvu == 7) lds_st32(posq + ((Q + 1) & 15), ninc);
    asm volatile("" ::: "memory");
#ifdef BWGR_STAMPS
    lds_st64(ctl + 2, __builtin_amdgcn_s_memtime());
#endif
    lds_st64(ctl, ((unsigned long long)ninc << 32) | (unsigned long long)(mystep + 1u));
    // the far rows of the next quad: every quad they come from is complete (the last one ended before this wave's step)
    if (Q + 1 < nq) S4_FAR_REQUEST(Q + 1);
    // ---- (9) off the chain: this wave's outputs, its entries of the quad's list ----
    if (valid && !(A.dbg & 4096)) {
      const int jg = (a.blk_begin + c) * m + t;
      const bool inc = ((am >> lane) & 1ull) != 0ull;
      const float bn = inc ? (float)fma(r + (double)c_xxb0, c_rden, c_sdz1) : c_b2;
      const float dn = inc ? 1.0f : 0.0f;
      a.b[jg] = bn;
      a.d[jg] = dn;
      if (vbv) a.vb[jg] = (float)((double)(Sb + bn * bn) / c_chi);
      sum_d += (double)dn;
      sum_b2 = fma((double)bn, (double)bn, sum_b2);
    }
    {
      const uint32_t p0 = lds_ld32(posq + (Q & 15));
      unsigned long long *L = A.lists + (size_t)Q * S4_LSTRIDE;
      const int cnt = (int)(ninc - n0);
      if (lane < cnt) {
        const int sl = (int)((n0 + (uint32_t)lane) & (ring - 1));
        const float2 st2 = accS[sl];   // the two float steps {included, rejected}: the streamers fold in their difference on the fixed-point grid
        const uint32_t k9 = (uint32_t)(accK[sl] - S4_QM * Q);
        const uint32_t idx = n0 + (uint32_t)lane - p0;
        st_agent_raw64(L + 1 + 2 * idx, ((unsigned long long)A.epoch << 40) | ((unsigned long long)(k9 & 0xFFu) << 32) | (unsigned long long)__float_as_uint(st2.x));
        st_agent_raw64(L + 2 + 2 * idx, ((unsigned long long)A.epoch << 40) | ((unsigned long long)(0xE0u | (k9 >> 8)) << 32) | (unsigned long long)__float_as_uint(st2.y));
      }
      if (wvu == 7 && lane == 0) st_agent_raw64(L, s3_hdr(A.epoch, (int)(ninc - p0)));
    }
    // the next quad's constants become the current ones
    c_b0 = n_b0; c_b2 = n_b2; c_drej = n_drej; c_xxb0 = n_xxb0; c_tacc = n_tacc; c_trej = n_trej;
    c_rden = n_rden; c_sdz1 = n_sdz1; c_chi = n_chi; c_spec = n_spec; c_xspec = n_xspec; c_gjj = n_gjj; c_zc = n_zc; c_ha = n_ha; c_hr = n_hr;
    c_qlo = n_qlo; c_qhi = n_qhi;
    parp = par; par = (par == S4_NPAR - 1) ? 0 : par + 1;
  }
#undef S4_LOAD_CONSTS
#undef S4_FAR_REQUEST
  S4ST_FLUSH(64 + 8 * wave, stq);
#ifdef BWGR_STAMPS
  if (stq && a.stamps) for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&a.stamps[128 + 8 * wave + k_], xtr[k_]);
  if (stq && a.stamps) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&a.stamps[192 + k_], xtr2[k_]);
  if (stq && a.stamps) { atomicAdd(&a.stamps[200], slow_ticks); atomicAdd(&a.stamps[201], slow_n); }
#endif
  // posterior sums, in wave order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
  if (lane == 0) { red[2 * wave] = sum_d; red[2 * wave + 1] = sum_b2; }
  __syncthreads();
  if ((uint32_t)lds_ld64(ctl + 1) != 0u) { if (tid == 0) a.sc->error = 1u; return; }
  if (tid == 0) {
    double sd = 0.0, sb2 = 0.0;
    for (int w8 = 0; w8 < 8; ++w8) { sd += red[2 * w8]; sb2 += red[2 * w8 + 1]; }
    a.sc->sum_d += sd; a.sc->sum_b2 += sb2;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer, second form (A.seq == 2): ONE wave runs the whole chain.
//
// The token sequencer above pays an LDS hand-off per 64 markers and per included marker (publish -> the next wave's poll -> its
// row read -> its token poll: ~1.5k cycles each).  Here wave 0 (the CHAIN wave) keeps the residual dots of the four lane groups
// (64 markers each) it will decide next in its own registers: an included marker is applied to all of them by four LDS row reads
// and four FMAs, no other wave is waited for, and the rounds of group T + 1 start the instruction after those of group T end.
// Everything else belongs to the other waves and reaches the chain wave through LDS long before it is needed:
//   helpers (waves 1..6)  take the lane groups ("tasks", eight per quad) in turn: constants, q, the window's Gram rows from global
//       memory -> r; they publish {r, radii} as a task record and then FOLLOW the chain (every newly included marker is applied
//       and the record rewritten under a sequence lock) until the chain wave adopts the group, three groups before it decides it;
//       they announce the group's candidates (row sets of four rows, distances 0..3, landed in LDS by DMA, plus the candidate's
//       constants as a record); after the chain wave is through with the group they write its outputs.
//   servicer (wave 7)     lands the row sets of candidates the chain wave itself discovers in its adopted groups.
// A candidate nobody announced in time costs the chain wave a wait for its row set (L2 or HBM latency); a marker with no free
// row-set slot (dense chains) goes through scratch rows fetched on the spot.
// ------------------------------------------------------------------------------------------------------------------
static constexpr int S4C_NS = 48;       // row-set slots: 0 .. NSH - 1 the helpers' (a shared counter), the rest the chain wave's own
static constexpr int S4C_NSH = 32;
static constexpr int S4C_NROW = 4;      // rows of a set: towards the marker's own block and the three behind it
static constexpr int S4C_NTR = 16;      // task records (two quads)
static constexpr int S4C_LEAD = 8;      // a helper publishes task T when the chain has finished task T - LEAD - 1
static constexpr int S4C_LOOK = 3;      // groups the chain wave holds beyond the one it decides
static constexpr int S4C_NH = 6;        // helper waves
static constexpr int S4C_QN = 32;       // announce requests in flight from the chain wave to the servicer
struct S4CRec {                          // a candidate's constants (what an inclusion needs of its marker), 64 bytes
  int kk, state;                         // marker index within the launch; kk + 1 once the row set has landed
  float b0, xxb0, drej, b2, tacc, trej;
  double rden, sdz1, gjj, ha;
};
struct S4CTask {                         // one lane group as the chain wave adopts it
  double r[64], zc[64], ha[64], hr[64];
  int gs[64];                            // the lane's row-set slot, or -1
  uint32_t seq, W, ready, pad;           // sequence lock of {r, W, an}; ring entries below W are inside r; task + 1 when published
  unsigned long long an;                 // lanes that have a row set
  unsigned long long pad2;
};
struct S4CGroup { unsigned long long am; uint32_t n0, n1; };   // what the chain wave decided: included lanes, their ring entries [n0, n1)
__host__ __device__ inline size_t s4c_seq_lds() {
  return (size_t)S4_RING * (8 + 4 + 4 + 4) + 64 + 64 + 128 + 128 + (size_t)S4C_NTR * sizeof(S4CTask) + (size_t)S4C_NTR * sizeof(S4CGroup) + (size_t)(S4C_NS + 1) * sizeof(S4CRec)
       + (size_t)S4C_QN * 8 + (size_t)(S4C_NS + 1) * S4C_NROW * SW_MAXM * 2 + (size_t)8 * S4C_NROW * SW_MAXM * 2 + 64;
}
__device__ __forceinline__ void s4_sequencer_cw(const Sweep4Args &A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  typedef uint16_t GT;
  typedef const __attribute__((address_space(1))) GT gGT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wave);
  const int m = SW_MAXM, nb = a.blk_end - a.blk_begin, DQ = A.DQ;
  const int nq = (nb + S4_QB - 1) / S4_QB, ntask = 8 * nq;
  constexpr int ring = S4_RING;
  size_t off = 0;
  double *accC = reinterpret_cast<double *>(smem + off); off += (size_t)S4_RING * 8;      // what marker k changed beyond drej
  int *accK = reinterpret_cast<int *>(smem + off); off += (size_t)S4_RING * 4;            // marker index within the launch
  int *accT = reinterpret_cast<int *>(smem + off); off += (size_t)S4_RING * 4;            // its row-set slot, or -1
  float *accB = reinterpret_cast<float *>(smem + off); off += (size_t)S4_RING * 4;        // its new effect
  unsigned long long *ctl = reinterpret_cast<unsigned long long *>(smem + off); off += 64;   // [0] {tasks decided, inclusions so far}, [1] failure, [2] {slot counter, -}, [3] {queue tail, queue head}
  uint32_t *posq = reinterpret_cast<uint32_t *>(smem + off); off += 64;                    // [Q & 15]: inclusions before quad Q
  const unsigned char **tab = reinterpret_cast<const unsigned char **>(smem + off); off += 128;   // [0] diagonal blocks in full, [d] distance-d cross blocks
  double *red = reinterpret_cast<double *>(smem + off); off += 128;                        // [wave][2] posterior sums
  S4CTask *TR = reinterpret_cast<S4CTask *>(smem + off); off += (size_t)S4C_NTR * sizeof(S4CTask);
  S4CGroup *GR = reinterpret_cast<S4CGroup *>(smem + off); off += (size_t)S4C_NTR * sizeof(S4CGroup);
  S4CRec *crec = reinterpret_cast<S4CRec *>(smem + off); off += (size_t)(S4C_NS + 1) * sizeof(S4CRec);   // (the last one: the chain wave's slot for a marker nobody announced)
  unsigned long long *queue = reinterpret_cast<unsigned long long *>(smem + off); off += (size_t)S4C_QN * 8;   // {slot << 32 | kk}
  GT *nrow = reinterpret_cast<GT *>(smem + off); off += (size_t)(S4C_NS + 1) * S4C_NROW * SW_MAXM * 2;        // [slot][distance 0..3][128]
  GT *srow = reinterpret_cast<GT *>(smem + off) + (size_t)wave * S4C_NROW * SW_MAXM;                            // this wave's scratch rows
  const uint32_t nrow_la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)reinterpret_cast<unsigned char *>(nrow);
  const uint32_t srow_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)reinterpret_cast<unsigned char *>(srow));
  uint32_t *ctl32 = reinterpret_cast<uint32_t *>(ctl);
  const float Cc = a.sc->C, odds = a.sc->odds, one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const bool vbv = (a.flags & SWF_VB_VEC) != 0;

  for (int i = tid; i < (int)((S4C_NTR * sizeof(S4CTask) + S4C_NTR * sizeof(S4CGroup) + (S4C_NS + 1) * sizeof(S4CRec)) / 4); i += SW_THREADS) reinterpret_cast<uint32_t *>(TR)[i] = 0u;   // adjacent
  if (tid < 16) { ctl32[tid] = 0u; posq[tid] = 0u; }
  if (tid >= 64 && tid < 64 + 16) tab[tid - 64] = (tid == 64) ? reinterpret_cast<const unsigned char *>(A.gd) : reinterpret_cast<const unsigned char *>(A.gx[tid - 65]);
  __syncthreads();
  if (tid < S4C_NS) crec[tid].kk = -(1 << 20);   // (no slot holds a marker)
  __syncthreads();

  auto tab_base = [&](int d) -> const unsigned char * {
    const unsigned long long bu = (unsigned long long)(uintptr_t)tab[d];
    return reinterpret_cast<const unsigned char *>((uintptr_t)(
        (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bu) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bu >> 32)) << 32)));
  };
  // a row-set slot for marker kk, or -1: slots go round; one is free once the chain is through with the last block its rows lead to
  auto slot_alloc = [&](int kk) -> int {
    uint32_t sN = 0u;
    if (lane == 0) sN = __hip_atomic_fetch_add((lu32_t *)(ctl32 + 4), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int sl = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)sN) % (uint32_t)S4C_NSH);
    const int prev = __builtin_amdgcn_readfirstlane(crec[sl].kk);
    const uint32_t done = (uint32_t)lds_ld64(ctl);                      // tasks the chain has decided
    if ((int)done < 2 * ((prev >> 7) + S4C_NROW)) return -1;            // (two tasks per block)
    if (lane == 0) { crec[sl].kk = kk; crec[sl].state = 0; }
    return sl;
  };
  // the row set of marker kk into slot sl: rows towards blocks ck .. ck + 3 (those that exist), by DMA
  auto rows_request = [&](int kk, int sl) {
    const int ck = kk >> 7, kl = kk & 127;
    for (int d = 0; d < S4C_NROW; ++d) if (ck + d < nb) {
      const unsigned char *rb = tab_base(d) + (((size_t)(a.blk_begin + ck + d) * m + (size_t)kl) * m) * sizeof(GT);
      s4_dma4u(rb, (uint32_t)lane * 4u, nrow_la + (uint32_t)(((sl * S4C_NROW + d) * SW_MAXM) * (int)sizeof(GT)));
    }
  };
  bool failed = false;

  if (wvu == 0) {
    // ==============================================================================================================
    // the chain wave: everything here is counted in instructions (one wave issues one every ~7 cycles).  Per lane group it keeps
    // r, the rounds' centre and the certain-reject radius in registers; what only an INCLUDED marker needs (the accept radius, its
    // row-set slot, its constants) is read from LDS when a marker is included
    // ==============================================================================================================
    double R[4] = {0, 0, 0, 0}, ZC[4] = {0, 0, 0, 0}, HR[4];
    unsigned long long AN[4] = {0, 0, 0, 0};                             // lanes of the set that have a row set (wave-uniform)
    bool LV[4] = {false, false, false, false};                           // the set holds a group that exists
#pragma unroll
    for (int i = 0; i < 4; ++i) HR[i] = INFINITY;
    uint32_t ninc = 0u, qtail = 0u;
    int own_kk = -(1 << 20);   // lane i: the marker that holds the chain wave's own slot NSH + i (i < NS - NSH)
    int own_next = 0;
    uint32_t qhead_seen = 0u, qpos = 0u;   // (qpos: inclusions before the quad in hand)
#ifdef BWGR_STAMPS
    unsigned long long xt4[4] = {0, 0, 0, 0};
#endif
    S4ST_DECL;
    const bool stq = (lane == 0);
    __builtin_amdgcn_s_setprio(3);
    // this lane's element of row d of slot sl, half hwj
    auto lds_row = [&](int sl, int d, int hwj) -> double { return (double)nrow[(size_t)((sl * S4C_NROW + d) * SW_MAXM) + 64 * hwj + lane]; };
    auto push_request = [&](int kk, int sl) {
      if (lane == 0) { crec[sl].kk = kk; crec[sl].state = 0; queue[qtail & (S4C_QN - 1)] = ((unsigned long long)(uint32_t)sl << 32) | (unsigned long long)(uint32_t)kk; }
      ++qtail;
      asm volatile("" ::: "memory");
      lds_st32(ctl32 + 6, qtail);
    };
    // adopt task Tn into register set si: its record (under the sequence lock), then the ring entries its helper had not seen
    auto adopt = [&](int Tn, auto si_c) {
      constexpr int si = decltype(si_c)::value;
      LV[si] = false; R[si] = 0.0; ZC[si] = 0.0; HR[si] = INFINITY; AN[si] = 0ull;
      if (Tn >= ntask) return;
      const S4CTask &tr = TR[Tn & (S4C_NTR - 1)];
      uint64_t t0 = 0;
      unsigned spins = 0;
      uint32_t W;
#ifdef BWGR_STAMPS
      const unsigned long long tw_ = __builtin_amdgcn_s_memtime();
#endif
      double zc_n = 0.0, hr_n = INFINITY;
      for (;;) {
        // every read of the record in one batch (LDS returns them in order): ready, sequence, {r, W, an}, sequence again, centre, radius
        const uint32_t rdy = lds_ld32(&tr.ready);
        const uint32_t s1 = lds_ld32(&tr.seq);
        const double rr = tr.r[lane];
        W = lds_ld32(&tr.W);
        const unsigned long long an = lds_ld64(&tr.an);
        const uint32_t s2 = lds_ld32(&tr.seq);
        zc_n = tr.zc[lane]; hr_n = tr.hr[lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (rdy == (uint32_t)(Tn + 1) && s1 == s2 && !(s1 & 1u)) { R[si] = rr; AN[si] = an; break; }
        if ((++spins & 255u) == 0u) {
          if ((uint32_t)lds_ld64(ctl + 1) != 0u) { failed = true; return; }   // (the helpers watch the launch's abort word)
          if (t0 == 0) t0 = wall_clock64();                             // (the clock is read only once a wait has become long)
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); failed = true; return; }
        }
      }
#ifdef BWGR_STAMPS
      if (stq) { ph4[3] += __builtin_amdgcn_s_memtime() - tw_; }
#endif
      ZC[si] = zc_n; HR[si] = hr_n;
      LV[si] = ((Tn >> 1) < nb);
      // catch up: entries [W, ninc) (the last few the chain published)
      const int cj = Tn >> 1, hwj = Tn & 1;
      for (uint32_t e = W; LV[si] && e != ninc; ++e) {
        const int sl = (int)(e & (ring - 1));
        const int kk = __builtin_amdgcn_readfirstlane(accK[sl]), slot = __builtin_amdgcn_readfirstlane(accT[sl]);
        const double cf = accC[sl];
        const int ck = kk >> 7, kl = kk & 127, d = cj - ck;
        double g;
        if (slot >= 0 && d < S4C_NROW) g = lds_row(slot, d, hwj);
        else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const unsigned char *rb = tab_base(d) + (((size_t)(a.blk_begin + cj) * m + (size_t)kl) * m) * sizeof(GT);
          s4_dma4u(rb, (uint32_t)lane * 4u, srow_la);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          g = (double)srow[64 * hwj + lane];
        }
        R[si] = fma(-g, (ck != cj || 64 * hwj + lane > kl) ? cf : 0.0, R[si]);
      }
    };
    // new candidates of an adopted set that nobody announced yet: one request per call goes to the servicer, into one of the chain
    // wave's own slots (taken in turn; free once the chain is through with the last block its rows lead to: no LDS traffic to find one)
    auto look = [&](int Tj, auto sj_c) {
      constexpr int sj = decltype(sj_c)::value;
      const unsigned long long cand = __ballot(!(fabs(R[sj] - ZC[sj]) < HR[sj])) & ~AN[sj];
      if (cand == 0ull || !LV[sj]) return;
      const int js = (int)__builtin_ctzll(cand);
      const int kk = 64 * Tj + js;
      const int prev = __builtin_amdgcn_readlane(own_kk, own_next);
      if (Tj - S4C_LOOK < 2 * ((prev >> 7) + S4C_NROW) || qtail - qhead_seen >= (uint32_t)S4C_QN) {
        qhead_seen = lds_ld32(ctl32 + 7);
        return;                                                         // (the marker stays unannounced for now)
      }
      const int sl = S4C_NSH + own_next;
      if (lane == own_next) own_kk = kk;
      own_next = (own_next + 1 == S4C_NS - S4C_NSH) ? 0 : own_next + 1;
      S4ST_ADD(5, 1, stq);
      AN[sj] |= (1ull << js);
      if (lane == 0) TR[Tj & (S4C_NTR - 1)].gs[js] = sl;
      push_request(kk, sl);
    };
    // the rounds of task T held in set si; sets si + 1 .. si + 3 hold tasks T + 1 .. T + 3
    auto decide = [&](int T, auto si_c) {
      constexpr int si = decltype(si_c)::value;
      constexpr int s1 = (si + 1) & 3, s2 = (si + 2) & 3, s3 = (si + 3) & 3;
      const int Q = T >> 3, c = T >> 1, hw = T & 1;
      const uint32_t n0 = ninc;
      unsigned long long am = 0ull;
      if (LV[si] && !(A.dbg & 128)) {
        const int mBc = min(m, a.p - (a.blk_begin + c) * m);
        unsigned long long livem = (mBc - 64 * hw >= 64) ? ~0ull : ((mBc - 64 * hw > 0) ? ((1ull << (mBc - 64 * hw)) - 1ull) : 0ull);
        const S4CTask &tr = TR[T & (S4C_NTR - 1)];
        const int d1 = ((T + 1) >> 1) - c, d2 = ((T + 2) >> 1) - c, d3 = ((T + 3) >> 1) - c;
        for (;;) {
          const unsigned long long cand = livem & ~__ballot(fabs(R[si] - ZC[si]) < HR[si]);
          if (cand == 0ull) break;
          const int js = (int)__builtin_ctzll(cand);
          const int kk = 64 * T + js, kl = 64 * hw + js;
          livem &= (~1ull << js);
          // what only an included marker needs: its accept radius, its centre, its row-set slot
          const double zc_j = tr.zc[js];
          int slot = __builtin_amdgcn_readfirstlane(tr.gs[js]);
          bool own_set = true;
          if (__builtin_expect(slot < 0, 0)) {
            // nobody announced this marker (it became a candidate this very moment, or no slot was free -- dense chains): its row set and
            // constants come through the servicer now, into the spare slot, and the chain waits for them.  (No load the compiler knows of
            // in this wave: a vector-memory load here would put s_waitcnt vmcnt(0) on the common path, behind the list words' stores.)
            slot = S4C_NS; own_set = false;
            uint64_t t0 = 0;
            unsigned spins = 0;
            while (qtail - lds_ld32(ctl32 + 7) >= (uint32_t)S4C_QN) {
              if ((++spins & 255u) == 0u) { if (t0 == 0) t0 = wall_clock64(); if ((uint32_t)lds_ld64(ctl + 1) != 0u || wall_clock64() - t0 > SW_TIMEOUT_TICKS) { failed = true; break; } }
            }
            if (failed) break;
            push_request(kk, slot);
            S4ST_ADD(4, 1, stq);
          }
          const S4CRec &cr = crec[slot];
          {
            uint64_t t0 = 0;
            unsigned spins = 0;
#ifdef BWGR_STAMPS
            const unsigned long long tc_ = __builtin_amdgcn_s_memtime();
#endif
            while (lds_ld32(reinterpret_cast<const uint32_t *>(&cr.state)) != (uint32_t)(kk + 1)) {   // its row set is still on the way
              if ((++spins & 255u) == 0u) {
                if (t0 == 0) t0 = wall_clock64();
                if ((uint32_t)lds_ld64(ctl + 1) != 0u || wall_clock64() - t0 > SW_TIMEOUT_TICKS) { failed = true; break; }
              }
            }
            if (failed) break;
#ifdef BWGR_STAMPS
            if (stq) { ph4[6] += __builtin_amdgcn_s_memtime() - tc_; }
#endif
            asm volatile("" ::: "memory");
          }
          const float k_b0 = cr.b0, k_xxb0 = cr.xxb0, k_drej = cr.drej;
          const double k_rden = cr.rden, k_sdz1 = cr.sdz1, ha_j = cr.ha;
          const double g0 = lds_row(slot, 0, hw), g1 = lds_row(slot, d1, (T + 1) & 1), g2 = lds_row(slot, d2, (T + 2) & 1), g3 = lds_row(slot, d3, (T + 3) & 1);
#ifdef BWGR_STAMPS
          unsigned long long tq_ = __builtin_amdgcn_s_memtime();
#endif
          const double rj = readlane_f64(R[si], js);
          const float b1 = (float)fma(rj + (double)k_xxb0, k_rden, k_sdz1);
          const float d1f = b1 - k_b0;
          if (__builtin_expect(!(fabs(rj - zc_j) > ha_j), 0)) {     // between the radii: the full test decides; a reject leaves its speculated step standing
            LaneConst lc;
            lc.b0 = k_b0; lc.xxb0 = k_xxb0; lc.b2 = cr.b2; lc.drej = k_drej; lc.rden = k_rden; lc.sdz1 = k_sdz1; lc.gjj = cr.gjj; lc.tacc = cr.tacc; lc.trej = cr.trej;
            lc.mk = a.marker0 + (uint32_t)((a.blk_begin + c) * m + kl);
            if (!__builtin_amdgcn_readfirstlane((int)lane_accept(rj, b1, lc, a.flags, Cc, odds, one_minus_pi, a.rng, a.iter))) continue;
          }
          const double corr = (double)d1f - (double)k_drej;
          am |= (1ull << js);
          {   // publish: the helpers that follow the chain apply it to the groups not yet adopted; the list words go straight out
            const int sl = (int)(ninc & (ring - 1));
            const uint32_t k9 = (uint32_t)(kk - S4_QM * Q), idx = ninc - qpos;
            unsigned long long *L = A.lists + (size_t)Q * S4_LSTRIDE;
            if (lane == 0) {
              accK[sl] = kk; accT[sl] = own_set ? slot : -1; accC[sl] = corr; accB[sl] = b1;
              const int jg = (a.blk_begin + c) * m + kl;           // the included marker's outputs (every other marker's are k_sweep4_finish's)
              a.b[jg] = b1; a.d[jg] = 1.0f;
              st_agent_raw64(L + 1 + 2 * idx, ((unsigned long long)A.epoch << 40) | ((unsigned long long)(k9 & 0xFFu) << 32) | (unsigned long long)__float_as_uint(d1f));
              st_agent_raw64(L + 2 + 2 * idx, ((unsigned long long)A.epoch << 40) | ((unsigned long long)(0xE0u | (k9 >> 8)) << 32) | (unsigned long long)__float_as_uint(k_drej));
            }
            ++ninc;
            asm volatile("" ::: "memory");
            lds_st64(ctl, ((unsigned long long)ninc << 32) | (unsigned long long)(uint32_t)T);
          }
#ifdef BWGR_STAMPS
          { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); if (stq) xt4[0] += tn_ - tq_; tq_ = tn_; }
#endif
          R[si] = fma(-((lane > js) ? g0 : 0.0), corr, R[si]);
          R[s1] = fma(-g1, corr, R[s1]); R[s2] = fma(-g2, corr, R[s2]); R[s3] = fma(-g3, corr, R[s3]);
          // candidates this step created in the sets ahead get their row sets requested now
          if (!(A.dbg & 64)) { look(T + 1, std::integral_constant<int, s1>{}); look(T + 2, std::integral_constant<int, s2>{}); if (!(A.dbg & 32)) look(T + 3, std::integral_constant<int, s3>{}); }
#ifdef BWGR_STAMPS
          { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); if (stq) xt4[1] += tn_ - tq_; }
#endif
        }
      }
      // the group is decided
      if (lane == 0) { S4CGroup &gr = GR[T & (S4C_NTR - 1)]; gr.am = am; gr.n0 = n0; gr.n1 = ninc; }
      if ((T & 7) == 7) {
        if (lane == 0) { posq[(Q + 1) & 15] = ninc; st_agent_raw64(A.lists + (size_t)Q * S4_LSTRIDE, s3_hdr(A.epoch, (int)(ninc - qpos))); }
        qpos = ninc;
      }
      asm volatile("" ::: "memory");
      lds_st64(ctl, ((unsigned long long)ninc << 32) | (unsigned long long)(uint32_t)(T + 1));
      S4ST_ADD(7, ninc - n0, stq);
    };
    adopt(0, std::integral_constant<int, 0>{}); adopt(1, std::integral_constant<int, 1>{}); adopt(2, std::integral_constant<int, 2>{});
    for (int T0 = 0; T0 < ntask && !failed; T0 += 4) {
      S4ST(0, stq);
      adopt(T0 + 3, std::integral_constant<int, 3>{}); if (failed) break;
      S4ST(1, stq);
      decide(T0, std::integral_constant<int, 0>{}); if (failed) break;
      S4ST(2, stq);
      adopt(T0 + 4, std::integral_constant<int, 0>{}); if (failed) break;
      decide(T0 + 1, std::integral_constant<int, 1>{}); if (failed) break;
      adopt(T0 + 5, std::integral_constant<int, 1>{}); if (failed) break;
      decide(T0 + 2, std::integral_constant<int, 2>{}); if (failed) break;
      adopt(T0 + 6, std::integral_constant<int, 2>{}); if (failed) break;
      decide(T0 + 3, std::integral_constant<int, 3>{}); if (failed) break;
    }
    if (failed) lds_st64(ctl + 1, 1ull);
    S4ST_FLUSH(64, stq);
#ifdef BWGR_STAMPS
    if (stq && a.stamps) for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&a.stamps[72 + k_], xt4[k_]);
#endif
  } else if (wvu == 7) {
    // ==============================================================================================================
    // the servicer: row sets and constants of the candidates the chain wave found in its adopted groups
    // ==============================================================================================================
    uint32_t head = 0u;
    const uint64_t t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
      const uint32_t done = (uint32_t)lds_ld64(ctl);
      const uint32_t tail = lds_ld32(ctl32 + 6);
      if (head == tail) {
        if ((int)done >= ntask || (uint32_t)lds_ld64(ctl + 1) != 0u) break;
        if ((++spins & 1023u) == 0u && wall_clock64() - t0 > 64 * SW_TIMEOUT_TICKS) break;
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      asm volatile("" ::: "memory");
      const unsigned long long rq = queue[head & (S4C_QN - 1)];
      const int kk = __builtin_amdgcn_readfirstlane((int)(uint32_t)rq), sl = __builtin_amdgcn_readfirstlane((int)(uint32_t)(rq >> 32));
      rows_request(kk, sl);
      const int blk = a.blk_begin + (kk >> 7), kl = kk & 127;
      const StageBuf &st = a.ps.blocks[blk];
      const float v_b0 = st.b0[kl], v_xxb0 = st.xxb0[kl], v_drej = st.drej[kl], v_b2 = st.b2[kl], v_tacc = st.tacc[kl], v_trej = st.trej[kl];
      const double v_rden = st.rden[kl], v_sdz1 = st.sdz1[kl], v_gjj = a.ps.spec[blk].gjj[kl], v_ha = a.ps.quick[blk].ha[kl];
      S4CRec &cr = crec[sl];
      if (lane == 0) { cr.b0 = v_b0; cr.xxb0 = v_xxb0; cr.drej = v_drej; cr.b2 = v_b2; cr.tacc = v_tacc; cr.trej = v_trej; cr.rden = v_rden; cr.sdz1 = v_sdz1; cr.gjj = v_gjj; cr.ha = v_ha; }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the rows have landed, the record is written
      lds_st32(reinterpret_cast<uint32_t *>(&cr.state), (uint32_t)(kk + 1));
      ++head;
      lds_st32(ctl32 + 7, head);
    }
  } else {
    // ==============================================================================================================
    // helpers: a lane group's r as the chain wave adopts it.  Per lane only what r and the rounds' first test need (q, the speculative
    // terms, centre, certain-reject radius); a candidate's other constants are read when it is announced; the outputs of the markers
    // that are NOT included are written by k_sweep4_finish after the sweep
    // ==============================================================================================================
    const int hi = wvu - 1;
    double c_sx = 0.0, c_zc = 0.0, c_hr = INFINITY; unsigned long long c_qlo = 0ull, c_qhi = 0ull;
    double n_sx = 0.0, n_zc = 0.0, n_hr = INFINITY; unsigned long long n_qlo = 0ull, n_qhi = 0ull;
#define S4C_LOAD_CONSTS(T_, P_) do {                                           /* unconditional loads (clamped block) */ \
    const int cl_ = min((T_) >> 1, nb - 1), bk_ = a.blk_begin + cl_, t_ = 64 * ((T_) & 1) + lane; \
    P_##sx = a.ps.spec[bk_].xspec[t_]; P_##zc = a.ps.quick[bk_].zc[t_]; P_##hr = a.ps.quick[bk_].hr[t_]; \
    const unsigned long long *g_ = A.qsum + ((size_t)bk_ * SW_MAXM + t_) * 2; \
    P_##qlo = ld_agent_raw64(g_); P_##qhi = ld_agent_raw64(g_ + 1); } while (0)
    // ring entries in batches: lane u forms the descriptor of entry e0 + u for a task in block cq (absolute block bq); the wave takes them
    // one by one through v_readlane.  Rows from the LDS row sets where one reaches the block, else from global memory
    unsigned long long d_gb = 0ull; double d_cf = 0.0; int d_loff = -1, d_klm = -1;
    auto describe_for = [&](uint32_t e0, int n, int cq, int bq) {
      const int sl = (int)((e0 + (uint32_t)min(lane, n - 1)) & (ring - 1));
      const int kk = accK[sl], slot = accT[sl];
      d_cf = (lane < n) ? accC[sl] : 0.0;
      const int ck = kk >> 7, kl = kk & 127, d = cq - ck;
      d_gb = (unsigned long long)(uintptr_t)tab[min(max(d, 0), 15)] + (((unsigned long long)bq * m + (unsigned long long)kl) * m) * sizeof(GT);
      d_loff = (slot >= 0 && d < S4C_NROW) ? ((slot * S4C_NROW + d) * SW_MAXM) * (int)sizeof(GT) : -1;
      d_klm = (ck == cq) ? kl : -1;
    };
    auto row_of_for = [&](int u, int tq) -> uint32_t {     // entry u of the described batch: this lane's element of its row
      const int lo = __builtin_amdgcn_readlane(d_loff, u);
      if (lo >= 0) return (uint32_t)*reinterpret_cast<const GT *>(reinterpret_cast<const unsigned char *>(nrow) + lo + 2 * tq);
      const unsigned long long gbu = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d_gb, u) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d_gb >> 32), u) << 32);
      return (uint32_t)((gGT *)(uintptr_t)gbu)[tq];
    };
    // the first 32 entries of a task's window, requested while the helper still works on its previous task
    uint32_t fg[32]; double fcf = 0.0; int fklm = -1, f_nf = 0; uint32_t f_napp = 0u;
#define S4C_BULK_REQUEST(T_) do { \
    const int Qn_ = (T_) >> 3, cn_ = (T_) >> 1, tn_ = 64 * ((T_) & 1) + lane; \
    f_napp = (Qn_ - DQ + 1 > 0) ? lds_ld32(posq + ((Qn_ - DQ + 1) & 15)) : 0u; \
    f_nf = (cn_ < nb) ? (int)min(32u, (uint32_t)(lds_ld64(ctl) >> 32) - f_napp) : 0; \
    fcf = 0.0; fklm = -1; \
    if (f_nf > 0) { \
      describe_for(f_napp, f_nf, cn_, a.blk_begin + cn_); \
      fcf = d_cf; fklm = d_klm; \
      _Pragma("unroll") for (int u_ = 0; u_ < 32; ++u_) fg[u_] = row_of_for(min(u_, f_nf - 1), tn_); \
    } else { _Pragma("unroll") for (int u_ = 0; u_ < 32; ++u_) fg[u_] = 0u; } } while (0)
#ifdef BWGR_STAMPS
    unsigned long long hst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, htl = __builtin_amdgcn_s_memtime();
#define S4H(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); hst[k] += t_ - htl; htl = t_; } while (0)
#else
#define S4H(k) do { } while (0)
#endif
    // stage 1 of task T_ from the constants in n_* and the rows in fg: q complete -> r, the first rows applied.  Returns false (nothing
    // done) while the streamers' sums are still incomplete and `block` is false
    double r_nx = 0.0; uint32_t napp_nx = 0u; bool have_nx = false;
    auto stage1 = [&](int Tn, bool block) -> bool {
      const int cn = Tn >> 1, tn = 64 * (Tn & 1) + lane;
      const bool lw = cn < nb;
      const int bk = a.blk_begin + min(cn, nb - 1);
      const bool vl = lw && tn < min(m, a.p - bk * m);
      if (lw) {
        const unsigned long long need = (unsigned long long)A.K3;
        const unsigned long long *gq = A.qsum + ((size_t)bk * SW_MAXM + tn) * 2;
        const uint64_t t0 = block ? wall_clock64() : 0;
        unsigned spins = 0;
        for (;;) {
          if (__ballot(vl && ((n_qlo & 0xFFull) != need || (n_qhi & 0xFFull) != need)) == 0ull) break;
          if (A.dbg & 8) break;
          n_qlo = ld_agent_raw64(gq); n_qhi = ld_agent_raw64(gq + 1);
          if (!block) { if (__ballot(vl && ((n_qlo & 0xFFull) != need || (n_qhi & 0xFFull) != need)) == 0ull) break; return false; }
          if ((++spins & 63u) == 0u) {
            if (ld_agent_u32(abortw) != 0u || (uint32_t)lds_ld64(ctl + 1) != 0u) { failed = true; return false; }
            if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); failed = true; return false; }
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      double r = vl ? fma((double)((long long)n_qhi >> 8), 16777216.0, (double)((long long)n_qlo >> 8)) * invS - n_sx : 0.0;
      uint32_t na = f_napp;
      if (f_nf > 0) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const double cf = readlane_f64(fcf, u);
          const int km = __builtin_amdgcn_readlane(fklm, min(u, f_nf - 1));
          r = fma(-(double)fg[u], (tn > km) ? cf : 0.0, r);
        }
        na += (uint32_t)f_nf;
      }
      r_nx = r; napp_nx = na; have_nx = true;
      return true;
    };
    if (hi < ntask) { S4C_LOAD_CONSTS(hi, n_); S4C_BULK_REQUEST(hi); }
    for (int T = hi; T < ntask && !failed; T += S4C_NH) {
      S4H(0);
      const int Q = T >> 3, c = T >> 1, hw = T & 1, t = 64 * hw + lane;
      const bool live_w = c < nb;
      const int blk = a.blk_begin + min(c, nb - 1);
      const int mBc = live_w ? min(m, a.p - blk * m) : 0;
      const bool valid = live_w && t < mBc;
      if (!have_nx) { stage1(T, true); if (failed) break; }
      S4H(3);
      double r = r_nx; uint32_t napp = napp_nx; have_nx = false;
      c_sx = n_sx; c_zc = n_zc; c_hr = n_hr; c_qlo = n_qlo; c_qhi = n_qhi;
      (void)c_sx; (void)c_qlo; (void)c_qhi; (void)Q;
      const double hr_l = valid ? c_hr : INFINITY;
      bool next_loaded = false;      // the next task's constants requested (after this task's record is out)
      // ring entries [napp, to) into r, in ring order
      auto apply = [&](uint32_t to) {
        while (live_w && napp != to) {
          const int n = (int)min(8u, to - napp);
          describe_for(napp, n, c, blk);
          uint32_t g[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) g[u] = row_of_for(min(u, n - 1), t);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const double cf = readlane_f64(d_cf, u);
            const int km = __builtin_amdgcn_readlane(d_klm, min(u, n - 1));
            r = fma(-(double)g[u], (t > km) ? cf : 0.0, r);
          }
          napp += (uint32_t)n;
        }
        if (!live_w) napp = to;
      };
      S4H(4);
      // ---- follow the chain until it is LEAD tasks away, then publish and keep following until the group is adopted ----
      S4CTask &tr = TR[T & (S4C_NTR - 1)];
      int myslot = -1;                      // this lane's row-set slot
      int nann = 0;
      bool published = false, next_requested = false;
      uint32_t seq = 0u;
#ifdef BWGR_STAMPS
      bool first_pub = true;
#endif
      {
        const uint64_t t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
          const unsigned long long cw = lds_ld64(ctl);
          asm volatile("" ::: "memory");
          const int done = (int)(uint32_t)cw;
          const uint32_t nn = (uint32_t)(cw >> 32);
          const bool fresh = (napp != nn);
          apply(nn);
          const bool due = done >= T - S4C_LEAD;
          if (due && (fresh || !published)) {
            // candidates without a row set: announce (up to four per task from here; the chain wave finds later ones itself).  The rows
            // are requested and the lanes' slots recorded before the record goes out; their landing is waited for after it
            int newk[4] = {-1, -1, -1, -1}, news[4] = {-1, -1, -1, -1};
            float v_b0[4], v_xxb0[4], v_drej[4], v_b2[4], v_tacc[4], v_trej[4]; double v_rden[4], v_sdz1[4], v_gjj[4], v_ha[4];
            int nnew = 0;
            if (live_w) {
              unsigned long long cand = __ballot(valid && !(fabs(r - c_zc) < hr_l) && myslot < 0);
              while (cand != 0ull && nann < 4) {
                const int js = (int)__builtin_ctzll(cand);
                cand &= cand - 1ull;
                const int kk = 64 * T + js, kl = 64 * hw + js;
                const int sl = slot_alloc(kk);
                if (sl < 0) break;
                if (lane == js) myslot = sl;
                rows_request(kk, sl);
                const StageBuf &st = a.ps.blocks[blk];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q == nnew) {
                  newk[q] = kk; news[q] = sl;
                  v_b0[q] = st.b0[kl]; v_xxb0[q] = st.xxb0[kl]; v_drej[q] = st.drej[kl]; v_b2[q] = st.b2[kl]; v_tacc[q] = st.tacc[kl]; v_trej[q] = st.trej[kl];
                  v_rden[q] = st.rden[kl]; v_sdz1[q] = st.sdz1[kl]; v_gjj[q] = a.ps.spec[blk].gjj[kl]; v_ha[q] = a.ps.quick[blk].ha[kl];
                }
                ++nnew; ++nann;
              }
            }
            // the record, under its sequence lock
            seq += 1u; lds_st32(&tr.seq, seq);
            asm volatile("" ::: "memory");
            tr.r[lane] = r; tr.gs[lane] = myslot;
            if (!published) { tr.zc[lane] = c_zc; tr.hr[lane] = hr_l; }
            lds_st32(&tr.W, napp);
            lds_st64(&tr.an, __ballot(myslot >= 0));
            asm volatile("" ::: "memory");
            seq += 1u; lds_st32(&tr.seq, seq);
            if (!published) { asm volatile("" ::: "memory"); lds_st32(&tr.ready, (uint32_t)(T + 1)); published = true; }
#ifdef BWGR_STAMPS
            if (first_pub) { first_pub = false; S4H(5); hst[7] += (unsigned long long)(T - done); }
#endif
            if (nnew > 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (q < nnew) {
                S4CRec &cr = crec[news[q]];
                if (lane == 0) { cr.b0 = v_b0[q]; cr.xxb0 = v_xxb0[q]; cr.drej = v_drej[q]; cr.b2 = v_b2[q]; cr.tacc = v_tacc[q]; cr.trej = v_trej[q]; cr.rden = v_rden[q]; cr.sdz1 = v_sdz1[q]; cr.gjj = v_gjj[q]; cr.ha = v_ha[q]; }
              }
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the rows have landed, the records are written
#pragma unroll
              for (int q = 0; q < 4; ++q) if (q < nnew) lds_st32(reinterpret_cast<uint32_t *>(&crec[news[q]].state), (uint32_t)(newk[q] + 1));
            }
          }
          // this helper's next task while this one is followed: its constants and first rows are requested once the record is out, and its
          // stage 1 is done as soon as they and the streamers' sums are there -- the switch to it then costs nothing
          if (published && !next_requested && T + S4C_NH < ntask) { next_requested = true; next_loaded = true; S4C_LOAD_CONSTS(T + S4C_NH, n_); S4C_BULK_REQUEST(T + S4C_NH); }
          else if (next_loaded && !have_nx && !fresh) { stage1(T + S4C_NH, false); if (failed) break; }
          if (done >= T - S4C_LOOK + 1 && published) break;   // adopted (the chain wave adopts task T before it decides task T - 3)
          if ((++spins & 255u) == 0u) {
            if ((uint32_t)lds_ld64(ctl + 1) != 0u) { failed = true; break; }
            if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); failed = true; break; }
          }
          if (!due) __builtin_amdgcn_s_sleep(1);
        }
      }
      if (failed) break;
      S4H(6);
    }
#undef S4C_LOAD_CONSTS
#undef S4C_BULK_REQUEST
#ifdef BWGR_STAMPS
    if (lane == 0 && a.stamps) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&a.stamps[80 + 8 * wave + k_], hst[k_]);
#endif
    if (failed) lds_st64(ctl + 1, 1ull);
  }
  __syncthreads();
  if ((uint32_t)lds_ld64(ctl + 1) != 0u) { if (tid == 0) a.sc->error = 1u; return; }
  // (the posterior sums and the outputs of the markers that were not included: k_sweep4_finish)
}

// ------------------------------------------------------------------------------------------------------------------
// prefetcher: one workgroup on the sequencer's XCD that walks a few quads ahead of the sequencer and touches one dword per 128-byte
// line of what the sequencer's waves will load (constants, speculative terms, radii) and of the packed diagonal Gram blocks (the rows
// of just-included markers are read on demand, inside the chain), so that those reads are L2 hits.  Paced by the lists; speed only.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s4_prefetcher(const Sweep4Args &A, int pi) {
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, nb = a.blk_end - a.blk_begin;
  constexpr int AHEADQ = 3;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  uint32_t sink = 0u;
  constexpr size_t gbytes = (size_t)SW_MAXM * SW_MAXM * 2;   // one 16-bit Gram block
  for (int c = pi; c < nb; c += A.npf) {
    const int Q = c / S4_QB;
    if (Q >= AHEADQ) {   // wait (one lane polls) until the sequencer has published the list of quad Q - AHEADQ
      const unsigned long long *L = A.lists + (size_t)(Q - AHEADQ) * S4_LSTRIDE;
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        if (s3_epoch_is(ld_agent_raw64(L), A.epoch)) break;
        if ((++spins & 63u) == 0u && (ld_agent_u32(abortw) != 0u || wall_clock64() - t0 > SW_TIMEOUT_TICKS)) return;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    const int blk = a.blk_begin + c;
    // every load unconditional (clamped), all of them in flight together: a load under a branch makes the compiler wait for the one before
    const unsigned char *stb = reinterpret_cast<const unsigned char *>(a.ps.blocks + blk);
    const unsigned char *spb = reinterpret_cast<const unsigned char *>(a.ps.spec + blk);
    const unsigned char *qkb = reinterpret_cast<const unsigned char *>(a.ps.quick + blk);
    const size_t o = (size_t)tid * 128;
    const uint32_t v0 = *reinterpret_cast<const uint32_t *>(stb + min(o, sizeof(StageBuf) - 4));
    const uint32_t v1 = *reinterpret_cast<const uint32_t *>(spb + min(o, sizeof(SpecBuf) - 4));
    const uint32_t v2 = *reinterpret_cast<const uint32_t *>(qkb + min(o, sizeof(QuickBuf) - 4));
    // the Gram rows towards this block from the blocks at distance 0..3: 4 x 256 lines of 128 bytes, two per thread
    uint32_t v3 = 0u, v4 = 0u;
    if (!(A.dbg & 8192)) {
      const int a0 = tid >> 8, a1 = 2 + (tid >> 8);                       // which array: 0 the diagonal blocks, d the distance-d cross blocks
      const size_t lo = (size_t)(tid & 255) * 128;
      const unsigned char *g0 = reinterpret_cast<const unsigned char *>(a0 == 0 ? A.gd : A.gx[a0 - 1]);
      const unsigned char *g1 = reinterpret_cast<const unsigned char *>(A.gx[a1 - 1]);
      const bool h0 = g0 != nullptr && c >= a0, h1 = g1 != nullptr && c >= a1;
      v3 = *reinterpret_cast<const uint32_t *>((h0 ? g0 : stb) + (h0 ? (size_t)blk * gbytes + lo : 0));
      v4 = *reinterpret_cast<const uint32_t *>((h1 ? g1 : stb) + (h1 ? (size_t)blk * gbytes + lo : 0));
    }
    sink += v0 + v1 + v2 + v3 + v4;
  }
  if (sink == 0x9E3779B9u && a.stamps) a.stamps[255] = sink;   // (keeps the loads alive)
}

// After a sweep whose sequencer was the chain wave: the markers that were NOT included (d = 0: the launch zeroes d, the chain wave
// writes b and d = 1 of the included ones) take their rejected draw b2, every marker its variance draw, and the two posterior sums of
// the launch's markers are formed in a fixed order (256 partial sums, then one thread block).
__global__ __launch_bounds__(256) void k_sweep4_finish(const SweepArgs a, double *part) {
  if (!(a.sc->inc_rate < a.gate3)) return;
  const int j0 = a.blk_begin * a.m, j1 = min(a.p, a.blk_end * a.m);
  const bool vbv = (a.flags & SWF_VB_VEC) != 0;
  const float Sb = a.sc->Sb;
  double sd = 0.0, sb2 = 0.0;
  for (int j = j0 + (int)(blockIdx.x * blockDim.x + threadIdx.x); j < j1; j += (int)(gridDim.x * blockDim.x)) {
    const StageBuf &st = a.ps.blocks[j / a.m];
    const int t = j % a.m;
    const float dn = a.d[j];
    float bn;
    if (dn != 0.0f) bn = a.b[j]; else { bn = st.b2[t]; a.b[j] = bn; }
    if (vbv) a.vb[j] = (float)((double)(Sb + bn * bn) / st.chi[t]);
    sd += (double)dn;
    sb2 = fma((double)bn, (double)bn, sb2);
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = sd; red[1][threadIdx.x] = sb2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = red[0][0]; part[2 * blockIdx.x + 1] = red[1][0]; }
}
__global__ __launch_bounds__(64) void k_sweep4_finish2(const SweepArgs a, const double *part, int nparts) {
  if (!(a.sc->inc_rate < a.gate3)) return;
  if (threadIdx.x != 0) return;
  double sd = 0.0, sb2 = 0.0;
  for (int i = 0; i < nparts; ++i) { sd += part[2 * i]; sb2 += part[2 * i + 1]; }
  a.sc->sum_d += sd; a.sc->sum_b2 += sb2;
}

template <int SS>
__global__ __launch_bounds__(SW_THREADS) void k_sweep4(const Sweep4Args A) {
  if (!(A.a.sc->inc_rate < A.a.gate3)) return;   // this sweep is k_sweep2's (dense inclusion: every workgroup sees the same scalar)
  const int b = (int)blockIdx.x;
  if (b > 0 && (b & 7) == 0 && (b >> 3) <= A.npf) { s4_prefetcher(A, (b >> 3) - 1); return; }
  if (blockIdx.x == 0) { if (!(A.dbg & 1024)) { if (A.seq == 2) s4_sequencer_cw(A); else s4_sequencer(A); } }
  else if ((A.a.flags & SWF_DEBUG_WITHHOLD) && blockIdx.x == 1) return;   // test hook: a streamer that never shows up
  else if (!(A.dbg & 2048)) s4_streamer<SS>(A);
}

}  // namespace bwgr
