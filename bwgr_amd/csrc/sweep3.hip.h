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

#ifdef BWGR_STAMPS
// diagnostic build: per-phase s_memtime sums of streamer 0's first update wave (stamps[0..7]) and first dots wave (8..15), of the
// sequencer's wave 0 (16..23) and of its helper waves (24..31), flushed once at the end
#define S3ST_DECL unsigned long long ph3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl3 = __builtin_amdgcn_s_memtime()
// (BWGR_STAMPS=2: only the stamps around the block barrier, slots 0 and 4 -- "busy" and "waiting" per role; a stamp costs a few
// hundred cycles because it drains the wave's LDS / scalar counter, so the full set distorts the roles it measures)
#define S3ST(k, cond) do { if ((BWGR_STAMPS == 1 || (BWGR_STAMPS == 2 && ((k) == 0 || (k) == 1 || (k) == 4)) || (BWGR_STAMPS == 3 && ((k) == 0 || (k) == 4) && blockIdx.x == 0 && threadIdx.x == 0) || (BWGR_STAMPS == 4 && (k) <= 5 && blockIdx.x == 0 && threadIdx.x == 0)) && (cond)) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph3[k] += t_ - tl3; tl3 = t_; } } while (0)
#define S3ST_FLUSH(base, cond) do { if ((cond) && a.stamps) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&a.stamps[(base) + k_], ph3[k_]); } while (0)
#else
#define S3ST_DECL do { } while (0)
#define S3ST(k, cond) do { } while (0)
#define S3ST_FLUSH(base, cond) do { } while (0)
#endif

static constexpr int S3_MAXD = 16;                    // deepest fold-in lag, blocks
static constexpr int S3_LSTRIDE = 2 * SW_MAXM + 2;    // 8-byte words of one block's list: header, two per entry, one spare
static constexpr int S3_ND = 7;                       // signed base-256 digits of the fixed-point residual and steps (|q| < 2^55)
static constexpr int S3_RING = S3_MAXD * SW_MAXM;     // flat ring of included markers in the sequencer's LDS (a power of two)
static constexpr int S3_GPD_BYTES = 16384;            // LDS copy of a packed diagonal Gram block, 16-bit entries (8128 of them at m = 128)
static constexpr int S3_ROWSLOT = 1024;               // bytes of LDS per such marker: a 64-lane x 16-byte DMA (the two rows are its first 4 m bytes)
static constexpr int S3_NRX = 8;                      // distance-1 / 2 rows of the first S3_NRX included markers of a block land in LDS by DMA
static constexpr int S3_OS = 12;                      // dwords per row of the int32 recombination scratch (8 used; b128 reads conflict-free)
static constexpr int S3_NFW = 4;                      // far-field waves of the sequencer: 5, 6 and the two staging waves (2, 3)
static constexpr int S3_NFL = 8;                      // rows a far-field wave keeps in flight (the rest of a dense block's rows are read in place)
static constexpr int S3_QRAW_BYTES = SW_MAXM * 16;    // a block's slab-dot sums as the streamers' atomics leave them: two 8-byte words per marker

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
  int dbg;                       // experiment switches (BWGR_DBG3)
  int pf;                        // blockIdx of the prefetcher workgroup (shares the sequencer's XCD), or -1
  int pf2;                       // ... of the second one (the included markers' distance-1 / 2 rows), or -1
  int qsplit;                    // 1: a marker's two slab-dot words hold digits 0-3 and 4-6 (k_sweep3's streamers: no LDS round trip between MFMA and atomics); 0: digits 0-2 and 3-6 (k_sweep3p)
  int skip_vb;                   // 1: the per-marker variances are formed after the sweep (k_vb_fill), not by the sequencer's wave 7
  const unsigned char *gx12;     // 16-bit panels: [nblocks][m][2][m] uint16, marker k of block b against blocks b+1 and b+2 side by side (k_near_rows):
                                 // an included marker's distance-1 and distance-2 rows in ONE LDS-DMA; nullptr: two requests from gx[0], gx[1]
};

// gx12[b][k][d-1][j] = gx[d-1][b+d][k][j] (zero past the last block)
__global__ void k_near_rows(const uint16_t *g1, const uint16_t *g2, uint16_t *out, int m, int64_t nblocks) {
  const int64_t total = nblocks * m * 2 * m;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(t % m), d = (int)((t / m) & 1), k = (int)((t / (2 * m)) % m);
    const int64_t b = t / ((int64_t)2 * m * m);
    const uint16_t *g = d ? g2 : g1;
    out[t] = (g && b + d + 1 < nblocks) ? g[((b + d + 1) * m + k) * m + j] : (uint16_t)0;
  }
}

// ---- fixed-point scale of one sweep.  With 2^k above both the largest |e_i| at the start of the sweep and the largest step
// |x| * |drej_j| a marker can apply to a row, e_fixed = e * 2^sh with sh = 44 - k: eleven bits of headroom below the 55-bit
// digit range (within a sweep the residual random-walks over p steps: |e| may pass its starting maximum several times over,
// and where p > n the steps themselves are larger than the residual), and a grid 2^-44 relative to that scale, far below the
// last bit of a float step.  k_prestage leaves the largest exponent field of drej in sc->e3_dex (reset here after use). ----
__global__ void k_escale_reset(ChainScalars *sc) { sc->e3_dex = 0u; }
// (sh_add: test hook, BWGR_DEBUG_SH_ADD -- a finer grid with that many bits less headroom, so that a test can leave the range on purpose)
__global__ __launch_bounds__(1024) void k_escale(const double *e, int64_t ld, ChainScalars *sc, int xbits, float gate3, int sh_add = 0) {
  if (!(sc->inc_rate < gate3)) { if (threadIdx.x == 0) sc->e3_dex = 0u; return; }   // this sweep is k_sweep2's
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
    sc->e3_sh = 44 - k + sh_add;
    sc->e3_dex = 0u;
  }
}
__device__ __forceinline__ double s3_pow2(int k) { return __hiloint2double((1023 + k) << 20, 0); }   // 2^k, |k| < 1022

// ---- implicitly centred sweeps (SWF_CENTRE; the reference's own sweep on centred columns, src/Rcpp20260726ai.cpp:668-682 with X_j -> X_j - mean_j) ----
// With s_j = sum_i x_ij and e = e_stored + shift * 1 (shift = sum over the markers swept so far of (s_k / n) delta_k),
//   (x_j - (s_j / n) 1)' e = x_j' e_stored - (s_j / n) E,   E = sum(e_stored) = E_0 - sum_{k < j} s_k delta_k,
// so the streamers, the Gram arrays and the lists stay those of the raw int8 columns and the sequencer adds scalars: the rejected steps' part of
// E is known before the sweep (cpre: running block sums; the in-block part goes into spec), the included markers' part is one running scalar.
// Which engine a centring helper serves (every launch of an iteration is enqueued; the device decides which side runs):
//   mode 0: k_sweep3 (rejected steps on the sweep's fixed-point grid)  -- when the chain's inclusion rate is below the gate
//   mode 1: k_sweep2 (the float steps themselves)                     -- the other side of the gate, or no gate at all
//   mode 2: k_sweep2 redoing a fixed-point sweep that left its range   -- when sc->redo is set
__device__ __forceinline__ bool cen_active(const SweepArgs &a, int mode) {
  if (mode == 2) return a.sc->redo != 0u;
  const bool three = a.gate3 > 0.0f && a.sc->inc_rate < a.gate3;
  return mode == 0 ? three : !three;
}
// k_cen_tot: cpre[b + 1] = sum over block b of s_k * drej_k (mode 0: on the sweep's grid);  k_cen_scan: the inclusive scan, cpre[0] = 0
__global__ __launch_bounds__(128) void k_cen_tot(const SweepArgs a, int blk_begin, int mode) {
  if (!cen_active(a, mode)) return;
  const int blk = blk_begin + blockIdx.x, j = threadIdx.x, m = a.m;
  const int mB = min(m, a.p - blk * m);
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(mode == 0 ? sh : 0), invS = s3_pow2(mode == 0 ? -sh : 0);
  __shared__ double red[128];
  const double dj = (j < mB) ? (double)a.ps.blocks[blk].drej[j] : 0.0;
  red[j] = (j < mB) ? (double)a.csum[blk * m + j] * (mode == 0 ? rint(dj * S) * invS : dj) : 0.0;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) { if (j < o) red[j] += red[j + o]; __syncthreads(); }   // (a fixed tree: the same bits on every run)
  if (j == 0) a.cpre[blk + 1] = red[0];
}
__global__ __launch_bounds__(1024) void k_cen_scan(const SweepArgs a, int nblocks, int mode) {
  if (!cen_active(a, mode)) return;
  __shared__ double part[1024];
  const int t = threadIdx.x, per = (nblocks + 1023) / 1024;
  const int b0 = min(nblocks, t * per), b1 = min(nblocks, b0 + per);
  double s = 0.0;
  for (int b = b0; b < b1; ++b) s += a.cpre[b + 1];
  part[t] = s;
  __syncthreads();
  if (t == 0) { double run = 0.0; for (int i = 0; i < 1024; ++i) { const double v = part[i]; part[i] = run; run += v; } }   // (serial: 1 024 adds, once per iteration)
  __syncthreads();
  double run = part[t];
  for (int b = b0; b < b1; ++b) { run += a.cpre[b + 1]; a.cpre[b + 1] = run; }
  if (t == 0) a.cpre[0] = 0.0;
}
// before a centred launch over [blk_begin, blk_end): u0 = -(sum(e) + cpre[blk_begin]) / n  (one workgroup; the residual's padded rows are zero)
__global__ __launch_bounds__(1024) void k_cen_begin(const SweepArgs a, int mode) {
  if (!cen_active(a, mode)) return;
  __shared__ double red[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int64_t i = t; i < a.ld; i += 1024) s += a.e[i];
  red[t] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) { a.sc->cen_u0 = -(red[0] + a.cpre[a.blk_begin]) * a.ninv; a.sc->cen_c = 0.0; }
}
// after it: e = e_stored + shift on the n real rows, shift = (cpre[blk_end] - cpre[blk_begin]) / n + cen_c  (not after a sweep that left the range:
// the recovery restores the state)
__global__ void k_cen_end(const SweepArgs a, int mode) {
  if (!cen_active(a, mode) || a.sc->error != 0u) return;
  const double shift = (a.cpre[a.blk_end] - a.cpre[a.blk_begin]) * a.ninv + a.sc->cen_c;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) a.e[i] += shift;
}

// k_spec3: spec_j = sum_{k<j, same block} G_kj * drej_k with drej on the sweep's fixed-point grid (what the streamers apply),
// and the Gram diagonal.  One workgroup of 128 threads per block, thread = marker j, four partial sums.
// SWF_CENTRE: spec_j -= (s_j / n) * (cpre[blk] + sum_{k<j, same block} s_k drej_k) -- the rejected steps' share of -(s_j / n) E -- and the
// Gram-diagonal slot carries {G_jj, s_j} as two 32-bit integers (the sequencer forms G_jj - s_j^2 / n).
__global__ __launch_bounds__(128) void k_spec3(const SweepArgs a, int blk_begin, const uint16_t *gp16 = nullptr) {
  if (!(a.sc->inc_rate < a.gate3)) return;   // this sweep is k_sweep2's
  const int blk = blk_begin + blockIdx.x, j = threadIdx.x, m = a.m;
  const int mB = min(m, a.p - blk * m);
  const int32_t *G = reinterpret_cast<const int32_t *>(a.gram) + (size_t)blk * m * m;
  SpecBuf &sp = a.ps.spec[blk];
  __shared__ double dr[128];
  __shared__ double cs[128];
  const bool cen = (a.flags & SWF_CENTRE) != 0;
  const int sh = a.sc->e3_sh;
  const double S = s3_pow2(sh), invS = s3_pow2(-sh);
  dr[j] = (j < mB) ? rint((double)a.ps.blocks[blk].drej[j] * S) * invS : 0.0;
  const int sj = (cen && j < mB) ? a.csum[blk * m + j] : 0;
  cs[j] = (double)sj * dr[j];
  __syncthreads();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, gjj = 0.0, ib = 0.0;
  if (j < mB) {
    gjj = (double)G[(size_t)j * m + j];
    int k = 0;
    if (gp16) {
      // 16-bit panels: the same entries from the packed strict upper triangle (16 KB a block instead of the 64 KB square: the kernel is bound by
      // HBM, 0.5 GB a C4 iteration through the square); entry (k, j > k) at prow(k) + j - k - 1, the same summation order
      const uint16_t *Gp = gp16 + (size_t)blk * a.pstride + (j - 1);
      int pr = 0;   // prow(k) - k, stepped: prow(k) = k (m - 1) - k (k - 1) / 2
      for (; k + 4 <= j; k += 4) {
        const int p0 = pr, p1 = p0 + (m - 2 - k), p2 = p1 + (m - 3 - k), p3 = p2 + (m - 4 - k);
        s0 = fma((double)Gp[p0], dr[k], s0);
        s1 = fma((double)Gp[p1], dr[k + 1], s1);
        s2 = fma((double)Gp[p2], dr[k + 2], s2);
        s3 = fma((double)Gp[p3], dr[k + 3], s3);
        pr = p3 + (m - 5 - k);
      }
      for (; k < j; ++k) { s0 = fma((double)Gp[pr], dr[k], s0); pr += m - 2 - k; }
    } else {
    for (; k + 4 <= j; k += 4) {
      s0 = fma((double)G[(size_t)k * m + j], dr[k], s0);
      s1 = fma((double)G[(size_t)(k + 1) * m + j], dr[k + 1], s1);
      s2 = fma((double)G[(size_t)(k + 2) * m + j], dr[k + 2], s2);
      s3 = fma((double)G[(size_t)(k + 3) * m + j], dr[k + 3], s3);
    }
    for (; k < j; ++k) s0 = fma((double)G[(size_t)k * m + j], dr[k], s0);
    }
    if (cen) for (int k2 = 0; k2 < j; ++k2) ib += cs[k2];
  }
  double spec = (s0 + s1) + (s2 + s3);
  if (cen) {
    spec -= ((double)sj * a.ninv) * (a.cpre[blk] + ib);
    gjj = __hiloint2double(sj, (j < mB) ? (int)G[(size_t)j * m + j] : 0);
  }
  sp.spec[j] = spec; sp.xspec[j] = 0.0; sp.gjj[j] = gjj;
}

// k_vb_fill: the per-marker variances of the markers a k_sweep3 launch swept, vb_j = (Sb + b_j^2) / chi_j (src/Rcpp20260726ai.cpp, BayesA / B) -- the
// expression every sweep engine uses, formed chip-wide after the sweep instead of by the sequencer's wave 7 (Sweep3Args::skip_vb)
__global__ void k_vb_fill(const SweepArgs a, int j_begin, int j_end) {
  const float Sb = a.sc->Sb;
  for (int j = j_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x); j < j_end; j += (int)(gridDim.x * blockDim.x)) {
    const float bn = a.b[j];
    a.vb[j] = (float)((double)(Sb + bn * bn) / a.ps.blocks[j / a.m].chi[j % a.m]);
  }
}

// The seven signed base-256 digits of q (|q| < 2^55) as bytes: adding 0x80 to each of the seven low bytes turns the signed digits
// into the plain bytes of the sum (q + sum 128 * 256^n = sum (d_n + 128) 256^n), and d_n = byte_n - 128 = byte_n ^ 0x80 as int8.
__device__ __forceinline__ void s3_put_digits7(long long q, int8_t *dst, int stride) {
  const unsigned long long u = ((unsigned long long)q + 0x0080808080808080ull) ^ 0x0080808080808080ull;
  const uint32_t lo = (uint32_t)u, hi = (uint32_t)(u >> 32);
  dst[0] = (int8_t)lo; dst[stride] = (int8_t)(lo >> 8); dst[2 * stride] = (int8_t)(lo >> 16); dst[3 * stride] = (int8_t)(lo >> 24);
  dst[4 * stride] = (int8_t)hi; dst[5 * stride] = (int8_t)(hi >> 8); dst[6 * stride] = (int8_t)(hi >> 16);
}

// list words
__device__ __forceinline__ unsigned long long s3_hdr(uint32_t epoch, int count) { return ((unsigned long long)epoch << 40) | (0xFFull << 32) | (unsigned long long)(uint32_t)count; }
__device__ __forceinline__ bool s3_epoch_is(unsigned long long w, uint32_t epoch) { return (uint32_t)(w >> 40) == epoch; }

typedef unsigned int s3_u4 __attribute__((ext_vector_type(4)));

// LDS-DMA as inline asm (16 or 4 bytes per lane to lds_base + 16 / 4 * lane): hipcc tracks the builtin's LDS writes and puts a
// vmcnt(0) in front of the next LDS access it cannot prove disjoint, which is exactly the wait these copies are meant to avoid
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void s3_dma16(const void *gsrc, const void *lds_base) {
  const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)lds_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ void s3_dma4(const void *gsrc, const void *lds_base) {
  const uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)lds_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(la) : "memory", "m0");
}
// ... and from a wave-uniform base (SGPR pair) plus a per-lane byte offset, to a uniform LDS address: no 64-bit vector arithmetic
__device__ __forceinline__ void s3_dma4s(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(gbase), "s"(la) : "memory", "m0");
}
// 16 bytes per lane
__device__ __forceinline__ void s3_dma16s(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  // (the base is wave-uniform; say so, for the callers whose pointer reaches here through a struct built in registers)
  const unsigned long long gb_ = (unsigned long long)(uintptr_t)gbase;
  const unsigned char *sb_ = reinterpret_cast<const unsigned char *>((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb_) |
                                                                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb_ >> 32)) << 32));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sb_), "s"(la) : "memory", "m0");
}
// ... bypassing this CU's L1 (sc1): words that other CUs' atomics complete
__device__ __forceinline__ void s3_dma16s_sc1(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  const unsigned long long gb_ = (unsigned long long)(uintptr_t)gbase;
  const unsigned char *sb_ = reinterpret_cast<const unsigned char *>((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb_) |
                                                                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(gb_ >> 32)) << 32));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" : : "v"(voff), "s"(sb_), "s"(la) : "memory", "m0");
}
#pragma clang diagnostic pop

// wait until at most n of this wave's memory operations are still in flight (n known only at run time: the instruction takes an immediate)
__device__ __forceinline__ void s3_wait_vmcnt(int n) {
  switch (n) {
#define S3_WV(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
    S3_WV(1) S3_WV(2) S3_WV(3) S3_WV(4) S3_WV(5) S3_WV(6) S3_WV(7) S3_WV(8) S3_WV(9) S3_WV(10) S3_WV(11) S3_WV(12) S3_WV(13) S3_WV(14) S3_WV(15) S3_WV(16)
    S3_WV(17) S3_WV(18) S3_WV(19) S3_WV(20) S3_WV(21) S3_WV(22) S3_WV(23) S3_WV(24) S3_WV(25) S3_WV(26) S3_WV(27) S3_WV(28) S3_WV(29) S3_WV(30) S3_WV(31) S3_WV(32)
#undef S3_WV
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (0, or more than the cases cover: everything)
  }
}

// which streamer a workgroup is (blockIdx 0 is the sequencer, A.pf the prefetcher)
__device__ __forceinline__ int s3_stream_index(const Sweep3Args &A) {
  const int x = (int)blockIdx.x;
  return x - 1 - ((A.pf >= 0 && x > A.pf) ? 1 : 0) - ((A.pf2 >= 0 && x > A.pf2) ? 1 : 0);
}

__host__ __device__ inline size_t s3_streamer_dma_lds(int R3 = 128) {   // (four unpadded tiles at 128 rows, three at 256)
  const size_t Rp = (size_t)R3 + 16;
  return (R3 == 128 ? 4 : 3) * (size_t)SW_MAXM * R3 + 2 * 16 * Rp + 2 * 16 * (size_t)S2_DP + (size_t)64 * S3_OS * 4 * 4 + (size_t)8 * 32 * S3_OS * 4 + 64 + (size_t)4 * SW_MAXM * 4;
}
__host__ __device__ inline size_t s3_streamer_lds(int R3) {
  const size_t Rp = (size_t)R3 + 16;
  return 2 * (size_t)SW_MAXM * Rp + 2 * 16 * Rp + 2 * 16 * (size_t)S2_DP + (size_t)64 * S3_OS * 4 * 4 + (size_t)8 * 32 * S3_OS * 4 + 64;
}
__host__ __device__ inline size_t s3_seq_lds(int D, bool g16) {
  size_t s = 3 * sizeof(StageBuf) + 3 * 2 * SW_MAXM * sizeof(double);            // constants, spec + gjj: rings of three blocks (landed by DMA)
  s += 2 * SW_MAXM * sizeof(double) + 2 * S3_NFW * SW_MAXM * sizeof(double);     // q sums, far-field partial sums
  s += 2 * 3 * SW_MAXM * sizeof(float);                                          // state of a block
  (void)D;
  s += (size_t)S3_RING * (sizeof(double) + sizeof(long long) + sizeof(int));       // the included markers of the last D blocks
  s += (size_t)2 * S3_NFW * S3_NFL * SW_MAXM * (g16 ? 2 : 4);                      // far-field rows in flight
  if (g16) s += (size_t)3 * S3_GPD_BYTES + (size_t)S3_NRX * S3_ROWSLOT;       // the packed diagonal block of three blocks, distance-1 / 2 rows of this block's included markers
  s += (size_t)3 * S3_QRAW_BYTES;                                                  // the slab-dot sums of three blocks as they land (two phases ahead)
  return s + 256 + 512;   // ... the far-field touches' landing pad, the control words and the array bases
}

// ------------------------------------------------------------------------------------------------------------------
// streamer
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s3_streamer(const Sweep3Args &A) {
#ifdef BWGR_EXPERIMENTS
  const int SDBG = A.dbg;
#else
  constexpr int SDBG = 0;   // (the experiment switches: -DBWGR_EXPERIMENTS)
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, grp = lane >> 4;
  const int w = s3_stream_index(A);
  const unsigned long long *lists_w = A.lists;
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
  int *outd = reinterpret_cast<int *>(smem + off); off += (size_t)8 * 32 * S3_OS * 4;     // [wave][marker 32][n]
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
  // two tiles in flight in two register sets (tile t travels in set t & 1): with one, a tile had a single block period to
  // arrive, and under load an HBM round trip on a streaming CU is longer than that -- the period could not drop below it
  s3_u4 ta0 = {0, 0, 0, 0}, ta1 = ta0, ta2 = ta0, ta3 = ta0, tb0 = ta0, tb1 = ta0, tb2 = ta0, tb3 = ta0;
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
    const unsigned long long *L = lists_w + (size_t)Bs * S3_LSTRIDE;
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

  // ---- prologue: tile 0 into LDS, tiles 1 and 2 in flight ----
  {
    s3_u4 &tp0 = ta0, &tp1 = ta1, &tp2 = ta2, &tp3 = ta3;
    S3_TILE_ISSUE(0);
    S3_TILE_COMMIT(0);
    if (nb > 2) S3_TILE_ISSUE(2);
  }
  if (nb > 1) { s3_u4 &tp0 = tb0, &tp1 = tb1, &tp2 = tb2, &tp3 = tb3; S3_TILE_ISSUE(1); }
  float drej_pre = a.ps.blocks[a.blk_begin].drej[tid & (SW_MAXM - 1)];   // (used by the last two waves)
  unsigned long long lpre = 0ull;
  __syncthreads();
  S3ST_DECL;
  const bool st_u = (w == 0 && tid == 0), st_d = (w == 0 && tid == 64 * NU);

  // one block; tp0..tp3: the register set of tile b+1 (committed here) and then of tile b+3 (requested here)
  auto step = [&](int b, s3_u4 &tp0, s3_u4 &tp1, s3_u4 &tp2, s3_u4 &tp3) -> bool {
    const int mB = blk_m(b), par = b & 1;
    S3ST(0, st_u || st_d);
    int8_t *tile = tile0 + (size_t)par * tile_b;
    int8_t *edig = edig0 + (size_t)par * 16 * Rp;
    int8_t *ddig = ddig0 + (size_t)par * 16 * S2_DP;
    // A: what the included markers of block b - D changed
    if (b >= D && upd && !(SDBG & 512)) { if (!fold_list(b - D, lpre)) ctl_s[0] = 1u; }
    S3ST(1, st_u);
    // B: digits of the residual rows and of this block's rejected steps
    if (upd) {
      if ((unsigned long long)(e_own + (1ll << 54)) >> 55) ctl_s[1] = 1u;       // left the 55-bit range
      s3_put_digits7(e_own, edig + 64 * wave + lane, Rp);
    } else if (tid >= SW_THREADS - SW_MAXM) {                                    // the last two waves (never update waves)
      const double qd = (tid - (SW_THREADS - SW_MAXM) < mB) ? rint((double)drej_pre * S) : 0.0;   // (unused markers: zero steps)
      if (!(fabs(qd) < 18014398509481984.0)) ctl_s[1] = 1u;                      // 2^54
      s3_put_digits7((long long)qd, ddig + (tid - (SW_THREADS - SW_MAXM)), S2_DP);
    }
    S3ST(2, st_u);
#if defined(BWGR_STAMPS) && BWGR_STAMPS == 2
    S3ST(0, st_u || st_d);                      // lite: slot 0 = busy (barrier exit .. barrier entry), slot 4 = waiting at the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    S3ST(4, st_u || st_d);
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    S3ST(3, st_u || st_d);
#endif
    if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return false; }
    // C: tile b+1 (in registers for two iterations) lands in the other buffer, whose last reader was block b-1; the loads of
    // tile b+3 go out into the registers just freed; the list of block b+1-D and the rejected steps of block b+1 are requested
    if (b + 1 < nb) S3_TILE_COMMIT(b + 1);
    S3ST(6, st_u || st_d);
    {   // the small requests first (older than the tile loads on the in-order memory counter, so waiting for them does not wait
        // for the tile), every one unconditional: a load under a branch makes the compiler drain the counter in front of it
      const int bn1 = min(b + 1, nb - 1);
      drej_pre = a.ps.blocks[a.blk_begin + bn1].drej[tid & (SW_MAXM - 1)];
      lpre = ld_agent_raw64(lists_w + (size_t)(a.blk_begin + max(bn1 - D, 0)) * S3_LSTRIDE + lane);
    }
    S3ST(7, st_u || st_d);
    if (b + 3 < nb && !(SDBG & 16)) S3_TILE_ISSUE(b + 3);
    S3ST(4, st_u || st_d);
    if (upd) {
      // ---- slab update with the rejected steps: out[row][n] = sum_markers x[row][marker] * digit_n(drej[marker]) ----
      // lane (m16, grp): row quad 16 wave + m16 (rows 4 * that + k for accumulator k); k slots (dword u, byte q) of step s0 are the
      // markers s0 + 16 u + 4 grp + q (the interleave keeps the four lane groups on different LDS banks)
      const int rowoff = 4 * (16 * wave + m16);
      s2_v4i acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      for (int s0 = 0; s0 < ((SDBG & 256) ? 0 : mB); s0 += 64) {
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
      S3ST(5, st_u);
    } else {
      // ---- slab dots of block b against the digits of e: markers in groups of 16, groups gm and gm + ND together on wave
      // NU + gm (the two groups' MFMAs, LDS round trips and atomics overlap) ----
      for (int gm = wave - NU; 16 * gm < m; gm += 2 * ND) {
        const int gm2 = gm + ND;
        const bool two = 16 * gm2 < m;
        const int8_t *bp = edig + (size_t)m16 * Rp + 16 * grp;
        const int8_t *ap = tile + (size_t)(16 * gm + m16) * Rp + 16 * grp;
        const int8_t *ap2 = tile + (size_t)(16 * (two ? gm2 : gm) + m16) * Rp + 16 * grp;
        s2_v4i acc = {0, 0, 0, 0}, acc2 = acc;
        for (int r = 0; r < ((SDBG & 256) ? 0 : R3); r += 64) {
          const s2_v4i bv = *reinterpret_cast<const s2_v4i *>(bp + r);
          // (digits as the A operand, genotypes as B: the output tile is [digit][marker], so that lane (marker m16, group grp) holds the marker's digits
          // 4 grp .. 4 grp + 3 and the two words are formed in registers -- the other order scattered a marker's digits over seven lanes and went through LDS)
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(bv, *reinterpret_cast<const s2_v4i *>(ap + r), acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bv, *reinterpret_cast<const s2_v4i *>(ap2 + r), acc2, 0, 0, 0);
        }
        if (grp < 2) {   // word grp of the marker: digits 0-3 (weight 1) or 4-6 (weight 2^32, the sequencer's), with the arrival count in the low byte
          const long long w1 = (long long)acc[0] + ((long long)acc[1] << 8) + ((long long)acc[2] << 16) + ((long long)acc[3] << 24);
          const long long w2 = (long long)acc2[0] + ((long long)acc2[1] << 8) + ((long long)acc2[2] << 16) + ((long long)acc2[3] << 24);
          unsigned long long *qs = A.qsum + ((size_t)(a.blk_begin + b) * SW_MAXM + 16 * gm + m16) * 2 + grp;
          if (!(SDBG & 8)) {
            __hip_atomic_fetch_add((gu64_t *)qs, (unsigned long long)((w1 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (two) __hip_atomic_fetch_add((gu64_t *)(qs + (size_t)(16 * (gm2 - gm)) * 2), (unsigned long long)((w2 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      S3ST(5, st_d);
    }
    return true;
  };
  for (int b = 0; b < nb; b += 2) {
    if (!step(b, tb0, tb1, tb2, tb3)) return;                     // b even: tile b+1 travels in set 1
    if (b + 1 < nb && !step(b + 1, ta0, ta1, ta2, ta3)) return;
  }
  S3ST_FLUSH(0, st_u); S3ST_FLUSH(8, st_d);
  // the lists of the last D blocks
  if (upd && !(SDBG & 512)) for (int bs = max(0, nb - D); bs < nb; ++bs) {
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

// The same streamer with its tiles landed by LDS-DMA (128-row streamers; BWGR_STREAM3=dma)
template <int R3, int NTB>
__device__ __forceinline__ void s3_streamer_dma(const Sweep3Args &A) {
#ifdef BWGR_EXPERIMENTS
  const int SDBG = A.dbg;
#else
  constexpr int SDBG = 0;   // (the experiment switches: -DBWGR_EXPERIMENTS)
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, grp = lane >> 4;
  const int w = s3_stream_index(A);
  const unsigned long long *lists_w = A.lists;
  const int m = a.m, R = a.R, D = A.D;
  constexpr int Rp = R3 + 16;         // (the digit rows' stride; the tiles are unpadded)
  constexpr int CH = R3 / 16;         // 16-byte chunks per marker
  constexpr int MPP = 1024 / R3;      // markers per 1 KiB piece
  constexpr int NPC = SW_MAXM / MPP;  // pieces per tile
  static_assert(R3 == 128 || R3 == 256, "sixteen or thirty-two pieces");
  const int slab = w / A.sub, hsub = w - slab * A.sub;
  const int nb = a.blk_end - a.blk_begin;
  constexpr int NU = R3 >> 6, ND = 8 - NU;   // update waves (64 rows each); the other waves form the dots
  const int8_t *Xs = reinterpret_cast<const int8_t *>(a.X) + (size_t)slab * a.p * R + (size_t)hsub * R3;   // marker j: Xs + j * R
  const int64_t row0 = (int64_t)slab * R + (int64_t)hsub * R3;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  constexpr size_t tile_b = (size_t)SW_MAXM * R3;
  int8_t *tile0 = reinterpret_cast<int8_t *>(smem);
  size_t off = NTB * tile_b;
  const uint32_t tile_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem);
  int8_t *edig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * Rp;     // [parity][n][row]
  int8_t *ddig0 = reinterpret_cast<int8_t *>(smem + off); off += (size_t)2 * 16 * S2_DP;  // [parity][n][marker]
  int *outu = reinterpret_cast<int *>(smem + off); off += (size_t)64 * S3_OS * 4 * 4;     // [update wave][row 64][n]
  int *outd = reinterpret_cast<int *>(smem + off); off += (size_t)8 * 32 * S3_OS * 4;     // [wave][marker 32][n]
  uint32_t *ctl_s = reinterpret_cast<uint32_t *>(smem + off); off += 64;                 // [0] failure, [1] overflow
  float *drej_s = reinterpret_cast<float *>(smem + off);                                 // [block % NTB][marker]: the blocks' rejected steps, landed by DMA with the tiles
  const uint32_t drej_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)drej_s);
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

  // tile moves: LDS-DMA, NPC 1 KiB pieces per tile (MPP markers each), PPW = NPC / 2 per DMA wave (waves 6 and 7: every DMA wave issues the same number of
  // requests per step, the wait count relies on it).  Lane l of piece pc fills LDS slot (marker 8 pc + (l >> 3), position l & 7) with the marker's
  // 16-byte chunk (l & 7) ^ ((l >> 3) & 7): chunk c of marker jj sits at position c ^ (jj & (CH - 1)), so that sixteen markers' equal chunks fall on
  // different banks without padding.  NTB tile buffers: tile t + NTB - 1 is requested right after the barrier of step t, into the buffer tile t - 1 left a
  // whole step ago, and is first read after the barrier of step t + NTB - 1 -- NTB - 2 block periods (and the rest of a step) to land; four buffers at
  // 128 rows.  Before a step's barrier a counted wait on the DMA waves (S3_DMA_BARRIER) covers the tile the step reads.
  // Roles by what a wave keeps on its (in-order) memory counter -- round 4: update waves 0-1 (the list words, compiler-managed waits); dots waves 2-5
  // (no load at all: their atomics are fire-and-forget, nothing ever waits behind them); DMA waves 6-7 (nothing but LDS-DMA requests, eight tile
  // pieces and the next block's rejected steps each per step, behind ONE counted wait).  Round 3 had all six non-update waves issue pieces, small
  // loads AND atomics: the counted wait then also waited for atomics (2 800 cycles under load), and the compiler's own wait for the small loads
  // drained the pieces a wave had just issued -- streamers alone 1.34 us per block, 1.03 with the atomics switched off.
  static_assert(NU == 2, "waves 0-1 update, 2-5 dots, 6-7 DMA");
  constexpr int NDOT = 4, PPW = NPC / 2;       // pieces per DMA wave and tile
  constexpr int LPS = PPW + 2;                 // DMA requests per DMA wave and step: the tile's pieces and two 256-byte halves of a block's rejected steps
  const uint32_t lane_mk = (uint32_t)(lane / CH), lane_cr = (uint32_t)(lane % CH);
  const int j_lo = a.blk_begin * m, j_hi = min(a.p, a.blk_end * m);
  const int wvs = __builtin_amdgcn_readfirstlane(wave);   // (a scalar: the LDS address goes into M0)
  auto drej_issue = [&](int t) {   // block t's rejected steps (clamped past the end), both DMA waves the same two requests: equal counts, the same bytes
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.ps.blocks[a.blk_begin + max(0, min(t, nb - 1))].drej);
    const uint32_t la = drej_la + (uint32_t)((max(t, 0) % NTB) * SW_MAXM * 4);
    s3_dma4s(src, (uint32_t)lane * 4u, la);
    s3_dma4s(src, 256u + (uint32_t)lane * 4u, la + 256u);
  };
  auto tile_issue = [&](int t) {   // set t = tile t and the rejected steps of block t + 1 (they are digitised BEFORE the barrier of step t + 1)
    if (wvs < 6) return;
    const int jb = blk_j0(min(t, nb - 1));
    const uint32_t la0 = tile_la + (uint32_t)__builtin_amdgcn_readfirstlane((int)((t % NTB) * (int)tile_b));
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
      const int pc = (wvs - 6) + 2 * u;
      const int jl = MPP * pc + (int)lane_mk;                      // the marker's index in the tile
      const int jj = min(jb + jl, j_hi - 1);
      const uint32_t voff = (uint32_t)(jj - j_lo) * (uint32_t)R + ((lane_cr ^ ((uint32_t)jl & (uint32_t)(CH - 1))) * 16u);   // (a launch's slab of the panel stays below 4 GiB: sweep3_args selects this streamer only then)
      s3_dma16s(reinterpret_cast<const unsigned char *>(Xs + (size_t)j_lo * R), voff, la0 + (uint32_t)pc * 1024u);
    }
    drej_issue(t + 1);
  };

  // the included markers of block bs (relative index): e -= x_k * corr_k for this wave's rows.  `pre` holds words 0..63 of
  // the list as requested one iteration ago (a list of up to 31 entries arrives with that single load)
  auto fold_list = [&](int bs, unsigned long long pre, int e_first = 0) -> int {   // (entries before e_first are folded already: fold_pre)
    const int Bs = a.blk_begin + bs;
    const unsigned long long *L = lists_w + (size_t)Bs * S3_LSTRIDE;
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
      for (int e0 = (c0 == 0) ? e_first : 0; e0 < nw / 2; e0 += 8) {              // eight columns in flight
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

  // The fold a step ahead.  A streamer that has fallen behind the sequencer finds the next step's list published already; its fold is then two
  // dependent round trips (the list's words, then the included markers' column bytes) in front of the step's barrier, on the update waves, while the
  // other six waves wait -- 0.6 us of a 1.5 us step, and the streamers, not the sequencer, set the period.  So the words are requested TWO steps ahead;
  // a step ahead (fold_prefetch, after the barrier) they are looked at, and if the list is complete the column bytes of its first NPRE markers are
  // requested and stay in flight through the step's matrix work; the next step's fold (fold_pre) is then a few multiply-adds.  A list that is not
  // there yet a step ahead takes the old path (fold_list) -- the streamer is ahead of the sequencer then, and waiting anyway.
  constexpr int NPRE = 4;
  int xp0 = 0, xp1 = 0, xp2 = 0, xp3 = 0;                 // the prefetched column bytes (this lane's row)
  long long cp0 = 0, cp1 = 0, cp2 = 0, cp3 = 0;           // their steps on the fixed-point grid
  int np = -1;                                            // entries of the list the next fold takes (-1: not seen a step ahead)
#ifndef BWGR_WORDS2
#define BWGR_WORDS2 0
#endif
  // (BWGR_WORDS2: the words requested TWO steps before they are looked at -- they are read past the caches, about 2 us, longer than a step: two registers
  // by the step's parity, the step compiled twice, so that a request lands in the register its consumer reads)
  unsigned long long lmid = 0ull, lw1 = 0ull;             // the words requested one step ago (no copy of them is kept for fold_list: a register rotation
                                                          // makes the compiler land the load in a temporary and wait for it at once)
#define S3_PAR(A0_, A1_) (*((!BWGR_WORDS2 || PP == 0) ? &(A0_) : &(A1_)))
  auto fold_prefetch = [&](int bs, auto par_c) {          // bs: the block whose list the NEXT step folds (relative; < 0: none)
    constexpr int PP = decltype(par_c)::value;
    // (straight-line code, every load unconditional: where a loaded value meets another one at the join of a branch the compiler copies registers
    // and waits for the load in front of the copy -- right after its issue)
    const unsigned long long wv = S3_PAR(lmid, lw1);
    const unsigned long long hv = __builtin_amdgcn_readfirstlane((uint32_t)wv) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(wv >> 32)) << 32);
    const int cnt = (int)(uint32_t)hv, nw = min(62, 2 * max(cnt, 0));
    const bool ok = bs >= 0 && s3_epoch_is(hv, A.epoch) && __ballot(lane >= 1 && lane <= nw && !s3_epoch_is(wv, A.epoch)) == 0ull;   // (all its words there)
    const int8_t *col = Xs + (size_t)((a.blk_begin + max(bs, 0)) * m) * R + 64 * wave + lane;
    const uint32_t wlo = (uint32_t)wv, whi = (uint32_t)(wv >> 32);
    const int ne = ok ? nw / 2 : 0;
#define S3_PRE1(U_, XP_, CP_) { \
      const int ee = min(U_, max(ne - 1, 0)); \
      const uint32_t a0 = __builtin_amdgcn_readlane(wlo, 1 + 2 * ee), a1 = __builtin_amdgcn_readlane(whi, 1 + 2 * ee), b0 = __builtin_amdgcn_readlane(wlo, 2 + 2 * ee); \
      CP_ = (U_ < ne) ? (long long)(((unsigned long long)b0 << 32) | (unsigned long long)a0) : 0ll; \
      XP_ = (int)col[(size_t)(ne > 0 ? (int)(a1 & 0xFFu) : 0) * R]; }
    S3_PRE1(0, xp0, cp0) S3_PRE1(1, xp1, cp1) S3_PRE1(2, xp2, cp2) S3_PRE1(3, xp3, cp3)
#undef S3_PRE1
    np = (bs < 0) ? 0 : (ok ? cnt : -1);
  };
  auto fold_pre = [&](int bs) -> int {                    // the fold of block bs's list with what fold_prefetch left
#ifdef BWGR_EXPERIMENTS
    if (w == 0 && tid == 0 && a.stamps && (SDBG & (1 << 21))) { atomicAdd(&a.stamps[203], 1ull); if (np < 0) atomicAdd(&a.stamps[202], 1ull); }   // folds; ... whose list was not there a step ahead
#endif
    if (np < 0) return fold_list(bs, 0ull);               // (reads the words itself)
    e_own -= (long long)xp0 * cp0 + (long long)xp1 * cp1 + (long long)xp2 * cp2 + (long long)xp3 * cp3;
    return (np > NPRE) ? fold_list(bs, 0ull, NPRE) : 1;
  };

  // ---- prologue: tiles 0 .. NTB - 2 requested and waited for.  The counted wait in front of a step's barrier (S3_DMA_BARRIER) proves tile b landed
  // only when at least WN younger loads have been issued, which holds from step NTB - 1 on (per step a dots wave issues PPW pieces and two small
  // loads: younger than tile b at the barrier of step b are (NTB - 1 - b) * PPW prologue pieces less tile b's own, one prologue load and b * (PPW + 2)
  // loop loads -- 7 at step 0, 9 at step 1, 12 from step 2 on for <128, 4>, against WN = 10).  The prologue's __syncthreads() compiles to
  // lgkmcnt(0) + s_barrier and drains no DMA, so the prologue tiles are drained here, once per launch. ----
  if (wvs >= 6) drej_issue(0);
  for (int t = 0; t < NTB - 1; ++t) tile_issue(t);
  if (wvs < NU) {   // (what steps 0 and 1 look at: never folded before step D)
    lmid = ld_agent_raw64(lists_w + (size_t)(a.blk_begin + max(min(1, nb - 1) - D, 0)) * S3_LSTRIDE + lane);
    if (BWGR_WORDS2) lw1 = ld_agent_raw64(lists_w + (size_t)(a.blk_begin + max(min(2, nb - 1) - D, 0)) * S3_LSTRIDE + lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  S3ST_DECL;
  const bool st_u = (w == 0 && tid == 0), st_d = (w == 0 && tid == 64 * NU);

  // Before a step's barrier the tile the step reads must have landed.  Loads (the DMA requests are loads) return in order among themselves,
  // so "at most WN outstanding", with WN = the loads a dots wave issues in NTB - 2 steps (per step: two small requests and PPW pieces), holds only
  // once every load older than those -- the step's tile among them -- is back, whatever the atomics (no order against loads) are doing; and it
  // leaves the younger tiles in flight.  (The update waves issue no pieces: the count is harmless there.)
  constexpr int WN = (NTB - 2) * LPS;
#define S3_DMA_BARRIER() do { if (wvs >= 6) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WN) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
  auto step = [&](int b, auto par_c) -> bool {
    constexpr int PP = decltype(par_c)::value;   // b & 1 (BWGR_WORDS2)
    const int mB = blk_m(b), par = b & 1;
    S3ST(0, st_u || st_d);
    int8_t *tile = tile0 + (size_t)(b % NTB) * tile_b;
    int8_t *edig = edig0 + (size_t)par * 16 * Rp;
    int8_t *ddig = ddig0 + (size_t)par * 16 * S2_DP;
    // A: what the included markers of block b - D changed
    if (b >= D && upd && !(SDBG & 512)) { if (!fold_pre(b - D)) ctl_s[0] = 1u; }
#ifndef BWGR_FOLD_EARLY
#define BWGR_FOLD_EARLY 1
#endif
    if (BWGR_FOLD_EARLY && wvs < NU) {   // (the update waves only: no other wave keeps a compiler-visible load)
      // right behind this step's fold: the words of the list step b + 1 folds were requested a step ago -- look at them, request its columns (they
      // get the whole step to land); then the words for step b + 2
      if (!(SDBG & 512)) fold_prefetch((b + 1 < nb) ? b + 1 - D : -1, par_c);
      S3_PAR(lmid, lw1) = ld_agent_raw64(lists_w + (size_t)(a.blk_begin + max(min(b + 2 + BWGR_WORDS2, nb - 1) - D, 0)) * S3_LSTRIDE + lane);
    }
    S3ST(1, st_u);
    // B: digits of the residual rows and of this block's rejected steps
    if (upd) {
      if ((unsigned long long)(e_own + (1ll << 54)) >> 55) ctl_s[1] = 1u;       // left the 55-bit range
      s3_put_digits7(e_own, edig + 64 * wave + lane, Rp);
    } else if (tid >= SW_THREADS - SW_MAXM) {                                    // the last two waves (never update waves)
      const double qd = (tid - (SW_THREADS - SW_MAXM) < mB) ? rint((double)drej_s[(b % NTB) * SW_MAXM + (tid - (SW_THREADS - SW_MAXM))] * S) : 0.0;   // (unused markers: zero steps; set b - 1 landed behind the last barrier)
      if (!(fabs(qd) < 18014398509481984.0)) ctl_s[1] = 1u;                      // 2^54
      s3_put_digits7((long long)qd, ddig + (tid - (SW_THREADS - SW_MAXM)), S2_DP);
    }
    S3ST(2, st_u);
#if defined(BWGR_STAMPS) && BWGR_STAMPS == 2
    S3ST(0, st_u || st_d);                      // lite: slot 0 = busy (barrier exit .. barrier entry), slot 4 = waiting at the barrier
    S3_DMA_BARRIER();
    S3ST(4, st_u || st_d);
#else
    S3_DMA_BARRIER();
    S3ST(3, st_u || st_d);
#endif
    if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return false; }
    tile_issue(b + NTB - 1);   // (always: the wait counts rely on it; past the end the last tile again)
    // C: tile b+1 (in registers for two iterations) lands in the other buffer, whose last reader was block b-1; the loads of
    // tile b+3 go out into the registers just freed; the list of block b+1-D and the rejected steps of block b+1 are requested
    S3ST(6, st_u || st_d);
    {   // the small requests first (older than the tile loads on the in-order memory counter, so waiting for them does not wait
        // for the tile), every one unconditional: a load under a branch makes the compiler drain the counter in front of it
      if (!BWGR_FOLD_EARLY && wvs < NU) {   // (the update waves only: no other wave keeps a compiler-visible load)
        // the words of the list step b + 1 folds were requested a step ago: look at them, request its columns; then the words for step b + 2
        if (!(SDBG & 512)) fold_prefetch((b + 1 < nb) ? b + 1 - D : -1, par_c);
        S3_PAR(lmid, lw1) = ld_agent_raw64(lists_w + (size_t)(a.blk_begin + max(min(b + 2 + BWGR_WORDS2, nb - 1) - D, 0)) * S3_LSTRIDE + lane);
      }
    }
    S3ST(7, st_u || st_d);
    S3ST(4, st_u || st_d);
    if (upd) {
      // ---- slab update with the rejected steps: out[row][n] = sum_markers x[row][marker] * digit_n(drej[marker]) ----
      // lane (m16, grp): row quad 16 wave + m16 (rows 4 * that + k for accumulator k); k slots (dword u, byte q) of step s0 are the
      // markers s0 + 16 u + 4 grp + q (the interleave keeps the four lane groups on different LDS banks)
      // The A operand is x[row][marker] with the markers as the MFMA's k, and the tile is [marker][row]: gfx950's transposing LDS read
      // (ds_read_b64_tr_b8: per sixteen lanes a block of 8 LDS rows x 16 bytes; lane 2 q + p supplies the address of row q, bytes 8 p .. 8 p + 7;
      // lane i receives byte i of the eight rows) hands lane (m16, grp) row m16 of a 16-row chunk for eight markers at once -- two reads per
      // operand instead of sixteen dword reads and thirty-two byte permutes.  Accumulator k holds rows 64 wave + 16 k + m16 (chunk 4 wave + k of a
      // marker's bytes, at position chunk ^ (marker & 7) in the swizzled tile); the k slots (lane group grp, byte t of the operand) are the markers
      // s0 + 16 grp + t, and the B operand -- the steps' digits -- is read in the same order (one b128 read).  EXEC is all ones here (a wave-uniform
      // branch): the gather crosses lanes.
      s2_v4i acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      const uint32_t tile_lds = tile_la + (uint32_t)((b % NTB) * (int)tile_b);
      const uint32_t q8 = (uint32_t)((lane & 15) >> 1), p8 = (uint32_t)(lane & 1);
      const uint32_t cw = (uint32_t)(4 * wave);
      for (int s0 = 0; s0 < ((SDBG & 256) ? 0 : mB); s0 += 64) {
        const uint32_t rowA = tile_lds + (uint32_t)(s0 + 16 * grp + (int)q8) * (uint32_t)R3 + 8u * p8;   // markers s0 + 16 grp + q8 and + 8 (both have marker & 7 = q8)
        const uint32_t a0 = rowA + ((((cw + 0u) ^ q8) & (uint32_t)(CH - 1)) << 4), a1 = rowA + ((((cw + 1u) ^ q8) & (uint32_t)(CH - 1)) << 4);
        const uint32_t a2 = rowA + ((((cw + 2u) ^ q8) & (uint32_t)(CH - 1)) << 4), a3 = rowA + ((((cw + 3u) ^ q8) & (uint32_t)(CH - 1)) << 4);
        typedef int s3_v2i __attribute__((ext_vector_type(2)));
        s3_v2i x0, x1, x2, x3, y0, y1, y2, y3;
        asm volatile("ds_read_b64_tr_b8 %0, %8\n\tds_read_b64_tr_b8 %4, %8 offset:1024\n\t"
                     "ds_read_b64_tr_b8 %1, %9\n\tds_read_b64_tr_b8 %5, %9 offset:1024\n\t"
                     "ds_read_b64_tr_b8 %2, %10\n\tds_read_b64_tr_b8 %6, %10 offset:1024\n\t"
                     "ds_read_b64_tr_b8 %3, %11\n\tds_read_b64_tr_b8 %7, %11 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        static_assert(R3 == 128, "offset:1024 = eight markers of 128 bytes");
        const s2_v4i bv = *reinterpret_cast<const s2_v4i *>(ddig + (size_t)m16 * S2_DP + s0 + 16 * grp);
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{x0[0], x0[1], y0[0], y0[1]}, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{x1[0], x1[1], y1[0], y1[1]}, bv, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{x2[0], x2[1], y2[0], y2[1]}, bv, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(s2_v4i{x3[0], x3[1], y3[0], y3[1]}, bv, acc3, 0, 0, 0);
      }
      int *ou = outu + (size_t)wave * 64 * S3_OS;
      if (m16 < 8) {      // lane: digit n = m16; acc_k[reg] belongs to local row 16 k + 4 grp + reg
        int *op = ou + (size_t)(4 * grp) * S3_OS + m16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          op[(reg + 0) * S3_OS] = acc0[reg]; op[(reg + 16) * S3_OS] = acc1[reg];
          op[(reg + 32) * S3_OS] = acc2[reg]; op[(reg + 48) * S3_OS] = acc3[reg];
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
      S3ST(5, st_u);
    } else {
      // ---- slab dots of block b against the digits of e: markers in groups of 16, groups gm and gm + ND together on wave
      // NU + gm (the two groups' MFMAs, LDS round trips and atomics overlap) ----
      for (int gm = wave - NU; wvs < NU + NDOT && 16 * gm < m; gm += 2 * NDOT) {
        const int gm2 = gm + NDOT;
        const bool two = 16 * gm2 < m;
        const int8_t *bp = edig + (size_t)m16 * Rp + 16 * grp;
        const int8_t *ap = tile + (size_t)(16 * gm + m16) * R3;           // (both groups' markers have jj & 15 = m16)
        const int8_t *ap2 = tile + (size_t)(16 * (two ? gm2 : gm) + m16) * R3;
        s2_v4i acc = {0, 0, 0, 0}, acc2 = acc;
        for (int r = 0; r < ((SDBG & 256) ? 0 : R3); r += 64) {
          const s2_v4i bv = *reinterpret_cast<const s2_v4i *>(bp + r);
          const int so = (((grp + (r >> 4)) ^ (m16 & (CH - 1))) << 4);   // rows r + 16 grp .. + 15 = chunk grp + r / 16
          // (digits as the A operand, genotypes as B: the output tile is [digit][marker], so that lane (marker m16, group grp) holds the marker's digits
          // 4 grp .. 4 grp + 3 and the two words are formed in registers -- the other order scattered a marker's digits over seven lanes and went through LDS)
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(bv, *reinterpret_cast<const s2_v4i *>(ap + so), acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bv, *reinterpret_cast<const s2_v4i *>(ap2 + so), acc2, 0, 0, 0);
        }
        if (grp < 2) {   // word grp of the marker: digits 0-3 (weight 1) or 4-6 (weight 2^32, the sequencer's), with the arrival count in the low byte
          const long long w1 = (long long)acc[0] + ((long long)acc[1] << 8) + ((long long)acc[2] << 16) + ((long long)acc[3] << 24);
          const long long w2 = (long long)acc2[0] + ((long long)acc2[1] << 8) + ((long long)acc2[2] << 16) + ((long long)acc2[3] << 24);
          unsigned long long *qs = A.qsum + ((size_t)(a.blk_begin + b) * SW_MAXM + 16 * gm + m16) * 2 + grp;
          if (!(SDBG & 8)) {
            __hip_atomic_fetch_add((gu64_t *)qs, (unsigned long long)((w1 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (two) __hip_atomic_fetch_add((gu64_t *)(qs + (size_t)(16 * (gm2 - gm)) * 2), (unsigned long long)((w2 << 8) + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      S3ST(5, st_d);
    }
    return true;
  };
  for (int b = 0; b < nb; b += 2) {
    if (!step(b, std::integral_constant<int, 0>{})) return;
    if (b + 1 < nb && !step(b + 1, std::integral_constant<int, 1>{})) return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the requests past the end)
#undef S3_DMA_BARRIER
#undef S3_PAR
  S3ST_FLUSH(0, st_u); S3ST_FLUSH(8, st_d);
  // the lists of the last D blocks
  if (upd && !(SDBG & 512)) for (int bs = max(0, nb - D); bs < nb; ++bs) {
    if (!fold_list(bs, 0ull)) { ctl_s[0] = 1u; break; }
  }
  if (upd && ((unsigned long long)(e_own + (1ll << 54)) >> 55)) ctl_s[1] = 1u;
  __syncthreads();
  if (ctl_s[0]) { if (tid == 0) a.sc->error = 1u; return; }
  if (ctl_s[1] && tid == 0) a.sc->error = 2u;
  if (upd) a.e[row0 + 64 * wave + lane] = (double)e_own * invS;
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer
// ------------------------------------------------------------------------------------------------------------------
template <typename GT, bool CEN = false>
__device__ __forceinline__ void s3_sequencer(const Sweep3Args &A) {
  // the experiment switches (BWGR_DBG3) cost instructions and branches inside the rounds: compiled in only with -DBWGR_EXPERIMENTS (tools/ab3_probe.py builds that library)
#ifdef BWGR_EXPERIMENTS
  const int SDBG = A.dbg;
#else
  constexpr int SDBG = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = a.m, nb = a.blk_end - a.blk_begin, D = A.D, pstride = a.pstride;
  size_t off = 0;
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 3 * sizeof(StageBuf);                        // [block % 3]
  double *spec_s = reinterpret_cast<double *>(smem + off); off += (size_t)3 * 2 * SW_MAXM * sizeof(double);   // [block % 3][spec | gjj][marker]
  double *q_s = reinterpret_cast<double *>(smem + off); off += (size_t)2 * SW_MAXM * sizeof(double);
  double *far_s = reinterpret_cast<double *>(smem + off); off += (size_t)2 * S3_NFW * SW_MAXM * sizeof(double);    // [parity][far wave][marker]
  float *state_s = reinterpret_cast<float *>(smem + off); off += (size_t)2 * 3 * SW_MAXM * sizeof(float);    // [parity][b | d][marker]
  double *accC = reinterpret_cast<double *>(smem + off); off += (size_t)S3_RING * sizeof(double);         // included markers of the last D blocks: what marker k changed beyond drej
  float2 *accS = reinterpret_cast<float2 *>(smem + off); off += (size_t)S3_RING * sizeof(float2);          // ... as the two float steps {included, rejected}
  int *accK = reinterpret_cast<int *>(smem + off); off += (size_t)S3_RING * sizeof(int);                    // k | source block << 8
  unsigned char *rowf_s = smem + off; off += (size_t)2 * S3_NFW * S3_NFL * SW_MAXM * sizeof(GT);            // far-field rows landing by LDS-DMA [parity][wave][row]
  constexpr bool G16 = (sizeof(GT) == 2);
#ifndef BWGR_GPD3
#define BWGR_GPD3 1
#endif
  constexpr bool GPD = G16 && (BWGR_GPD3 != 0);   // the packed diagonal block staged through LDS (else: an included marker's row is read from global memory inside the chain)
  // 16-bit panels: the packed diagonal block of the block in flight and of the next one (LDS-DMA by the staging waves, a phase
  // ahead), so that an included marker's row is an LDS read inside the chain instead of an HBM miss; and the distance-1 / 2 rows
  // of the block's included markers, requested by DMA when the marker is included and consumed after the block's last round
  unsigned char *gpd_s = smem + off; if (G16) off += (size_t)3 * S3_GPD_BYTES;
  unsigned char *rowx_s = smem + off; if (G16) off += (size_t)S3_NRX * S3_ROWSLOT;
  unsigned char *qraw_s = smem + off; off += (size_t)3 * S3_QRAW_BYTES;   // [block % 3]: the block's slab-dot words, landed by LDS-DMA two phases before they are adopted
  unsigned char *touch_s = smem + off; off += 256;     // where wave 0's far-field touches land (never read)
  int *ctrl_s = reinterpret_cast<int *>(smem + off);   // [0] ok flag
  int *pos_s = ctrl_s + 8;                             // [b & 31]: ring position where block b's entries begin
  const unsigned char **gx_s = reinterpret_cast<const unsigned char **>(ctrl_s + 40);   // the cross Gram arrays' base addresses (a table in LDS: indexed per entry)
  S3ST_DECL;
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
  // Waves 2-3, staging.  Everything block c needs that does not depend on the chain -- its constants (StageBuf, 6 KB), speculative terms and Gram
  // diagonal (2 KB) and, 16-bit panels, its packed diagonal Gram block (16 KB) -- lands by LDS-DMA straight in rings of three blocks: requested while
  // wave 0 runs block c - 2, waited for (a counted wait, behind the requests for block c + 1) before the barrier that ends block c - 1.  No registers in
  // flight, no commit phase, two phases for a request to land.  (Round 3 moved the constants through registers, requested in one phase and stored in
  // the next: one phase to land, and every such round trip that is longer than the period IS the period -- profiles/NOTES.md, round 4.)
  // Each wave issues S3_STG_PIECES pieces per block, whatever the block size (sources clamped inside the block's data): the wait count relies on it.
  constexpr int S3_STG_PIECES = 4 + (GPD ? 8 : 0);
  const uint32_t stage_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)stage);
  const uint32_t spec_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)spec_s);
  static_assert(sizeof(StageBuf) == 6 * 1024, "six 1 KiB pieces of constants");
  auto stage_issue = [&](int c, int part = -1) {   // part 0: the constants and speculative terms, 1 / 2: the packed block's first / second half; -1: all
    const int blk = a.blk_begin + min(c, nb - 1), slot = c % 3, wv2 = __builtin_amdgcn_readfirstlane(wave) - 2;
    const unsigned char *stb = reinterpret_cast<const unsigned char *>(a.ps.blocks + blk);
    const unsigned char *spb = reinterpret_cast<const unsigned char *>(a.ps.spec + blk);
    const uint32_t lo16 = (uint32_t)lane * 16u;
    // pieces 0-5: the constants; 6: spec; 7: the Gram diagonal (SpecBuf: spec at 0, gjj at 2 KB) -- even pieces on wave 2, odd ones on wave 3
    if (part <= 0) {
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int pc = wv2 + 2 * u; s3_dma16s(stb, (uint32_t)pc * 1024u + lo16, stage_la + (uint32_t)slot * (uint32_t)sizeof(StageBuf) + (uint32_t)pc * 1024u); }
    s3_dma16s(spb, (uint32_t)wv2 * 2048u + lo16, spec_la + (uint32_t)slot * 2048u + (uint32_t)wv2 * 1024u);
    }
    if constexpr (GPD) {
      const int gpbytes = pstride * 2;   // (pstride is a multiple of 8 entries: whole 16-byte chunks)
      const unsigned char *src = reinterpret_cast<const unsigned char *>(gp_all + (size_t)blk * pstride);
      const uint32_t gla = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)gpd_s) + (uint32_t)slot * (uint32_t)S3_GPD_BYTES;
      if (part < 0 || part == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int pc = wv2 + 2 * u; s3_dma16s(src, (uint32_t)min(pc * 1024 + lane * 16, gpbytes - 16), gla + (uint32_t)pc * 1024u); }
      }
      if (part < 0 || part == 2) {
#pragma unroll
      for (int u = 4; u < 8; ++u) { const int pc = wv2 + 2 * u; s3_dma16s(src, (uint32_t)min(pc * 1024 + lane * 16, gpbytes - 16), gla + (uint32_t)pc * 1024u); }
      }
    }
  };
#define S3_STG_WAIT() asm volatile("s_waitcnt vmcnt(%0)" : : "n"(S3_STG_PIECES) : "memory")   /* all but the block just requested have landed */
  // wave 1: the K3 slab dots of block c, summed by the streamers' atomics; two markers per lane.  The four words a lane needs are
  // requested one phase ahead (pq_*): the streamers run blocks ahead of the sequencer, so in steady state the words are complete
  // when they are first looked at and no memory round trip sits in the block period.
  // The 2 KB of a block's words are requested by two LDS-DMA pieces (sc1: past this CU's L1) TWO phases before the block is adopted: one phase
  // ahead -- round 3, through registers -- the period could not drop below that round trip (the words live at the memory side: the streamers'
  // atomics are performed there), and every helper role's request -> consume chain was one phase long (profiles/NOTES.md, round 4).  The counted
  // wait leaves the younger block's two pieces in flight; an incomplete sum (the streamers are D blocks ahead: rare) falls back to polling.
  const uint32_t qraw_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)qraw_s);
  const double qweight = A.qsplit ? 4294967296.0 : 16777216.0;   // the second word's weight: digits 4-6 (k_sweep3's streamers) or 3-6 (k_sweep3p's)
  auto poll_request = [&](int c) {   // (clamped block: the requests past the end re-read the last block)
    const unsigned char *g = reinterpret_cast<const unsigned char *>(A.qsum + (size_t)(a.blk_begin + min(c, nb - 1)) * SW_MAXM * 2);
    const uint32_t la = qraw_la + (uint32_t)((c % 3) * S3_QRAW_BYTES);
    s3_dma16s_sc1(g, (uint32_t)lane * 16u, la);
    s3_dma16s_sc1(g, 1024u + (uint32_t)lane * 16u, la + 1024u);
  };
  auto poll_q = [&](int c) -> int {
    const int mBc = blk_m(c);
    const unsigned long long *g = A.qsum + (size_t)(a.blk_begin + c) * SW_MAXM * 2;
    const unsigned long long need = (unsigned long long)A.K3;
    const bool n0 = lane < mBc, n1 = 64 + lane < mBc;
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // block c's two pieces (two phases old) have landed; block c + 1's may still be in flight
    const ulonglong2 v0 = reinterpret_cast<const ulonglong2 *>(qraw_s + (size_t)(c % 3) * S3_QRAW_BYTES)[lane];
    const ulonglong2 v1 = reinterpret_cast<const ulonglong2 *>(qraw_s + (size_t)(c % 3) * S3_QRAW_BYTES)[64 + lane];
    unsigned long long l0 = n0 ? v0.x : need, h0 = n0 ? v0.y : need, l1 = n1 ? v1.x : need, h1 = n1 ? v1.y : need;
    if (__builtin_expect(!(SDBG & 8) && __ballot(!((l0 & 0xFFull) == need && (h0 & 0xFFull) == need && (l1 & 0xFFull) == need && (h1 & 0xFFull) == need)) != 0ull, 0)) {   // (experiment 8, the streamers publish nothing: taken as complete -- the fall-back's clock read alone cost 0.2 us a block in those timings)
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
#ifdef BWGR_EXPERIMENTS
      if (lane == 0 && a.stamps && (SDBG & (1 << 21))) atomicAdd(&a.stamps[200], 1ull);   // (BWGR_DBG3 bit 21) blocks whose slab-dot words were not complete when they landed
#endif
      for (;;) {
        if ((l0 & 0xFFull) == need && (h0 & 0xFFull) == need && (l1 & 0xFFull) == need && (h1 & 0xFFull) == need) break;
        if (SDBG & 8) break;   // (timing experiment only: the streamers publish nothing)
#ifdef BWGR_EXPERIMENTS
        if (lane == 0 && a.stamps && (SDBG & (1 << 21))) atomicAdd(&a.stamps[201], 1ull);   // ... and the polls that took
#endif
        if ((++spins & 63u) == 0u) {
          if (ld_agent_u32(abortw) != 0u) return 0;
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
        }
        __builtin_amdgcn_s_sleep(1);
        if (n0) { l0 = ld_agent_raw64(g + 2 * lane); h0 = ld_agent_raw64(g + 2 * lane + 1); }
        if (n1) { l1 = ld_agent_raw64(g + 2 * (64 + lane)); h1 = ld_agent_raw64(g + 2 * (64 + lane) + 1); }
      }
    }
    double *qd = q_s + (size_t)(c & 1) * SW_MAXM;
    qd[lane] = n0 ? fma((double)((long long)h0 >> 8), qweight, (double)((long long)l0 >> 8)) * invS : 0.0;
    qd[64 + lane] = n1 ? fma((double)((long long)h1 >> 8), qweight, (double)((long long)l1 >> 8)) * invS : 0.0;
    poll_request(c + 2);
    return 1;
  };
  // The included markers of all blocks live in one flat ring of D * SW_MAXM entries {k | source block << 8, corr, corr on the grid}
  // (the entries of the last D blocks can never exceed it); pos_s[b & 31] is where block b's entries begin.
  constexpr int ring = S3_RING;
  static_assert((S3_RING & (S3_RING - 1)) == 0, "ring positions wrap with a mask");
  // Waves 4-6, far field: sum over the included markers k of blocks c-D+1 .. c-3 of G_kj corr_k for destination block c (wave 0
  // itself takes distances 1 and 2 as the markers appear).  Those rows are HBM misses about 2 us away on this CU, so the work is
  // cut in two phases: far_issue(c) -- while wave 0 is two blocks short of c, when the last of those lists is final -- requests
  // up to NFL rows per wave by LDS-DMA (no registers in flight, a dynamic number of rows, nothing else on these waves' memory
  // counter); far_consume(c), one block period later, waits for them, multiplies and leaves the sum in LDS.  Every third entry
  // of the flat list per wave; lane = the row's dword(s) lane (markers 2 lane, 2 lane + 1 for 16-bit entries; lane and 64 + lane
  // for 32-bit ones).
  constexpr int NFL = S3_NFL;
  constexpr int NFW = S3_NFW;                       // far-field shares: waves 5 and 6 (0, 1) and the staging waves 2 and 3 (2, 3), whose phase is mostly waiting;
                                                    // wave 4, on wave 0's SIMD, must stay light (a third of the far field there cost 30 % of the sweep)
  constexpr int ROWB = SW_MAXM * (int)sizeof(GT);   // bytes reserved per row (256 or 512)
  constexpr int NPC = ROWB / 256;
  int f_p0 = 0, f_cnt = 0, f_n = 0;
  const uint32_t rowf_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)rowf_s);
  const int rowbytes = m * (int)sizeof(GT);
  auto far_src = [&](int c, int sl) -> const unsigned char * {
    const int kb = accK[sl];                         // k | (source block, relative) << 8
    const int d = c - (kb >> 8);
    return gx_s[d - 1] + ((size_t)(a.blk_begin + c) * m * m + (size_t)(kb & 0xFF) * m) * sizeof(GT);
  };
  auto far_add = [&](const uint32_t *row, double cf, double &s0, double &s1) {   // row: 64 or 128 dwords in LDS or registers' image
    if constexpr (sizeof(GT) == 2) { const uint32_t v = row[lane]; s0 = fma((double)(v & 0xFFFFu), cf, s0); s1 = fma((double)(v >> 16), cf, s1); }
    else { s0 = fma((double)(int)row[lane], cf, s0); s1 = fma((double)(int)row[64 + lane], cf, s1); }
  };
  // (Per entry the work is a handful of instructions: the entries' addresses and coefficients are formed by the lanes in
  // parallel -- lane i = this wave's i-th entry -- and handed to the wave through readlane; a loop that chases accK -> gx_s ->
  // address one entry at a time costs two dependent LDS latencies per entry.)
  unsigned long long f_ptr = 0ull;   // lane i: source row of this wave's i-th entry
  double f_cf = 0.0;                 // lane i: its coefficient
  // Wave 4 makes the far-field rows L2 hits.  While wave 0 runs block c - 1, the markers block c - 2 included are final: their rows against blocks
  // c + 1 .. c - 3 + D -- what the far-field waves request from the end of this phase on, HBM misses of about a block period each, waited for inside
  // those waves' phase -- are touched here, one 128-byte line per lane (32 lanes per marker, two markers per request), landing in an LDS pad nobody
  // reads.  Nothing ever waits for a touch.
#ifndef BWGR_FARTOUCH
#define BWGR_FARTOUCH 1
#endif
  auto far_touch = [&](int c) {
    if (!G16 || !BWGR_FARTOUCH || c < 2 || D < 4 || (SDBG & 32)) return;
    const int s_blk = c - 2;
    const int p0 = pos_s[s_blk & 31], cnt = (pos_s[(s_blk + 1) & 31] - p0) & (ring - 1);
    const int q = lane & 31, d = 3 + (q >> 1);
    for (int e0 = 0; e0 < cnt; e0 += 2) {
      const int e = e0 + (lane >> 5);
      if (e < cnt && d <= D - 1 && s_blk + d < nb && (q & 1) * 128 < rowbytes) {   // (a row of a small block is one line: nothing past the array's end)
        const int k = accK[(p0 + e) & (ring - 1)] & 0xFF;
        s3_dma4(gx_s[d - 1] + ((size_t)(a.blk_begin + s_blk + d) * m * m + (size_t)k * m) * sizeof(GT) + (size_t)(q & 1) * 128, touch_s);
      }
    }
  };
  // far_plan(c): which rows (LDS reads only: the lists' positions, entries, array bases) -- this runs under the wait for the previous request's rows;
  // far_request(c): the requests themselves, after far_consume, at the end of the wave's phase
  // (the plan is three dependent LDS round trips -- the lists' positions, the entries, the array bases -- some 500 cycles of latency; the staging waves
  // take them one at a time between the thirds of their twelve requests, whose issue is a thousand cycles of its own: far_plan_a / _b / _c)
  int fp_a = 0, fp_b = 0, fp_kb = 0; double fp_cf = 0.0; bool fp_on = false;
  auto far_plan_a = [&](int c) {
    fp_on = !(c < 3 || D < 4 || c >= nb || (SDBG & 32));
    fp_a = pos_s[max(c - D + 1, 0) & 31]; fp_b = pos_s[(max(c, 2) - 2) & 31];
  };
  auto far_plan_b = [&](int c, int hw) {
    f_p0 = fp_a;
    f_cnt = fp_on ? ((fp_b - fp_a) & (ring - 1)) : 0;    // [f_p0, +f_cnt): blocks c-D+1 .. c-3
    f_n = (f_cnt > hw) ? min(NFL, (f_cnt - hw + NFW - 1) / NFW) : 0;
    const int sl = (f_p0 + hw + NFW * min(lane, max(f_n - 1, 0))) & (ring - 1);
    fp_kb = accK[sl]; fp_cf = accC[sl];
  };
  auto far_plan_c = [&](int c) {
    const int d = min(max(c - (fp_kb >> 8), 1), S3_MAXD);   // (clamped: without entries the slot holds anything)
    const unsigned char *src = gx_s[d - 1] + ((size_t)(a.blk_begin + c) * m * m + (size_t)(fp_kb & 0xFF) * m) * sizeof(GT);
    f_ptr = (f_n > 0) ? (unsigned long long)src : 0ull;
    f_cf = (lane < f_n) ? fp_cf : 0.0;
  };
  auto far_plan = [&](int c, int hw) { far_plan_a(c); far_plan_b(c, hw); far_plan_c(c); };
  auto far_request = [&](int c, int hw) {
    // (inline-asm requests: the compiler must not see them, or it drains them in front of the next LDS read -- the wait is the caller's, counted)
    const uint32_t dst_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rowf_la + (uint32_t)(((c & 1) * NFW + hw) * NFL) * (uint32_t)ROWB));
    for (int n = 0; n < f_n; ++n) {
      const unsigned char *src = reinterpret_cast<const unsigned char *>(
          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f_ptr, n) |
          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(f_ptr >> 32), n) << 32));
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) s3_dma4s(src, (uint32_t)min(lane * 4 + 256 * pc, rowbytes - 4), dst_la + (uint32_t)(n * ROWB + 256 * pc));
    }
  };
  // (the caller has waited for the rows)
  auto far_consume = [&](int c, int hw, int f_p0, int f_cnt, int f_n, double f_cf) {
    double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
    const unsigned char *src = rowf_s + (size_t)(((c & 1) * NFW + hw) * NFL) * ROWB;
    int n = 0;
    if constexpr (sizeof(GT) == 2) {
      // four rows per trip, their LDS reads issued together (one round trip where the two-row loop had two); the sums keep the two-row loop's order
      // -- even rows into (s0, s1), odd ones into (t0, t1) -- so the chain is the same bit for bit
      for (; n + 4 <= f_n; n += 4) {
        const uint32_t *r_ = reinterpret_cast<const uint32_t *>(src + (size_t)n * ROWB);
        const uint32_t v0 = r_[lane], v1 = r_[ROWB / 4 + lane], v2 = r_[2 * (ROWB / 4) + lane], v3 = r_[3 * (ROWB / 4) + lane];
        const double c0 = readlane_f64(f_cf, n), c1 = readlane_f64(f_cf, n + 1), c2 = readlane_f64(f_cf, n + 2), c3 = readlane_f64(f_cf, n + 3);
        s0 = fma((double)(v0 & 0xFFFFu), c0, s0); s1 = fma((double)(v0 >> 16), c0, s1);
        t0 = fma((double)(v1 & 0xFFFFu), c1, t0); t1 = fma((double)(v1 >> 16), c1, t1);
        s0 = fma((double)(v2 & 0xFFFFu), c2, s0); s1 = fma((double)(v2 >> 16), c2, s1);
        t0 = fma((double)(v3 & 0xFFFFu), c3, t0); t1 = fma((double)(v3 >> 16), c3, t1);
      }
    }
    for (; n + 2 <= f_n; n += 2) {                      // two rows per trip: their LDS reads overlap
      far_add(reinterpret_cast<const uint32_t *>(src + (size_t)n * ROWB), readlane_f64(f_cf, n), s0, s1);
      far_add(reinterpret_cast<const uint32_t *>(src + (size_t)(n + 1) * ROWB), readlane_f64(f_cf, n + 1), t0, t1);
    }
    if (n < f_n) far_add(reinterpret_cast<const uint32_t *>(src + (size_t)n * ROWB), readlane_f64(f_cf, n), s0, s1);
    s0 += t0; s1 += t1;
    for (int i = hw + NFW * NFL; i < f_cnt; i += NFW) {     // more included markers than rows in flight (dense blocks): the rest in place
      const int sl = (f_p0 + i) & (ring - 1);
      const unsigned char *g = far_src(c, sl);
      uint32_t v0 = *reinterpret_cast<const uint32_t *>(g + min(lane * 4, rowbytes - 4)), v1 = 0u;
      if constexpr (sizeof(GT) == 4) v1 = *reinterpret_cast<const uint32_t *>(g + min(lane * 4 + 256, rowbytes - 4));
      const double cf = accC[sl];
      if constexpr (sizeof(GT) == 2) { s0 = fma((double)(v0 & 0xFFFFu), cf, s0); s1 = fma((double)(v0 >> 16), cf, s1); }
      else { s0 = fma((double)(int)v0, cf, s0); s1 = fma((double)(int)v1, cf, s1); }
    }
    double *fd = far_s + ((size_t)(c & 1) * NFW + hw) * SW_MAXM;
    if constexpr (sizeof(GT) == 2) { if (2 * lane < m) { fd[2 * lane] = s0; fd[2 * lane + 1] = s1; } }
    else { fd[lane] = s0; fd[64 + lane] = s1; }
  };
  // wave 7: the Gram rows wave 0 may ask for PF blocks from now -- the packed diagonal block and the distance-1 cross block,
  // 48 KB -- are touched (one dword per 128-byte line) so that they sit in this XCD's L2 when an included marker needs its row
  constexpr int PF = 8;
  uint32_t pf0 = 0u, pf1 = 0u, pf2 = 0u, pf3 = 0u, pf4 = 0u, pf5 = 0u;
  auto touch = [&](int c) {
    if (c >= nb || !(SDBG & 2)) return;   // (off by default: measured slower at C4; BWGR_DBG3=2 turns the touches on)
    const int blk = a.blk_begin + c;
    const unsigned char *gpb = reinterpret_cast<const unsigned char *>(gp_all + (size_t)blk * pstride);
    const size_t gpbytes = (size_t)pstride * sizeof(GT), gxbytes = (size_t)m * m * sizeof(GT);
    const unsigned char *gxb = (D >= 2 && c >= 1) ? reinterpret_cast<const unsigned char *>(reinterpret_cast<const GT *>(A.gx[0]) + (size_t)blk * m * m) : gpb;
    const size_t gxl = (D >= 2 && c >= 1) ? gxbytes : gpbytes;
    // (the values are consumed one phase later, so the loads have a whole period to land)
    asm volatile("" :: "v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3), "v"(pf4), "v"(pf5));
    const size_t o = (size_t)lane * 128;
    pf0 = *reinterpret_cast<const uint32_t *>(gpb + min(o, gpbytes - 4));
    pf1 = *reinterpret_cast<const uint32_t *>(gpb + min(o + 8192, gpbytes - 4));
    pf2 = *reinterpret_cast<const uint32_t *>(gxb + min(o, gxl - 4));
    pf3 = *reinterpret_cast<const uint32_t *>(gxb + min(o + 8192, gxl - 4));
    pf4 = *reinterpret_cast<const uint32_t *>(gxb + min(o + 16384, gxl - 4));
    pf5 = *reinterpret_cast<const uint32_t *>(gxb + min(o + 24576, gxl - 4));
  };
  // wave 7: everything about a finished block that is off the chain -- its marker state (with the per-marker variance draw),
  // the posterior sums, and its list for the streamers (the fixed-point corrections are formed here, one entry per lane)
  double sum_d = 0.0, sum_b2 = 0.0;
  double chn0 = 1.0, chn1 = 1.0;   // the chi-square variates of the block finish_block handles next, requested one phase ahead
  auto chi_request = [&](int c) {   // (unconditional, clamped; a ragged block's unused entries are staged as 1)
    const StageBuf &sb = a.ps.blocks[a.blk_begin + max(0, min(c, nb - 1))];
    chn0 = sb.chi[lane]; chn1 = sb.chi[64 + lane];
  };
  auto finish_block = [&](int c) {
    const int blk = a.blk_begin + c, j0c = blk * m, mBc = blk_m(c);
    const float *sp = state_s + (size_t)(c & 1) * 2 * SW_MAXM;
    const bool vbv = (a.flags & SWF_VB_VEC) != 0 && !A.skip_vb;   // (skip_vb: k_vb_fill forms the variances after the sweep -- two fp64 divisions a lane less on this wave)
    const double ch0 = chn0, ch1 = chn1;
    if (vbv) chi_request(c + 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = lane + 64 * h;
      if (t < mBc) {
        const float bn = sp[t], dn = sp[SW_MAXM + t];
        a.b[j0c + t] = bn;
        a.d[j0c + t] = dn;
        if (vbv) a.vb[j0c + t] = (float)((double)(Sb + bn * bn) / (h ? ch1 : ch0));
        sum_d += (double)dn;
        sum_b2 = fma((double)bn, (double)bn, sum_b2);
      }
    }
    const int p0 = pos_s[c & 31], cnt = (pos_s[(c + 1) & 31] - p0) & (ring - 1);
    unsigned long long *L = A.lists + (size_t)blk * S3_LSTRIDE;
    for (int w0 = 0; w0 < 1 + 2 * cnt; w0 += 64) {
      const int wi = w0 + lane;
      if (wi < 1 + 2 * cnt) {
        unsigned long long v;
        if (wi == 0) v = s3_hdr(A.epoch, cnt);
        else {
          const int sl = (p0 + ((wi - 1) >> 1)) & (ring - 1);
          const float2 st2 = accS[sl];                                    // {included step, rejected step}
          const long long cq = (long long)rint((double)st2.x * S) - (long long)rint((double)st2.y * S);   // what the streamers fold in
          v = ((wi - 1) & 1) ? (((unsigned long long)A.epoch << 40) | (0xEEull << 32) | ((unsigned long long)cq >> 32))
                             : (((unsigned long long)A.epoch << 40) | ((unsigned long long)(uint32_t)(accK[sl] & 0xFF) << 32) | ((unsigned long long)cq & 0xFFFFFFFFull));
        }
        st_agent_raw64(L + wi, v);
      }
    }
  };
#ifndef BWGR_ROLEBAR
#define BWGR_ROLEBAR 1
#endif
  // BWGR_ROLEBAR: every role reaches the block barrier inside its own branch.  The roles are exec-masked branches of one function, and
  // where they join the compiler copies role-specific registers and puts an s_waitcnt vmcnt(0) in front of the copies -- which every
  // wave executes: with the barrier behind the join, a staging wave waited out the round trip of the requests it had just issued,
  // every block, before it reached the barrier.  With the barrier in front of the join that wait falls into the next phase, where the
  // wave waits for those requests anyway.
#define S3_ROLE_BARRIER() do { if (BWGR_ROLEBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
  const int wvu = __builtin_amdgcn_readfirstlane(wave);   // the role as a scalar: real branches, every wave runs its own role's code (and barrier) only
  auto helper_phase = [&](int c) {   // block c >= 1, while wave 0 runs block c-1
    if (wvu == 1) { if (!(SDBG & 32768)) { if (!poll_q(c)) ctrl_s[0] = 0; } S3ST(2, tid == 64); S3_ROLE_BARRIER(); }
    else if (wvu <= 3) { if (!(SDBG & 8192)) {
      // the requests for block c + 1 (always: past the end the last block again, the wait counts rely on it), then the wait for block c's, which were
      // issued a phase ago and are used by wave 0 after this phase's barrier
      const int o_p0 = f_p0, o_cnt = f_cnt, o_n = f_n; const double o_cf = f_cf;
      far_plan_a(c + 1);             // (far-field shares 2 and 3: the plan's LDS round trips under the requests' issue)
      stage_issue(c + 1, 0);
      far_plan_b(c + 1, wave);
      stage_issue(c + 1, 1);
      far_plan_c(c + 1);
      stage_issue(c + 1, 2);
      S3_STG_WAIT();                 // ... whose rows, requested at the end of the last phase, are older than the pieces just requested: landed too
      far_consume(c, wave, o_p0, o_cnt, o_n, o_cf);
      far_request(c + 1, wave);
    } S3ST(2, tid == 128 || tid == 192); S3_ROLE_BARRIER(); }
    else if (wvu == 4) { far_touch(c); S3_ROLE_BARRIER(); }   // (the fourth SIMD's other wave is wave 0, whose dependent chain wants the issue slots: a handful of instructions only)
    else if (wvu <= 6) {
      const int o_p0 = f_p0, o_cnt = f_cnt, o_n = f_n; const double o_cf = f_cf;
      far_plan(c + 1, wave - 5);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      S3ST(5, tid == 320);
      far_consume(c, wave - 5, o_p0, o_cnt, o_n, o_cf);
      S3ST(6, tid == 320);
      far_request(c + 1, wave - 5);
      S3ST(7, tid == 320);
      S3ST(2, tid == 384);
      S3_ROLE_BARRIER();
    }
    else { if (c >= 2 && !(SDBG & 4096)) finish_block(c - 2); touch(c + PF); S3ST(2, tid == 448); S3_ROLE_BARRIER(); }
  };

  // ---- prologue: block 0 ----
  if (tid < 40) ctrl_s[tid] = (tid == 0) ? 1 : 0;   // (pos_s: block 0's entries begin at ring position 0)
  if (tid >= 64 && tid < 64 + S3_MAXD) gx_s[tid - 64] = reinterpret_cast<const unsigned char *>(A.gx[tid - 64]);
  __syncthreads();
  if (wave == 1) { poll_request(0); poll_request(1); if (!poll_q(0)) ctrl_s[0] = 0; }
  else if (wave == 2 || wave == 3) { stage_issue(0); stage_issue(1); S3_STG_WAIT(); far_consume(0, wave, 0, 0, 0, 0.0); far_plan(1, wave); far_request(1, wave); }
  else if (wave >= 5 && wave <= 6) { far_consume(0, wave - 5, 0, 0, 0, 0.0); far_plan(1, wave - 5); far_request(1, wave - 5); }
  else if (wave == 7) { for (int c = 0; c < PF; ++c) touch(c); if ((a.flags & SWF_VB_VEC) && !A.skip_vb) chi_request(0); }
  __syncthreads();
  if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }

  const bool sq0 = (tid == 0), sq1 = (tid == 64), sq2 = (tid == 128), sq4 = (tid == 320), sq3 = (tid == 192), sq6 = (tid == 384), sq7 = (tid == 448);
  // wave 0's included-marker path, counted in instructions: the packed row's byte offset from a lane-indexed table by one v_readlane
  // (lane = the marker within its group; entry j of packed row k sits at prow(k) + j - k - 1), the lane's dword of a cross row, and the
  // LDS addresses of the packed blocks and of the row slots as scalars
  const int tabp0 = 2 * (lane * (m - 1) - lane * (lane - 1) / 2 - lane - 1);
  const int tabp1 = 2 * ((64 + lane) * (m - 1) - (64 + lane) * (63 + lane) / 2 - (64 + lane) - 1);
  const uint32_t rolane = (uint32_t)min(lane * 4, rowbytes - 4);
  const uint32_t rolane16 = (uint32_t)min(lane * 16, 2 * rowbytes - 16);   // (gx12: a marker's two rows are 2 * rowbytes contiguous bytes, a multiple of 16)
  constexpr bool near12 = G16;   // (16-bit panels always carry gx12; block sizes are multiples of 16)
  const unsigned char *gpd_lane = gpd_s + 2 * lane;   // this lane's entry of a packed row, before the row's and the buffer's offsets
  int gpd_off = 0;                                     // (b % 3) * S3_GPD_BYTES, stepped once per block
  const uint32_t rowx_la = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)rowx_s);
  const bool altb2 = (a.flags & SWF_ALT_B2) != 0;
  if (wave == 0 && !(SDBG & 1)) __builtin_amdgcn_s_setprio(3);   // the chain's wave goes first wherever it shares an issue port
  // wave 0: what the included markers of block b change in blocks b+1 and b+2 (distances 1 and 2), accumulated as they appear
  double rnext0 = 0.0, rnext1 = 0.0, rnxt20 = 0.0, rnxt21 = 0.0;
  int pend_n = 0, pend_p = 0;              // row slots requested by the block before and not applied yet; ring position of their first entry
  bool pend_u1 = false, pend_u2 = false;   // ... and whether that block had a block at distance 1 / 2
  // implicitly centred columns (CEN): wave 0 carries U = -(E_0 + cpre[first block]) / n + sum over the included markers so far of (s_k / n) corr_k;
  // a lane's centred dot is its raw one plus s_j * U (the rejected steps' share sits in the staged spec), an included marker's row G_kj becomes
  // G_kj - s_j s_k / n for the later lanes of its own block, and every later block sees it through U
  const double ninv = a.ninv;
  const double cen_u0 = CEN ? a.sc->cen_u0 : 0.0;
  double cenU = cen_u0;
  const GT *gx0_w0 = reinterpret_cast<const GT *>(A.gx[0]) + (size_t)a.blk_begin * m * m, *gx1_w0 = reinterpret_cast<const GT *>(A.gx[1]) + (size_t)a.blk_begin * m * m;
  const unsigned char *g12_w0 = A.gx12 + (size_t)a.blk_begin * m * 2 * rowbytes;
  for (int b = 0; b < nb; ++b) {
    const int mB = blk_m(b), blk = a.blk_begin + b;
    const bool have_next = (b + 1 < nb);
    S3ST(0, sq0 || sq1 || sq2 || sq4 || sq3 || sq6 || sq7);
    if (wvu == 0 && !(SDBG & 16384)) {
      // Wave 0 is one long dependent chain, so everything here is counted in instructions.  Dead lanes of a ragged last block need
      // no masks: k_prestage fills their constants so that they reject for certain, and their q, spec and far terms are zero.
      const StageBuf &st = stage[b % 3];
      const double *sps = spec_s + (size_t)(b % 3) * 2 * SW_MAXM;
      const double *qd = q_s + (size_t)(b & 1) * SW_MAXM;
      const double *fd = far_s + (size_t)(b & 1) * S3_NFW * SW_MAXM;
      const int pos0 = pos_s[b & 31];
      const GT *gp = gp_all + (size_t)blk * pstride;
      const bool use1 = have_next && D >= 2;   // (D = 1: the streamers fold block b's list in before the dots of block b+1)
      // (bases hoisted out of the block loop: no kernel-argument reload, a 32 x 32 bit product; 16-bit panels touch g1 / g2 only under
      // use1 / use2, the 32-bit path reads "a harmless in-bounds" entry otherwise and wants the fallback)
      const GT *g1 = (G16 || use1) ? gx0_w0 + (size_t)(b + 1) * (uint32_t)(m * m) : gp;
      const bool use2 = (b + 2 < nb) && D >= 3;
      const GT *g2 = (G16 || use2) ? gx1_w0 + (size_t)(b + 2) * (uint32_t)(m * m) : gp;
      const unsigned char *g12b = g12_w0 + (size_t)b * (uint32_t)(m * 2 * rowbytes);
      const int l0 = lane, l1 = 64 + lane;
      const int l1c = min(l1, m - 1), l0c = min(l0, m - 1);
      // constants of this lane's two markers
      const float b0a = st.b0[l0], b0b = st.b0[l1], b2a = st.b2[l0], b2b = st.b2[l1], dra = st.drej[l0], drb = st.drej[l1];
      const float taa = st.tacc[l0], tab = st.tacc[l1], tra = st.trej[l0], trb = st.trej[l1];
      const double xba = (double)st.xxb0[l0], xbb = (double)st.xxb0[l1];
      const double rda = st.rden[l0], rdb = st.rden[l1], sza = st.sdz1[l0], szb = st.sdz1[l1];
      double gja = sps[SW_MAXM + l0], gjb = sps[SW_MAXM + l1];
      double r0 = (qd[l0] - sps[l0]) - ((fd[l0] + fd[SW_MAXM + l0]) + (fd[2 * SW_MAXM + l0] + fd[3 * SW_MAXM + l0]));
      double r1 = (qd[l1] - sps[l1]) - ((fd[l1] + fd[SW_MAXM + l1]) + (fd[2 * SW_MAXM + l1] + fd[3 * SW_MAXM + l1]));
      // The distance-1 / 2 rows of the PREVIOUS block's included markers (requested by DMA as the markers appeared) are waited for here, at their
      // point of use: their round trip runs under the block's outputs, the barrier and this block's constants instead of ending the last round.
      if constexpr (G16) {
        if (pend_n > 0) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          for (int i_ = 0; i_ < pend_n; ++i_) {
            const double cf_ = accC[(pend_p + i_) & (ring - 1)];
            const uint16_t *rw1_ = reinterpret_cast<const uint16_t *>(rowx_s + (size_t)i_ * S3_ROWSLOT);
            const uint16_t *rw2_ = rw1_ + m;
            const uint32_t x1a_ = rw1_[l0c], x1b_ = rw1_[l1c], x2a_ = rw2_[l0c], x2b_ = rw2_[l1c];
            const double c1_ = pend_u1 ? cf_ : 0.0, c2_ = pend_u2 ? cf_ : 0.0;
            rnext0 = fma(-(double)x1a_, c1_, rnext0); rnext1 = fma(-(double)x1b_, c1_, rnext1);
            rnxt20 = fma(-(double)x2a_, c2_, rnxt20); rnxt21 = fma(-(double)x2b_, c2_, rnxt21);
          }
          pend_n = 0;
        }
      }
      r0 += rnext0; r1 += rnext1;
      int csa = 0, csb = 0;
      double sA = 0.0, sB = 0.0;
      if constexpr (CEN) {   // the Gram-diagonal slot holds {G_jj, s_j} as two 32-bit integers (k_spec3)
        csa = __double2hiint(gja); csb = __double2hiint(gjb);
        sA = (double)csa; sB = (double)csb;
        gja = (double)(uint32_t)__double2loint(gja) - sA * sA * ninv; gjb = (double)(uint32_t)__double2loint(gjb) - sB * sB * ninv;
        r0 = fma(sA, cenU, r0); r1 = fma(sB, cenU, r1);
      }
      const double D2a = (double)((altb2 ? b2a : 0.0f) - b0a), D2b = (double)((altb2 ? b2b : 0.0f) - b0b);   // the alternative's step
      const double D2sa = D2a * D2a, D2sb = D2b * D2b;
      rnext0 = rnxt20; rnext1 = rnxt21; rnxt20 = 0.0; rnxt21 = 0.0;
      S3ST(1, sq0);
      unsigned long long am0 = 0ull, am1 = 0ull;
      int nacc = 0, nslot = 0, napp = 0;   // included markers of this block; row slots in use; entries whose rows are applied already
      // One evaluation: the in-model draw b1 given r and whether the Bernoulli step includes the marker (the algebra of lane_b1 /
      // lane_accept with the r-independent factors hoisted: 11 dependent operations).
#define S3_EVAL(R_, XB_, RD_, SZ_, B0_, D2_, D2S_, GJ_, TA_, TR_, MKOFF_, D1F_, ACC_) { \
        const float b1_ = (float)fma((R_) + (XB_), (RD_), (SZ_)); \
        D1F_ = b1_ - (B0_); \
        const double D1_ = (double)D1F_; \
        const double diffd_ = fma((GJ_), (D2S_) - D1_ * D1_, (2.0 * (R_)) * (D1_ - (D2_)));   /* |e2|^2 - |e1|^2 */ \
        const float x_ = Cc * (float)diffd_; \
        /* the decisions as scalar masks (two compares straight into SGPR pairs; a per-lane bool made the compiler shuttle it through \
           v_cndmask / v_cmp and EXEC-masked regions every round) */ \
        const unsigned long long ma_ = __ballot(x_ < (TA_)); \
        const unsigned long long mu_ = ~(ma_ | __ballot(x_ > (TR_)));   /* the lanes between the two thresholds (a NaN too): the exact test decides */ \
        ACC_ = ma_; \
        if (__builtin_expect(mu_ != 0ull, 0)) { \
          const bool ex_ = lane_accept_exact(diffd_, a.marker0 + (uint32_t)(blk * m + (MKOFF_) + lane), a.flags, Cc, odds, one_minus_pi, a.rng, a.iter); \
          ACC_ |= __ballot(ex_) & mu_; \
        } }
      // An included marker k = KOFF_ + js: its step beyond the speculated one, corr = (b1 - b0) - drej (both floats, exact in
      // fp64; the streamers fold in the same difference on the fixed-point grid, 2^-44 of the scale away), and its Gram rows, read
      // on demand: the packed diagonal block for the later markers of this block, the distance-1 cross block for the next one.
      // the distance-1 / 2 rows requested so far (16-bit panels: S3_NRX pairs of LDS slots): waited for and applied to the next two
      // blocks' dots -- after the block's last round, and when a block includes more markers than there are slots
#define S3_APPLY_ROWS() { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        /* every LDS read of a slot unconditional and ahead of the first use (a read under "if (use1)" compiled into read, wait, convert, \
           fma, read, wait, ...: four LDS round trips per slot on the chain's wave); a row that was not requested -- no next block -- holds \
           stale 16-bit values and meets a zero coefficient */ \
        for (int i_ = 0; i_ < nslot; ++i_) { \
          const double cf_ = accC[(pos0 + napp + i_) & (ring - 1)]; \
          const uint16_t *rw1_ = reinterpret_cast<const uint16_t *>(rowx_s + (size_t)i_ * S3_ROWSLOT); \
          const uint16_t *rw2_ = rw1_ + m; \
          const uint32_t x1a_ = rw1_[l0c], x1b_ = rw1_[l1c], x2a_ = rw2_[l0c], x2b_ = rw2_[l1c]; \
          const double c1_ = use1 ? cf_ : 0.0, c2_ = use2 ? cf_ : 0.0; \
          rnext0 = fma(-(double)x1a_, c1_, rnext0); rnext1 = fma(-(double)x1b_, c1_, rnext1); \
          rnxt20 = fma(-(double)x2a_, c2_, rnxt20); rnxt21 = fma(-(double)x2b_, c2_, rnxt21); \
        } \
        napp += nslot; nslot = 0; }
#define S3_INCLUDE(KOFF_, D1F_, DR_) { \
        const int k_ = (KOFF_) + js; \
        GT ga_ = (GT)1, gb_ = (GT)1; \
        if constexpr (G16) { \
          if (!(SDBG & 64)) { \
          if constexpr (GPD) { \
            const unsigned char *gpl_ = gpd_lane + gpd_off + __builtin_amdgcn_readlane((KOFF_) ? tabp1 : tabp0, js); \
            ga_ = *reinterpret_cast<const GT *>(gpl_); gb_ = *reinterpret_cast<const GT *>(gpl_ + 128);   /* (lanes at or before k_: masked below) */ \
          } else { \
            const int pr_ = prow(k_); \
            ga_ = gp[min(pr_ + max(l0 - k_ - 1, 0), pstride - 1)]; gb_ = gp[min(pr_ + max(l1c - k_ - 1, 0), pstride - 1)]; \
          } \
          if (__builtin_expect(nslot == S3_NRX, 0)) S3_APPLY_ROWS()   /* every slot taken: the rows so far first */ \
          if (use1) s3_dma16s(g12b, rolane16 + (uint32_t)(k_ * 2 * rowbytes), rowx_la + (uint32_t)nslot * S3_ROWSLOT); \
          ++nslot; } \
        } else if (!(SDBG & 64)) {   /* 32-bit Gram entries: the rows straight from global memory */ \
          const int pr_ = prow(k_); \
          ga_ = gp[min(pr_ + max(l0 - k_ - 1, 0), pstride - 1)]; gb_ = gp[min(pr_ + max(l1c - k_ - 1, 0), pstride - 1)];   /* (the last row is empty) */ \
        } \
        const float dacc_ = readlane_f32(D1F_, js), drj_ = readlane_f32(DR_, js); \
        const double corr_ = (double)dacc_ - (double)drj_; \
        if (lane == 0) { const int sl_ = (pos0 + nacc) & (ring - 1); accK[sl_] = k_ | (b << 8); accC[sl_] = corr_; accS[sl_] = make_float2(dacc_, drj_); } \
        ++nacc; \
        if constexpr (CEN) { \
          const double xk_ = (double)__builtin_amdgcn_readlane((KOFF_) ? csb : csa, js) * ninv;   /* mean of column k */ \
          cenU = fma(xk_, corr_, cenU); \
          r0 = fma(-((l0 > k_) ? ((double)ga_ - sA * xk_) : 0.0), corr_, r0); \
          r1 = fma(-((l1 > k_ && l1 < m) ? ((double)gb_ - sB * xk_) : 0.0), corr_, r1); \
        } else { \
        r0 = fma(-(double)((l0 > k_) ? ga_ : (GT)0), corr_, r0); \
        r1 = fma(-(double)((l1 > k_ && l1 < m) ? gb_ : (GT)0), corr_, r1); \
        } \
        if constexpr (!G16) { if (!(SDBG & 64)) { \
          const GT *row_ = g1 + (use1 ? (size_t)k_ * m : (size_t)0);   /* (without a next block: a harmless in-bounds read) */ \
          const GT xa_ = row_[use1 ? min(l0, m - 1) : 0], xb_ = row_[use1 ? l1c : 0]; \
          const GT *row2_ = g2 + (use2 ? (size_t)k_ * m : (size_t)0); \
          const GT ya_ = row2_[use2 ? min(l0, m - 1) : 0], yb_ = row2_[use2 ? l1c : 0]; \
          if (use1) { rnext0 = fma(-(double)xa_, corr_, rnext0); rnext1 = fma(-(double)xb_, corr_, rnext1); } \
          if (use2) { rnxt20 = fma(-(double)ya_, corr_, rnxt20); rnxt21 = fma(-(double)yb_, corr_, rnxt21); } } } }
      // exact speculative rounds, first over markers 0..63, then 64..127: every lane assumes "nobody before me is included"
      if (!(SDBG & 128)) {
        const int cnt0 = min(64, mB);
        unsigned long long live0 = (cnt0 >= 64) ? ~0ull : ((1ull << cnt0) - 1ull);   // lanes not yet passed
        for (;;) {
          float d1f; unsigned long long acc;
          S3_EVAL(r0, xba, rda, sza, b0a, D2a, D2sa, gja, taa, tra, 0, d1f, acc)
          const unsigned long long bal = acc & live0;
          if (bal == 0ull) break;
          const int js = (int)__builtin_ctzll(bal);
          live0 &= (~1ull << js);
          am0 |= 1ull << js;
          S3_INCLUDE(0, d1f, dra)
        }
      }
      if (mB > 64 && (!(SDBG & 128))) {
        const int cnt1 = mB - 64;
        unsigned long long live1 = (cnt1 >= 64) ? ~0ull : ((1ull << cnt1) - 1ull);
        for (;;) {
          float d1f; unsigned long long acc;
          S3_EVAL(r1, xbb, rdb, szb, b0b, D2b, D2sb, gjb, tab, trb, 64, d1f, acc)
          const unsigned long long bal = acc & live1;
          if (bal == 0ull) break;
          const int js = (int)__builtin_ctzll(bal);
          live1 &= (~1ull << js);
          am1 |= 1ull << js;
          S3_INCLUDE(64, d1f, drb)
        }
      }
#undef S3_EVAL
#undef S3_INCLUDE
      S3ST(5, sq0);
      if constexpr (G16) { pend_n = nslot; pend_p = pos0 + napp; pend_u1 = use1; pend_u2 = use2; }   // the rows requested as the markers appeared: applied where the next block needs them
#undef S3_APPLY_ROWS
      S3ST(2, sq0);
      // the block's new effects (every lane's r is final for its own marker); the rest of the outputs is wave 7's
      {
        float *sp = state_s + (size_t)(b & 1) * 2 * SW_MAXM;
        const float b1a = (float)fma(r0 + xba, rda, sza), b1b = (float)fma(r1 + xbb, rdb, szb);
        const bool ia = ((am0 >> lane) & 1ull) != 0ull, ib = ((am1 >> lane) & 1ull) != 0ull;
        sp[l0] = ia ? b1a : b2a; sp[l1] = ib ? b1b : b2b;
        sp[SW_MAXM + l0] = ia ? 1.0f : 0.0f; sp[SW_MAXM + l1] = ib ? 1.0f : 0.0f;
      }
      if (lane == 0) pos_s[(b + 1) & 31] = (pos0 + nacc) & (ring - 1);
      S3ST(3, sq0);
      gpd_off = (gpd_off == 2 * S3_GPD_BYTES) ? 0 : gpd_off + S3_GPD_BYTES;
      S3_ROLE_BARRIER();
    } else if (have_next && wvu != 0) {
      helper_phase(b + 1);
      S3ST(1, sq1 || sq2 || sq4 || sq3 || sq6 || sq7);
    } else S3_ROLE_BARRIER();
    // block b's rounds are done, its list is in LDS; everything block b+1 needs from the helpers is in LDS.  (A bare barrier:
    // __syncthreads() would drain the far-field rows that are meant to stay in flight across it.)
    if (!BWGR_ROLEBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    S3ST(4, sq0 || sq1 || sq2 || sq4 || sq3 || sq6 || sq7);
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }   // (no DMA may land after the workgroup has gone)
  }
  S3ST_FLUSH(16, sq0); S3ST_FLUSH(24, sq1); S3ST_FLUSH(32, sq2); S3ST_FLUSH(40, sq4); S3ST_FLUSH(48, sq3); S3ST_FLUSH(56, sq6); S3ST_FLUSH(64, sq7);
  if constexpr (CEN) { if (tid == 0) a.sc->cen_c = cenU - cen_u0; }   // the included markers' share of the shift (k_cen_end)
  if (wave >= 1 && wave <= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the poll and staging waves' requests past the end; wave 4's touches)
  if (wave == 7) {   // the last two blocks
    if (nb >= 2) finish_block(nb - 2);
    finish_block(nb - 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
    if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// prefetcher: one workgroup on the sequencer's XCD (equal blockIdx mod 8) that walks a dozen blocks ahead of the sequencer and
// touches one dword per 128-byte line of the Gram rows wave 0 may ask for on demand -- the packed diagonal block and the cross
// blocks of distance 1 and 2, 80 KB per block -- so that those reads are L2 hits (a few hundred cycles) instead of HBM misses
// (about 2 us on a loaded chip, on the chain's critical path).  Paced by the lists the sequencer publishes; speed only.
// ------------------------------------------------------------------------------------------------------------------
template <typename GT, bool ROWS>
__device__ __forceinline__ void s3_prefetcher(const Sweep3Args &A) {
  const SweepArgs &a = A.a;
  const int tid = threadIdx.x, nb = a.blk_end - a.blk_begin;
  // Four blocks per trip, their touches in flight together, one poll per trip: a trip costs one poll round trip plus one HBM round trip, so the
  // prefetcher keeps ahead of a sequencer that takes a block every 1.2 - 1.6 us.  (One block per trip -- a poll and a dependent load wave each --
  // took about 2 us a block: the prefetcher fell behind the sequencer it serves, whose staging loads then missed L2 again.)
  constexpr int AHEAD = 16, NPB = 4;
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  const size_t gpbytes = (size_t)a.pstride * sizeof(GT);
  uint32_t sink = 0u;
  const size_t o = (size_t)tid * 128;
  for (int c0 = 0; c0 < nb; c0 += NPB) {
    const int clast = min(c0 + NPB, nb) - 1;
    if (clast >= AHEAD) {   // wait (one lane polls) until the sequencer has published the list of block clast - AHEAD
      const unsigned long long *L = A.lists + (size_t)(a.blk_begin + clast - AHEAD) * S3_LSTRIDE;
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        if (s3_epoch_is(ld_agent_raw64(L), A.epoch)) break;
        if ((++spins & 63u) == 0u && (ld_agent_u32(abortw) != 0u || wall_clock64() - t0 > SW_TIMEOUT_TICKS)) return;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    // what the staging waves will ask for: the blocks' constants, speculative terms and packed diagonal Gram blocks, one dword per 128-byte line
    // (their loads are HBM misses ~2 us away on the sequencer's CU, and they must be back within one block period)
    uint32_t v[NPB][3];
    const size_t g12bytes = (size_t)a.m * 2 * a.m * sizeof(GT);
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int blk = a.blk_begin + min(c0 + u, nb - 1);
      if (ROWS) {
        // the second prefetcher: what wave 0 asks for when it includes a marker -- the marker's distance-1 / 2 rows (16-bit panels: one record of
        // 4 m bytes per marker, 64 KB a block), an HBM miss of a block period otherwise, waited for at the top of the next block
        v[u][0] = *reinterpret_cast<const uint32_t *>(A.gx12 + (size_t)blk * g12bytes + min(o, g12bytes - 4));
        v[u][1] = 0u; v[u][2] = 0u;
      } else {
        const unsigned char *gpb = reinterpret_cast<const unsigned char *>(A.gp) + (size_t)blk * gpbytes;
        const unsigned char *stb = reinterpret_cast<const unsigned char *>(a.ps.blocks + blk);
        const unsigned char *spb = reinterpret_cast<const unsigned char *>(a.ps.spec + blk);
        v[u][0] = *reinterpret_cast<const uint32_t *>(stb + min(o, sizeof(StageBuf) - 4));
        v[u][1] = *reinterpret_cast<const uint32_t *>(spb + min(o, sizeof(SpecBuf) - 4));
        v[u][2] = *reinterpret_cast<const uint32_t *>(gpb + min(o, gpbytes - 4));
      }
    }
#pragma unroll
    for (int u = 0; u < NPB; ++u) sink += v[u][0] + v[u][1] + v[u][2];
  }
  if (sink == 0x9E3779B9u && a.stamps) a.stamps[255] = sink;   // (keeps the loads alive)
}

template <typename GT, bool CEN = false>
__global__ __launch_bounds__(SW_THREADS) void k_sweep3(const Sweep3Args A) {
  if (!(A.a.sc->inc_rate < A.a.gate3)) return;   // this sweep is k_sweep2's (dense inclusion: every workgroup sees the same scalar)
  if ((int)blockIdx.x == A.pf) { s3_prefetcher<GT, false>(A); return; }
  if ((int)blockIdx.x == A.pf2) { s3_prefetcher<GT, true>(A); return; }
  if (blockIdx.x == 0) { if (!(A.dbg & 1024)) s3_sequencer<GT, CEN>(A); }
  else if ((A.a.flags & SWF_DEBUG_WITHHOLD) && blockIdx.x == 1 && A.pf != 1) return;   // test hook: a streamer that never shows up
  else if (A.dbg & 2048) return;
  else if ((A.dbg & (1 << 22)) && !(A.dbg & (1 << 23)) && A.R3 == 128) s3_streamer_dma<128, 4>(A);
  else if ((A.dbg & (1 << 23)) && A.R3 == 128) s3_streamer_dma<128, 3>(A);
  else s3_streamer(A);   // (a second streamer whose every load was inline asm with hand-counted waits measured no faster and was removed: DESIGN 9.0)
}

}  // namespace bwgr
