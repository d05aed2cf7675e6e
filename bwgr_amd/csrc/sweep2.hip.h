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
// diagnostic build: per-phase cycle sums of streamer 0 (stamps[0..15]) and of the sequencer (stamps[16..31]), and
// wall-clock sums for the hand-off latencies (stamps[32..])
#define S2STAMP_DECL unsigned long long ph2[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, tl2 = __builtin_amdgcn_s_memtime()
#define S2STAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph2[k] += t_ - tl2; tl2 = t_; } } while (0)
#define S2STAMP_FLUSH(base) do { if (tid == 0 && a.stamps) for (int k_ = 0; k_ < 16; ++k_) a.stamps[(base) + k_] += ph2[k_]; } while (0)
// wall-clock sums are kept in registers and flushed once at the end (an atomic per sample would sit in front of the next vmcnt(0))
#define S2WALL_DECL unsigned long long wl2[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}
#define S2WALL(slot, cond) do { if (cond) wl2[slot] += (unsigned long long)wall_clock64(); } while (0)
#define S2WALL_FLUSH do { if (a.stamps) for (int k_ = 0; k_ < 16; ++k_) if (wl2[k_]) atomicAdd(&a.stamps[32 + k_], wl2[k_]); } while (0)
#define S2ONE(dst, cond) do { if ((cond) && a.stamps) a.stamps[dst] = (unsigned long long)wall_clock64(); } while (0)
#define S2MINMAX(dmin, dmax, cond) do { if ((cond) && a.stamps) { const unsigned long long t_ = (unsigned long long)wall_clock64(); atomicMax(&a.stamps[dmax], t_); atomicMax(&a.stamps[dmin], ~t_); } } while (0)
#define S2WALL_FLUSH_AT(base) do { if (a.stamps) for (int k_ = 4; k_ < 8; ++k_) if (wl2[k_]) { atomicAdd(&a.stamps[(base) + k_ - 4], wl2[k_]); wl2[k_] = 0; } } while (0)
#else
#define S2STAMP_DECL do { } while (0)
#define S2STAMP(k) do { } while (0)
#define S2STAMP_FLUSH(base) do { } while (0)
#define S2WALL_DECL do { } while (0)
#define S2WALL(slot, cond) do { } while (0)
#define S2WALL_FLUSH do { } while (0)
#define S2ONE(dst, cond) do { } while (0)
#define S2MINMAX(dmin, dmax, cond) do { } while (0)
#define S2WALL_FLUSH_AT(base) do { } while (0)
#endif

static constexpr int S2_NSLOT = 8;   // ring depth of the q / delta buffers (the affine sweeps' pipeline runs up to seven blocks deep)

template <typename XT> __host__ __device__ inline size_t s2_seq_lds_bytes(int m);
__host__ __device__ inline size_t s2i_lds_bytes(int m, int R, int lag = 3);
__host__ __device__ inline size_t s2_seq16_lds_bytes(int m);   // (defined with s2_sequencer_sel16)
template <typename XT> __host__ __device__ inline size_t sweep2_lds_bytes(int m, int R) {
  size_t streamer = (size_t)3 * m * tile_rp<XT>(R) * sizeof(XT);
  streamer = (streamer + 15) & ~(size_t)15;
  streamer += (size_t)R * sizeof(double) + SW_MAXM * sizeof(double) + (size_t)(SW_THREADS / 64) * SW_MAXM * sizeof(double) + 64;
  if (sizeof(XT) == 1) streamer = s2i_lds_bytes(m, R, 3);
  size_t seq = s2_seq_lds_bytes<XT>(m);
  if (sizeof(XT) == 1 && s2_seq16_lds_bytes(m) > seq) seq = s2_seq16_lds_bytes(m);   // small blocks: the 16-bit sequencer's fixed arrays outweigh its Gram buffers
  return streamer > seq ? streamer : seq;
}

// delta hand-off: one 8-byte granule per marker, {hi: block epoch (24 bits) | largest float exponent field of the
// block's deltas (8 bits), lo: the float delta}: the datum is the flag, and the int8 streamers get the scale of their
// fixed-point digits without a reduction of their own
__device__ __forceinline__ unsigned long long s2_dgranule(int b, uint32_t exmax, float dl) {
  return ((unsigned long long)((((uint32_t)(b + 1)) & 0xFFFFFFu) | (exmax << 24)) << 32) | (unsigned long long)__float_as_uint(dl);
}
__device__ __forceinline__ bool s2_dgranule_is(unsigned long long v, int b) { return (((uint32_t)(v >> 32)) & 0xFFFFFFu) == (((uint32_t)(b + 1)) & 0xFFFFFFu); }

// q hand-off: each slab dot travels as ONE 8-byte word whose low 8 mantissa bits carry the block's tag (1..255; the
// slots are zeroed before every launch and a slot's consecutive users differ in tag), so a reader needs a single round
// trip and the writer no payload fence.  The value loses 8 of its 53 mantissa bits (relative 2^-45).
__device__ __forceinline__ unsigned long long s2_qtag(int b) { return (unsigned long long)(b % 255 + 1); }
__device__ __forceinline__ void s2_put_q(double *slot, double v, int b) {
  st_agent_raw64(reinterpret_cast<unsigned long long *>(slot), ((unsigned long long)__double_as_longlong(v) & ~0xFFull) | s2_qtag(b));
}

// tile macros with an explicit thread numbering (streamers use all 512 threads)
#define S2_ISSUE1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) name = src_[c_]; }
// (the vmcnt(0) in front: as for S2_TILE_COMMIT; the previous tile's loads have been committed by then)
#define S2_TILE_ISSUE(j0_, mB_) do { const int tot_ = (mB_) * (R / PER); __builtin_amdgcn_s_waitcnt(0x0F70); \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(X + (size_t)(j0_) * R); BWGR_TILE_EACH(S2_ISSUE1) } while (0)
#define S2_COMMIT1(u, name) { const int c_ = tid + (u) * SW_THREADS; if (c_ < tot_) { \
    const int jj_ = (int)(((float)c_ + 0.5f) * rcpr_), ii_ = c_ - jj_ * cpr_; \
    *reinterpret_cast<uint4 *>((dst_) + (size_t)jj_ * Rp + ii_ * PER) = name; } }
// The unconditional vmcnt(0) in front tells the compiler that none of the (conditionally issued) tile loads is in flight
// when the registers are written again; without it every load of the next S2_TILE_ISSUE waits for the one before it.
#define S2_TILE_COMMIT(dstp, mB_) do { XT *dst_ = (dstp); const int cpr_ = R / PER; const int tot_ = (mB_) * cpr_; \
    const float rcpr_ = 1.0f / (float)cpr_; __builtin_amdgcn_s_waitcnt(0x0F70); BWGR_TILE_EACH(S2_COMMIT1) } while (0)

// int8 streamer: the tile moves belong to waves 4-7 (256 threads, eight chunks each).  Waves 0-1 poll the delta granules and
// must never have a tile load in flight (the vmcnt counter is in order: a poll behind tile loads waits for them); waves 0-3
// carry the e update and the digit split, so with the movers kept apart the commit runs beside the e update and the
// ~600 cycles of load issue beside the e digits, instead of both in front of the e-update barrier
#define S2I_MOVERS 256
static constexpr size_t S2I_TILE_BYTES_MAX = (size_t)8 * 16 * (SW_THREADS - S2I_MOVERS);   // eight 16-byte chunks per mover thread
#define S2I_TILE_EACH(X) X(0, tp0) X(1, tp1) X(2, tp2) X(3, tp3) X(4, tp4) X(5, tp5) X(6, tp6) X(7, tp7)
#define S2I_ISSUE1(u, name) { const int c_ = (tid - S2I_MOVERS) + (u) * (SW_THREADS - S2I_MOVERS); if (c_ < tot_) name = src_[c_]; }
#define S2I_TILE_ISSUE(j0_, mB_) do { if (tid >= S2I_MOVERS) { const int tot_ = (mB_) * (R / PER); __builtin_amdgcn_s_waitcnt(0x0F70); \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(X + (size_t)(j0_) * R); S2I_TILE_EACH(S2I_ISSUE1) } } while (0)
#define S2I_COMMIT1(u, name) { const int c_ = (tid - S2I_MOVERS) + (u) * (SW_THREADS - S2I_MOVERS); if (c_ < tot_) { \
    const int jj_ = (int)(((float)c_ + 0.5f) * rcpr_), ii_ = c_ - jj_ * cpr_; \
    *reinterpret_cast<uint4 *>((dst_) + (size_t)jj_ * Rp + ii_ * PER) = name; } }
#define S2I_TILE_COMMIT(dstp, mB_) do { if (tid >= S2I_MOVERS) { XT *dst_ = (dstp); const int cpr_ = R / PER; const int tot_ = (mB_) * cpr_; \
    const float rcpr_ = 1.0f / (float)cpr_; __builtin_amdgcn_s_waitcnt(0x0F70); S2I_TILE_EACH(S2I_COMMIT1) } } while (0)

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
  int *fail_s = reinterpret_cast<int *>(smem + off);   // set once, on a failed wait
  const XT *X = reinterpret_cast<const XT *>(a.X) + (size_t)wg * a.p * R;
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  const int mpad = (m <= 64) ? 64 : 128;
  const int ngroups = SW_THREADS / mpad, rows_per_group = R / ngroups;
  auto blk_j0 = [&](int b) { return (a.blk_begin + b) * m; };
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };

  uint4 tp0 = make_uint4(0, 0, 0, 0), tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0;
  for (int i = tid; i < R; i += SW_THREADS) e_s[i] = a.e[row0 + i];
  if (tid == 0) fail_s[0] = 0;
  const int L = a.lag < 3 ? a.lag : 3;   // q_b is computed after delta_{b-L}: tiles i .. i+L-1 sit in the ring of three
  for (int b = 0; b < L && b < nb; ++b) { S2_TILE_ISSUE(blk_j0(b), blk_m(b)); S2_TILE_COMMIT(S2_TILE(b), blk_m(b)); }
  if (nb > L) S2_TILE_ISSUE(blk_j0(L), blk_m(L));
  __syncthreads();

  // publish the slab dots of block b (computed against the current e_s): payload write-through, then the epoch flag
  auto publish = [&](int b, const XT *tile) {
    const int mB = blk_m(b);
    s2_slab_dots<XT>(tile, e_s, part_s, mB, Rp, mpad, rows_per_group, tid);
    __syncthreads();
    if (tid < mB) {
      double mine = 0.0;
      for (int g = 0; g < ngroups; ++g) mine += part_s[g * SW_MAXM + tid];
      s2_put_q(a.qpart + ((size_t)(b % S2_NSLOT) * K + wg) * SW_MAXM + tid, mine, b);
    }
    __syncthreads();   // part_s is free again
  };
  for (int b = 0; b < L && b < nb; ++b) publish(b, S2_TILE(b));
  S2STAMP_DECL;

  for (int i = 0; i < nb; ++i) {
    const int mB = blk_m(i);
    S2STAMP(0);
    // delta_i: one 8-byte {epoch, float} granule per marker, polled by the thread that needs it
    int bad = 0;
    if (tid < mB) {
      const unsigned long long *g = a.dgran + (size_t)(i % S2_NSLOT) * SW_MAXM + tid;
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        const unsigned long long v = ld_agent_raw64(g);
        if (s2_dgranule_is(v, i)) { delta_s[tid] = (double)__uint_as_float((uint32_t)v); break; }
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
    // tile(i) is spent: tile(i+L), in registers since the previous iteration, takes a free ring slot (for L = 3 the one of
    // tile(i)), and the loads of tile(i+L+1) go out (see s2_streamer_i8)
    if (i + L < nb) S2_TILE_COMMIT(S2_TILE(i + L), blk_m(i + L));
    if (i + L + 1 < nb) S2_TILE_ISSUE(blk_j0(i + L + 1), blk_m(i + L + 1));
    __syncthreads();
    if (i + L < nb) publish(i + L, S2_TILE(i + L));
    S2STAMP(4);
  }
  if (wg == 0) S2STAMP_FLUSH(0);
  __syncthreads();
  for (int i = tid; i < R; i += SW_THREADS) a.e[row0 + i] = e_s[i];
}


// ------------------------------------------------------------------------------------------------------------------
// int8 streamer: both slab phases as exact integer matrix products on v_mfma_i32_16x16x64_i8
// ------------------------------------------------------------------------------------------------------------------
// A value v (fp64 residual, or the float delta) is taken relative to a power-of-two scale S above the block maximum,
// rounded to an integer of < 55 (e) / 47 (delta) bits and split into signed base-256 digits; int8 x times a digit,
// summed over the slab rows / the block's markers, is exact in the MFMA's int32 accumulators, and sum_n acc_n 256^n / S
// is formed in fp64.  The only inexact step is the rounding of v to 2^-54 (e) / 2^-46 (delta) of its block maximum.
//   dots:    M = 16 markers, K = 64 slab rows, N = digit n of e       (A dword = 4 rows of one marker, as stored)
//   update:  M = 16 rows,    K = 64 markers,   N = digit n of delta   (A dword = 4 markers of one row: 4x4 byte
//                                                                      transposes of the stored micro-tiles, v_perm)
// Both operands of a product use the same (lane group, dword, byte) -> k map, so the k order inside the instruction
// does not matter; the M / N / accumulator maps were checked with exact integer data (tools/mfma_probe.hip).
typedef int s2_v4i __attribute__((ext_vector_type(4)));
static constexpr int S2_NDE = 7;     // digits of e:     |q| < 2^54
static constexpr int S2_NDD = 6;     // digits of delta: |q| < 2^46
static constexpr int S2_OS = 12;     // dwords per row of the int32 output array (8 used; 12 keeps its b128 reads conflict-free)
static constexpr int S2_DP = SW_MAXM + 16;   // bytes per digit row of the delta digits
__host__ __device__ inline size_t s2i_lds_bytes(int m, int R, int lag) {   // ring of max(3, lag) tiles
  const size_t Rp = (size_t)R + 16;
  return (size_t)(lag > 3 ? lag : 3) * m * Rp + (size_t)R * 8 + 16 * Rp + 16 * S2_DP + (size_t)(R > SW_MAXM ? R : SW_MAXM) * S2_OS * 4 + 64;
}
__device__ __forceinline__ double pow2_field(int field) { return __hiloint2double(field << 20, 0); }   // 2^(field-1023)
// maximum over the wave (DPP row shifts + row broadcasts, result broadcast from lane 63); v >= 0 as int
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  int x = (int)v;
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true));   // row_shr:1
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true));   // row_shr:2
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true));   // row_shr:4
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true));   // row_shr:8  -> lane 15 of each row holds the row maximum
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true));   // row_bcast:15 into rows 1 and 3
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true));   // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// sum over the wave of int32 values whose total fits 32 bits (the same DPP ladder; result broadcast from lane 63)
__device__ __forceinline__ int wave_sum_i32(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);
  return __builtin_amdgcn_readlane(x, 63);
}
// signed base-256 digits of q into bytes dst[n*stride], n < ND
template <int ND> __device__ __forceinline__ void put_digits(long long q, int8_t *dst, int stride) {
#pragma unroll
  for (int n = 0; n < ND; ++n) {
    const int dg = (int)((q + 128) & 255) - 128;
    dst[n * stride] = (int8_t)dg;
    q = (q - dg) >> 8;
  }
}

__device__ __forceinline__ void s2_streamer_i8(const SweepArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // declared here so that LDS accesses stay ds_*
  using XT = int8_t;
  constexpr int PER = 16;
  const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, grp = lane >> 4;   // MFMA lane coordinates
  const int m = a.m, R = a.R, K = a.K;
  const int Rp = R + 16;                        // bytes per marker in an LDS tile, and per digit row of e
  const int row0 = wg * R;
  const int nb = a.blk_end - a.blk_begin;
  const int L = a.lag;   // q_b is computed after delta_{b-L} has been applied (2..4): tiles i .. i+L-1 sit in a ring of max(3, L)
  const int NR = L > 3 ? L : 3;
  const size_t tile_b = (size_t)m * Rp;
#define S2I_TILE(i_) reinterpret_cast<int8_t *>(smem + (size_t)((i_) % NR) * tile_b)
  size_t off = (size_t)NR * tile_b;
  double *e_s = reinterpret_cast<double *>(smem + off); off += (size_t)R * sizeof(double);
  int8_t *edig_s = reinterpret_cast<int8_t *>(smem + off); off += (size_t)16 * Rp;        // [n][row]: digit n of e[row]; rows n >= S2_NDE stay 0
  int8_t *ddig_s = reinterpret_cast<int8_t *>(smem + off); off += 16 * S2_DP;             // [n][marker]: digit n of delta[marker]
  const size_t out_n = (size_t)(R > SW_MAXM ? R : SW_MAXM) * S2_OS;
  int *out_s = reinterpret_cast<int *>(smem + off); off += out_n * 4;   // [row or marker][n]
  uint32_t *ctl_s = reinterpret_cast<uint32_t *>(smem + off);   // [0],[1]: max exponent of delta (by block parity); [2],[3]: of e; [8]: failure
  const int8_t *X = reinterpret_cast<const int8_t *>(a.X) + (size_t)wg * a.p * R;
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  auto blk_j0 = [&](int b) { return (a.blk_begin + b) * m; };
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };

  uint4 tp0 = make_uint4(0, 0, 0, 0), tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0, tp5 = tp0, tp6 = tp0, tp7 = tp0;
  for (int i = tid; i < 4 * Rp; i += SW_THREADS) reinterpret_cast<uint32_t *>(edig_s)[i] = 0u;
  for (int i = tid; i < 4 * S2_DP; i += SW_THREADS) reinterpret_cast<uint32_t *>(ddig_s)[i] = 0u;
  if (tid < 16) ctl_s[tid] = 0u;
  for (int b = 0; b < L && b < nb; ++b) { S2I_TILE_ISSUE(blk_j0(b), blk_m(b)); S2I_TILE_COMMIT(S2I_TILE(b), blk_m(b)); }
  if (nb > L) S2I_TILE_ISSUE(blk_j0(L), blk_m(L));
  __syncthreads();
  S2STAMP_DECL;
  S2WALL_DECL;
#ifdef BWGR_STAMPS
  unsigned long long wseen = 0, wq = 0;   // per-streamer wall-clock sums: delta_i seen, q_{i+L} stored
#endif

  // digits of the e slab relative to the maximum exponent field recorded in ctl_s[2 + epar]; returns 1/S
  auto e_digits = [&](int epar) -> double {
    const int E = max((int)ctl_s[2 + epar], 100);            // max|e| < 2^(E-1022);  S = 2^(1076-E): |q| < 2^54
    const double S = pow2_field(2099 - E);
    for (int r = tid; r < R; r += SW_THREADS) put_digits<S2_NDE>(__double2ll_rn(e_s[r] * S), edig_s + r, Rp);
    return pow2_field(E - 53);
  };
  // slab dots of block b against the current e digits, payload write-through, then the epoch flag
  auto publish = [&](int b, const int8_t *tile, double invSe) {
    const int mB = blk_m(b);
    __syncthreads();                                          // e digits visible; out_s free
    if (wave * 16 < mB) {
      const int8_t *ap = tile + (size_t)(16 * wave + m16) * Rp + 16 * grp;
      const int8_t *bp = edig_s + (size_t)m16 * Rp + 16 * grp;
      s2_v4i acc = {0, 0, 0, 0};
      for (int r = 0; r < R; r += 64)                         // 64 rows per MFMA: lane group grp holds rows r+16grp .. +15
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const s2_v4i *>(ap + r), *reinterpret_cast<const s2_v4i *>(bp + r), acc, 0, 0, 0);
      if (m16 < 8) {                                          // lane: digit n = m16 of markers 16 wave + 4 grp + reg
        int *op = out_s + (size_t)(16 * wave + 4 * grp) * S2_OS + m16;
        op[0] = acc[0]; op[S2_OS] = acc[1]; op[2 * S2_OS] = acc[2]; op[3 * S2_OS] = acc[3];
      }
    }
    __syncthreads();
    S2STAMP(7);
    // out_s must be all zero when the next update starts (its partial products are added with LDS atomics): the threads that
    // recombine a marker's row clear it behind them, waves 4-7 clear the rows past the markers (last read by the e update)
    if (tid >= 256) for (int c = tid - 256; c < (R - SW_MAXM) * S2_OS / 4; c += 256) reinterpret_cast<uint4 *>(out_s + (size_t)SW_MAXM * S2_OS)[c] = make_uint4(0, 0, 0, 0);
    const int tq = tid - 128;   // waves 2-3 store q: waves 0-1 keep nothing but polls on their memory counter
    if (tq >= 0 && tq < mB) {
      const int4 o0 = *reinterpret_cast<const int4 *>(out_s + (size_t)tq * S2_OS), o1 = *reinterpret_cast<const int4 *>(out_s + (size_t)tq * S2_OS + 4);
      double w = invSe, v = (double)o0.x * w;
      w *= 256.0; v = fma((double)o0.y, w, v); w *= 256.0; v = fma((double)o0.z, w, v); w *= 256.0; v = fma((double)o0.w, w, v);
      w *= 256.0; v = fma((double)o1.x, w, v); w *= 256.0; v = fma((double)o1.y, w, v); w *= 256.0; v = fma((double)o1.z, w, v);
      s2_put_q(a.qpart + ((size_t)(b % S2_NSLOT) * K + wg) * SW_MAXM + tq, v, b);
    }
    if (tq >= 0 && tq < SW_MAXM) {
      uint4 *rowp = reinterpret_cast<uint4 *>(out_s + (size_t)tq * S2_OS);
      rowp[0] = make_uint4(0, 0, 0, 0); rowp[1] = make_uint4(0, 0, 0, 0); rowp[2] = make_uint4(0, 0, 0, 0);
    }
    S2WALL(2, wg == 0 && tid == 0 && b >= a.lag);
#ifdef BWGR_STAMPS
    if (tid == 0 && b >= a.lag) wq += (unsigned long long)wall_clock64();
#endif
    S2WALL(6, wg == 0 && tid == 0 && b == 100 + a.lag);
    S2MINMAX(56, 57, tid == 128 && b == 100 + a.lag);
  };
  static_assert(S2_NDE == 7 && S2_NDD == 6, "the digit recombinations are written out for 7 / 6 digits");

  {
    uint32_t ex = 0u;
    for (int r = tid; r < R; r += SW_THREADS) { const double v = a.e[row0 + r]; e_s[r] = v; ex = max(ex, (uint32_t)((__double2hiint(v) & 0x7FFFFFFF) >> 20)); }
    ex = wave_max_u32(ex);
    if (lane == 0) atomicMax(&ctl_s[2], ex);
    __syncthreads();
    const double invSe = e_digits(0);
    for (int b = 0; b < L && b < nb; ++b) publish(b, S2I_TILE(b), invSe);
  }

  unsigned long long pre = 0ull;   // early-requested granule of the next block's delta (tid < SW_MAXM)
  for (int i = 0; i < nb; ++i) {
    const int mB = blk_m(i);
    const int par = i & 1;
    if (tid == 0) ctl_s[2 + (par ^ 1)] = 0u;      // next block's e maximum (this block's was reset one iteration ago)
    S2STAMP(0);
    // delta_i: one granule per marker, polled by the thread that needs it; it also carries the block's largest exponent
    // field ex: |delta| < 2^(ex-126), so S = 2^(172-ex) gives |q| < 2^46 and the digits follow at once
    int bad = 0;
    if (tid < SW_MAXM) {
      uint32_t dbits = 0u, dex = 0u;
      if (tid < mB) {
        const unsigned long long *g = a.dgran + (size_t)(i % S2_NSLOT) * SW_MAXM + tid;
        const uint64_t t0 = wall_clock64();
        unsigned spins = 0;
        unsigned long long v = pre;   // requested during the previous iteration: when this streamer runs behind, delta_i is
                                      // already there and the poll's round trip (~0.6 us) is off its iteration
        for (;;) {
          if (s2_dgranule_is(v, i)) break;
          v = ld_agent_raw64(g);
          if (s2_dgranule_is(v, i)) break;
          if ((++spins & 63u) == 0u) {
            if (ld_agent_u32(abortw) != 0u) { bad = 1; break; }
            if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); bad = 1; break; }
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!bad) { dbits = (uint32_t)v; dex = (uint32_t)(v >> 56); }
      }
      S2STAMP(10);
      put_digits<S2_NDD>(__double2ll_rn((double)__uint_as_float(dbits) * pow2_field(1195 - (int)dex)), ddig_s + tid, S2_DP);
      if (tid == 0) ctl_s[par] = dex;
    }
    if (bad) ctl_s[8] = 1u;
    S2STAMP(11);
    __syncthreads();
    S2WALL(1, wg == 0 && tid == 0 && i + L < nb);
#ifdef BWGR_STAMPS
    if (tid == 0 && i + L < nb) wseen += (unsigned long long)wall_clock64();
#endif
    S2WALL(4, wg == 0 && tid == 0 && i == 100);
    S2MINMAX(58, 59, tid == 0 && i == 100);
    S2WALL(5, wg == 0 && tid == 0 && i == 101);
    if (ctl_s[8]) { if (tid == 0) a.sc->error = 1u; return; }
    // first request for delta_{i+1}, a whole iteration ahead of its use: a streamer that runs behind the sequencer (the ones
    // that set the pace) finds it there already, and the load's latency -- ~2 us on a CU whose memory queue also carries the
    // tile stream -- is off its iteration; the others ask again after the update (below)
    if (tid < SW_MAXM && i + 1 < nb) pre = (tid < blk_m(i + 1)) ? ld_agent_raw64(a.dgran + (size_t)((i + 1) % S2_NSLOT) * SW_MAXM + tid) : 0ull;
    const double invSd = pow2_field(851 + (int)ctl_s[par]);
    S2STAMP(1);
    S2STAMP(2);
    // ---- slab update with tile(i): out_s[row][n] = sum_markers x[row][marker] * digit_n(delta[marker]) ----
    // lane (m16, grp) of a wave pass owns the row quad r4 = 16 rg + m16; its k slots (dword u, byte q) of MFMA step s are the
    // markers 64 s + 16 u + 4 grp + q (this interleave keeps the four lane groups on different LDS banks)
    {
      const int8_t *tile = S2I_TILE(i);
      // tasks: (64-row group rg, half of the block's 64-marker steps); the two halves add into the same int32 entries
      const int nrg = R / 64, nhalf = 2;
      for (int task = wave; task < nhalf * nrg; task += SW_THREADS / 64) {
        const int half = task / nrg, rg = task - half * nrg;
        const int rowoff = 4 * (16 * rg + m16);
        s2_v4i acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        for (int s0 = 64 * half; s0 < mB; s0 += 64 * nhalf) {
          // c[u][q]: rows rowoff..+3 of marker s0 + 16u + 4grp + q.  For m % 64 != 0 the last step reads up to 48 markers
          // past the tile: still inside this workgroup's LDS (the arrays behind the tiles are larger), and those k slots
          // meet zero digits (delta digits are written for all SW_MAXM markers of every block).
          const int8_t *tp = tile + __mul24(s0 + 4 * grp, Rp) + rowoff;
          uint32_t c[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              c[u][q] = *reinterpret_cast<const uint32_t *>(tp + (16 * u + q) * Rp);
          const int8_t *bp = ddig_s + (size_t)m16 * S2_DP + s0 + 4 * grp;
          const s2_v4i bv = {*reinterpret_cast<const int *>(bp), *reinterpret_cast<const int *>(bp + 16),
                             *reinterpret_cast<const int *>(bp + 32), *reinterpret_cast<const int *>(bp + 48)};
          uint32_t rw[4][4];                                               // rw[k][u]: row k, bytes = the 4 markers of chunk u
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
        if (m16 < 8) {      // lane: digit n = m16; acc_k[reg] belongs to row 4 (16 rg + 4 grp + reg) + k
          int *op = out_s + (size_t)(4 * (16 * rg + 4 * grp)) * S2_OS + m16;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            atomicAdd(&op[(4 * reg + 0) * S2_OS], acc0[reg]); atomicAdd(&op[(4 * reg + 1) * S2_OS], acc1[reg]);
            atomicAdd(&op[(4 * reg + 2) * S2_OS], acc2[reg]); atomicAdd(&op[(4 * reg + 3) * S2_OS], acc3[reg]);
          }
        }
      }
    }
    S2STAMP(3);
    __syncthreads();
    S2STAMP(4);
    // tile(i) is spent: tile(i+L), in registers since the previous iteration, takes a free ring slot (for L = 3 the one of
    // tile(i)), and the loads of tile(i+L+1) go out at once: they must have landed before the next iteration's poll (the
    // vmcnt counter is in order, so a poll behind 32 KB of tile loads pays for them; issued after the publish they cost
    // 10 % of the sweep)
#ifdef BWGR_STAMPS
    unsigned long long w4t = 0; if (wg == 0 && tid == 256 && i == 100) w4t = wall_clock64();
#endif
    if (i + L < nb) S2I_TILE_COMMIT(S2I_TILE(i + L), blk_m(i + L));
#ifdef BWGR_STAMPS
    if (wg == 0 && tid == 256 && i == 100) wl2[8] += (wall_clock64() - w4t);
#endif
    // early request for delta_{i+1} (older than the tile loads below, so the in-order vmcnt lets it be consumed first)
    if (tid < SW_MAXM && i + 1 < nb && tid < blk_m(i + 1) && !s2_dgranule_is(pre, i + 1)) pre = ld_agent_raw64(a.dgran + (size_t)((i + 1) % S2_NSLOT) * SW_MAXM + tid);
    S2STAMP(9);
    S2STAMP(0);
    {
      uint32_t ex = 0u;
      for (int r = tid; r < R; r += SW_THREADS) {
        const int4 o0 = *reinterpret_cast<const int4 *>(out_s + (size_t)r * S2_OS);
        const int2 o1 = *reinterpret_cast<const int2 *>(out_s + (size_t)r * S2_OS + 4);
        double w = invSd, v = (double)o0.x * w;
        w *= 256.0; v = fma((double)o0.y, w, v); w *= 256.0; v = fma((double)o0.z, w, v); w *= 256.0; v = fma((double)o0.w, w, v);
        w *= 256.0; v = fma((double)o1.x, w, v); w *= 256.0; v = fma((double)o1.y, w, v);
        const double en = e_s[r] - v;
        e_s[r] = en;
        ex = max(ex, (uint32_t)((__double2hiint(en) & 0x7FFFFFFF) >> 20));
      }
      ex = wave_max_u32(ex);
      if (lane == 0) atomicMax(&ctl_s[2 + par], ex);
    }
    __syncthreads();
#ifdef BWGR_STAMPS
    if (wg == 0 && tid == 256 && i == 100) wl2[15] += (wall_clock64() - w4t);
#endif
    // the loads of tile(i+L+1) go out behind the barrier, under the e digits (waves 0-3)
    if (i + L + 1 < nb) S2I_TILE_ISSUE(blk_j0(i + L + 1), blk_m(i + L + 1));
#ifdef BWGR_STAMPS
    if (wg == 0 && tid == 256 && i == 100) wl2[8] += (wall_clock64() - w4t) << 32;
#endif
    S2STAMP(5);
    if (i + L < nb) {
      const double invSe = e_digits(par);
      S2STAMP(6);
      publish(i + L, S2I_TILE(i + L), invSe);
    } else {   // no publish to clear out_s behind it
      for (int c = tid; c < R * S2_OS / 4; c += SW_THREADS) reinterpret_cast<uint4 *>(out_s)[c] = make_uint4(0, 0, 0, 0);
    }
    S2STAMP(8);
  }
  if (wg == 0) S2STAMP_FLUSH(0);
  S2WALL_FLUSH;
#ifdef BWGR_STAMPS
  if (tid == 0 && a.stamps) { a.stamps[64 + wg] += wseen; a.stamps[128 + wg] += wq; }
#endif
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
  using GT = typename XTraits<XT>::GT;   // (the 16-bit staging variant needs less; the launch reserves for the wide one)
  const int pstride = ((m * (m - 1) / 2 + 7) / 8) * 8;
  size_t s = ((size_t)2 * ((pstride < 8 ? 8 : pstride) + 256) * sizeof(GT) + 15) & ~(size_t)15;   // 64 + 192 entries of slack per buffer
  s += ((size_t)m * m * sizeof(GT) + 15) & ~(size_t)15;
  s += 2 * sizeof(StageBuf) + 2 * sizeof(SpecBuf);
  s += 5 * SW_MAXM * sizeof(double);      // r0, carry2, delta, acc_corr[2]
  s += 2 * SW_MAXM * sizeof(float);       // bnew, dnew
  s += 2 * SW_MAXM * sizeof(int);         // acc_k[2]
  s += 4 * SW_MAXM * sizeof(double);      // part
  s += 64;
  return s;
}

// sum over a third of the streamers of q_b[t], fixed order, into dst[part][t]; every word is polled until it carries
// block b's tag.  Returns 0 on abort / timeout.
__device__ __forceinline__ int s2_gather_q(const SweepArgs &a, int b, int part, int t, int mB, double *dst) {
  const int K = a.K;
  uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
  const unsigned long long *slot = reinterpret_cast<const unsigned long long *>(a.qpart + (size_t)(b % S2_NSLOT) * K * SW_MAXM);
  const unsigned long long tag = s2_qtag(b);
  const int wq = (K + 2) / 3;
  const uint64_t t0 = wall_clock64();
  double r = 0.0;
  for (int wbase = 0; wbase < wq; wbase += 16) {
    unsigned long long v[16];
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int w = part * wq + wbase + u;
        const bool need = (t < mB && wbase + u < wq && w < K);
        v[u] = need ? ld_agent_raw64(slot + (size_t)w * SW_MAXM + t) : tag;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) ok = ok && ((v[u] & 0xFFull) == tag);
      if (ok) break;
      if ((++spins & 63u) == 0u) {
        if (ld_agent_u32(abortw) != 0u) return 0;
        if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int w = part * wq + wbase + u;
      const bool need = (t < mB && wbase + u < wq && w < K);
      r += need ? __longlong_as_double((long long)(v[u] & ~0xFFull)) : 0.0;
    }
  }
  dst[part * SW_MAXM + t] = r;
  return 1;
}

// GT: element type of the packed diagonal and the distance-1 cross Gram blocks as staged through LDS (the panel's Gram
// type, or uint16_t when every entry of the panel's Gram blocks fits: half the bytes through this CU per block);
// the distance-2 blocks, read sparsely from global memory, keep the panel's type
template <typename XT, bool SELECT, typename GT>
__device__ __forceinline__ void s2_sequencer(const SweepArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using G2T = typename XTraits<XT>::GT;
  constexpr int GPT = 16 / sizeof(GT);
  constexpr int MAXMX = XTraits<XT>::MAXM;
  constexpr int NHELP = SW_THREADS - 64;
  constexpr int XCH = (MAXMX * MAXMX / GPT + NHELP - 1) / NHELP;                    // chunks of Gx per helper thread
  constexpr int PFULLCH = ((MAXMX * (MAXMX - 1) / 2 + 7) / 8 * 8) / GPT, XFULLCH = MAXMX * MAXMX / GPT;   // chunk counts at m = MAXM
  constexpr int PCH = (PFULLCH + NHELP - 1) / NHELP;   // chunks of packed G per helper thread
  constexpr int PSURE = PFULLCH / NHELP, XSURE = XFULLCH / NHELP;   // at m = MAXM every helper thread's first PSURE / XSURE chunks exist
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = a.m;
  const int nb = a.blk_end - a.blk_begin;
  const int pstride = a.pstride;
  size_t off = 0;
  // each packed buffer has 64 entries of slack in front and 192 behind, so the recurrence's row loads need no clamp
  GT *gp_base = reinterpret_cast<GT *>(smem) + 64;
  const int gp_elems = (pstride < 8 ? 8 : pstride) + 256;
#define S2_GP(i_) (gp_base + (size_t)((i_) & 1) * gp_elems)
  off = ((size_t)2 * gp_elems * sizeof(GT) + 15) & ~(size_t)15;
  // 16-bit staging: Gx is double-buffered too (same LDS as one 32-bit copy) and both blocks arrive by LDS-DMA, see below
  constexpr bool DMA = (sizeof(GT) == 2);
  GT *gx_base = reinterpret_cast<GT *>(smem + off);
  const size_t gx_elems = (((size_t)m * m * sizeof(GT) + 15) & ~(size_t)15) / sizeof(GT);
  off += (DMA ? 2 : 1) * gx_elems * sizeof(GT);
#define S2_GX(i_) (gx_base + (DMA ? (size_t)((i_) & 1) * gx_elems : (size_t)0))
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 2 * sizeof(StageBuf);
  SpecBuf *specb = reinterpret_cast<SpecBuf *>(smem + off); off += 2 * sizeof(SpecBuf);
  double *r0_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  double *carry2_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);   // lag 3: Gx2_{b+1}' delta_{b-1}
  double *delta_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  // what the accepted markers of a block changed beyond the speculated step: lists by block parity (block b's list
  // serves r0_{b+1} right away and, with lag 3, r0_{b+2} one iteration later); counts in ctrl_s[2 + parity]
  double *acc_corr2 = reinterpret_cast<double *>(smem + off); off += 2 * SW_MAXM * sizeof(double);
  float *bnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  float *dnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  int *acc_k2 = reinterpret_cast<int *>(smem + off); off += 2 * SW_MAXM * sizeof(int);
  double *part_s = reinterpret_cast<double *>(smem + off); off += 4 * SW_MAXM * sizeof(double);
  // [0] ok flag, [2], [3] number of accepted markers (by block parity); not volatile: a volatile access stays a flat_ one and waits on vmcnt
  int *ctrl_s = reinterpret_cast<int *>(smem + off);
  const GT *gramp = reinterpret_cast<const GT *>(a.gramp);
  const GT *gramx = reinterpret_cast<const GT *>(a.gramx);
  const float Cc = a.sc->C, odds = a.sc->odds, one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;
  const double dscale = (a.flags & SWF_DELTA2) ? 2.0 : 1.0;   // emBA's doubled residual update (affine path)
  const int emflags = SELECT ? 0 : (a.flags & SWF_EM_ANY);     // the EM family's non-affine updates (lane_em)
  const float L1 = a.sc->lam;
  const int pchunks = pstride / GPT, xchunks = m * m / GPT;
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };
  auto prow = [&](int k) { return k * (m - 1) - k * (k - 1) / 2; };   // offset of packed row k

  // q_b summed over the streamers into part_s[0..2][t] (added up by the reader): from the feeders' tagged sums (one word
  // per marker, polled by the threads of part 0) or, without feeders, gathered here from the streamers' words
  auto gather_q = [&](int b, int part, int t, int mB) -> int {
    if (!SELECT || a.nfeed <= 0) return s2_gather_q(a, b, part, t, mB, part_s);
    double v = 0.0;
    if (part == 0 && t < mB) {
      const unsigned long long *g = reinterpret_cast<const unsigned long long *>(a.qpart + (size_t)S2_NSLOT * a.K * SW_MAXM + (size_t)(b % S2_NSLOT) * SW_MAXM + t);
      const unsigned long long tag = s2_qtag(b);
      uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
      const uint64_t t0 = wall_clock64();
      unsigned spins = 0;
      for (;;) {
        const unsigned long long w = ld_agent_raw64(g);
        if ((w & 0xFFull) == tag) { v = __longlong_as_double((long long)(w & ~0xFFull)); break; }
        if ((++spins & 63u) == 0u) {
          if (ld_agent_u32(abortw) != 0u) return 0;
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    part_s[part * SW_MAXM + t] = v;
    return 1;
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
    if (tid == 0) { ctrl_s[0] = 1; ctrl_s[2] = 0; ctrl_s[3] = 0; }
    if (tid < SW_MAXM) carry2_s[tid] = 0.0;
    __syncthreads();
    if (wave >= 1 && wave <= 6) {
      if (!gather_q(0, (tid - 64) >> 7, (tid - 64) & 127, mB)) ctrl_s[0] = 0;
    }
    __syncthreads();
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }
    if (tid < mB) r0_s[tid] = (part_s[tid] + part_s[SW_MAXM + tid]) + part_s[2 * SW_MAXM + tid];
  }
  double sum_d = 0.0, sum_b2 = 0.0;
  // the selection rounds' centres and radii (k_spec's QuickBuf) of the next block: wave 0 reads them itself, a block ahead
  double nzc[2] = {0.0, 0.0}, nha[2] = {INFINITY, INFINITY}, nhr[2] = {-1.0, -1.0};
  const QuickBuf *quick = (SELECT && a.ps.quick) ? a.ps.quick + a.blk_begin : nullptr;
  if (SELECT && wave == 0 && quick) {
#pragma unroll
    for (int g = 0; g < 2; ++g) { nzc[g] = quick[0].zc[64 * g + lane]; nha[g] = quick[0].ha[64 * g + lane]; nhr[g] = quick[0].hr[64 * g + lane]; }
  }
  // implicitly centred columns (SWF_CENTRE, selection sweeps of int8 panels; sweep3.hip.h "implicitly centred sweeps"): the lanes' column sums of the
  // next block, read a block ahead, and wave 0's running scalar
  const bool cen = SELECT && sizeof(XT) == 1 && (a.flags & SWF_CENTRE) != 0;
  const double ninv = a.ninv, cen_u0 = cen ? a.sc->cen_u0 : 0.0;
  double cenU = cen_u0;
  auto cs_load = [&](int b, int q) -> double { const int jj = (a.blk_begin + b) * m + 64 * q + lane; return (cen && 64 * q + lane < min(m, a.p - (a.blk_begin + b) * m)) ? (double)a.csum[min(jj, a.p - 1)] : 0.0; };
  double ncs[2] = {0.0, 0.0};
  if (cen && wave == 0) { ncs[0] = cs_load(0, 0); ncs[1] = cs_load(0, 1); }
  // helper threads' prefetch registers (plain named locals: an aggregate would end up in scratch memory)
  static_assert(PCH <= 5 && XCH <= 10, "named prefetch registers cover 5 + 10 chunks per helper thread");
  constexpr int NCH = (int)(sizeof(StageBuf) / 16), NSP = (int)(sizeof(SpecBuf) / 16);
  static_assert(NCH <= NHELP && NSP <= NHELP, "one chunk per helper thread");
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  uint4 gq0 = z4, gq1 = z4, gq2 = z4, gq3 = z4, gq4 = z4;
  uint4 xq0 = z4, xq1 = z4, xq2 = z4, xq3 = z4, xq4 = z4, xq5 = z4, xq6 = z4, xq7 = z4, xq8 = z4, xq9 = z4;
  uint4 spre = z4, cpre = z4;
#define S2_G_EACH(X) X(0, gq0) X(1, gq1) X(2, gq2) X(3, gq3) X(4, gq4)
#define S2_X_EACH(X) X(0, xq0) X(1, xq1) X(2, xq2) X(3, xq3) X(4, xq4) X(5, xq5) X(6, xq6) X(7, xq7) X(8, xq8) X(9, xq9)
#define S2_GLD(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < pchunks) name = gsrc[c_]; }
#define S2_XLD(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < xchunks) name = xsrc[c_]; }
#define S2_GST(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < pchunks) gdst[c_] = name; }
#define S2_XST(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < xchunks) xdst[c_] = name; }
  // at m == MAXM every helper thread's first PSURE / XSURE chunks exist and only the last one needs a guard (unguarded
  // accesses spare the exec-mask juggling of up to 15 conditional loads and stores)
#define S2_GLD_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < PSURE || c_ < pchunks) name = gsrc[c_]; }
#define S2_XLD_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < XSURE || c_ < xchunks) name = xsrc[c_]; }
#define S2_GST_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < PSURE || c_ < pchunks) gdst[c_] = name; }
#define S2_XST_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < XSURE || c_ < xchunks) xdst[c_] = name; }
  const bool fullm = (m == MAXMX);
  if (wave >= 1 && nb > 1) {   // block 1
    const uint4 *gsrc = reinterpret_cast<const uint4 *>(gramp + (size_t)(a.blk_begin + 1) * pstride);
    const uint4 *xsrc = reinterpret_cast<const uint4 *>(gramx + (size_t)(a.blk_begin + 1) * m * m);
    if constexpr (!DMA) { if (fullm) { S2_G_EACH(S2_GLD_F) S2_X_EACH(S2_XLD_F) } else { S2_G_EACH(S2_GLD) S2_X_EACH(S2_XLD) } }
    if (tid - 64 < NCH) spre = reinterpret_cast<const uint4 *>(a.ps.blocks + a.blk_begin + 1)[tid - 64];
    if (tid - 64 < NSP) cpre = reinterpret_cast<const uint4 *>(a.ps.spec + a.blk_begin + 1)[tid - 64];
  }
  S2STAMP_DECL;
  S2WALL_DECL;

  for (int b = 0; b < nb; ++b) {
    const int blk = a.blk_begin + b;
    const int j0 = blk * m;
    const int mB = blk_m(b);
    const bool have_next = (b + 1 < nb);
    const int mBn = have_next ? blk_m(b + 1) : 0;
    const StageBuf &st = stage[b & 1];
    const SpecBuf &sb = specb[b & 1];
    const GT *gp = S2_GP(b);
    int *acc_k = acc_k2 + (b & 1) * SW_MAXM;
    double *acc_corr = acc_corr2 + (b & 1) * SW_MAXM;
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
        if constexpr (SELECT && sizeof(XT) == 1) { if (cen) r[q] = fma(ncs[q], cenU, r[q]); }
        lc[q].b0 = live ? st.b0[t] : 0.0f; lc[q].xxb0 = live ? st.xxb0[t] : 0.0f;
        lc[q].b2 = live ? st.b2[t] : 0.0f; lc[q].drej = live ? st.drej[t] : 0.0f;
        lc[q].rden = live ? st.rden[t] : 1.0; lc[q].sdz1 = live ? st.sdz1[t] : 0.0;
        lc[q].gjj = live ? sb.gjj[t] : 0.0; lc[q].mk = a.marker0 + (uint32_t)(j0 + t);
        lc[q].tacc = live ? st.tacc[t] : -INFINITY; lc[q].trej = live ? st.trej[t] : -INFINITY;   // dead lanes: certain reject
      }
      S2STAMP(5);
      // affine models: the recurrence's variable, the un-rounded draw t = (r + xx*b0)*rden + sd*z
      double u[2] = {fma(r[0] + (double)lc[0].xxb0, lc[0].rden, lc[0].sdz1), fma(r[1] + (double)lc[1].xxb0, lc[1].rden, lc[1].sdz1)};
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
            const int oA = prow(base) - 1;   // + lane: own-group entry of packed row `base`; the other group's is 64 further
            if (emflags) {
              // the EM family's non-affine updates (lane_em): the same lane-ordered recurrence on r, four markers per trip, Gram rows
              // read four markers ahead, loops specialised for one / two lane groups (as the affine loop below)
              const bool two = (q == 0 && ngrp > 1);
              double ra = r[q], rb = two ? r[1] : 0.0;
              int oL = oA;
              GT ga0, ga1, ga2, ga3, gb0 = (GT)0, gb1 = (GT)0, gb2 = (GT)0, gb3 = (GT)0;
#define S2_EM_STEP(TWO_, MODE_, l_, GA, GB) { \
                const GT g0 = (lane > (l_)) ? GA : (GT)0, g1 = GB; \
                GA = gp[oL + lane]; if (TWO_) GB = gp[oL + 64 + lane]; oL += m - 2 - (base + (l_) + 4); \
                float dtmp; \
                const float dl = lane_em(ra, lc[q], (MODE_), Cc, odds, L1, &dtmp) - lc[q].b0; \
                const double dd = (double)readlane_f32(dl, (l_)); \
                ra = fma(-(double)g0, dd, ra); \
                if (TWO_) rb = fma(-(double)g1, dd, rb); }
#define S2_EM_LOOP(TWO_, MODE_) { \
                ga0 = gp[oL + lane]; if (TWO_) gb0 = gp[oL + 64 + lane]; oL += m - 2 - (base + 0); \
                ga1 = gp[oL + lane]; if (TWO_) gb1 = gp[oL + 64 + lane]; oL += m - 2 - (base + 1); \
                ga2 = gp[oL + lane]; if (TWO_) gb2 = gp[oL + 64 + lane]; oL += m - 2 - (base + 2); \
                ga3 = gp[oL + lane]; if (TWO_) gb3 = gp[oL + 64 + lane]; oL += m - 2 - (base + 3); \
                int l = 0; \
                for (; l + 4 <= cnt; l += 4) { S2_EM_STEP(TWO_, MODE_, l, ga0, gb0) S2_EM_STEP(TWO_, MODE_, l + 1, ga1, gb1) S2_EM_STEP(TWO_, MODE_, l + 2, ga2, gb2) S2_EM_STEP(TWO_, MODE_, l + 3, ga3, gb3) } \
                for (; l < cnt; ++l) { \
                  S2_EM_STEP(TWO_, MODE_, l, ga0, gb0) \
                  const GT ta = ga0, tb = gb0; \
                  ga0 = ga1; ga1 = ga2; ga2 = ga3; ga3 = ta; gb0 = gb1; gb1 = gb2; gb2 = gb3; gb3 = tb; \
                } }
              // one copy of the loop per member class, so that lane_em's mode tests fold away
#define S2_EM_BOTH(MODE_) { if (two) S2_EM_LOOP(1, MODE_) else S2_EM_LOOP(0, MODE_) }
              if (emflags & SWF_EM_SEL) S2_EM_BOTH(SWF_EM_SEL)
              else if (emflags & SWF_EM_EN) S2_EM_BOTH(SWF_EM_EN)
              else if (emflags & SWF_EM_BL) S2_EM_BOTH(SWF_EM_BL)
              else S2_EM_BOTH(SWF_EM_LASSO)
#undef S2_EM_BOTH
#undef S2_EM_LOOP
#undef S2_EM_STEP
              r[q] = ra; if (two) r[1] = rb;
            } else {
              // Affine models.  The recurrence runs on t = (r + xx*b0)*rden + sd*z, the un-rounded draw itself: a step's update
              // r -= g*delta becomes t -= (g*rden)*delta with g*rden formed off the chain, which leaves five dependent
              // operations per marker (fma, cvt, sub, readlane, cvt) instead of seven; eight markers per trip (with one per trip its speed followed the loop's placement in the
              // instruction-fetch lines), and the packed Gram rows are read FOUR markers ahead: with the next row requested only
              // one step ahead, every step waited out the LDS latency (~100 cycles) instead of its own ~50-cycle chain.
              // Row k's offsets advance by m - 2 - k; the buffers' slack makes the reads past the last row harmless.
              const bool two = (q == 0 && ngrp > 1);
              double ua = u[q], ub = two ? u[1] : 0.0;
              const double sclA = dscale * lc[q].rden, sclB = two ? dscale * lc[1].rden : 0.0; const float b0q = lc[q].b0;
              int oL = oA;   // offset (+ lane) of the next row to request
              GT ga0, ga1, ga2, ga3, gb0 = (GT)0, gb1 = (GT)0, gb2 = (GT)0, gb3 = (GT)0;
#define S2_AFFINE_STEP(TWO_, l_, GA, GB) { \
                const GT g0 = (lane > (l_)) ? GA : (GT)0, g1 = GB; \
                GA = gp[oL + lane]; if (TWO_) GB = gp[oL + 64 + lane]; oL += m - 2 - (base + (l_) + 4);   /* row l + 4 */ \
                const float dl = (float)ua - b0q; \
                const double dd = (double)readlane_f32(dl, (l_)); \
                ua = fma(-(double)g0 * sclA, dd, ua); \
                if (TWO_) ub = fma(-(double)g1 * sclB, dd, ub); }
#define S2_AFFINE_LOOP(TWO_) { \
                ga0 = gp[oL + lane]; if (TWO_) gb0 = gp[oL + 64 + lane]; oL += m - 2 - (base + 0); \
                ga1 = gp[oL + lane]; if (TWO_) gb1 = gp[oL + 64 + lane]; oL += m - 2 - (base + 1); \
                ga2 = gp[oL + lane]; if (TWO_) gb2 = gp[oL + 64 + lane]; oL += m - 2 - (base + 2); \
                ga3 = gp[oL + lane]; if (TWO_) gb3 = gp[oL + 64 + lane]; oL += m - 2 - (base + 3); \
                int l = 0; \
                for (; l + 8 <= cnt; l += 8) { \
                  S2_AFFINE_STEP(TWO_, l, ga0, gb0) S2_AFFINE_STEP(TWO_, l + 1, ga1, gb1) S2_AFFINE_STEP(TWO_, l + 2, ga2, gb2) S2_AFFINE_STEP(TWO_, l + 3, ga3, gb3) \
                  S2_AFFINE_STEP(TWO_, l + 4, ga0, gb0) S2_AFFINE_STEP(TWO_, l + 5, ga1, gb1) S2_AFFINE_STEP(TWO_, l + 6, ga2, gb2) S2_AFFINE_STEP(TWO_, l + 7, ga3, gb3) \
                } \
                for (; l < cnt; ++l) {   /* the last, partial block: rotate the four rows through ga0 / gb0 */ \
                  S2_AFFINE_STEP(TWO_, l, ga0, gb0) \
                  const GT ta = ga0, tb = gb0; \
                  ga0 = ga1; ga1 = ga2; ga2 = ga3; ga3 = ta; gb0 = gb1; gb1 = gb2; gb2 = gb3; gb3 = tb; \
                } }
              if (two) S2_AFFINE_LOOP(1) else S2_AFFINE_LOOP(0)
#undef S2_AFFINE_LOOP
#undef S2_AFFINE_STEP
              u[q] = ua; if (two) u[1] = ub;   // (group 1 continues from here)
            }
          } else if (q == 0) {
            // exact speculative rounds, both lane groups, decided by lane_quick's two compares (sweep.hip.h)
            double zc[2] = {nzc[0], nzc[1]}, ha[2] = {nha[0], nha[1]}, hr[2] = {nhr[0], nhr[1]};
            if (quick) {
              if (have_next) {
#pragma unroll
                for (int g = 0; g < 2; ++g) { nzc[g] = quick[b + 1].zc[64 * g + lane]; nha[g] = quick[b + 1].ha[64 * g + lane]; nhr[g] = quick[b + 1].hr[64 * g + lane]; }
              }
            } else {
#pragma unroll
              for (int g = 0; g < 2; ++g) lane_quick(lc[g], a.flags, Cc, zc[g], ha[g], hr[g]);
            }
            bool done_cen = false;
            if constexpr (SELECT && sizeof(XT) == 1) {
              if (cen) {
                const double cs[2] = {ncs[0], ncs[1]};
                if (have_next) { ncs[0] = cs_load(b + 1, 0); ncs[1] = cs_load(b + 1, 1); }
                quick_rounds<GT, true>(r, lc, zc, ha, hr, accmask, gp, m, mB, lane, a.flags, Cc, odds, one_minus_pi, a.rng, a.iter, cs, ninv, &cenU);
                done_cen = true;
              }
            }
            if (!done_cen) quick_rounds(r, lc, zc, ha, hr, accmask, gp, m, mB, lane, a.flags, Cc, odds, one_minus_pi, a.rng, a.iter);
          }
        }
      }
      S2STAMP(6);
      // outputs; delta_b goes out at once as {epoch, float} granules (one 8-byte write-through store per marker)
      unsigned long long *gslot = a.dgran + (size_t)(b % S2_NSLOT) * SW_MAXM;
      const int nacc0 = __popcll(accmask[0]);
      float dl_own[2] = {0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        if (t < mB) {
          float dem = 1.0f;
          const float b1 = SELECT ? lane_b1(r[q], lc[q])
                         : emflags ? lane_em(r[q], lc[q], emflags, Cc, odds, L1, &dem)
                                   : (float)u[q];   // exactly the value the recurrence used
          const bool inc = SELECT ? (((accmask[q] >> lane) & 1ull) != 0ull) : true;
          const float bn = inc ? b1 : lc[q].b2;
          const float dn = inc ? dem : 0.0f;
          const float dl = (bn - lc[q].b0) * (float)dscale;
          dl_own[q] = dl;
          delta_s[t] = (double)dl; bnew_s[t] = bn; dnew_s[t] = dn;
          if (SELECT && inc) {   // what this marker changed relative to the speculated step
            const int idx = (q ? nacc0 : 0) + __popcll(accmask[q] & ((1ull << lane) - 1ull));
            acc_k[idx] = t; acc_corr[idx] = (double)(b1 - lc[q].b0) - (double)lc[q].drej;
          }
          sum_d += (double)dn;
          sum_b2 = fma((double)bn, (double)bn, sum_b2);
        }
      }
      {   // the granules: every delta of the block is known here, so is their largest exponent
        const uint32_t exmax = wave_max_u32(max((__float_as_uint(dl_own[0]) >> 23) & 0xFFu, (__float_as_uint(dl_own[1]) >> 23) & 0xFFu));
        if (lane < mB) st_agent_raw64(gslot + lane, s2_dgranule(b, exmax, dl_own[0]));
        if (64 + lane < mB) st_agent_raw64(gslot + 64 + lane, s2_dgranule(b, exmax, dl_own[1]));
      }
      if (SELECT && lane == 0) ctrl_s[2 + (b & 1)] = nacc0 + __popcll(accmask[1]);
      S2WALL(0, lane == 0 && b + 3 < nb);
      S2STAMP(1);
    } else if (have_next) {
      const int gpart = (tid - 64) >> 7, gt = (tid - 64) & 127;
      auto carry2_term = [&]() {
        // wave 7: r0_{b+1}'s cross term with block b-1 = speculated part (k_spec) + the rows of Gx2_{b+1} that block
        // b-1's accepted markers touch, straight from global memory (the list has been final since the last barrier A)
        const int *pk = acc_k2 + ((b + 1) & 1) * SW_MAXM;
        const double *pc = acc_corr2 + ((b + 1) & 1) * SW_MAXM;
        const int npre = (b >= 1) ? ctrl_s[2 + ((b + 1) & 1)] : 0;
        const G2T *gx2 = reinterpret_cast<const G2T *>(a.gramx2) + (size_t)(blk + 1) * m * m;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = lane + 64 * h;
          double c = 0.0;
          if (b >= 1 && j < mBn) {
            c = a.xspec2[(size_t)(blk + 1) * SW_MAXM + j];
            for (int i0 = 0; i0 < npre; i0 += 8) {
              G2T gv[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) gv[u] = (i0 + u < npre) ? gx2[(size_t)pk[i0 + u] * m + j] : (G2T)0;
#pragma unroll
              for (int u = 0; u < 8; ++u) if (i0 + u < npre) c = fma((double)gv[u], pc[i0 + u], c);
            }
          }
          carry2_s[j] = c;
        }
      };
      if constexpr (DMA) {
        // ---- helpers, 16-bit staging.  Block b+1's packed G and Gx go straight from global memory into LDS by LDS-DMA (no
        // VGPRs, no store pass; one 1 KiB piece per instruction, ~7 per helper wave: one wave alone sustains only ~25 GB/s).
        // The pieces land under the recurrence and are drained by the vmcnt(0) that __syncthreads() puts in front of
        // barrier A; both destination buffers were last read two blocks ago.  Then waves 1-6 gather q_{b+1}, store block
        // b+1's constants (in registers since the last iteration) and request block b+2's; wave 7 forms the lag-3 cross term. ----
        S2WALL(4, tid == 64);
        {   // the 1 KiB pieces of block b+1's packed G (first) and Gx, dealt round-robin to the 7 helper waves
          const uint4 *gsrc = reinterpret_cast<const uint4 *>(gramp + (size_t)(blk + 1) * pstride);
          const uint4 *xsrc = reinterpret_cast<const uint4 *>(gramx + (size_t)(blk + 1) * m * m);
          unsigned char *gl = reinterpret_cast<unsigned char *>(S2_GP(b + 1)), *xl = reinterpret_cast<unsigned char *>(S2_GX(b + 1));
          const int gpieces = (pchunks + 63) >> 6, xpieces = (xchunks + 63) >> 6;
          for (int pc = wave - 1; pc < gpieces + xpieces; pc += 7) {
            const bool isg = pc < gpieces;
            const int c0 = (isg ? pc : pc - gpieces) << 6;
            const uint4 *src = (isg ? gsrc : xsrc) + c0 + lane;
            unsigned char *dst = (isg ? gl : xl) + (size_t)c0 * 16;
            if (c0 + lane < (isg ? pchunks : xchunks))
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
          }
        }
        S2WALL(5, tid == 64);
        if (wave <= 6) {
          if (!gather_q(b + 1, gpart, gt, mBn)) ctrl_s[0] = 0;
          S2WALL(6, tid == 64);
          if (tid - 64 < NCH) reinterpret_cast<uint4 *>(&stage[(b + 1) & 1])[tid - 64] = spre;
          if (tid - 64 < NSP) reinterpret_cast<uint4 *>(&specb[(b + 1) & 1])[tid - 64] = cpre;
          if (b + 2 < nb) {
            if (tid - 64 < NCH) spre = reinterpret_cast<const uint4 *>(a.ps.blocks + blk + 2)[tid - 64];
            if (tid - 64 < NSP) cpre = reinterpret_cast<const uint4 *>(a.ps.spec + blk + 2)[tid - 64];
          }
          S2WALL(7, tid == 64);
        } else if (SELECT && a.lag == 3) carry2_term();
      } else {
        // ---- helpers, register staging.  Block b+1's Gram blocks and constants were loaded into registers one iteration
        // ago.  Order: (1) the registers go to LDS, (2) the loads of block b+2 are issued (they have a whole iteration to
        // land), (3) q_{b+1} is gathered: its round trip runs under the prefetch traffic instead of in front of it ----
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no prefetch load is in flight
        uint4 *gdst = reinterpret_cast<uint4 *>(S2_GP(b + 1));
        uint4 *xdst = reinterpret_cast<uint4 *>(S2_GX(b + 1));
        if (fullm) { S2_G_EACH(S2_GST_F) S2_X_EACH(S2_XST_F) } else { S2_G_EACH(S2_GST) S2_X_EACH(S2_XST) }
        if (tid - 64 < NCH) reinterpret_cast<uint4 *>(&stage[(b + 1) & 1])[tid - 64] = spre;
        if (tid - 64 < NSP) reinterpret_cast<uint4 *>(&specb[(b + 1) & 1])[tid - 64] = cpre;
        if (b + 2 < nb) {
          const uint4 *gsrc = reinterpret_cast<const uint4 *>(gramp + (size_t)(blk + 2) * pstride);
          const uint4 *xsrc = reinterpret_cast<const uint4 *>(gramx + (size_t)(blk + 2) * m * m);
          if (fullm) { S2_G_EACH(S2_GLD_F) S2_X_EACH(S2_XLD_F) } else { S2_G_EACH(S2_GLD) S2_X_EACH(S2_XLD) }
          if (tid - 64 < NCH) spre = reinterpret_cast<const uint4 *>(a.ps.blocks + blk + 2)[tid - 64];
          if (tid - 64 < NSP) cpre = reinterpret_cast<const uint4 *>(a.ps.spec + blk + 2)[tid - 64];
        }
        if (wave <= 6) {
          if (!gather_q(b + 1, gpart, gt, mBn)) ctrl_s[0] = 0;
        } else if (SELECT && a.lag == 3) carry2_term();
      }
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
          const int nacc = ctrl_s[2 + (b & 1)];
          const GT *gxn = S2_GX(b + 1);
          for (int i = 0; i < nacc; ++i) r = fma(-(double)gxn[(size_t)acc_k[i] * m + tid], acc_corr[i], r);
          if (a.lag == 3) r -= carry2_s[tid];
          r0_s[tid] = r;
        }
        S2STAMP(3);
      } else {
        // r0_{b+1} = sum_w q - Gx' delta_b  (dense; 4 k-ranges per marker, LDS reads batched)
        const int sp = tid >> 7, j = tid & 127;
        const GT *gxn = S2_GX(b + 1);
        double xa = 0.0;
        if (j < mBn) {
          const int k0 = sp * 32, kx1 = min(k0 + 32, mB);
          int k = k0;
          for (; k + 8 <= kx1; k += 8) {
            double gv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { gv[u] = (double)gxn[(size_t)(k + u) * m + j]; dv[u] = delta_s[k + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) xa = fma(gv[u], dv[u], xa);
          }
          for (; k < kx1; ++k) xa = fma((double)gxn[(size_t)k * m + j], delta_s[k], xa);
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
  S2STAMP_FLUSH(16);
  S2WALL_FLUSH;
  if (wave == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
    if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; if (cen) a.sc->cen_c = cenU - cen_u0; }
  }
}



// q feeders: a.nfeed workgroups of their own, hence their own CUs' memory paths.  Gathers the K
// streamers' slab dots of each block (the 40 KB of write-through words per block that otherwise go through the
// sequencer's CU, whose ingest is what bounds the chain), sums them in the fixed order and hands the sequencer one
// tagged word per marker.  Costs one more hop on a path that the lag-3 pipeline keeps off the critical cycle.
__device__ __forceinline__ void s2_feeder(const SweepArgs &a, int f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int m = a.m, nb = a.blk_end - a.blk_begin;
  double *part_s = reinterpret_cast<double *>(smem);
  int *ok_s = reinterpret_cast<int *>(smem + 3 * SW_MAXM * sizeof(double));
  double *qsum = a.qpart + (size_t)S2_NSLOT * a.K * SW_MAXM;
  S2WALL_DECL;
  if (a.nfeed <= 0 || f >= a.nfeed) return;
  if (tid == 0) ok_s[0] = 1;
  __syncthreads();
  for (int b = f; b < nb; b += a.nfeed) {   // one gather + sum takes about a block period: the feeders take turns
    const int mB = min(m, a.p - (a.blk_begin + b) * m);
    if (wave >= 1 && wave <= 6) {
      if (!s2_gather_q(a, b, (tid - 64) >> 7, (tid - 64) & 127, mB, part_s)) ok_s[0] = 0;
    }
    __syncthreads();
    if (ok_s[0] == 0) return;   // the abort word is set; the sequencer reports the error
    if (tid < mB) s2_put_q(qsum + (size_t)(b % S2_NSLOT) * SW_MAXM + tid, (part_s[tid] + part_s[SW_MAXM + tid]) + part_s[2 * SW_MAXM + tid], b);
    S2WALL(3, tid == 0 && b >= a.lag);
    S2WALL(13, tid == 0 && b == 100 + a.lag);
    __syncthreads();
  }
  S2WALL_FLUSH;
}

// ------------------------------------------------------------------------------------------------------------------
// sequencer, selection models on an int8 panel whose Gram blocks fit 16 bits: ONE barrier per block.
// Wave 0 owns the chain and issues no global memory operation but the delta granules: recurrence of block b, then (after
// the barrier) r0_{b+1} for its own lanes, then straight into block b+1.  Waves 1-7 run one block ahead: while wave 0 is
// in block b they bring in block b+1 (the feeder's q sums, packed G, Gx and the constants through registers, the lag-3
// cross term) and write out block b-1's state; every LDS object they fill is double-buffered by block parity, so they
// start on block b+2 the moment barrier b has passed.
// Period = max(recurrence + post, helpers); the generic sequencer pays max(recurrence, helpers) + post + a barrier.
// ------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t s2_seq16_lds_bytes(int m) {
  const int pstride = ((m * (m - 1) / 2 + 7) / 8) * 8;
  size_t s = (size_t)2 * ((pstride < 8 ? 8 : pstride) + 256) * 2;     // packed G, two buffers with 64 + 192 entries of slack
  s = (s + 15) & ~(size_t)15;
  s += (size_t)2 * m * m * 2;                                          // Gx, two buffers
  s += 2 * sizeof(StageBuf) + 2 * sizeof(SpecBuf);
  s += (size_t)2 * SW_MAXM * sizeof(double);                           // sum_w q [parity]
  s += (size_t)2 * 3 * SW_MAXM * sizeof(float);                        // state of a block [parity]
  s += (size_t)2 * SW_MAXM * sizeof(double);                           // lag-3 cross term [parity]
  s += (size_t)4 * SW_MAXM * (sizeof(double) + sizeof(int));           // accepted lists, ring of four blocks
  s += (size_t)8 * SW_MAXM * sizeof(double);                           // dense blocks: per-wave partial sums
  return s + 64;
}
__device__ __forceinline__ void s2_sequencer_sel16(const SweepArgs &a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using GT = uint16_t;
  constexpr int GPT = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  S2WALL_DECL;
  const int m = a.m, nb = a.blk_end - a.blk_begin, pstride = a.pstride;
  const int gp_elems = (pstride < 8 ? 8 : pstride) + 256;
  GT *gp_base = reinterpret_cast<GT *>(smem) + 64;
  size_t off = ((size_t)2 * gp_elems * sizeof(GT) + 15) & ~(size_t)15;
  GT *gx_base = reinterpret_cast<GT *>(smem + off); off += (size_t)2 * m * m * sizeof(GT);
  StageBuf *stage = reinterpret_cast<StageBuf *>(smem + off); off += 2 * sizeof(StageBuf);
  SpecBuf *specb = reinterpret_cast<SpecBuf *>(smem + off); off += 2 * sizeof(SpecBuf);
  double *qs2 = reinterpret_cast<double *>(smem + off); off += (size_t)2 * SW_MAXM * sizeof(double);       // sum_w q [parity]
  float *state2 = reinterpret_cast<float *>(smem + off); off += (size_t)2 * 3 * SW_MAXM * sizeof(float);   // b, d, vb of a block [parity]
  double *carry2 = reinterpret_cast<double *>(smem + off); off += (size_t)2 * SW_MAXM * sizeof(double);
  // what the accepted markers of a block changed beyond the speculated step: ring of four blocks (block b's list serves
  // r0_{b+1} from wave 0's registers and, through these lists, the cross terms of blocks b+2 and b+3)
  double *acc_corr2 = reinterpret_cast<double *>(smem + off); off += (size_t)4 * SW_MAXM * sizeof(double);
  int *acc_k2 = reinterpret_cast<int *>(smem + off); off += (size_t)4 * SW_MAXM * sizeof(int);
  double *dpost = reinterpret_cast<double *>(smem + off); off += (size_t)8 * SW_MAXM * sizeof(double);   // dense blocks: per-wave partial sums of the distance-1 term
  int *ctrl_s = reinterpret_cast<int *>(smem + off);   // [0] ok flag, [4 + (b & 3)] number of accepted markers of block b
#define Q16_GP(i_) (gp_base + (size_t)((i_) & 1) * gp_elems)
#define Q16_GX(i_) (gx_base + (size_t)((i_) & 1) * m * m)
#define Q16_QS(i_) (qs2 + (size_t)((i_) & 1) * SW_MAXM)
#define Q16_STATE(i_) (state2 + (size_t)((i_) & 1) * 3 * SW_MAXM)
  const GT *gramp = reinterpret_cast<const GT *>(a.gramp);
  const GT *gramx = reinterpret_cast<const GT *>(a.gramx);
  const float Cc = a.sc->C, odds = a.sc->odds, one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;
  const int pchunks = pstride / GPT, xchunks = m * m / GPT;
  const int L = a.lag;
  auto blk_m = [&](int b) { return min(m, a.p - (a.blk_begin + b) * m); };
  constexpr int NCH = (int)(sizeof(StageBuf) / 16), NSP = (int)(sizeof(SpecBuf) / 16);
  static_assert(NCH <= 2 * 320 && NSP <= 320, "at most two StageBuf chunks and one SpecBuf chunk per thread of waves 1-5");

  double *qsum = a.qpart + (size_t)S2_NSLOT * a.K * SW_MAXM;   // the feeder's sums, [S2_NSLOT][SW_MAXM] tagged words
  uint32_t *abortw = a.xflags + (size_t)a.K * SW_FLAG_STRIDE;
  auto poll_qsum = [&](int c, int mBc) -> int {   // one wave, two markers per lane
    const unsigned long long *g = reinterpret_cast<const unsigned long long *>(qsum + (size_t)(c % S2_NSLOT) * SW_MAXM);
    const unsigned long long tag = s2_qtag(c);
    const bool n0 = lane < mBc, n1 = 64 + lane < mBc;
    const uint64_t t0 = wall_clock64();
    unsigned spins = 0;
    unsigned long long w0 = tag, w1 = tag;
    for (;;) {
      if (n0) w0 = ld_agent_raw64(g + lane);
      if (n1) w1 = ld_agent_raw64(g + 64 + lane);
      if ((w0 & 0xFFull) == tag && (w1 & 0xFFull) == tag) break;
      if ((++spins & 63u) == 0u) {
        if (ld_agent_u32(abortw) != 0u) return 0;
        if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) { st_agent_u32(abortw, 1u); return 0; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    Q16_QS(c)[lane] = n0 ? __longlong_as_double((long long)(w0 & ~0xFFull)) : 0.0;
    Q16_QS(c)[64 + lane] = n1 ? __longlong_as_double((long long)(w1 & ~0xFFull)) : 0.0;
    return 1;
  };
  auto store_state = [&](int c) {   // one wave
    const int j0c = (a.blk_begin + c) * m, mBc = blk_m(c);
    const float *sp = Q16_STATE(c);
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
  // the helpers' work for block c (c >= 1): everything block c needs that does not depend on block c-1's recurrence
  uint4 spre = make_uint4(0, 0, 0, 0), spre2 = spre, cpre = spre;   // block c's constants, requested one phase earlier
  // block c's Gram chunks, requested one phase earlier (plain named locals: an aggregate would end up in scratch memory)
  // one kind of memory traffic per helper wave (the memory counter returns in order, so whatever a wave has in flight is in
  // front of its next result): waves 1-5 stage Gram blocks and constants (prefetch, a whole period to land), wave 6 reads
  // the sparse cross-term rows, wave 7 polls the feeders' sums
  constexpr int NHELP = SW_THREADS - 192;
  constexpr int PFULLCH = ((SW_MAXM * (SW_MAXM - 1) / 2 + 7) / 8 * 8) / GPT, XFULLCH = SW_MAXM * SW_MAXM / GPT;
  constexpr int PSURE = PFULLCH / NHELP, XSURE = XFULLCH / NHELP;   // at m = 128 every helper thread's first PSURE / XSURE chunks exist
  static_assert((PFULLCH + NHELP - 1) / NHELP <= 4 && (XFULLCH + NHELP - 1) / NHELP <= 7, "named prefetch registers cover 4 + 7 chunks per helper thread");
  static_assert(NCH >= NHELP, "every staging thread owns one StageBuf chunk unconditionally");
  uint4 gq0 = spre, gq1 = spre, gq2 = spre, gq3 = spre, xq0 = spre, xq1 = spre, xq2 = spre, xq3 = spre, xq4 = spre, xq5 = spre, xq6 = spre;
  const bool fullm = (m == SW_MAXM);
#define S16_G_EACH(X) X(0, gq0) X(1, gq1) X(2, gq2) X(3, gq3)
#define S16_X_EACH(X) X(0, xq0) X(1, xq1) X(2, xq2) X(3, xq3) X(4, xq4) X(5, xq5) X(6, xq6)
#define S16_GLD(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < pchunks) name = gsrc[c_]; }
#define S16_XLD(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < xchunks) name = xsrc[c_]; }
#define S16_GST(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < pchunks) gdst[c_] = name; }
#define S16_XST(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if (c_ < xchunks) xdst[c_] = name; }
#define S16_GLD_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < PSURE || c_ < pchunks) name = gsrc[c_]; }
#define S16_XLD_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < XSURE || c_ < xchunks) name = xsrc[c_]; }
#define S16_GST_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < PSURE || c_ < pchunks) gdst[c_] = name; }
#define S16_XST_F(u, name) { const int c_ = (tid - 64) + (u) * NHELP; if ((u) < XSURE || c_ < xchunks) xdst[c_] = name; }
  auto helper_phase = [&](int c) {
    const int blk = a.blk_begin + c, mBc = blk_m(c);
    S2ONE(48, tid == 64 && c == 100 + L); S2ONE(51, tid == 448 && c == 100 + L);
    if (wave <= 5) {
      // waves 1-5: block c's packed G, Gx and constants, in registers since the previous phase, go to LDS; block c+1's are
      // requested and have a whole period to land.  (Register staging moves ~60 GB/s through this CU, LDS-DMA only ~25.)
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): tells the compiler that no prefetch load is in flight
      uint4 *gdst = reinterpret_cast<uint4 *>(Q16_GP(c));
      uint4 *xdst = reinterpret_cast<uint4 *>(Q16_GX(c));
      if (fullm) { S16_G_EACH(S16_GST_F) S16_X_EACH(S16_XST_F) } else { S16_G_EACH(S16_GST) S16_X_EACH(S16_XST) }
      reinterpret_cast<uint4 *>(&stage[c & 1])[tid - 64] = spre;                                   // NCH >= NHELP: every thread has one
      if (tid - 64 < NCH - NHELP) reinterpret_cast<uint4 *>(&stage[c & 1])[NHELP + tid - 64] = spre2;
      if (tid - 64 < NSP) reinterpret_cast<uint4 *>(&specb[c & 1])[tid - 64] = cpre;
      S2ONE(49, tid == 64 && c == 100 + L);
      if (c + 1 < nb) {
        const uint4 *gsrc = reinterpret_cast<const uint4 *>(gramp + (size_t)(blk + 1) * pstride);
        const uint4 *xsrc = reinterpret_cast<const uint4 *>(gramx + (size_t)(blk + 1) * m * m);
        if (fullm) { S16_G_EACH(S16_GLD_F) S16_X_EACH(S16_XLD_F) } else { S16_G_EACH(S16_GLD) S16_X_EACH(S16_XLD) }
        spre = reinterpret_cast<const uint4 *>(a.ps.blocks + blk + 1)[tid - 64];
        if (tid - 64 < NCH - NHELP) spre2 = reinterpret_cast<const uint4 *>(a.ps.blocks + blk + 1)[NHELP + tid - 64];
        if (tid - 64 < NSP) cpre = reinterpret_cast<const uint4 *>(a.ps.spec + blk + 1)[tid - 64];
      }
      S2ONE(50, tid == 64 && c == 100 + L);
    } else if (wave == 6) {
      // wave 6: block c's cross terms with blocks c-2 .. c-L+1 = speculated parts (k_spec) + the rows of Gx2_c / Gx3_c that
      // those blocks' accepted markers touch, straight from global memory (the lists have been final since their barriers,
      // the last one since this phase began).  Every load unconditional (clamped indices) and issued before the first use:
      // one round trip per 8 accepted markers; nothing else is on this wave's memory counter.
      double *dst = carry2 + (size_t)(c & 1) * SW_MAXM;
      const int ja = min(lane, m - 1), jb = min(64 + lane, m - 1);
      const bool on2 = (L > 2 && c >= 2), on3 = (L > 3 && c >= 3);
      const int n2 = on2 ? ctrl_s[4 + ((c - 2) & 3)] : 0, n3 = on3 ? ctrl_s[4 + ((c - 3) & 3)] : 0, ntot = n2 + n3;
      const int *pk2 = acc_k2 + (size_t)((c - 2) & 3) * SW_MAXM, *pk3 = acc_k2 + (size_t)((c - 3) & 3) * SW_MAXM;
      const double *pc2 = acc_corr2 + (size_t)((c - 2) & 3) * SW_MAXM, *pc3 = acc_corr2 + (size_t)((c - 3) & 3) * SW_MAXM;
      const int32_t *gx2 = reinterpret_cast<const int32_t *>(a.gramx2) + (size_t)blk * m * m;
      const int32_t *gx3 = reinterpret_cast<const int32_t *>(a.gramx3) + (size_t)blk * m * m;
      // the speculated parts: four loads in flight together with the first batch of rows
      const double x2a = on2 ? a.xspec2[(size_t)blk * SW_MAXM + ja] : 0.0, x2b = on2 ? a.xspec2[(size_t)blk * SW_MAXM + jb] : 0.0;
      const double x3a = on3 ? a.xspec3[(size_t)blk * SW_MAXM + ja] : 0.0, x3b = on3 ? a.xspec3[(size_t)blk * SW_MAXM + jb] : 0.0;
      double cva = 0.0, cvb = 0.0;
      int32_t ga[8], gb[8];
      double cf[8];
      auto batch_load = [&](int i0) {   // the two blocks' accepted markers as one list, eight rows per batch
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = max(0, min(i0 + u, ntot - 1));
          const bool live = idx < ntot, from2 = idx < n2;              // ntot == 0: one harmless in-bounds load of Gx2's row 0
          const int k = live ? (from2 ? pk2[idx] : pk3[idx - n2]) : 0;
          const int32_t *row = ((from2 || !live) ? gx2 : gx3) + (size_t)k * m;
          ga[u] = row[ja]; gb[u] = row[jb];
          cf[u] = (i0 + u < ntot) ? (from2 ? pc2[idx] : pc3[idx - n2]) : 0.0;
        }
      };
      auto batch_use = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) { cva = fma((double)ga[u], cf[u], cva); cvb = fma((double)gb[u], cf[u], cvb); }
      };
      if (on2) batch_load(0);   // (gx3 is only dereferenced for list entries of block c-3, i.e. when on3)
      else { for (int u = 0; u < 8; ++u) { ga[u] = 0; gb[u] = 0; cf[u] = 0.0; } }
      batch_use();
      for (int i0 = 8; i0 < ntot; i0 += 8) { batch_load(i0); batch_use(); }
      cva += x2a; cvb += x2b; cva += x3a; cvb += x3b;
      dst[lane] = (lane < mBc) ? cva : 0.0;
      dst[64 + lane] = (64 + lane < mBc) ? cvb : 0.0;
      S2ONE(53, tid == 384 && c == 100 + L);
    } else {
      // wave 7: the feeders' sums (nothing else is on its memory counter but the state stores of the previous phase), then
      // block c-2's state (left in LDS by wave 0, which issues no global memory operation but the delta granules)
      if (!poll_qsum(c, mBc)) ctrl_s[0] = 0;
      S2WALL(10, tid == 448 && c >= L);
      S2WALL(14, tid == 448 && c == 100 + L);
      S2ONE(52, tid == 448 && c == 100 + L);
      if (c >= 2) store_state(c - 2);   // last: the next phase's poll is behind these stores only
      S2ONE(54, tid == 448 && c == 100 + L);
    }
  };

  // ---- prologue: block 0 through registers, block 1's constants requested ----
  {
    const int mB0 = blk_m(0);
    for (int c = tid; c < pchunks; c += SW_THREADS) reinterpret_cast<uint4 *>(Q16_GP(0))[c] = reinterpret_cast<const uint4 *>(gramp + (size_t)a.blk_begin * pstride)[c];
    for (int c = tid; c < NCH; c += SW_THREADS) reinterpret_cast<uint4 *>(&stage[0])[c] = reinterpret_cast<const uint4 *>(a.ps.blocks + a.blk_begin)[c];
    for (int c = tid; c < NSP; c += SW_THREADS) reinterpret_cast<uint4 *>(&specb[0])[c] = reinterpret_cast<const uint4 *>(a.ps.spec + a.blk_begin)[c];
    if (tid < 16) ctrl_s[tid] = (tid == 0) ? 1 : 0;
    __syncthreads();
    if (wave >= 1 && wave <= 5) {
      if (nb > 1) {
        spre = reinterpret_cast<const uint4 *>(a.ps.blocks + a.blk_begin + 1)[tid - 64];
        if (tid - 64 < NCH - NHELP) spre2 = reinterpret_cast<const uint4 *>(a.ps.blocks + a.blk_begin + 1)[NHELP + tid - 64];
        if (tid - 64 < NSP) cpre = reinterpret_cast<const uint4 *>(a.ps.spec + a.blk_begin + 1)[tid - 64];
      }
    }
    if (wave == 7) { if (!poll_qsum(0, mB0)) ctrl_s[0] = 0; }
    if (wave >= 1 && wave <= 5 && nb > 1) {
      const uint4 *gsrc = reinterpret_cast<const uint4 *>(gramp + (size_t)(a.blk_begin + 1) * pstride);
      const uint4 *xsrc = reinterpret_cast<const uint4 *>(gramx + (size_t)(a.blk_begin + 1) * m * m);
      if (fullm) { S16_G_EACH(S16_GLD_F) S16_X_EACH(S16_XLD_F) } else { S16_G_EACH(S16_GLD) S16_X_EACH(S16_XLD) }
    }
    __syncthreads();
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }
  }
  double sum_d = 0.0, sum_b2 = 0.0;
  double rnext[2] = {0.0, 0.0};
  // implicitly centred columns (SWF_CENTRE; sweep3.hip.h "implicitly centred sweeps"): the lanes' column sums of the next block, read a block ahead like
  // the radii below, and wave 0's running scalar
  const bool cen = (a.flags & SWF_CENTRE) != 0;
  const double ninv = a.ninv, cen_u0 = cen ? a.sc->cen_u0 : 0.0;
  double cenU = cen_u0;
  double ncs[2] = {0.0, 0.0};
  auto cs_load = [&](int b, int q) -> double { const int jj = (a.blk_begin + b) * m + 64 * q + lane; return (cen && 64 * q + lane < blk_m(b)) ? (double)a.csum[min(jj, a.p - 1)] : 0.0; };
  // the rounds' centres and radii (k_spec's QuickBuf) of the next block: wave 0 reads them itself, a block ahead, into registers
  double nzc[2] = {0.0, 0.0}, nha[2] = {INFINITY, INFINITY}, nhr[2] = {-1.0, -1.0};
  const QuickBuf *quick = a.ps.quick ? a.ps.quick + a.blk_begin : nullptr;
  if (wave == 0) {
    const double *ps = Q16_QS(0);
#pragma unroll
    for (int q = 0; q < 2; ++q) { rnext[q] = ps[64 * q + lane]; ncs[q] = cs_load(0, q); }
    if (quick) {
#pragma unroll
      for (int q = 0; q < 2; ++q) { nzc[q] = quick[0].zc[64 * q + lane]; nha[q] = quick[0].ha[64 * q + lane]; nhr[q] = quick[0].hr[64 * q + lane]; }
    }
  }
  S2STAMP_DECL;

  for (int b = 0; b < nb; ++b) {
    const int mB = blk_m(b);
    const bool have_next = (b + 1 < nb);
    const int mBn = have_next ? blk_m(b + 1) : 0;
    float bn[2] = {0.0f, 0.0f}, dn[2] = {0.0f, 0.0f};
    double chi[2] = {1.0, 1.0};
    int nacc = 0;
    if (wave == 0) {
      const StageBuf &st = stage[b & 1];
      const SpecBuf &sb = specb[b & 1];
      const GT *gp = Q16_GP(b);
      int *acc_k = acc_k2 + (size_t)(b & 3) * SW_MAXM;
      double *acc_corr = acc_corr2 + (size_t)(b & 3) * SW_MAXM;
      double r[2];
      LaneConst lc[2];
      {   // all loads unconditional (the buffers hold SW_MAXM entries) and issued before the first use; dead lanes masked after
        double spc[2], rd[2], sz[2], gj[2], ch[2];
        float fb0[2], fxx[2], fb2[2], fdr[2], fta[2], ftr[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = 64 * q + lane;
          spc[q] = sb.spec[t]; gj[q] = sb.gjj[t];
          fb0[q] = st.b0[t]; fxx[q] = st.xxb0[t]; fb2[q] = st.b2[t]; fdr[q] = st.drej[t];
          rd[q] = st.rden[t]; sz[q] = st.sdz1[t]; ch[q] = st.chi[t]; fta[q] = st.tacc[t]; ftr[q] = st.trej[t];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bool live = 64 * q + lane < mB;
          r[q] = live ? (rnext[q] - spc[q]) : 0.0;
          if (cen) r[q] = fma(ncs[q], cenU, r[q]);   // (dead lanes: a zero column sum)
          lc[q].b0 = live ? fb0[q] : 0.0f; lc[q].xxb0 = live ? fxx[q] : 0.0f;
          lc[q].b2 = live ? fb2[q] : 0.0f; lc[q].drej = live ? fdr[q] : 0.0f;
          lc[q].rden = live ? rd[q] : 1.0; lc[q].sdz1 = live ? sz[q] : 0.0;
          lc[q].gjj = live ? gj[q] : 0.0; lc[q].mk = a.marker0 + (uint32_t)((a.blk_begin + b) * m + 64 * q + lane);
          lc[q].tacc = live ? fta[q] : -INFINITY; lc[q].trej = live ? ftr[q] : -INFINITY;   // dead lanes: certain reject
          chi[q] = live ? ch[q] : 1.0;
        }
      }
      S2STAMP(5);
      unsigned long long accmask[2] = {0ull, 0ull};
      // exact speculative rounds (all lanes assume "nobody before me is accepted"), decided by lane_quick's two compares
      double zc[2] = {nzc[0], nzc[1]}, ha[2] = {nha[0], nha[1]}, hr[2] = {nhr[0], nhr[1]};
      if (quick) {
        if (have_next) {
#pragma unroll
          for (int q = 0; q < 2; ++q) { nzc[q] = quick[b + 1].zc[64 * q + lane]; nha[q] = quick[b + 1].ha[64 * q + lane]; nhr[q] = quick[b + 1].hr[64 * q + lane]; }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) lane_quick(lc[q], a.flags, Cc, zc[q], ha[q], hr[q]);
      }
      if (cen) {
        const double cs[2] = {ncs[0], ncs[1]};
        if (have_next) { ncs[0] = cs_load(b + 1, 0); ncs[1] = cs_load(b + 1, 1); }
        quick_rounds<GT, true>(r, lc, zc, ha, hr, accmask, gp, m, mB, lane, a.flags, Cc, odds, one_minus_pi, a.rng, a.iter, cs, ninv, &cenU);
      } else
      quick_rounds(r, lc, zc, ha, hr, accmask, gp, m, mB, lane, a.flags, Cc, odds, one_minus_pi, a.rng, a.iter);
      S2STAMP(6);
      // outputs; delta_b goes out at once as {epoch, float} granules (one 8-byte write-through store per marker)
      unsigned long long *gslot = a.dgran + (size_t)(b % S2_NSLOT) * SW_MAXM;
      const int nacc0 = __popcll(accmask[0]);
      nacc = nacc0 + __popcll(accmask[1]);
      float dl_own[2] = {0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        if (t < mB) {
          const float b1 = lane_b1(r[q], lc[q]);
          const bool inc = ((accmask[q] >> lane) & 1ull) != 0ull;
          bn[q] = inc ? b1 : lc[q].b2;
          dn[q] = inc ? 1.0f : 0.0f;
          dl_own[q] = bn[q] - lc[q].b0;
          if (inc) {   // what this marker changed relative to the speculated step
            const int idx = (q ? nacc0 : 0) + __popcll(accmask[q] & ((1ull << lane) - 1ull));
            acc_k[idx] = t; acc_corr[idx] = (double)(b1 - lc[q].b0) - (double)lc[q].drej;
          }
          float *sp = Q16_STATE(b);
          sp[t] = bn[q]; sp[SW_MAXM + t] = dn[q];
          if (a.flags & SWF_VB_VEC) sp[2 * SW_MAXM + t] = (float)((double)(Sb + bn[q] * bn[q]) / chi[q]);
          sum_d += (double)dn[q];
          sum_b2 = fma((double)bn[q], (double)bn[q], sum_b2);
        }
      }
      {   // the granules: every delta of the block is known here, so is their largest exponent
        const uint32_t exmax = wave_max_u32(max((__float_as_uint(dl_own[0]) >> 23) & 0xFFu, (__float_as_uint(dl_own[1]) >> 23) & 0xFFu));
        if (lane < mB) st_agent_raw64(gslot + lane, s2_dgranule(b, exmax, dl_own[0]));
        if (64 + lane < mB) st_agent_raw64(gslot + 64 + lane, s2_dgranule(b, exmax, dl_own[1]));
      }
      if (lane == 0) ctrl_s[4 + (b & 3)] = nacc;
      S2WALL(0, lane == 0 && b + L < nb);
      S2WALL(7, lane == 0 && b == 100); S2WALL(9, lane == 0 && b == 101); S2WALL(11, lane == 0 && b == 102); S2WALL(12, lane == 0 && b == 103);
      S2STAMP(1);
    } else if (have_next) {
      helper_phase(b + 1);
    }
    __syncthreads();   // block b's recurrence is done; everything block b+1 needs is in LDS (parity (b+1)&1)
    S2ONE(55, tid == 0 && b == 99 + L);
    S2STAMP(2);
    if (ctrl_s[0] == 0) { if (tid == 0) a.sc->error = 1u; return; }
    // Dense inclusion (BayesCpi / BayesDpi keep about half the markers in the model): the distance-1 term -- one row of Gx_{b+1}
    // per accepted marker of block b -- is 250 cycles per marker as a loop of wave 0's; above 12 accepted markers all eight
    // waves take every eighth list entry and wave 0 adds the eight partial sums behind one more barrier.
    const int naccb = ctrl_s[4 + (b & 3)];
    const bool dense = have_next && naccb > 12;   // (measured at 21 % inclusion: 6.52 -> 6.29 us per block against the threshold 24; no difference at 6 %)
    if (dense) {
      const int *lk = acc_k2 + (size_t)(b & 3) * SW_MAXM;
      const double *lcf = acc_corr2 + (size_t)(b & 3) * SW_MAXM;
      const GT *gxn = Q16_GX(b + 1);
      const int ja = min(lane, m - 1), jb = min(64 + lane, m - 1);
      double xa = 0.0, xb = 0.0;
      for (int idx = wave; idx < naccb; idx += 8) {
        const int k = lk[idx];
        const double cf = lcf[idx];
        const GT *row = gxn + (size_t)k * m;
        xa = fma((double)row[ja], cf, xa);
        xb = fma((double)row[jb], cf, xb);
      }
      dpost[(size_t)wave * SW_MAXM + lane] = xa;
      dpost[(size_t)wave * SW_MAXM + 64 + lane] = xb;
      __syncthreads();
    }
    if (wave == 0 && have_next) {
      // r0_{b+1} = sum_w q - Gx' drej_b (precomputed) - what block b's accepted markers changed beyond drej - the cross terms of
      // distance 2 .. L-1, for this lane's two markers; sparse blocks: the accepted markers come from the masks (no list
      // reads), their Gx rows from LDS
      const double *ps = Q16_QS(b + 1);
      const double *cr = carry2 + (size_t)((b + 1) & 1) * SW_MAXM;
      const GT *gxn = Q16_GX(b + 1);
      const SpecBuf &sn = specb[(b + 1) & 1];
      const bool l0 = lane < mBn, l1 = 64 + lane < mBn;
      double r0 = l0 ? ps[lane] - sn.xspec[lane] : 0.0, r1 = l1 ? ps[64 + lane] - sn.xspec[64 + lane] : 0.0;
      if (L > 2) { r0 -= l0 ? cr[lane] : 0.0; r1 -= l1 ? cr[64 + lane] : 0.0; }
      if (dense) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) { s0 += dpost[(size_t)w * SW_MAXM + lane]; s1 += dpost[(size_t)w * SW_MAXM + 64 + lane]; }
        r0 -= s0; r1 -= s1;
      } else if (naccb > 0) {
        // sparse blocks: this block's list (acc_k / acc_corr, written with the outputs) read once, lane i = entry i, and handed to
        // the wave by readlane -- a dozen instructions per accepted marker instead of the two dozen of a loop over the masks
        const int *lk = acc_k2 + (size_t)(b & 3) * SW_MAXM;
        const double *lcf = acc_corr2 + (size_t)(b & 3) * SW_MAXM;
        const int ei = min(lane, naccb - 1);                       // (naccb <= 12 here)
        const int rowoff = lk[ei] * m * (int)sizeof(GT);
        const double cfl = lcf[ei];
        const unsigned char *gl0 = reinterpret_cast<const unsigned char *>(gxn + lane);
        const unsigned char *gl1 = reinterpret_cast<const unsigned char *>(gxn + min(64 + lane, m - 1));
        for (int i = 0; i < naccb; ++i) {
          const int ro = __builtin_amdgcn_readlane(rowoff, i);
          const double corr = readlane_f64(cfl, i);
          const GT ga = *reinterpret_cast<const GT *>(gl0 + ro), gb = *reinterpret_cast<const GT *>(gl1 + ro);
          r0 = fma(-(double)ga, corr, r0);
          r1 = fma(-(double)gb, corr, r1);
        }
      }
      rnext[0] = l0 ? r0 : 0.0; rnext[1] = l1 ? r1 : 0.0;
      S2STAMP(3);
    }
  }
  S2STAMP_FLUSH(16);
  S2WALL_FLUSH;
  if (wave == 7) {   // the state of the last two blocks is still in LDS
    if (nb >= 2) store_state(nb - 2);
    store_state(nb - 1);
  }
  if (wave == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
    if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; if (cen) a.sc->cen_c = cenU - cen_u0; }
  }
}

template <typename XT, bool SELECT, typename GT = typename XTraits<XT>::GT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep2(const SweepArgs a) {
  if (a.redo_only ? (a.sc->redo == 0u) : (a.gate3 > 0.0f && a.sc->inc_rate < a.gate3)) return;   // this sweep is k_sweep3's, or nothing to redo (every workgroup sees the same scalar)
  if ((int)blockIdx.x > a.K) {
    if constexpr (SELECT) s2_feeder(a, (int)blockIdx.x - a.K - 1);   // (the affine variants are launched without feeders)
  } else if ((int)blockIdx.x == a.K) {
    if constexpr (SELECT && sizeof(XT) == 1 && sizeof(GT) == 2) s2_sequencer_sel16(a);
    else s2_sequencer<XT, SELECT, GT>(a);
  }
  else if ((a.flags & SWF_DEBUG_WITHHOLD) && blockIdx.x == 0) return;   // test hook: a streamer that never shows up
  else if constexpr (sizeof(XT) == 1) s2_streamer_i8(a);
  else s2_streamer<XT>(a);
}

}  // namespace bwgr
