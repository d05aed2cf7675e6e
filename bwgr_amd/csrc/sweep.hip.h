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
  SWF_VB_VEC = 16,    // per-marker variance draw vb_j = (Sb + b_j^2)/chisq(df+1)
  SWF_KMUP2 = 32,     // KMUP2's conditional mean: numerator + b0 (not xx*b0), denominator xx*bg + L (src/Rcpp20260726ai.cpp:59)
  SWF_DELTA2 = 64,    // emBA applies every marker's step to the residual twice (src/Rcpp20260726ai.cpp:108, :111): affine sweeps only
  // deterministic EM coordinate updates that are not affine in the marker's dot product (k_sweep2's generic sequencer only):
  SWF_EM_SEL = 128,   // emBB / emBC / emBCpi: b = b1 * d, d = 1/(1 + Pi0 exp(C(|e2|^2-|e1|^2)))  (:165-170, :224-229, :1526-1532)
  SWF_EM_EN = 256,    // emEN: soft threshold (OLS -/+ Lmb1)/(Lmb2 + xx), clamped at 0 (:433-438); Lmb1 rides in sc->lam
  SWF_EM_BL = 512,    // emBL: G + Half_L2 (:379-387); Lmb1 in sc->lam, 1/(xx + cxx) in the sdz1 slot (cxx in sc->Sb)
  SWF_EM_LASSO = 1024, // lasso: yx = (e + x b0).x, soft threshold (yx -/+ Lmb)/xx clamped at 0 (:1477-1485); Lmb in sc->lam; yx_j leaves in d[j]
  SWF_EM_ANY = SWF_EM_SEL | SWF_EM_EN | SWF_EM_BL | SWF_EM_LASSO,
  SWF_SERIAL = 1 << 19,          // affine sweep that must keep the lane-ordered recurrence (wgr's de: Vb_j = |b_j| sqrt(Ve/MSx) feeds rounding-level
                                 // differences of b back into the next sweep's shrinkage, amplified; R/wgr.R:118)
  SWF_DEBUG_WITHHOLD = 1 << 20,  // test hook (bwgr_debug_withhold): slab workgroup 0 leaves at once, so every wait on it must time out
  SWF_CENTRE = 1 << 21           // the sweep is over the IMPLICITLY CENTRED columns x_j - mean(x_j) of an int8 panel (bwgr_panel_set_centred): the streamers
                                 // keep sweeping the raw int8 columns (e_stored = e - shift * 1), the sequencer adds the scalar terms: with s_j = colsum,
                                 // (x_j - s_j/n 1)' e = x_j' e_stored - (s_j/n) sum(e_stored), and sum(e_stored) moves by -s_k delta_k per marker
};

// scalars produced on the device by the per-iteration tail kernel (or filled by the host for KMUP)
struct ChainScalars {
  float ve, vb, lam, pi;        // current residual / common marker variance, common lambda, pi
  float Sb, Se, C, odds;        // priors; C = -0.5/sqrt(ve); odds = pi/(1-pi) at chain start
  float mu, dfp1, vy, MSx;
  float MU, VE, VBs, Pi;        // posterior sums
  double sum_d, sum_b2;         // written by the sweep (marker order, fp64)
  uint32_t error;               // 1: an exchange gave up; 2: the fixed-point residual of k_sweep3 left its range
  float bg;                     // KMUP2: n0/n of the row subsample
  int e3_sh;                    // k_sweep3: the sweep's fixed-point scale, e_fixed = e * 2^e3_sh (k_escale)
  uint32_t e3_dex;              // k_sweep3: largest float exponent field among this sweep's rejected steps (k_prestage -> k_escale)
  float inc_rate;               // share of markers in the model, as far as the chain knows (start: 1 - pi; then mean(d) of the last sweep):
                                // picks the sweep engine of a selection model on the device (SweepArgs::gate3)
  uint32_t redo;                // a fixed-point sweep left its range (error 2): the state it started from is back and the launches queued
                                // behind it with SweepArgs::redo_only run the same sweep on the fp64 residual (k_range_recover ... k_redo_clear)
  double snap_sum_d, snap_sum_b2;   // sum_d / sum_b2 before that sweep
  uint32_t nredo;               // sweeps redone so far (bwgr_chain_redo_count)
  // implicitly centred sweeps (SWF_CENTRE): u0 = -(sum(e) at the start of the launch + cpre[blk_begin]) / n, written by k_cen_begin before the
  // sweep kernel; cen_c = sum over the launch's included markers of (s_k / n) corr_k, written by the sequencer; k_cen_end turns e_stored back into e
  double cen_u0, cen_c;
};

// per-marker constants of one sweep, produced chip-wide by k_prestage before the sweep kernel starts (they depend on
// the previous iteration's b and lambda and on the RNG counters only, never on the residual)
struct StageBuf;
struct SpecBuf {                // r-independent per-marker terms of one block (k_sweep2), filled by k_spec
  double spec[128];             // sum_{k<j} G_bb[k][j] * drej[k]        (selection models)
  double xspec[128];            // sum_k Gx_b[k][j] * drej_{b-1}[k]      (selection models; unused for the first block of a launch)
  double gjj[128];              // G_bb[j][j]
};
struct QuickBuf {               // lane_quick's centre and radii of one block (k_sweep2's selection rounds), filled by k_spec
  double zc[128], ha[128], hr[128];
};
struct PreStage {
  StageBuf *blocks;   // one StageBuf per marker block, filled by k_prestage
  SpecBuf *spec;      // one SpecBuf per marker block, filled by k_spec
  QuickBuf *quick;    // one per marker block, filled by k_spec (selection sweeps of k_sweep2)
};

struct SweepArgs {
  const void *X; int64_t ld;    // slab-major: element (row i, marker j) at ((i/R)*p + j)*R + i%R; ld = K*R padded rows
  const void *gram;             // [nblocks][m][m]  diagonal blocks X_b' X_b
  const void *gramx;            // [nblocks][m][m]  off-diagonal blocks X_{b-1}' X_b (entry 0 unused)
  const void *gramx2;           // [nblocks][m][m]  X_{b-2}' X_b (entries 0, 1 unused); null when the panel has < 3 blocks
  double *xspec2;               // [nblocks][SW_MAXM]  sum_k gramx2_b[k][j] * drej_{b-2}[k] (k_spec, lag >= 3)
  const void *gramx3;           // [nblocks][m][m]  X_{b-3}' X_b (lag 4; null otherwise)
  double *xspec3;               // [nblocks][SW_MAXM]  sum_k gramx3_b[k][j] * drej_{b-3}[k] (k_spec, lag 4)
  int lag;                      // k_sweep2: q_b is taken against e after delta_{b-lag}; 2, or 3 / 4 for selection models
  int nfeed;                    // k_sweep2: number of q feeder workgroups behind the sequencer (feeder f serves blocks b = f mod nfeed)
  const void *gramp;            // [nblocks][pstride] strict upper triangle of the diagonal blocks, row k = entries (k, k+1..m-1)
  int pstride;
  int n, p, m, K, R;
  int blk_begin, blk_end;
  int flags;
  double *e;                    // residual, ld entries (padding rows stay 0)
  float *b, *d, *vb;
  const float *xx, *lam;
  ChainScalars *sc;
  uint32_t iter;
  uint32_t marker0;             // global id of local marker 0 (RNG counters use global ids)
  Rng rng;
  PreStage ps;
  double *xpart;                // k_sweep: [2][K][SW_MAXM]
  double *qpart;                // k_sweep2: [S2_NSLOT][K][SW_MAXM] streamer slab dots
  unsigned long long *dgran;    // k_sweep2: [S2_NSLOT][SW_MAXM] {epoch, float delta} granules
  uint32_t *xflags;             // [K*SW_FLAG_STRIDE] epochs, then the abort word
  unsigned long long *stamps;   // diagnostic build only (-DBWGR_STAMPS): per-phase cycle sums of workgroup 0
  const double *draws;          // k_draws' output for THIS iteration (five arrays of p: z1, z2, the two threshold logits, chi), or null: k_prestage draws itself
  int redo_only;                // this launch is the fp64 fallback of a fixed-point sweep: it runs only when sc->redo is set
  float gate3;                  // > 0: both engines of the selection models are launched and the device picks one -- k_sweep3 (and its
                                // k_escale / k_spec3) runs when sc->inc_rate < gate3, k_sweep2 (and k_spec) otherwise; 0: no gating
  // SWF_CENTRE: the panel's column sums (p int32), the running sums cpre[b] = sum over the markers of blocks < b of s_k * (rejected step on the
  // sweep's fixed-point grid) with cpre[nblocks] the total (k_cen_tot / k_cen_scan, per iteration), and 1 / n
  const int32_t *csum; double *cpre; double ninv;
};

template <typename XT> struct XTraits;
template <> struct XTraits<int8_t> { using GT = int32_t; static constexpr int PER16 = 16; static constexpr int MAXM = 128; };
template <> struct XTraits<float> { using GT = double; static constexpr int PER16 = 4; static constexpr int MAXM = 64; };

// padded LDS row length (elements) so that consecutive markers start on odd multiples of 16 B
template <typename XT> __host__ __device__ inline int tile_rp(int R) { return R + XTraits<XT>::PER16; }

struct StageBuf {               // per-block per-marker constants, lane = marker
  float b0[SW_MAXM], xxb0[SW_MAXM], b2[SW_MAXM], drej[SW_MAXM];
  double rden[SW_MAXM], sdz1[SW_MAXM], chi[SW_MAXM];
  float tacc[SW_MAXM], trej[SW_MAXM];   // selection models: the uniform turned into thresholds on C*(|e2|^2-|e1|^2), see lane_accept
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
  s += 4 * SW_MAXM * sizeof(double);                           // r0, speculative correction s, Gram diagonal, delta
  s += 2 * SW_MAXM * sizeof(float);                            // bnew, dnew
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

// hand-off words are always accessed as GLOBAL (address_space(1)) agent-scope atomics, never through flat_*
typedef __attribute__((address_space(1))) uint32_t gu32_t;
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
__device__ __forceinline__ uint32_t ld_agent_u32(const uint32_t *p) {
  return __hip_atomic_load((const gu32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u32(uint32_t *p, uint32_t v) {
  __hip_atomic_store((gu32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent_raw64(const unsigned long long *p) {
  return __hip_atomic_load((const gu64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_raw64(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store((gu64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u64(double *p, double v) {
  st_agent_raw64(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ double ld_agent_f64(const double *p) {
  return __longlong_as_double((long long)ld_agent_raw64(reinterpret_cast<const unsigned long long *>(p)));
}

// slab-major addressing of the resident panel: workgroup w's rows of marker j are the contiguous run
// X[(w*p + j)*R .. +R), so a block's tile is ONE contiguous mB*R run (one DRAM/TLB page stream per workgroup)
__host__ __device__ inline size_t xoff(int64_t i, int64_t j, int R, int64_t p) {
  const int64_t w = i / R;
  return (size_t)((w * p + j) * R + (i - w * R));
}

// cooperative copy of one block's slab tile into LDS (marker-major, padded rows), split in two halves so that the
// loads of block s+2 can be in flight (in registers) during the whole of block s+1: first-touch HBM latency on an
// almost idle fabric is ~3 us, a block period is ~10 us.  SW_TCH chunks of 16 B per thread bound the tile size
// (checked on the host: m*R*sizeof(x) <= SW_TCH*16*(SW_THREADS-64)).
static constexpr int SW_TCH = 5;
// Plain scalar locals tp0..tp4 (an aggregate here ends up in scratch memory).  TILE_ISSUE(j0, mB) starts the loads of
// the tile of block [j0, j0+mB); TILE_COMMIT(dst, mB) stores them to LDS.  Both use the prefetch waves' numbering.
#define BWGR_TILE_EACH(X) X(0, tp0) X(1, tp1) X(2, tp2) X(3, tp3) X(4, tp4)
#define BWGR_ISSUE1(u, name) { const int c_ = (tid - 64) + (u) * (SW_THREADS - 64); if (c_ < tot_) name = src_[c_]; }
#define TILE_ISSUE(j0_, mB_) do { const int tot_ = (mB_) * (R / PER); \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(X + (size_t)(j0_) * R); BWGR_TILE_EACH(BWGR_ISSUE1) } while (0)
#define BWGR_COMMIT1(u, name) { const int c_ = (tid - 64) + (u) * (SW_THREADS - 64); if (c_ < tot_) { \
    const int jj_ = (int)(((float)c_ + 0.5f) * rcpr_), ii_ = c_ - jj_ * cpr_;   /* exact: c < 2^14 */ \
    *reinterpret_cast<uint4 *>((dst_) + (size_t)jj_ * Rp + ii_ * PER) = name; } }
#define TILE_COMMIT(dstp, mB_) do { XT *dst_ = (dstp); const int cpr_ = R / PER; const int tot_ = (mB_) * cpr_; \
    const float rcpr_ = 1.0f / (float)cpr_; BWGR_TILE_EACH(BWGR_COMMIT1) } while (0)

// k_prestage: one thread per (marker, piece); piece 0: in-model normal, 1: alternative normal, 2: Bernoulli uniform,
// 3: chi-square.  Restates the per-marker scalar set-up of src/Rcpp20260726ai.cpp:20-21 / :615-617 / :670-680.
__global__ void k_prestage(const SweepArgs a, int j_begin, int j_end) {
  const ChainScalars &sc = *a.sc;
  const float ve = sc.ve, lam_common = sc.lam, dfp1 = sc.dfp1;
  const int64_t nm = j_end - j_begin;
  __shared__ uint32_t dex_s;   // largest float exponent field of the rejected steps this workgroup staged
  if (threadIdx.x == 0) dex_s = 0u;
  __syncthreads();
  for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < 4 * nm; task += (int64_t)gridDim.x * blockDim.x) {
    const int piece = (int)(task / nm);
    const int j = j_begin + (int)(task - (int64_t)piece * nm);
    StageBuf &st = a.ps.blocks[j / a.m];
    const int t = j % a.m;
    const float b0 = a.b[j];
    const float xxj = a.xx[j];
    const float lamj = (a.flags & SWF_LAM_VEC) ? a.lam[j] : lam_common;
    const bool k2 = (a.flags & SWF_KMUP2) != 0;
    const float den = k2 ? (xxj * sc.bg + lamj) : (xxj + lamj);
    const float sd = sqrtf(ve / den);
    const uint32_t mk = a.marker0 + (uint32_t)j;
    const bool sel = (a.flags & SWF_SELECT) != 0;
    if (piece == 0) {
      st.b0[t] = b0;
      st.xxb0[t] = k2 ? b0 : xxj * b0;
      st.rden[t] = 1.0 / (double)den;
      const double sdz1 = (a.flags & SWF_EM_BL) ? 1.0 / (double)(xxj + sc.Sb)             // emBL's second denominator xx + cxx, :380
                                                : (double)sd * (a.draws ? a.draws[j] : rng_normal(a.rng, mk, a.iter, RNG_Z1, 0));
      st.sdz1[t] = sdz1;
      // affine sweeps on the fixed-point residual (k_sweep2w): a marker's step is not known before the sweep, but |b0| and the noise
      // term bound what it can be when the residual itself is tiny (a KMUP call with e = 0); k_escale sizes the grid by the larger
      if (!sel && !(a.flags & SWF_EM_ANY)) atomicMax(&dex_s, (__float_as_uint(fmaxf(fabsf(b0), fabsf((float)sdz1))) >> 23) & 0xFFu);
    } else if (piece == 1) {
      const float b2 = sel ? (float)((double)0.0f + (double)sd * (a.draws ? a.draws[(size_t)a.p + j] : rng_normal(a.rng, mk, a.iter, RNG_Z2, 0))) : 0.0f;
      st.b2[t] = b2;
      const float drej = sel ? (b2 - b0) : 0.0f;   // the step a marker takes when it is NOT included
      st.drej[t] = drej;
      if (sel) atomicMax(&dex_s, (__float_as_uint(drej) >> 23) & 0xFFu);   // k_sweep3 sizes its fixed-point grid by the largest step
    } else if (piece == 2) {
      if (a.draws) {   // (k_draws: selection models other than BayesDpi) the uniform's share of the thresholds came ahead; the odds' is this iteration's
        float ta = -INFINITY, tr = INFINITY;
        const double la = a.draws[2 * (size_t)a.p + j], lr = a.draws[3 * (size_t)a.p + j];   // log1p(-u(1 +- eta)) - log(u(1 +- eta)); NaN: no threshold
        if (sc.odds > 0.0f && lr == lr) {
          const double lo = log((double)sc.odds);
          if (la == la) ta = (float)(la - lo);
          tr = (float)(lr - lo);
          ta = nextafterf(ta, -INFINITY); tr = nextafterf(tr, INFINITY);
          if (!(ta < tr)) { ta = -INFINITY; tr = INFINITY; }
        }
        st.tacc[t] = ta; st.trej[t] = tr;
        continue;
      }
      const double uj = sel ? rng_uniform(a.rng, mk, a.iter, RNG_U, 0) : 0.0;
      // The Bernoulli step accepts iff u < pj with pj a float function of x = C*(|e2|^2 - |e1|^2):
      //   pj = 1/(1 + odds*expf(x))  or (BayesDpi)  min(1, (1-pi)*expf(-x)).
      // Both are decreasing in x, so u maps to a threshold on x.  pj's float evaluation is within a relative 1e-6 of the exact
      // function; x below tacc is therefore a certain accept, x above trej a certain reject, and only the sliver between them
      // (about one decision in 10^5) needs pj itself -- the exponential and the division leave the recurrence's dependent chain.
      float ta = -INFINITY, tr = INFINITY;
      if (sel && uj > 0.0) {
        const double eta = 1e-6, ua = uj * (1.0 + eta), ur = uj * (1.0 - eta);
        if (a.flags & SWF_MH) {
          const double omp = (double)(1.0f - sc.pi);
          if (omp > 0.0) { ta = (float)log(omp / ua); tr = (float)log(omp / ur); }
        } else if (sc.odds > 0.0f) {
          const double lo = log((double)sc.odds);
          if (ua < 1.0) ta = (float)(log1p(-ua) - log(ua) - lo);
          tr = (float)(log1p(-ur) - log(ur) - lo);
        }
        ta = nextafterf(ta, -INFINITY); tr = nextafterf(tr, INFINITY);   // conservative after the rounding to float
        if (!(ta < tr)) { ta = -INFINITY; tr = INFINITY; }
      }
      st.tacc[t] = ta; st.trej[t] = tr;
    } else {
      st.chi[t] = a.draws ? a.draws[4 * (size_t)a.p + j] : ((a.flags & SWF_VB_VEC) ? rng_chisq(a.rng, (double)dfp1, mk, a.iter, RNG_CHI) : 1.0);
    }
  }
  // the unused entries of every block (blocks narrower than SW_MAXM, the ragged last block): constants with which a lane rejects
  // for certain and changes nothing (k_sweep3's recurrence wave runs without dead-lane masks, its streamers digitise all
  // SW_MAXM steps of a block)
  if (j_end > j_begin && (a.m < SW_MAXM || (j_end % a.m) != 0)) {
    const int blk0 = j_begin / a.m, blk1 = (j_end - 1) / a.m;
    for (int blk = blk0 + (int)blockIdx.x; blk <= blk1; blk += (int)gridDim.x) {
      StageBuf &st = a.ps.blocks[blk];
      for (int t = min(a.m, j_end - blk * a.m) + (int)threadIdx.x; t < SW_MAXM; t += (int)blockDim.x) {
        st.b0[t] = 0.0f; st.xxb0[t] = 0.0f; st.b2[t] = 0.0f; st.drej[t] = 0.0f;
        st.rden[t] = 0.0; st.sdz1[t] = 0.0; st.chi[t] = 1.0;
        st.tacc[t] = -INFINITY; st.trej[t] = -INFINITY;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && dex_s != 0u) atomicMax(&a.sc->e3_dex, dex_s);
}
// k_draws: the variates of iteration `iter` that do not depend on the chain's state -- the two normals, the Bernoulli uniform as the two threshold
// logits before the odds are subtracted, the chi-square -- for markers [j_begin, j_end): what k_prestage spends most of its 0.2 ms per C4 iteration on
// (fp64 logarithms, Box-Muller, the gamma sampler).  The streams are counter-based (marker, iteration, purpose), so they are drawn one iteration AHEAD,
// on a second stream, beside the sweep (which leaves two thirds of the chip idle); k_prestage then only combines them with ve, the variances and the odds.
// Selection models with the logistic step (not BayesDpi's Metropolis form, whose threshold takes pi inside the logarithm).  Same values bit for bit:
// the expressions are k_prestage's own, cut where the state enters.  A large dynamic LDS request keeps its workgroups off the sweep's compute units.
__global__ void k_draws(Rng rng, uint32_t marker0, uint32_t iter, int flags, const ChainScalars *sc_in, int64_t p, int j_begin, int j_end, double *out) {
  const float dfp1 = sc_in->dfp1;   // (df + 1: a constant of the chain)
  const int64_t nm = j_end - j_begin;
  const bool sel = (flags & SWF_SELECT) != 0;
  for (int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; task < 4 * nm; task += (int64_t)gridDim.x * blockDim.x) {
    const int piece = (int)(task / nm);
    const int j = j_begin + (int)(task - (int64_t)piece * nm);
    const uint32_t mk = marker0 + (uint32_t)j;
    if (piece == 0) out[j] = rng_normal(rng, mk, iter, RNG_Z1, 0);
    else if (piece == 1) out[(size_t)p + j] = sel ? rng_normal(rng, mk, iter, RNG_Z2, 0) : 0.0;
    else if (piece == 2) {
      const double uj = sel ? rng_uniform(rng, mk, iter, RNG_U, 0) : 0.0;
      double la = NAN, lr = NAN;
      if (sel && uj > 0.0) {
        const double eta = 1e-6, ua = uj * (1.0 + eta), ur = uj * (1.0 - eta);
        if (ua < 1.0) la = log1p(-ua) - log(ua);
        lr = log1p(-ur) - log(ur);
      }
      out[2 * (size_t)p + j] = la; out[3 * (size_t)p + j] = lr;
    } else out[4 * (size_t)p + j] = (flags & SWF_VB_VEC) ? rng_chisq(rng, (double)dfp1, mk, iter, RNG_CHI) : 1.0;
  }
}
// k_prestage_fin: k_prestage for an iteration whose variates k_draws drew ahead -- one thread per marker, the marker's scalars read and its
// denominator and standard deviation formed once for the four pieces, log(odds) once per workgroup.  The same expressions as k_prestage's, so the same
// StageBuf bit for bit (tests: the chains do not move).  Selection models with the logistic step only (launch_prestage).
__global__ __launch_bounds__(256) void k_prestage_fin(const SweepArgs a, int j_begin, int j_end) {
  const ChainScalars &sc = *a.sc;
  const float ve = sc.ve, lam_common = sc.lam;
  __shared__ uint32_t dex_s;
  __shared__ double lo_s;
  if (threadIdx.x == 0) { dex_s = 0u; lo_s = (sc.odds > 0.0f) ? log((double)sc.odds) : 0.0; }
  __syncthreads();
  const bool odds_pos = sc.odds > 0.0f;
  const double lo = lo_s;
  const bool k2 = (a.flags & SWF_KMUP2) != 0;
  const size_t p = (size_t)a.p;
  uint32_t dex = 0u;
  for (int j = j_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x); j < j_end; j += (int)(gridDim.x * blockDim.x)) {
    StageBuf &st = a.ps.blocks[j / a.m];
    const int t = j % a.m;
    const float b0 = a.b[j];
    const float xxj = a.xx[j];
    const float lamj = (a.flags & SWF_LAM_VEC) ? a.lam[j] : lam_common;
    const float den = k2 ? (xxj * sc.bg + lamj) : (xxj + lamj);
    const float sd = sqrtf(ve / den);
    st.b0[t] = b0;
    st.xxb0[t] = k2 ? b0 : xxj * b0;
    st.rden[t] = 1.0 / (double)den;
    st.sdz1[t] = (double)sd * a.draws[j];
    const float b2 = (float)((double)0.0f + (double)sd * a.draws[p + j]);
    st.b2[t] = b2;
    const float drej = b2 - b0;
    st.drej[t] = drej;
    dex = max(dex, (__float_as_uint(drej) >> 23) & 0xFFu);
    float ta = -INFINITY, tr = INFINITY;
    const double la = a.draws[2 * p + j], lr = a.draws[3 * p + j];   // log1p(-u(1 +- eta)) - log(u(1 +- eta)); NaN: no threshold
    if (odds_pos && lr == lr) {
      if (la == la) ta = (float)(la - lo);
      tr = (float)(lr - lo);
      ta = nextafterf(ta, -INFINITY); tr = nextafterf(tr, INFINITY);
      if (!(ta < tr)) { ta = -INFINITY; tr = INFINITY; }
    }
    st.tacc[t] = ta; st.trej[t] = tr;
    st.chi[t] = a.draws[4 * p + j];
  }
  if (dex) atomicMax(&dex_s, dex);
  if (j_end > j_begin && (a.m < SW_MAXM || (j_end % a.m) != 0)) {   // the unused entries of every block: as k_prestage
    const int blk0 = j_begin / a.m, blk1 = (j_end - 1) / a.m;
    for (int blk = blk0 + (int)blockIdx.x; blk <= blk1; blk += (int)gridDim.x) {
      StageBuf &st = a.ps.blocks[blk];
      for (int t = min(a.m, j_end - blk * a.m) + (int)threadIdx.x; t < SW_MAXM; t += (int)blockDim.x) {
        st.b0[t] = 0.0f; st.xxb0[t] = 0.0f; st.b2[t] = 0.0f; st.drej[t] = 0.0f;
        st.rden[t] = 0.0; st.sdz1[t] = 0.0; st.chi[t] = 1.0;
        st.tacc[t] = -INFINITY; st.trej[t] = -INFINITY;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && dex_s != 0u) atomicMax(&a.sc->e3_dex, dex_s);
}
// block constants global -> LDS: one StageBuf is sizeof(StageBuf)/16 chunks, at most one per thread
__device__ inline void stage_block(StageBuf &st, int blk, const SweepArgs &a, int tid0, int nthreads) {
  constexpr int NCH = (int)(sizeof(StageBuf) / 16);
  static_assert(sizeof(StageBuf) % 16 == 0, "StageBuf must be a multiple of 16 bytes");
  const uint4 *src = reinterpret_cast<const uint4 *>(a.ps.blocks + blk);
  uint4 *dst = reinterpret_cast<uint4 *>(&st);
  for (int c = tid0; c < NCH; c += nthreads) dst[c] = src[c];
}

#ifdef BWGR_STAMPS
#define STAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// Gram block -> LDS, keeping only entries (row, col > row): row = a finalized marker, col = a later marker it
// corrects.  A lane's r therefore stops changing once its own marker is final.  The diagonal goes to gdiag_s.
template <typename GT>
__device__ __forceinline__ void put_gram4(GT *gram_s, double *gdiag_s, int m, int e0, const GT v[4]) {
  const int row = (int)(((float)e0 + 0.5f) * (1.0f / (float)m)), col0 = e0 - row * m;   // exact for e0 < 2^14
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = col0 + c;
    gram_s[e0 + c] = (col > row) ? v[c] : (GT)0;
    if (col == row) gdiag_s[row] = (double)v[c];
  }
}

// what marker t does given its current r: the in-model draw b1 and (SELECT) the inclusion decision
struct LaneConst {
  float b0, xxb0, b2, drej;
  double rden, sdz1, gjj;
  float tacc, trej;   // the Bernoulli step's uniform as thresholds (k_prestage); the uniform itself is re-derived when needed
  uint32_t mk;        // global marker id (RNG counter word)
};
// The conditional mean stays in fp64 (wide contract): 2 dependent fp64 ops instead of the
// cvt/add/cvt/mul/cvt/cvt/add chain that the reference's float rounding points would force on the serial path.
__device__ __forceinline__ float lane_b1(double r, const LaneConst &c) {
  return (float)fma(r + (double)c.xxb0, c.rden, c.sdz1);
}
// The EM family's non-affine coordinate updates (deterministic): the new effect given the marker's current dot product r;
// *dout = the inclusion weight d_j of the soft-selection members (1 otherwise).  Like lane_b1, the dot product and what is
// formed from it stay in fp64 and only the stored effect is rounded to float.
__device__ __forceinline__ float lane_em(double r, const LaneConst &c, int flags, float Cc, float Pi0, float L1, float *dout) {
  const double ols = r + (double)c.xxb0;                    // gen.col(j).dot(e) + xx[j]*b0
  if (flags & SWF_EM_SEL) {
    const float b1 = (float)(ols * c.rden);
    const double D1 = (double)(b1 - c.b0), D2 = (double)(0.0f - c.b0);
    const double diffd = 2.0 * r * (D1 - D2) + c.gjj * (D2 * D2 - D1 * D1);   // |e2|^2 - |e1|^2, e_k = e - x d_k
    // std::exp(float) rounded from the fp64 exponential: emBC / emBCpi divide by (mean(d) - Pi), which amplifies an ulp of d
    const float LR = Pi0 * (float)exp((double)(Cc * (float)diffd));
    const float d = 1.0f / (1.0f + LR);
    *dout = d;
    return b1 * d;
  }
  *dout = 1.0f;
  // the soft thresholds, branch-free (the sign of the OLS term picks the side of the threshold and of the clamp)
  if (flags & SWF_EM_LASSO) {
    const double yx = fma(c.gjj, (double)c.b0, r);          // e += gen.col(j)*b[j]; yx[j] = e.dot(gen.col(j))  (x.x = G_jj)
    *dout = (float)yx;
    const bool pos = yx > 0.0;
    const float b1 = (float)((yx - (pos ? (double)L1 : -(double)L1)) * c.rden);
    return pos ? fmaxf(b1, 0.0f) : fminf(b1, 0.0f);
  }
  const bool pos = ols > 0.0;
  const double sl = pos ? (double)L1 : -(double)L1;
  if (flags & SWF_EM_EN) {
    const float b1 = (float)((ols - sl) * c.rden);
    return pos ? fmaxf(b1, 0.0f) : fminf(b1, 0.0f);
  }
  // SWF_EM_BL
  const double half = 0.5 * ols * c.sdz1;                   // Half_L2 = 0.5*OLS/(xx+cxx)
  const double G = 0.5 * (ols - sl) * c.rden;
  const bool keep = pos ? (G > 0.0) : (G < 0.0);
  return (float)(keep ? G + half : half);
}
// the Bernoulli step's literal form, u < pj: reached only when some lane's x falls between its two thresholds (about one
// decision in 10^5).  Not inlined, so that the Philox block that re-derives the uniform stays out of the recurrence's code.
__device__ __attribute__((noinline)) bool lane_accept_exact(double diffd, uint32_t mk, int flags, float Cc, float odds, float one_minus_pi,
                                                            Rng rng, uint32_t iter) {
  float pj;
  if (flags & SWF_MH) {
    const float diff = (float)(-diffd);
    pj = one_minus_pi * expf(Cc * diff);
    if (pj > 1.0f) pj = 1.0f;
  } else {
    const float diff = (float)diffd;
    const float LR = odds * expf(Cc * diff);
    pj = 1.0f / (1.0f + LR);
  }
  return rng_uniform(rng, mk, iter, RNG_U, 0) < (double)pj;   // the same uniform k_prestage drew
}
__device__ __forceinline__ bool lane_accept(double r, float b1, const LaneConst &c, int flags, float Cc, float odds,
                                            float one_minus_pi, const Rng &rng, uint32_t iter) {
  const float d1f = b1 - c.b0;
  const float d2f = (flags & SWF_ALT_B2) ? (c.b2 - c.b0) : (0.0f - c.b0);
  const double D1 = (double)d1f, D2 = (double)d2f;
  const double diffd = 2.0 * r * (D1 - D2) + c.gjj * (D2 * D2 - D1 * D1);   // |e2|^2 - |e1|^2, e_k = e - x d_k
  const float x = Cc * (float)diffd;                  // the argument of the exponential (BayesDpi: its negative, exactly)
  const bool sure_acc = x < c.tacc, sure_rej = x > c.trej;
  if (__builtin_expect(__ballot(!(sure_acc || sure_rej)) == 0ull, 1)) return sure_acc;   // every lane decided by its thresholds
  const bool exact = lane_accept_exact(diffd, c.mk, flags, Cc, odds, one_minus_pi, rng, iter);
  return sure_acc ? true : (sure_rej ? false : exact);
}

// The inclusion test as two compares on the marker's residual dot.  With D1 = rden*r + k1 the un-rounded step, lane_accept's
// x = C*(|e2|^2 - |e1|^2) is C*Q(r) with Q a parabola in r: Q = A*(r - zc)^2 + Qmin, A = rden*(2 - gjj*rden) > 0, and C < 0.  So
//   x < tacc   <=>  |r - zc| > sqrt((tacc/C - Qmin)/A),     x > trej   <=>  |r - zc| < sqrt((trej/C - Qmin)/A).
// lane_accept evaluates x through four float roundings (the draw, the step, the float of Q, the product with C); their sum is bounded
// (err below, in units of Q, with a fourfold margin on the dominant term) and the two radii are moved apart by it: |z| > ha is a
// certain accept of lane_accept, |z| < hr a certain reject, and the sliver between (about one decision in 10^6; tools/quick_accept_check.py
// straddles the radii at relative distances 1e-15 .. 1e-6 with lane_accept's float arithmetic restated) is left to lane_accept
// itself.  The decisions are therefore lane_accept's, bit for bit; what leaves the rounds' dependent chain is its eleven operations.
__device__ __forceinline__ void lane_quick(const LaneConst &c, int flags, float Cc, double &zc, double &ha, double &hr) {
  const double b0 = (double)c.b0, D2 = (double)((flags & SWF_ALT_B2) ? (c.b2 - c.b0) : (0.0f - c.b0));
  const double k1 = fma((double)c.xxb0, c.rden, c.sdz1) - b0;
  const double gr = c.gjj * c.rden;
  const double A = c.rden * (2.0 - gr), Bh = (k1 - D2) - gr * k1, C0 = c.gjj * (D2 * D2 - k1 * k1);
  zc = -Bh / A;
  const double Qmin = fma(Bh, zc, C0);
  const double qa = (double)c.tacc / (double)Cc, qr = (double)c.trej / (double)Cc;   // C < 0: qr <= qa; -inf / +inf thresholds map to +inf / -inf
  auto err = [&](double h, double q) {
    const double ra = fabs(zc) + h;                                   // the larger |r| of the two crossings
    const double d1 = fma(c.rden, ra, fabs(k1)), t = d1 + fabs(b0);   // bounds of |step| and |draw| there
    const double eD = 6.1e-8 * (t + d1);                              // the draw's and the step's roundings to float
    return 4.0 * (2.0 * (ra + c.gjj * d1) * eD + c.gjj * eD * eD) + 2.5e-7 * fabs(q)
           + 1e-13 * (2.0 * ra * (d1 + fabs(D2)) + c.gjj * (D2 * D2 + d1 * d1));
  };
  ha = INFINITY; hr = -1.0;   // neither test ever certain: every decision is lane_accept's
  if (A > 0.0 && fabs(zc) < 1e300 && Cc < 0.0f) {
    if (fabs(qa) < INFINITY) {
      const double num = (qa - Qmin) + err(sqrt(fmax(0.0, (qa - Qmin) / A)), qa);
      ha = (num > 0.0) ? sqrt(num / A) * (1.0 + 1e-12) : -1.0;
    }
    if (qr == INFINITY) hr = INFINITY;   // (dead lanes: trej = -inf)
    else if (fabs(qr) < INFINITY) {
      const double num = (qr - Qmin) - err(sqrt(fmax(0.0, (qr - Qmin) / A)), qr);
      hr = (num > 0.0) ? sqrt(num / A) * (1.0 - 1e-12) : -1.0;
    }
  }
}
// The speculative rounds of one 128-marker block on lane_quick's compares (wave-wide; lane = marker, two lane groups).  Every lane
// holds its marker's residual dot as if no undecided marker before it were included (their rejected steps are inside r); a round
// finds the first lane that is not a certain reject -- certain accepts are taken, the sliver asks lane_accept -- applies what that
// marker changes beyond its speculated step to the later lanes through its packed Gram row, and the next round re-tests.  The row
// of the likely next marker (the following candidate under the state before the step) is requested while the step is applied.
// Same decisions and the same arithmetic on r as rounds made of lane_accept.  gp: packed Gram block in LDS (row k's entry for
// marker j > k at prow(k) + j - k - 1; unconditional loads: the buffers carry slack).
// (One wave issues at most one instruction every four to five cycles, scalar ones included, so the rounds cost their instruction
// count, not their dependent chain: the loop below is written for few instructions -- the packed row's offset from a lane-indexed
// table by one v_readlane instead of the triangular-number arithmetic, one exit test, the correction formed per lane before the
// lane is known, the sliver's code out of line.)
// CEN (SWF_CENTRE, implicitly centred columns): cs = the two lanes' column sums s_j, cenU the running scalar -(E_0 + ...)/n + sum over the included
// markers of (s_k / n) corr_k (see sweep3.hip.h, "implicitly centred sweeps"); an included marker's packed row G_kj acts as G_kj - s_j s_k / n.
template <typename GT, bool CEN = false>
__device__ __forceinline__ void quick_rounds(double (&r)[2], const LaneConst (&lc)[2], const double (&zc)[2], const double (&ha)[2], const double (&hr)[2],
                                             unsigned long long (&accmask)[2], const GT *gp, int m, int mB,
                                             int lane, int flags, float Cc, float odds, float one_minus_pi, const Rng &rng, uint32_t iter,
                                             const double *cs = nullptr, double ninv = 0.0, double *cenU = nullptr) {
  const int ngrp = (mB + 63) >> 6;
  const GT *gl = gp + lane;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (q < ngrp) {
      const int base = 64 * q, cnt = min(64, mB - base);
      const int kk = base + lane;
      const int tab = (kk * (m - 1) - kk * (kk - 1) / 2 - lane - 1) * (int)sizeof(GT);   // byte offset (from gl) of this lane's packed row, entry "lane 0"
      unsigned long long livem = (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull);        // lanes not yet passed
      unsigned long long am = 0ull;
      double rq = r[q], ro = r[1], z = rq - zc[q];
      const double haq = ha[q], hrq = hr[q], drejd = (double)lc[q].drej;
      const float b0q = lc[q].b0;
      // (uniform gotos: one exit test at the top, the sliver's code out of line, one back edge)
      {
        unsigned long long cand, accm;
        int js;
        float b1;
        double cd, corr, gm0;
        GT g0, g1;
      qr_top:
        cand = livem & ~__ballot(fabs(z) < hrq);
        if (cand == 0ull) goto qr_done;
        js = (int)__builtin_ctzll(cand);
        {   // its packed row, requested at once (a reject in the sliver wastes the request, nothing else)
          const GT *row = reinterpret_cast<const GT *>(reinterpret_cast<const char *>(gl) + __builtin_amdgcn_readlane(tab, js));
          g0 = row[0]; g1 = row[64];
        }
        accm = __ballot(fabs(z) > haq);
        b1 = lane_b1(rq, lc[q]);
        cd = (double)(b1 - b0q) - drejd;                          // per lane: what its accepted step changes beyond the speculated one
        livem &= (~1ull << js);
        if (__builtin_expect(!((accm >> js) & 1ull), 0)) goto qr_sliver;
      qr_apply:
        corr = readlane_f64(cd, js);
        am |= (1ull << js);
        if constexpr (CEN) {
          const double xk = readlane_f64(cs[q], js) * ninv;   // mean of column k
          *cenU = fma(xk, corr, *cenU);
          gm0 = (lane > js) ? ((double)g0 - cs[q] * xk) : 0.0;
          rq = fma(-gm0, corr, rq);
          z = fma(-gm0, corr, z);
          if (q == 0) ro = fma(-((double)g1 - cs[1] * xk), corr, ro);
          goto qr_top;
        }
        gm0 = (double)((lane > js) ? g0 : (GT)0);
        rq = fma(-gm0, corr, rq);
        z = fma(-gm0, corr, z);
        if (q == 0) ro = fma(-(double)g1, corr, ro);   // (with one lane group r[1] is never read; the loads stay inside the buffer's slack)
        goto qr_top;
      qr_sliver:   // between the radii: the full test decides; a reject leaves its speculated step standing, nothing moves
        if ((__ballot(lane_accept(rq, b1, lc[q], flags, Cc, odds, one_minus_pi, rng, iter)) >> js) & 1ull) goto qr_apply;
        goto qr_top;
      qr_done:;
      }
      r[q] = rq;
      if (q == 0) r[1] = ro;
      accmask[q] |= am;
    }
  }
}

// k_spec: the r-independent speculative terms of every block of a sweep (one workgroup of 128 threads per block,
// thread = marker j; Gram rows are read coalesced across j; fixed summation order k ascending)
template <typename GT>
__global__ __launch_bounds__(128) void k_spec(const SweepArgs a, int blk_begin, int select) {
  if (a.redo_only ? (a.sc->redo == 0u) : (a.gate3 > 0.0f && a.sc->inc_rate < a.gate3)) return;   // this sweep is k_sweep3's (or: nothing to redo)
  const int blk = blk_begin + blockIdx.x, j = threadIdx.x, m = a.m;
  const int mB = min(m, a.p - blk * m);
  const GT *G = reinterpret_cast<const GT *>(a.gram) + (size_t)blk * m * m;
  SpecBuf &sp = a.ps.spec[blk];
  __shared__ float dr[128], drp[128];
  const StageBuf &st = a.ps.blocks[blk];
  dr[j] = (j < mB) ? st.drej[j] : 0.0f;
  drp[j] = (blk > blk_begin) ? a.ps.blocks[blk - 1].drej[j] : 0.0f;   // previous block is always a full block
  __syncthreads();
  double s = 0.0, xs = 0.0, gjj = 0.0;
  const bool cen = (a.flags & SWF_CENTRE) != 0 && select;   // implicitly centred columns (sweep3.hip.h): the rejected steps' share of -(s_j / n) sum(e) goes into spec,
  __shared__ double cs_s[128];                                // the Gram diagonal becomes |x_j - mean_j|^2; cpre: k_cen_tot / k_cen_scan on the UNROUNDED steps
  const double sj = (cen && j < mB) ? (double)a.csum[blk * m + j] : 0.0;
  if (cen) { cs_s[j] = sj * (double)dr[j]; __syncthreads(); }
  if (j < mB) {
    gjj = (double)G[(size_t)j * m + j];
    if (cen) {
      double ib = 0.0;
      for (int k = 0; k < j; ++k) ib += cs_s[k];
      s = -(sj * a.ninv) * (a.cpre[blk] + ib);   // (spec is subtracted from the dot: + (s_j / n) * the rejected steps' share of the drop of sum(e))
      gjj -= sj * sj * a.ninv;
    }
    if (select) {
      for (int k = 0; k < j; ++k) s = fma((double)G[(size_t)k * m + j], (double)dr[k], s);
      if (blk > blk_begin) {
        const GT *Gx = reinterpret_cast<const GT *>(a.gramx) + (size_t)blk * m * m;
        for (int k = 0; k < m; ++k) xs = fma((double)Gx[(size_t)k * m + j], (double)drp[k], xs);
      }
    }
  }
  sp.spec[j] = s; sp.xspec[j] = xs; sp.gjj[j] = gjj;
  if (select && a.ps.quick) {   // the rounds' two compares (lane_quick): off the sequencer's one serial wave, some 300 instructions per lane group
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
  for (int dist = 2; dist < a.lag; ++dist) {   // further cross terms of the deeper pipelines
    double xs2 = 0.0;
    __syncthreads();
    drp[j] = (blk - dist >= blk_begin) ? a.ps.blocks[blk - dist].drej[j] : 0.0f;
    __syncthreads();
    if (j < mB && select && blk - dist >= blk_begin) {
      const GT *Gxd = reinterpret_cast<const GT *>(dist == 2 ? a.gramx2 : a.gramx3) + (size_t)blk * m * m;
      for (int k = 0; k < m; ++k) xs2 = fma((double)Gxd[(size_t)k * m + j], (double)drp[k], xs2);
    }
    (dist == 2 ? a.xspec2 : a.xspec3)[(size_t)blk * SW_MAXM + j] = xs2;
  }
}

template <typename XT, bool SELECT>
__global__ __launch_bounds__(SW_THREADS) void k_sweep(const SweepArgs a) {
  using GT = typename XTraits<XT>::GT;
  constexpr int PER = XTraits<XT>::PER16;
  constexpr int GPT = 16 / sizeof(GT);            // Gram entries per 16-byte chunk
  constexpr int MAXMX = XTraits<XT>::MAXM;
  constexpr int GCH = (MAXMX * MAXMX / GPT + (SW_THREADS - 64) - 1) / (SW_THREADS - 64);  // chunks per prefetch thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int m = a.m, R = a.R, K = a.K;
  const int Rp = tile_rp<XT>(R);
  const int row0 = wg * R;
  if ((a.flags & SWF_DEBUG_WITHHOLD) && wg == 0 && K > 1) return;

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
  double *spec_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  double *gdiag_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  double *delta_s = reinterpret_cast<double *>(smem + off); off += SW_MAXM * sizeof(double);
  float *bnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  float *dnew_s = reinterpret_cast<float *>(smem + off); off += SW_MAXM * sizeof(float);
  int *ctrl_s = reinterpret_cast<int *>(smem + off);   // (not volatile: a volatile access stays a flat_ one and waits on vmcnt)

  const XT *X = reinterpret_cast<const XT *>(a.X) + (size_t)wg * a.p * R;   // this workgroup's slab stream
  const GT *gram = reinterpret_cast<const GT *>(a.gram);

  // scalars of this iteration
  const float Cc = a.sc->C, odds = a.sc->odds;
  const double dscale = (a.flags & SWF_DELTA2) ? 2.0 : 1.0;   // emBA's doubled residual update (affine path)
  const float one_minus_pi = 1.0f - a.sc->pi, Sb = a.sc->Sb;

  const int nb = a.blk_end - a.blk_begin;
  const int mpad = (m <= 64) ? 64 : 128;
  const int ngroups = SW_THREADS / mpad;     // row groups of the dot phase
  const int rows_per_group = R / ngroups;    // R is a multiple of 128, so a multiple of 16
  const int gchunks = m * m / GPT;           // 16-byte chunks of one Gram block

  uint4 tp0 = make_uint4(0, 0, 0, 0), tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0;   // tile prefetch registers
  for (int i = tid; i < R; i += SW_THREADS) e_s[i] = a.e[row0 + i];
  if (tid == 0) ctrl_s[0] = 1;

  // speculative correction of a block: spec[j] = sum_{k<j} G_kj * drej_k  (SELECT only; 4 k-ranges per marker)
  auto spec_matvec = [&](const StageBuf &stn, int mBn) {
    if (SELECT) {
      const int part = tid >> 7, j = tid & 127;
      if (j < mBn) {
        double acc = 0.0;
        const int k0 = part * 32, k1 = min(k0 + 32, min(j, mBn));
        for (int k = k0; k < k1; ++k) acc = fma((double)gram_s[(size_t)k * m + j], (double)stn.drej[k], acc);
        part_s[part * SW_MAXM + j] = acc;
      }
      __syncthreads();
      if (tid < mBn) spec_s[tid] = (part_s[tid] + part_s[SW_MAXM + tid]) + (part_s[2 * SW_MAXM + tid] + part_s[3 * SW_MAXM + tid]);
    }
  };

  // prologue: tile, constants and Gram block of the first block
  {
    const int j0 = a.blk_begin * m;
    const int mB = min(m, a.p - j0);
    if (wave != 0) { TILE_ISSUE(j0, mB); TILE_COMMIT(tile0, mB); }
    stage_block(stage[0], a.blk_begin, a, tid, SW_THREADS);
    const GT *gsrc = gram + (size_t)a.blk_begin * m * m;
    for (int c = tid; c < gchunks; c += SW_THREADS) {
      GT v[GPT];
      *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(gsrc + (size_t)c * GPT);
      if constexpr (GPT == 4) put_gram4<GT>(gram_s, gdiag_s, m, c * 4, v);
      else {
        const int e0 = c * 2, row = e0 / m, col = e0 - row * m;
        gram_s[e0] = (col > row) ? v[0] : (GT)0; gram_s[e0 + 1] = (col + 1 > row) ? v[1] : (GT)0;
        if (col == row) gdiag_s[row] = (double)v[0];
        if (col + 1 == row) gdiag_s[row] = (double)v[1];
      }
    }
    __syncthreads();
    spec_matvec(stage[0], mB);
  }
  double sum_d = 0.0, sum_b2 = 0.0;  // meaningful in wave 0 only
  if (wave != 0 && nb > 1) {          // prefetch waves: tile of block s+1 in flight, committed during block s
    const int j1 = (a.blk_begin + 1) * m;
    TILE_ISSUE(j1, min(m, a.p - j1));
  }
#ifdef BWGR_STAMPS
  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

  for (int s = 0; s < nb; ++s) {
    const int blk = a.blk_begin + s;
    const int j0 = blk * m;
    const int mB = min(m, a.p - j0);
    const int buf = s & 1;
    XT *tile = buf ? tile1 : tile0;
    XT *tile_next = buf ? tile0 : tile1;
    StageBuf &st = stage[buf];
    __syncthreads();  // tile, stage[buf], gram_s, spec_s, e_s of this block are in place
    STAMP(0);

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
    STAMP(1);

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
        st_agent_u32(a.xflags + (size_t)wg * SW_FLAG_STRIDE, epoch);
      if (wave == 0) {
        uint32_t *abortw = a.xflags + (size_t)K * SW_FLAG_STRIDE;
        const uint64_t t0 = wall_clock64();
        int ok = 0;
        for (;;) {
          bool all_here = true;
          for (int w = lane; w < K; w += 64) {
            const uint32_t f = ld_agent_u32(a.xflags + (size_t)w * SW_FLAG_STRIDE);
            all_here = all_here && (f >= epoch);
          }
          if (__all(all_here)) { ok = 1; break; }
          const uint32_t ab = ld_agent_u32(abortw);
          if (__any(ab != 0)) break;
          if (wall_clock64() - t0 > SW_TIMEOUT_TICKS) {
            if (lane == 0) st_agent_u32(abortw, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) ctrl_s[0] = ok;
      }
      __syncthreads();
      if (ctrl_s[0] == 0) {  // uniform: some workgroup never arrived; give up, report
        if (tid == 0) a.sc->error = 1u;
        return;
      }
      {   // 4 threads per marker, each gathers a quarter of the workgroups (<= 16 loads in one batch, fixed order);
          // quarters are combined in a fixed order too, so every workgroup forms the same bits
        const int qd = tid >> 7, t = tid & 127;
        const int wq = (K + 3) >> 2;
        for (int wbase = 0; wbase < wq; wbase += 16) {
          double v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int w = qd * wq + wbase + u;
            v[u] = (t < mB && wbase + u < wq && w < K) ? ld_agent_f64(slot + (size_t)w * SW_MAXM + t) : 0.0;
          }
          double r = 0.0;
#pragma unroll
          for (int u = 0; u < 16; ++u) r += v[u];
          if (wbase == 0) part_s[qd * SW_MAXM + t] = r; else part_s[qd * SW_MAXM + t] += r;
        }
        __syncthreads();
        if (tid < mB) r0_s[tid] = (part_s[tid] + part_s[SW_MAXM + tid]) + (part_s[2 * SW_MAXM + tid] + part_s[3 * SW_MAXM + tid]);
      }
    } else {
      if (tid < mB) r0_s[tid] = mine;
    }
    __syncthreads();
    STAMP(2);

    // ---- wave 0: the in-block recurrence; waves 1..7: stream in the next block ----
    GT gpre[GCH][GPT];   // next Gram block in flight (prefetch waves)
    const bool have_next = (s + 1 < nb);
    if (wave == 0) {
      const int ngrp = (mB + 63) >> 6;
      double r[2];
      LaneConst lc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        const bool live = t < mB;
        r[q] = live ? (SELECT ? (r0_s[t] - spec_s[t]) : r0_s[t]) : 0.0;
        lc[q].b0 = live ? st.b0[t] : 0.0f; lc[q].xxb0 = live ? st.xxb0[t] : 0.0f;
        lc[q].b2 = live ? st.b2[t] : 0.0f; lc[q].drej = live ? st.drej[t] : 0.0f;
        lc[q].rden = live ? st.rden[t] : 1.0; lc[q].sdz1 = live ? st.sdz1[t] : 0.0;
        lc[q].gjj = live ? gdiag_s[t] : 0.0; lc[q].mk = a.marker0 + (uint32_t)(j0 + t);
        lc[q].tacc = live ? st.tacc[t] : -INFINITY; lc[q].trej = live ? st.trej[t] : -INFINITY;   // dead lanes: certain reject
      }
      unsigned long long accmask[2] = {0ull, 0ull};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q < ngrp) {
          const int base = 64 * q;
          const int cnt = min(64, mB - base);
          if (!SELECT) {
            // every marker takes its in-model draw: finalize lanes in order, one broadcast per marker
            GT gnext0 = gram_s[(size_t)base * m + base + lane];
            GT gnext1 = (q == 0 && ngrp > 1) ? gram_s[(size_t)base * m + 64 + lane] : (GT)0;
            for (int l = 0; l < cnt; ++l) {
              const GT g0 = gnext0, g1 = gnext1;
              if (l + 1 < cnt) {   // Gram row of the next marker: independent of the recurrence, fetch early
                gnext0 = gram_s[(size_t)(base + l + 1) * m + base + lane];
                if (q == 0 && ngrp > 1) gnext1 = gram_s[(size_t)(base + l + 1) * m + 64 + lane];
              }
              const float b1 = lane_b1(r[q], lc[q]);
              const float dl = b1 - lc[q].b0;
              const double dd = (double)readlane_f32(dl, l);
              r[q] = fma(-(double)g0 * dscale, dd, r[q]);   // (the scaled Gram entry is ready before dd: off the chain)
              if (q == 0 && ngrp > 1) r[1] = fma(-(double)g1 * dscale, dd, r[1]);
            }
          } else {
            // speculative rounds: lanes >= front decide as if every earlier unfinalized marker is rejected
            // (their step drej is already inside r); the first accepted lane is fixed up, the rest re-decide.
            int front = 0;
            while (front < cnt) {
              const float b1 = lane_b1(r[q], lc[q]);
              const bool acc = lane_accept(r[q], b1, lc[q], a.flags, Cc, odds, one_minus_pi, a.rng, a.iter);
              const unsigned long long bal = __ballot(acc && lane >= front && lane < cnt);
              if (bal == 0ull) break;
              const int js = __ffsll((long long)bal) - 1;
              const float corr_f1 = b1 - lc[q].b0;          // the accepted step
              const double corr = (double)readlane_f32(corr_f1, js) - (double)readlane_f32(lc[q].drej, js);
              const GT *grow = gram_s + (size_t)(base + js) * m;
              r[q] = fma(-(double)grow[base + lane], corr, r[q]);
              if (q == 0 && ngrp > 1) r[1] = fma(-(double)grow[64 + lane], corr, r[1]);
              accmask[q] |= (1ull << js);
              front = js + 1;
            }
          }
        }
      }
      // every lane's r is now final for its own marker: produce the outputs in parallel
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 64 * q + lane;
        if (t < mB) {
          const float b1 = lane_b1(r[q], lc[q]);
          const bool inc = SELECT ? (((accmask[q] >> lane) & 1ull) != 0ull) : true;
          const float bn = inc ? b1 : lc[q].b2;
          const float dn = inc ? 1.0f : 0.0f;
          delta_s[t] = (double)((bn - lc[q].b0) * (float)dscale); bnew_s[t] = bn; dnew_s[t] = dn;
          sum_d += (double)dn;
          sum_b2 = fma((double)bn, (double)bn, sum_b2);
        }
      }
      STAMP(6);
    } else {
      if (have_next) {
        const int j1 = j0 + m;
        const int mB1 = min(m, a.p - j1);
        const GT *gsrc = gram + (size_t)(blk + 1) * m * m;
#ifdef BWGR_STAMPS
        unsigned long long w1t0 = 0;
        if (tid == 64) w1t0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int k = 0; k < GCH; ++k) {
          const int c = (tid - 64) + k * (SW_THREADS - 64);
          if (c < gchunks) *reinterpret_cast<uint4 *>(gpre[k]) = *reinterpret_cast<const uint4 *>(gsrc + (size_t)c * GPT);
        }
#ifdef BWGR_STAMPS
        if (tid == 64) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[10] += t_ - w1t0; }
#endif
        // constants of the next block: load issued before the tile's loads are consumed (one latency, not two)
        constexpr int NCH = (int)(sizeof(StageBuf) / 16);
        static_assert(NCH <= SW_THREADS - 64, "one StageBuf chunk per prefetch thread");
        uint4 spre = make_uint4(0, 0, 0, 0);
        if (tid - 64 < NCH) spre = reinterpret_cast<const uint4 *>(a.ps.blocks + blk + 1)[tid - 64];
#ifdef BWGR_STAMPS
        if (tid == 64) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[11] += t_ - w1t0; }
#endif
        TILE_COMMIT(tile_next, mB1);                                                     // issued one block ago
        if (s + 2 < nb) { const int j2 = j1 + m; TILE_ISSUE(j2, min(m, a.p - j2)); }
        if (tid - 64 < NCH) reinterpret_cast<uint4 *>(&stage[buf ^ 1])[tid - 64] = spre;
#ifdef BWGR_STAMPS
        if (tid == 64) {
          const unsigned long long t1 = __builtin_amdgcn_s_memtime();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned long long t2 = __builtin_amdgcn_s_memtime();
          ph[8] += t1 - w1t0; ph[9] += t2 - t1;
        }
#endif
      }
    }
    __syncthreads();
    STAMP(3);

    // ---- outputs of the block (workgroup 0 is the only writer of marker state) ----
    if (wg == 0 && tid < mB) {
      const float bn = bnew_s[tid];
      a.b[j0 + tid] = bn;
      a.d[j0 + tid] = dnew_s[tid];
      if (a.flags & SWF_VB_VEC) a.vb[j0 + tid] = (float)((double)(Sb + bn * bn) / st.chi[tid]);
    }
    // ---- next Gram block into LDS (the recurrence no longer reads gram_s) ----
    if (wave != 0 && have_next) {
#pragma unroll
      for (int k = 0; k < GCH; ++k) {
        const int c = (tid - 64) + k * (SW_THREADS - 64);
        if (c < gchunks) {
          if constexpr (GPT == 4) put_gram4<GT>(gram_s, gdiag_s, m, c * 4, gpre[k]);
          else {
            const int e0 = c * 2, row = e0 / m, col = e0 - row * m;
            gram_s[e0] = (col > row) ? gpre[k][0] : (GT)0; gram_s[e0 + 1] = (col + 1 > row) ? gpre[k][1] : (GT)0;
            if (col == row) gdiag_s[row] = (double)gpre[k][0];
            if (col + 1 == row) gdiag_s[row] = (double)gpre[k][1];
          }
        }
      }
    }
    __syncthreads();   // gram_s / gdiag_s of the next block and stage[buf^1] are complete
    // ---- slab update e_i -= sum_j x_ij delta_j (rows x marker parts, fp64, x*delta exact) fused with the next block's
    //      speculative correction spec[j] = sum_{k<j} G_kj * drej_k (SELECT): one pass, one reduction barrier ----
    {
      const int nparts = (R >= SW_THREADS) ? 1 : SW_THREADS / R;        // R is a multiple of 128; threads beyond nparts*R idle
      const int per = (mB + nparts - 1) / nparts;
      double *upd_s = part_s;                                           // nparts*R <= 512 doubles
      double *spp_s = part_s + (SW_THREADS / 64) * SW_MAXM / 2;         // 4 x 128 doubles
      const int mBn = have_next ? min(m, a.p - (j0 + m)) : 0;
      for (int i0 = 0; i0 < R; i0 += SW_THREADS) {
        const int part = (R >= SW_THREADS) ? 0 : tid / R;
        const int i = (R >= SW_THREADS) ? i0 + tid : tid - part * R;
        const bool active = (i < R) && (part < nparts);
        double acc = 0.0;
        if (active) {
          const int ja = part * per, jb = min(mB, ja + per);
          const XT *tp = tile + i;
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
        if (SELECT && i0 == 0 && have_next) {
          const StageBuf &stn = stage[buf ^ 1];
          const int sp = tid >> 7, j = tid & 127;
          if (j < mBn) {
            double sa = 0.0;
            const int k0 = sp * 32, k1 = min(k0 + 32, min(j, mBn));
            for (int k = k0; k < k1; ++k) sa = fma((double)gram_s[(size_t)k * m + j], (double)stn.drej[k], sa);
            spp_s[sp * SW_MAXM + j] = sa;
          }
        }
        if (nparts == 1 && !(SELECT && i0 == 0 && have_next)) { if (active) e_s[i] -= acc; }
        else {
          if (nparts > 1 && active) upd_s[part * R + i] = acc;
          __syncthreads();
          if (nparts == 1) { if (active) e_s[i] -= acc; }
          else if (tid < R) { double t = 0.0; for (int q = 0; q < nparts; ++q) t += upd_s[q * R + tid]; e_s[tid] -= t; }
          if (SELECT && i0 == 0 && have_next && tid < mBn)
            spec_s[tid] = (spp_s[tid] + spp_s[SW_MAXM + tid]) + (spp_s[2 * SW_MAXM + tid] + spp_s[3 * SW_MAXM + tid]);
          if (i0 + SW_THREADS < R) __syncthreads();   // upd_s is reused by the next row chunk
        }
      }
    }
    STAMP(4);
    STAMP(5);
  }
  __syncthreads();
  for (int i = tid; i < R; i += SW_THREADS) a.e[row0 + i] = e_s[i];
  if (wg == 0 && wave == 0) {   // per-lane partial sums of wave 0 -> chain scalars
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_d += __shfl_down(sum_d, o, 64); sum_b2 += __shfl_down(sum_b2, o, 64); }
    if (lane == 0) { a.sc->sum_d += sum_d; a.sc->sum_b2 += sum_b2; }
  }
#ifdef BWGR_STAMPS
  if (wg == 0 && tid == 0 && a.stamps) for (int k = 0; k < 8; ++k) a.stamps[k] += ph[k];
  if (wg == 0 && tid == 64 && a.stamps) for (int k = 8; k < 12; ++k) a.stamps[k] += ph[k];
#endif
}

}  // namespace bwgr
