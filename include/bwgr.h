/*
 * include/bwgr.h -- C ABI of libbwgr_hip.so, the MI355X (gfx950) Gibbs sweep engine that stands
 * behind bWGR's wgr()/KMUP and the standalone Bayes* samplers.
 *
 * Plain pointers and sizes only; no torch / Rcpp / Eigen types.  Every entry point names the
 * reference interface it replaces (paths relative to the bWGR source tree).  The reference-side
 * binding (an R .Call shim, plus the ctypes stub used by this repo's host layer) is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - All functions return 0 (BWGR_OK) or a bwgr_status code; bwgr_last_error() gives the text.
 *     HIP failures never abort the process (the reference's BEGIN_RCPP/END_RCPP turns C++
 *     exceptions into R conditions, src/RcppExports.cpp:17,30; the shim maps non-zero to Rf_error).
 *   - Inputs are never modified unless documented as in/out; outputs are caller-allocated
 *     (the Rcpp glue passes every Eigen argument by value, src/RcppExports.cpp:20-27).
 *   - X is column-major n x p with leading dimension ldx (R / Eigen::MatrixXf layout).
 *   - `seed` replaces R's global RNG stream (Rcpp::RNGScope, src/RcppExports.cpp:19); the R
 *     front-end derives it from unif_rand() so set.seed() still governs repeatability.
 *   - There is no CPU fallback: without a gfx950 device every compute entry returns BWGR_ENODEV.
 */
#ifndef BWGR_H
#define BWGR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BWGR_ABI_VERSION 1

enum bwgr_status {
  BWGR_OK = 0,
  BWGR_EINVAL = 1,   /* bad argument */
  BWGR_EHIP = 2,     /* a HIP runtime call failed */
  BWGR_ENOMEM = 3,
  BWGR_ETIMEOUT = 4, /* an in-kernel workgroup exchange gave up (bounded spin) */
  BWGR_ENODEV = 5,   /* no usable GPU */
  BWGR_ERANGE = 6    /* a fixed-point sweep left its range and could NOT be redone.  Since round 3 a sweep of the single-chain entry
                        points that leaves the range is redone on the fp64 residual from the state it started with and the chain goes
                        on (bwgr_chain_redo_count says how often); this status remains for the paths without that recovery: a pair
                        sweep (bwgr_chain_run_pair -- run the two chains unpaired instead; the host layer's fit_many does), the debug
                        abort hook, and a recovery whose snapshot could not be allocated.  The chain's state is then invalid. */
};
enum bwgr_xtype { BWGR_X_I8 = 0, BWGR_X_F32 = 1, BWGR_X_F64 = 2 }; /* F64 (an R numeric matrix) is narrowed
                                                                     to float on upload, as the Rcpp glue does
                                                                     on every call (src/RcppExports.cpp:20) */
enum bwgr_memloc { BWGR_HOST = 0, BWGR_DEVICE = 1 };
enum bwgr_model {
  BWGR_BAYESA = 0,   /* src/Rcpp20260726ai.cpp:589-635 */
  BWGR_BAYESB = 1,   /* :638-699 */
  BWGR_BAYESC = 2,   /* :702-759 */
  BWGR_BAYESL = 3,   /* :762-809 */
  BWGR_BAYESRR = 4,  /* :812-855 */
  BWGR_BAYESCPI = 5, /* :858-921 */
  BWGR_BAYESDPI = 6  /* :924-987 */
};
enum bwgr_rng_mode { BWGR_RNG_PHILOX = 0, BWGR_RNG_DEGENERATE = 1 /* z=0, chi2=mean, u=0.5 (tests) */ };

typedef struct bwgr_panel bwgr_panel; /* genotype matrix resident in HBM + per-marker setup */
typedef struct bwgr_chain bwgr_chain; /* one MCMC chain: residual, effects, variances, posterior sums */

int bwgr_abi_version(void);
const char *bwgr_last_error(void);
int bwgr_device_count(int *count);

/* ---- panel: X staged once, column-major in HBM ------------------------------------------------
 * Replaces the per-call SEXP -> Eigen::MatrixXf conversion of X (src/RcppExports.cpp:20, :198 ...).
 * block = markers per exact block (0 = auto, <= 128); nwg = row-slab workgroups (0 = auto).
 * Builds xx, vx, MSx (src/Rcpp20260726ai.cpp:593-598) and the block-diagonal Gram used by the
 * blocked sweep. */
int bwgr_panel_create(bwgr_panel **out, const void *X, int xtype, int memloc, int64_t n, int64_t p, int64_t ldx,
                      int device, int block, int nwg);
int bwgr_panel_destroy(bwgr_panel *P);
int bwgr_panel_set_stream(bwgr_panel *P, void *hip_stream); /* NULL = the default stream */
/* A second handle on the same resident genotypes: shares X, the Gram arrays and xx/vx with `src` (read-only during
 * sweeps) and owns its own sweep scratch and its own non-blocking stream, so chains on `src` and on its clones run
 * side by side on disjoint compute units (a sweep occupies nwg + 1 + feeders of the 256).  This is how the callers
 * that fit many models on one X -- mcmcCV's folds x models loop, R/cv.R:113-216 -- fill the chip.  Destroy the clones
 * before `src`.  bwgr_panel_max_concurrent: how many sweeps of this geometry fit at once; for selection models on a panel
 * with k_sweep3 it counts the larger of the two engines a chain may run (K3 + 1, or nwg + 1 + feeders above the engine
 * threshold).  bwgr_panel_max_pairs: how many pairs (bwgr_chain_run_pair, K3 + 2 units each) fit, 0 without k_sweep3.
 *
 * Occupancy guard: a sweep's workgroups wait for one another, so all of them have to be resident at once, beside the
 * sweeps other handles (panels, clones, pairs) have in flight on other streams of the same device.  Every sweep entry
 * point checks that before it enqueues anything -- grid against hipOccupancyMaxActiveBlocksPerMultiprocessor x compute
 * units, less the units of the sweeps in flight -- and returns BWGR_EINVAL if the launch would not fit (the chain is left
 * as it was; wait for the others and call again).  Without the guard such a launch spins to BWGR_ETIMEOUT.
 * BWGR_OCC_GUARD=0 switches it off.  bwgr_debug_occupancy_fits is the guard's arithmetic (a host function: `grid`
 * workgroups at `per_cu` per unit need ceil(grid / per_cu) units, which must fit `cus` less `busy`). */
int bwgr_panel_clone(bwgr_panel **out, bwgr_panel *src);
int bwgr_panel_max_concurrent(const bwgr_panel *P, int selection, int *count);
int bwgr_panel_max_pairs(const bwgr_panel *P, int *pairs);
int bwgr_debug_occupancy_fits(int grid, int per_cu, int cus, int busy, int *need);
/* host arithmetic of one more launch rule: the LDS-DMA streamers of k_sweep3 address a launch's columns by 32-bit lane offsets, so a
 * launch over `ncols` columns of `slab_rows`-row slabs takes them only while ncols * slab_rows < 2^32 (else the register-path streamers
 * with 64-bit offsets).  Returns 1 / 0. */
int bwgr_debug_stream3_dma(int64_t ncols, int64_t slab_rows);
/* geometry actually chosen: info[0]=n, [1]=p, [2]=ld (padded rows), [3]=block, [4]=nwg, [5]=slab rows,
 * [6]=bytes of X resident, [7]=bytes of Gram resident */
int bwgr_panel_info(const bwgr_panel *P, int64_t info[8]);
/* how a sweep over this panel is pipelined (no reference counterpart; reporting only): info[0]=kernel generation (1 k_sweep,
 * 2 k_sweep2, 3 k_sweep3 -- selection sweeps of sparse chains --, 4 k_sweep2w -- affine sweeps as a triangular product --),
 * [1]=pipeline depth in blocks (a block's dots lag the chain by this many blocks), [2]=q feeder workgroups,
 * [3]=bits of the Gram entries the sequencer stages (16|32; 0 for float panels).  selection != 0: BayesB/C/Cpi/Dpi-type
 * sweeps (inclusion indicators), else the affine ones (BayesA/L/RR). */
int bwgr_panel_pipeline(const bwgr_panel *P, int selection, int info[4]);
/* xx[j] = |X_j|^2, vx[j] = fvar(X_j), MSx = sum vx   (host outputs; any may be NULL) */
int bwgr_panel_stats(bwgr_panel *P, float *xx, float *vx, float *MSx);

/* ---- KMUP: one Gibbs sweep over all markers ---------------------------------------------------
 * Replaces SEXP KMUP(X,b,d,xx,e,L,Ve,pi), src/Rcpp20260726ai.cpp:12-38 / _bWGR_KMUP,
 * src/RcppExports.cpp:16-31.  b, d (p) and e (n) are in/out host vectors; xx, L (p) inputs.
 * `iter` is the iteration word of the RNG counter (wgr passes its loop index - 1).
 * Inclusion probability uses the stable form 1/(1+pi/(1-pi)*exp(C(|e2|^2-|e1|^2))), which equals the
 * reference's cj/(cj+dj) wherever that does not underflow to NaN. */
int bwgr_kmup(bwgr_panel *P, float *b, float *d, const float *xx, float *e, const float *L, float Ve, float pi,
              uint64_t seed, uint32_t iter, int rng_mode);

/* ---- KMUP2: the same sweep on a row subsample -----------------------------------------------------
 * Replaces SEXP KMUP2(X,Use,b,d,xx,E,L,Ve,pi), src/Rcpp20260726ai.cpp:41-77 / _bWGR_KMUP2,
 * src/RcppExports.cpp:34-50 (R/RcppExports.R:8-10).  Use: nuse 0-based row ids of the resident panel (wgr passes
 * sort(sample(n, n*bag, rp)) - 1, R/wgr.R:68); b, d (p) in/out; xx, L (p) and E (the panel's n rows) inputs; e_out
 * receives the nuse residuals of the subsample (:76).  Reference quirks kept: the conditional mean's numerator adds b0,
 * not xx*b0, and the denominator is xx*bg + L with bg = n/nuse (:47, :59). */
int bwgr_kmup2(bwgr_panel *P, const int *Use, int64_t nuse, float *b, float *d, const float *xx, const float *E,
               float *e_out, const float *L, float Ve, float pi, uint64_t seed, uint32_t iter, int rng_mode);

/* ---- fused chains: BayesA/B/C/L/RR/Cpi/Dpi ---------------------------------------------------------
 * Replaces SEXP Bayes*(y, X, it, bi, [pi,] df, R2), src/Rcpp20260726ai.cpp:589-987 /
 * _bWGR_BayesA.._bWGR_BayesDpi, src/RcppExports.cpp:177-290.  y: n floats (host or device per
 * memloc).  it/bi are floats cast to int, as in the reference (:611, :642).  pi is ignored by
 * A/L/RR/Cpi/Dpi. */
int bwgr_chain_create(bwgr_chain **out, bwgr_panel *P, int model, const float *y, int memloc, float it, float bi,
                      float pi, float df, float R2, uint64_t seed, int rng_mode);
int bwgr_chain_destroy(bwgr_chain *C);
/* run the next `iters` MCMC iterations (sweep + intercept + variance draws + posterior sums);
 * asynchronous on the panel's stream */
int bwgr_chain_run(bwgr_chain *C, int iters);
/* Two chains of one resident panel (one on the panel, one on a clone of it -- or on two clones) advanced in lockstep, `iters`
 * iterations each: one set of streamer workgroups, one pass over the genotypes, serves both (k_sweep3p); each chain's results are
 * bit-identical to a run of its own.  Selection models (BayesB/C/Cpi/Dpi) on int8 panels that have k_sweep3; BWGR_EINVAL otherwise.
 * No reference counterpart: this is how the callers that fit many models on one X (mcmcCV, /root/reference/R/cv.R:113-216; replicate
 * chains) use fewer compute units per chain. */
int bwgr_chain_run_pair(bwgr_chain *C0, bwgr_chain *C1, int iters);
/* wait for the stream and report in-kernel exchange failures */
int bwgr_chain_sync(bwgr_chain *C);
/* iterations completed so far */
int bwgr_chain_iterations(const bwgr_chain *C, int *done);
/* posterior means and fitted values, the reference's return list (host outputs, any may be NULL):
 *   mu, b[p], d[p], hat[n], vb[p] (A/B/L/Dpi) or vb[1] (C/RR/Cpi), ve, h2, MSx, pi (Cpi/Dpi), PVAL[p] */
int bwgr_chain_result(bwgr_chain *C, float *mu, float *b, float *d, float *hat, float *vb, float *ve, float *h2,
                      float *MSx, float *pi, float *pval);
/* current chain state (host outputs, any may be NULL): b[p], d[p], e[n], vb[p] (common variance
 * replicated), scal[4] = {mu, ve, vb_common, pi} */
int bwgr_chain_state(bwgr_chain *C, float *b, float *d, float *e, float *vb, float *scal);
/* device-time of the sweep kernel alone, averaged over the launches since the last call (ms);
 * measured with hipEvents on the stream the kernel runs on */
int bwgr_chain_sweep_ms(bwgr_chain *C, float *avg_ms, int *launches);
/* sweeps of this chain that left the fixed-point range of their engine and were redone on the fp64 residual (the reference's update,
 * src/Rcpp20260726ai.cpp:681, has no such failure: the redo keeps the chain the same chain; this only reports how often it happened) */
int bwgr_chain_redo_count(bwgr_chain *C, int *count);

/* ---- marker-sharded chains (one rank per GPU) ----------------------------------------
 * The panel holds this rank's columns [marker0, marker0 + p_local) of a p_total-marker panel; the residual is
 * replicated on every rank in e_ext (device, `ld` doubles from bwgr_panel_info, caller-owned, e.g. a torch tensor,
 * so that the caller can all-reduce residual deltas with RCCL between block ranges).  RNG counters use global marker
 * ids, MSx_total is the all-rank sum of the panels' MSx.  An iteration is then
 *     for each range: bwgr_chain_sweep_blocks(C, lo, hi); <caller: all-reduce (e - e_at_range_start)>;
 *     bwgr_chain_get_sums(C, s); <caller: all-reduce s>; bwgr_chain_end_iteration(C, s_total);
 * With one rank and one range this is exactly bwgr_chain_run(C, 1).  With several ranks markers on different ranks
 * are updated against a residual that is only synchronised at range boundaries: a partitioned Gibbs sampler, NOT
 * the reference's chain (parity is statistical). */
int bwgr_chain_create_sharded(bwgr_chain **out, bwgr_panel *P, int model, const float *y, int memloc, float it, float bi,
                              float pi, float df, float R2, uint64_t seed, int rng_mode, int64_t marker0,
                              int64_t p_total, float MSx_total, double *e_ext);
int bwgr_chain_sweep_blocks(bwgr_chain *C, int blk_begin, int blk_end);
/* one exchange round in two calls: round_sweep remembers e, sweeps [blk_begin, blk_end) (empty range: nothing) and writes
 * delta = e - e_before to delta_dev (ld doubles, device); the caller all-reduces delta; round_apply sets e = e_before + delta */
int bwgr_chain_round_sweep(bwgr_chain *C, int blk_begin, int blk_end, double *delta_dev);
/* device-side forms of get_sums / end_iteration(sums_total): sums_dev = two doubles on the chain's device, all-reduced in place */
int bwgr_chain_get_sums_dev(bwgr_chain *C, double *sums_dev);
int bwgr_chain_end_iteration_dev(bwgr_chain *C, const double *sums_total_dev);
int bwgr_chain_round_apply(bwgr_chain *C, const double *delta_dev);
int bwgr_chain_get_sums(bwgr_chain *C, double sums[2]);              /* {sum d, sum b^2} of this rank's sweep */
int bwgr_chain_end_iteration(bwgr_chain *C, const double sums_total[2]); /* NULL: use this rank's own sums */

/* one-call form: create + run(it) + result + destroy */
int bwgr_bayes(bwgr_panel *P, int model, const float *y, float it, float bi, float pi, float df, float R2,
               uint64_t seed, int rng_mode, float *mu, float *b, float *d, float *hat, float *vb, float *ve,
               float *h2, float *MSx, float *pi_out, float *pval);

/* ---- two-effect samplers ---------------------------------------------------------------------------
 * Replaces BayesA2 / BayesB2 / BayesRR2(y, X1, X2, it, bi, [pi,] df, R2), src/Rcpp20260726ai.cpp:990-1218: one
 * residual, two resident panels with the same rows swept one after the other in every iteration, each with its own
 * prior scale and variance(s).  base_model = BWGR_BAYESA, BWGR_BAYESB or BWGR_BAYESRR.  Both panels must have been
 * created on the same device with the same slab geometry (same n, block and nwg).  Panel-2 markers carry the RNG ids
 * p1 .. p1+p2-1.  Outputs as the reference's return lists: mu, b1[p1], b2[p2], vb1 / vb2 (p_k entries for A2 and B2,
 * one for RR2), d1[p1], d2[p2] (B2 only; may be NULL otherwise), ve, hat[n], h2. */
int bwgr_bayes2(bwgr_panel *P1, bwgr_panel *P2, int base_model, const float *y, float it, float bi, float pi, float df,
                float R2, uint64_t seed, int rng_mode, float *mu, float *b1, float *d1, float *vb1, float *b2, float *d2,
                float *vb2, float *ve, float *hat, float *h2);

/* Host-only helper of the RNG contract: the k row indices (0-based, ascending) that `sort(sample(n, k, rp))` selects
 * for (seed, iter) -- what wgr's bagging (R/wgr.R:68) and the cross-validation folds of mcmcCV (R/cv.R:118-121) draw
 * from R's stream in the reference.  Does not touch the GPU. */
int bwgr_sample_rows(uint64_t seed, uint32_t iter, int64_t n, int64_t k, int rp, int *rows);

/* ---- wgr(): the R-level driver, device-resident ---------------------------------------------------
 * Replaces the iteration body and setup/teardown of wgr(), R/wgr.R:41-168 (bag = 1, eigK = NULL in
 * this round).  y is the R numeric vector (double).  Outputs as wgr's return list: mu, b[p],
 * Vb[p] (iv/de) or Vb[1], d[p], Ve, hat[n], cxx. */
int bwgr_wgr(bwgr_panel *P, const double *y, int it, int bi, int th, int iv, int de, double pi, double df, double R2,
             uint64_t seed, int rng_mode, double *mu, double *b, double *Vb, double *d, double *Ve, double *hat,
             double *cxx);
/* wgr() with the polygenic kernel term (eigK, R/wgr.R:23-32,70-78,116-119,148-150): U = the first pk eigenvectors
 * of the kernel (n x pk doubles, column-major, host), V their eigenvalues (the caller applies VarK: pk =
 * which.max(cumsum(V)/length(V) > VarK)).  Each iteration first sweeps KMUP(U,h,dh,xxK = 1,e,Lk = Ve/(V*Vk),Ve,0),
 * then the markers.  Extra outputs as in wgr's list: u[n] = U %*% H, Vk.  U == NULL, bag == 1 is bwgr_wgr.
 * bag != 1 (R/wgr.R:20,46,68,85,121): every iteration sweeps KMUP2 (src/Rcpp20260726ai.cpp:41-77) on
 * sort(sample(n, n*bag, rp)) rows -- the subsample is drawn from the RNG contract (purpose 20), a panel of those rows
 * and its Gram blocks are rebuilt on the device per iteration, df is divided by bag^2 and xx multiplied by bag as in R.
 * bag != 1 together with eigK is refused: the reference indexes the subsampled residual out of bounds there. */
int bwgr_wgr_ex(bwgr_panel *P, const double *y, int it, int bi, int th, int iv, int de, double pi, double df, double R2,
                uint64_t seed, int rng_mode, const double *U, const double *V, int64_t pk, double bag, int rp, double *mu,
                double *b, double *Vb, double *d, double *Ve, double *hat, double *cxx, double *u, double *Vk);

/* ---- EM / Gauss-Seidel family --------------------------------------------------------
 * Replaces SEXP emRR(y,gen,df,R2) src/Rcpp20260726ai.cpp:308-354, emBA(y,gen,df,R2) :80-128, emBB(y,gen,df,R2,Pi) :131-187,
 * emBC(y,gen,df,R2,Pi) :190-247, emBCpi(y,gen,df,R2,Pi) :1502-1545, emDE(y,gen,R2) :250-305, emBL(y,gen,R2,alpha) :357-397,
 * emEN(y,gen,R2,alpha) :400-460, emML(y,gen,D) :463-521, lasso(y,gen) :1463-1500 (_bWGR_emRR ... in src/RcppExports.cpp).  Deterministic
 * coordinate updates in the marker order the reference re-shuffles before every sweep with std::shuffle(order,
 * std::mt19937(i)) (emBCpi sweeps in natural order): the library makes the same standard-library call, gathers the
 * resident panel into that order on the device, rebuilds the Gram blocks and runs the sweep kernel with the variates
 * switched off (affine members) or with the member's own coordinate update in the recurrence (soft selection: emBB /
 * emBC / emBCpi; soft threshold: emEN, emBL).  par = Pi (emBB / emBC / emBCpi; reference default 0.75) or alpha (emBL /
 * emEN; 0.02).  maxit = 0: the reference's count (200 sweeps; emDE / emML / emEN up to 300 with their convergence
 * tests).  D: emML's optional marker weights (p floats) or NULL.  Outputs (host): mu, b[p], d[p] (soft-selection
 * members; may be NULL), hat[n], vbvec[p] (emBA / emBB / emDE: Vb; may be NULL), scal[6] = emRR {Va, Ve, h2}, emBA / emBB /
 * emDE {0, Ve, h2}, emML {Vb, Ve, h2, Va}, emBC {Va, Ve, h2, Vg}, emBCpi {Va, Ve, h2, Vg, pi}, emBL {0, 0, h2},
 * emEN {Va, Ve, h2}, lasso {Lmb, 0, h2}; iters = sweeps run.  lasso(y,gen) :1463-1500 sweeps in natural order; its
 * penalty is re-estimated after every sweep from the per-marker yx the sweep hands back (host loop, as the reference). */
enum { BWGR_EM_RR = 0, BWGR_EM_BA = 1, BWGR_EM_DE = 2, BWGR_EM_ML = 3, BWGR_EM_BB = 4, BWGR_EM_BC = 5, BWGR_EM_BCPI = 6,
       BWGR_EM_BL = 7, BWGR_EM_EN = 8, BWGR_EM_LASSO = 9 };
int bwgr_em(bwgr_panel *P, int model, const float *y, float df, float R2, float par, const float *D, int maxit, float *mu,
            float *b, float *d, float *hat, float *vbvec, float *scal, int *iters);
/* the marker order of sweep `upto` (0-based): the identity shuffled with std::mt19937(0), (1), ... (upto) (host only) */
int bwgr_em_order(int64_t p, int upto, int32_t *order);

/* ---- synthetic panels ----------------------------------------------------------
 * X_ij ~ Binomial(2, f_j), f_j ~ U(0.05,0.5), int8 column-major written to device memory Xdev
 * (ldx >= n); freq (p floats, device, may be NULL) receives f_j.  The p columns written are columns
 * col0 .. col0+p-1 of the (conceptually unbounded) panel of this seed, so marker shards of one panel can be
 * generated independently on different GPUs. */
int bwgr_synth_genotypes(void *Xdev, int64_t n, int64_t p, int64_t ldx, int64_t col0, uint64_t seed, float *freq_dev,
                         int device, void *hip_stream);

/* ---- multi-GPU inside the library ----------------------------------------------------------------
 * (new; the reference is single-process, single-threaded R: R/wgr.R:2.)  One marker shard per device of THIS process, the
 * residual replicated, RCCL all-reduces of the residual delta (n fp64) at the exchange rounds, one host thread: what an R
 * .Call needs to use several GPUs.  G = 1 is the exact chain.  G > 1 is the partitioned sampler of DESIGN.md section 8: NOT the
 * reference's chain, and statistically SOUND ONLY ON CENTRED COLUMNS (x_j - mean(x_j); measured: 2 / 4 / 8 shards then follow the exact
 * chain's ve, mean(d) and hat; on uncentred genotypes -- what bWGR sweeps -- every shard corrects the same stale residual mean and the
 * sampler diverges: ve 15 against 1.45 with four shards).  bwgr_group_create therefore REFUSES G > 1 on uncentred columns (BWGR_EINVAL)
 * unless BWGR_GROUP_ALLOW_UNCENTRED=1 is set; bwgr_group_sound says which case a group is in.  Centring is a change
 * of the model's parametrisation under the flat intercept prior (src/Rcpp20260726ai.cpp:683-684; the intercept absorbs sum_j mean_j b_j), which an
 * exact Gibbs sampler would not notice; bWGR's own chain does, a little (its xx_j carry the squared means: DESIGN.md section 8 -- ve 1.47 uncentred
 * against 1.56 centred on the probe panel), so "sound" here means: follows the exact chain on the SAME centred panel.  X is a HOST matrix (column-major n x p, ldx >= n), y n host floats; device g
 * of `devices` takes the block-aligned column shard g.  markers_per_sync: markers swept per device between two all-reduces
 * (0: 131072 / ndev; 131072 when the shards share one device, where an exchange costs a launch boundary, not a ring).  bwgr_group_result returns the Bayes* return list over the whole panel (b, d, pval: p floats; vb: p
 * floats for BayesA/B/L/Dpi, else 1; hat: n floats).  info: {devices, exchange rounds per sweep, markers per round, RCCL in use}. */
typedef struct bwgr_group bwgr_group;
int bwgr_group_create(bwgr_group **out, int ndev, const int *devices, const void *X, int xtype, int64_t n, int64_t p, int64_t ldx,
                      int block, const float *y, int model, float it, float bi, float pi, float df, float R2, uint64_t seed,
                      int rng_mode, int64_t markers_per_sync);
/* the same on the IMPLICITLY centred columns of an int8 matrix (bwgr_panel_set_centred on every shard: the genotypes stay int8, k_sweep3 sweeps them):
 * sound with several devices; selection models only; the intercept returned is that of the centred parametrisation (mu_c = mu + sum_j mean_j b_j).
 * memloc: where X lives (BWGR_DEVICE: only when every shard sits on that device).
 * SHARDS SIDE BY SIDE ON ONE GPU: `devices` may name the same device ndev times (both create calls).  One exact chain is a latency-bound pipeline
 * that occupies a third of the chip; the shards of the partitioned sampler then run their sweeps concurrently on streams of their own, each on its own
 * compute units, and an exchange round is a sum kernel between events (no RCCL).  Same sampler, same soundness rule (centred columns) as across GPUs. */
int bwgr_group_create_centred(bwgr_group **out, int ndev, const int *devices, const void *X, int xtype, int64_t n, int64_t p, int64_t ldx,
                              int block, const float *y, int model, float it, float bi, float pi, float df, float R2, uint64_t seed,
                              int rng_mode, int64_t markers_per_sync, int memloc);
int bwgr_group_run(bwgr_group *G, int iters);
int bwgr_group_sync(bwgr_group *G);
int bwgr_group_info(const bwgr_group *G, int64_t info[4]);
int bwgr_group_sound(const bwgr_group *G, int *sound);      /* 1: one device (exact chain) or centred columns; 0: G > 1 on uncentred columns */
int bwgr_panel_centred(bwgr_panel *P, int *centred);        /* 1 when every column's |mean| <= 1e-3 sd (from the panel's own statistics), or after
                                                               bwgr_panel_set_centred(P, 1) */
/* Implicit centring of an int8 panel (no reference counterpart: bWGR sweeps whatever columns it is given, src/Rcpp20260726ai.cpp:668-682; centring
 * is the caller's preprocessing there).  on != 0: from now on the fused chains on this panel and its clones (bwgr_chain_*, bwgr_bayes, the chains
 * bwgr_group_* builds) sweep the columns x_j - mean(x_j) -- the same chain as on an explicitly centred float copy of the panel, to the float
 * rounding of that copy's entries -- while the genotypes stay int8 in HBM and every kernel keeps reading the raw columns: with s_j = sum_i x_ij and
 * the residual carried as e_stored = e - shift * 1,  (x_j - s_j/n 1)'e = x_j'e_stored - (s_j/n) sum(e_stored), and sum(e_stored) moves by -s_k delta_k
 * per marker: scalars on the sequencer, nothing on the streamers.  bwgr_panel_stats then returns xx_j = |x_j - mean_j|^2 (vx, MSx do not change),
 * hat = X_c B + mu, bwgr_panel_centred answers 1, and the group entry points accept several devices.  Selection models (BayesB / C /
 * Cpi / Dpi) on int8 panels that have k_sweep3, at every inclusion rate (both of their engines carry the terms, and so does the fp64 redo of a sweep
 * that leaves the fixed-point range); the affine models and the non-chain entry points (KMUP, wgr, EM, two-effect samplers, pairs) return BWGR_EINVAL
 * on a centred panel.  Refused while chains are alive on the panel; on == 0 switches back. */
int bwgr_panel_set_centred(bwgr_panel *P, int on);
int bwgr_group_result(bwgr_group *G, float *mu, float *b, float *d, float *hat, float *vb, float *ve, float *h2, float *MSx,
                      float *pi_out, float *pval);
int bwgr_group_destroy(bwgr_group *G);

/* ---- test hooks -----------------------------------------------------------------------------------
 * variates of the RNG contract computed on the device: kind 0 normal, 1 uniform, 2 chisq(nu);
 * out[i] for marker = marker0 + i. */
int bwgr_debug_variates(int device, uint64_t seed, int kind, double nu, uint32_t marker0, uint32_t iter,
                        uint32_t purpose, int count, double *out_host);
/* abort-path hook: while on != 0, sweeps launched on this panel run with slab workgroup 0 absent; every workgroup that waits
 * for it reaches its wall-clock bound (4 s), the shared abort word ends the launch and the call reports BWGR_ETIMEOUT.  The
 * panel stays usable: switch the hook off and launch again. */
int bwgr_debug_withhold(bwgr_panel *P, int on);

#ifdef __cplusplus
}
#endif
#endif
