// bwgr_amd/csrc/bwgr_hip.hip -- libbwgr_hip.so: C ABI (include/bwgr.h) + setup / per-iteration / finalisation
// kernels around the blocked sweep (sweep.hip.h).  gfx950 only; there is no CPU path in this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>
#include <chrono>
#include <algorithm>
#include <mutex>
#include <utility>
#include <type_traits>
#include "../../include/bwgr.h"
#include "rng.hip.h"
#include "sweep.hip.h"
#include "sweep3.hip.h"
#include "sweep2w.hip.h"
#include "sweep3p.hip.h"
#include <stdlib.h>

using namespace bwgr;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
  return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(BWGR_EHIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define CHK(x) do { int r_ = (x); if (r_ != BWGR_OK) return r_; } while (0)

// the status a sweep kernel left in ChainScalars::error
static int sweep_error(uint32_t code, const char *who) {
  // (since round 3 a sweep that leaves the range is redone on the fp64 residual, k_range_recover: this status only surfaces where no fallback
  // is queued -- chains advanced in pairs)
  if (code == 2u) return fail(BWGR_ERANGE, "%s: the residual left the fixed-point range of the sweep (it grew beyond the grid's headroom, about a thousandfold of its starting scale, within one sweep); the chain state is invalid", who);
  return fail(BWGR_ETIMEOUT, "%s: a workgroup exchange timed out inside the sweep kernel (the chain state is invalid)", who);
}
extern "C" const char *bwgr_last_error(void) { return g_err; }
extern "C" int bwgr_abi_version(void) { return BWGR_ABI_VERSION; }
extern "C" int bwgr_device_count(int *count) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
  if (count) *count = c;
  return BWGR_OK;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// block-wide sum; result valid in every thread.  red must hold >= 17 doubles.
__device__ inline double block_sum(double v, double *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0; for (int w = 0; w < nw; ++w) s += red[w]; red[16] = s; }
  __syncthreads();
  return red[16];
}

__device__ __forceinline__ float xval(const int8_t *X, int64_t i) { return (float)X[i]; }
__device__ __forceinline__ float xval(const float *X, int64_t i) { return X[i]; }

// ---- upload conversion: src (n x p, ldx) -> X (ld x p), zero padded rows ----
template <typename ST, typename XT>
__global__ void k_convert(const ST *src, int64_t ldx, XT *X, int64_t ld, int n, int64_t j0, int64_t ncols, int R, int64_t p) {
  const int64_t total = ncols * ld;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t jj = idx / ld, i = idx - jj * ld;
    XT v = (XT)0;
    if (i < n) v = (XT)src[jj * ldx + i];
    X[xoff(i, j0 + jj, R, p)] = v;
  }
}

__global__ void k_f2d(const float *src, double *dst, int64_t n, int64_t ld) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ld; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (i < n) ? (double)src[i] : 0.0;
}
__global__ void k_d2f(const double *src, float *dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

// ---- a10: xx[j] = |X_j|^2, vx[j] = fvar(X_j)   (src/Rcpp20260726ai.cpp:7-9, 593-597); one wave per column ----
template <typename XT>
__global__ void k_stats(const XT *X, int R, int n, int p, float *xx, float *vx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= p) return;
  double s1 = 0, s2 = 0;
  for (int i = lane; i < n; i += 64) { const float v = xval(X, xoff(i, j, R, p)); s1 += (double)v; s2 += (double)v * (double)v; }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  s1 = __shfl(s1, 0, 64); s2 = __shfl(s2, 0, 64);
  const float mean = (float)(s1 / (double)n);
  double sv = 0;
  for (int i = lane; i < n; i += 64) { const float dev = xval(X, xoff(i, j, R, p)) - mean; const float sq = dev * dev; sv += (double)sq; }
  sv = wave_sum(sv);
  if (lane == 0) { xx[j] = (float)s2; vx[j] = (float)(sv / (double)(float)(n - 1)); }
}

// ---- implicit centring of an int8 panel (bwgr_panel_set_centred): s_j = sum_i x_ij exactly, and the centred column's squared norm
// sum_i (x_ij - s_j / n)^2 = sum x^2 - s_j^2 / n from exact integer sums, rounded to float once (the reference would form it from the centred
// float column, X.colwise().squaredNorm(), src/Rcpp20260726ai.cpp:593-594); one wave per column ----
__global__ void k_colsum_i8(const int8_t *X, int R, int n, int p, int32_t *csum, float *xxc) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= p) return;
  long long s1 = 0, s2 = 0;
  for (int i = lane; i < n; i += 64) { const int v = (int)X[xoff(i, j, R, p)]; s1 += v; s2 += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
  if (lane == 0) { csum[j] = (int32_t)s1; xxc[j] = (float)((double)s2 - (double)s1 * (double)s1 / (double)n); }
}
// sum_j s_j coef_j / n  (one workgroup, fixed order): what the centred columns take off X * coef
__global__ __launch_bounds__(1024) void k_cen_dot(const int32_t *csum, const float *coef, int64_t p, double ninv, double *out) {
  __shared__ double red[1024];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int64_t j = t; j < p; j += 1024) s = fma((double)csum[j], (double)coef[j], s);
  red[t] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
  if (t == 0) *out = red[0] * ninv;
}

// deterministic two-stage sum of a float vector into a double
__global__ void k_sum_stage1(const float *v, int64_t n, double *part) {
  __shared__ double red[17];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += (double)v[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void k_sum_stage2(const double *part, int nparts, float *out_f) {
  __shared__ double red[17];
  double s = 0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out_f = (float)s;
}

// ---- block-diagonal Gram G_B = X_B' X_B (setup; exact int32 for int8 genotypes) ----
// 256 threads as a 16 x 16 grid, thread (tj,tk) owns G[tj+16a][tk+16c], a,c < m/16.
template <int TJ>
__global__ __launch_bounds__(256) void k_gram_i8(const int8_t *X, int64_t ld, int R, int p, int m, int32_t *gram) {
  constexpr int RC = 128, RW = RC / 4 + 1;  // rows per chunk, dwords per column in LDS (padded)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t *tile = reinterpret_cast<int32_t *>(smem);
  const int blk = blockIdx.x, j0 = blk * m, mB = min(m, p - j0);
  const int tj = threadIdx.x >> 4, tk = threadIdx.x & 15;
  int32_t acc[TJ][TJ];
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) acc[a][c] = 0;
  for (int64_t r0 = 0; r0 < ld; r0 += RC) {
    __syncthreads();
    for (int c = threadIdx.x; c < m * (RC / 4); c += 256) {
      const int jj = c / (RC / 4), w = c - jj * (RC / 4);
      int32_t v = 0;
      if (jj < mB) v = *reinterpret_cast<const int32_t *>(X + xoff(r0 + 4 * w, j0 + jj, R, p));
      tile[jj * RW + w] = v;
    }
    __syncthreads();
    for (int w = 0; w < RC / 4; ++w) {
      int32_t av[TJ], bv[TJ];
#pragma unroll
      for (int a = 0; a < TJ; ++a) { av[a] = tile[(tj + 16 * a) * RW + w]; bv[a] = tile[(tk + 16 * a) * RW + w]; }
#pragma unroll
      for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = __builtin_amdgcn_sdot4(av[a], bv[c], acc[a][c], false);
    }
  }
  int32_t *g = gram + (size_t)blk * m * m;
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) g[(size_t)(tj + 16 * a) * m + (tk + 16 * c)] = acc[a][c];
}

// m = 128: the same blocks on the matrix cores.  out[blk][i][j] = X_{(blk-dist)m+i} . X_{blk*m+j} (dist = 0: the diagonal block),
// exact in the int32 accumulators.  One workgroup of four waves per block; wave w owns the 64 x 64 quadrant (4 x 4 tiles of
// 16 x 16); v_mfma_i32_16x16x64_i8 takes, per lane (m16, grp), the 16 bytes of marker 16t + m16 at rows 64kk + 16grp .. +15 for
// both operands -- straight from the slab-major panel, whose markers are contiguous along the rows -- and returns
// out[16ti + 4grp + reg][16tj + m16].  Per 64 rows a wave issues 8 loads and 16 MFMAs (the sdot4 kernels above ran at
// 0.75 TB/s: one dword per thread per load and an LDS round trip).
__global__ __launch_bounds__(256) void k_gram_mfma_i8(const int8_t *X, int64_t ld, int R, int p, int32_t *out, int dist) {
  constexpr int m = 128;
  const int blk = blockIdx.x + dist, ia0 = (blk - dist) * m, jb0 = blk * m;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m16 = lane & 15, grp = lane >> 4;
  const int ti0 = 4 * (wave >> 1), tj0 = 4 * (wave & 1);
  s2_v4i acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = s2_v4i{0, 0, 0, 0};
  // markers past the panel (last block) are read at a clamped column and zeroed
  int ca[4], cb[4]; bool oka[4], okb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ja = ia0 + 16 * (ti0 + t) + m16, jb = jb0 + 16 * (tj0 + t) + m16;
    oka[t] = ja < p; okb[t] = jb < p; ca[t] = min(ja, p - 1); cb[t] = min(jb, p - 1);
  }
  const s2_v4i zero = {0, 0, 0, 0};
  // software pipeline: the operands of step k+1 are requested before the sixteen MFMAs of step k
  auto load_step = [&](int64_t r0, s2_v4i (&av)[4], s2_v4i (&bv)[4]) {
    const int64_t sl = r0 / R;
    const size_t roff = (size_t)(r0 - sl * R) + 16 * grp;
    const int8_t *sb = X + (size_t)sl * p * R + roff;       // slab base + row offset; marker j adds j*R
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      av[t] = *reinterpret_cast<const s2_v4i *>(sb + (size_t)ca[t] * R);
      bv[t] = *reinterpret_cast<const s2_v4i *>(sb + (size_t)cb[t] * R);
    }
  };
  s2_v4i av[4], bv[4], an[4], bn[4];
  load_step(0, av, bv);
  for (int64_t r0 = 0; r0 < ld; r0 += 64) {
    const bool more = r0 + 64 < ld;
    load_step(more ? r0 + 64 : r0, an, bn);                 // (the last step reloads its own operands: harmless)
#pragma unroll
    for (int t = 0; t < 4; ++t) { av[t] = oka[t] ? av[t] : zero; bv[t] = okb[t] ? bv[t] : zero; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[a], bv[c], acc[a][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) { av[t] = an[t]; bv[t] = bn[t]; }
  }
  int32_t *g = out + (size_t)blk * m * m;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        g[(size_t)(16 * (ti0 + a) + 4 * grp + reg) * m + 16 * (tj0 + c) + m16] = acc[a][c][reg];
}

// off-diagonal blocks for the pipelined sweep: gramx[blk][k][j] = X_{(blk-dist)m+k} . X_{blk*m+j}, blk >= dist (dist = 1, 2)
template <int TJ>
__global__ __launch_bounds__(256) void k_gramx_i8(const int8_t *X, int64_t ld, int R, int p, int m, int32_t *gramx, int dist) {
  constexpr int RC = 128, RW = RC / 4 + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t *ta = reinterpret_cast<int32_t *>(smem), *tb = ta + (size_t)m * RW;
  const int blk = blockIdx.x + dist, ja0 = (blk - dist) * m, jb0 = blk * m, mBb = min(m, p - jb0);
  const int tj = threadIdx.x >> 4, tk = threadIdx.x & 15;
  int32_t acc[TJ][TJ];
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) acc[a][c] = 0;
  for (int64_t r0 = 0; r0 < ld; r0 += RC) {
    __syncthreads();
    for (int c = threadIdx.x; c < m * (RC / 4); c += 256) {
      const int jj = c / (RC / 4), w = c - jj * (RC / 4);
      ta[jj * RW + w] = *reinterpret_cast<const int32_t *>(X + xoff(r0 + 4 * w, ja0 + jj, R, p));
      tb[jj * RW + w] = (jj < mBb) ? *reinterpret_cast<const int32_t *>(X + xoff(r0 + 4 * w, jb0 + jj, R, p)) : 0;
    }
    __syncthreads();
    for (int w = 0; w < RC / 4; ++w) {
      int32_t av[TJ], bv[TJ];
#pragma unroll
      for (int a = 0; a < TJ; ++a) { av[a] = ta[(tj + 16 * a) * RW + w]; bv[a] = tb[(tk + 16 * a) * RW + w]; }
#pragma unroll
      for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = __builtin_amdgcn_sdot4(av[a], bv[c], acc[a][c], false);
    }
  }
  int32_t *g = gramx + (size_t)blk * m * m;
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) g[(size_t)(tj + 16 * a) * m + (tk + 16 * c)] = acc[a][c];
}
template <int TJ>
__global__ __launch_bounds__(256) void k_gramx_f32(const float *X, int64_t ld, int R, int p, int m, double *gramx, int dist) {
  constexpr int RC = 64, RW = RC + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *ta = reinterpret_cast<float *>(smem), *tb = ta + (size_t)m * RW;
  const int blk = blockIdx.x + dist, ja0 = (blk - dist) * m, jb0 = blk * m, mBb = min(m, p - jb0);
  const int tj = threadIdx.x >> 4, tk = threadIdx.x & 15;
  double acc[TJ][TJ];
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) acc[a][c] = 0.0;
  for (int64_t r0 = 0; r0 < ld; r0 += RC) {
    __syncthreads();
    for (int c = threadIdx.x; c < m * RC; c += 256) {
      const int jj = c / RC, w = c - jj * RC;
      ta[jj * RW + w] = X[xoff(r0 + w, ja0 + jj, R, p)];
      tb[jj * RW + w] = (jj < mBb) ? X[xoff(r0 + w, jb0 + jj, R, p)] : 0.0f;
    }
    __syncthreads();
    for (int w = 0; w < RC; ++w) {
      double av[TJ], bv[TJ];
#pragma unroll
      for (int a = 0; a < TJ; ++a) { av[a] = (double)ta[(tj + 16 * a) * RW + w]; bv[a] = (double)tb[(tk + 16 * a) * RW + w]; }
#pragma unroll
      for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = fma(av[a], bv[c], acc[a][c]);
    }
  }
  double *g = gramx + (size_t)blk * m * m;
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) g[(size_t)(tj + 16 * a) * m + (tk + 16 * c)] = acc[a][c];
}

template <int TJ>
__global__ __launch_bounds__(256) void k_gram_f32(const float *X, int64_t ld, int R, int p, int m, double *gram) {
  constexpr int RC = 64, RW = RC + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *tile = reinterpret_cast<float *>(smem);
  const int blk = blockIdx.x, j0 = blk * m, mB = min(m, p - j0);
  const int tj = threadIdx.x >> 4, tk = threadIdx.x & 15;
  double acc[TJ][TJ];
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) acc[a][c] = 0.0;
  for (int64_t r0 = 0; r0 < ld; r0 += RC) {
    __syncthreads();
    for (int c = threadIdx.x; c < m * RC; c += 256) {
      const int jj = c / RC, w = c - jj * RC;
      tile[jj * RW + w] = (jj < mB) ? X[xoff(r0 + w, j0 + jj, R, p)] : 0.0f;
    }
    __syncthreads();
    for (int w = 0; w < RC; ++w) {
      double av[TJ], bv[TJ];
#pragma unroll
      for (int a = 0; a < TJ; ++a) { av[a] = (double)tile[(tj + 16 * a) * RW + w]; bv[a] = (double)tile[(tk + 16 * a) * RW + w]; }
#pragma unroll
      for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int c = 0; c < TJ; ++c) acc[a][c] = fma(av[a], bv[c], acc[a][c]);
    }
  }
  double *g = gram + (size_t)blk * m * m;
#pragma unroll
  for (int a = 0; a < TJ; ++a)
#pragma unroll
    for (int c = 0; c < TJ; ++c) g[(size_t)(tj + 16 * a) * m + (tk + 16 * c)] = acc[a][c];
}

// strict upper triangle of the diagonal Gram blocks, row-packed: entry (k, j>k) at k(m-1) - k(k-1)/2 + (j-k-1)
template <typename GT>
__global__ void k_gram_pack(const GT *gram, GT *gramp, int m, int pstride, int64_t nblocks) {
  const int64_t blk = blockIdx.x;
  const GT *G = gram + (size_t)blk * m * m;
  GT *P = gramp + (size_t)blk * pstride;
  for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
    const int k = e / m, j = e - k * m;
    if (j > k) P[k * (m - 1) - k * (k - 1) / 2 + (j - k - 1)] = G[e];
  }
  for (int e = m * (m - 1) / 2 + threadIdx.x; e < pstride; e += blockDim.x) P[e] = (GT)0;
}

// 16-bit copies of the packed and the distance-1 cross Gram blocks for the k_sweep2 sequencer (half the bytes through its
// CU per block); *bad is set when an entry does not fit, and the sequencer then stages the 32-bit arrays
// largest |x| of an int8 panel (k_sweep3's integer sums are sized by it)
__global__ void k_absmax_i8(const int8_t *X, size_t count, int *out) {
  int mx = 0;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < count; i += (size_t)gridDim.x * blockDim.x * 16) {
    const uint4 v = *reinterpret_cast<const uint4 *>(X + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) { const int x = (int)(int8_t)(w[q] >> (8 * c)); mx = max(mx, x < 0 ? -x : x); }
  }
  atomicMax(out, mx);
}
__global__ void k_gram_narrow(const int32_t *src, uint16_t *dst, int64_t count, int *bad) {
  int any = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = src[i];
    any |= (v < 0 || v > 65535);
    dst[i] = (uint16_t)v;
  }
  if (any) *bad = 1;
}

// ---- chain setup (src/Rcpp20260726ai.cpp:599-610 and the identical blocks of the other samplers) ----
struct InitArgs {
  const float *y; double *e; int n, p; int64_t ld; int model; float pi, df, R2; float MSx; ChainScalars *sc;
};
__global__ void k_chain_init(const InitArgs a) {
  __shared__ double red[17];
  double s = 0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) s += (double)a.y[i];
  s = block_sum(s, red);
  const float mu = (float)(s / (double)a.n);               // y.mean()
  double sv = 0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) { const float dev = a.y[i] - mu; const float sq = dev * dev; sv += (double)sq; }
  sv = block_sum(sv, red);
  const float vy = (float)(sv / (double)(float)(a.n - 1));   // fvar(y)
  for (int i = threadIdx.x; i < a.ld; i += blockDim.x) { const float t = (i < a.n) ? (a.y[i] - mu) : 0.0f; a.e[i] = (double)t; }
  if (threadIdx.x == 0) {
    float pi = a.pi;
    if (a.model == BWGR_BAYESCPI || a.model == BWGR_BAYESDPI) pi = 0.5f;
    float Sb;
    if (a.model == BWGR_BAYESC || a.model == BWGR_BAYESCPI) Sb = a.df * (a.R2) * vy / a.MSx / (1 - pi);
    else Sb = (a.R2) * a.df * vy / a.MSx;
    ChainScalars sc;
    memset(&sc, 0, sizeof(sc));
    sc.ve = vy; sc.vb = Sb; sc.lam = vy / Sb; sc.pi = pi;
    sc.Sb = Sb; sc.Se = (1 - a.R2) * a.df * vy; sc.C = -0.5f / sqrtf(vy); sc.odds = pi / (1.0f - pi);
    sc.mu = mu; sc.dfp1 = a.df + 1; sc.vy = vy; sc.MSx = a.MSx;
    sc.inc_rate = 1.0f - pi;
    *a.sc = sc;
  }
}
__global__ void k_marker_init(float *b, float *d, float *vb, float *lam, float *B, float *D, float *VB, int p,
                              const ChainScalars *sc) {
  const float Sb = sc->Sb, ve = sc->ve;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p; j += gridDim.x * blockDim.x) {
    b[j] = 0; d[j] = 0; B[j] = 0; D[j] = 0; VB[j] = 0;
    vb[j] = Sb;
    lam[j] = ve * (1.0f / Sb);   // ve * vb.cwiseInverse()
  }
}

// ---- per-iteration tail: intercept, residual / marker variances, pi (one workgroup) ----
struct TailArgs {
  double *e; int n, p; int model; float df, R2, Phi; int accumulate; uint32_t iter; Rng rng; ChainScalars *sc;
};
__global__ __launch_bounds__(1024) void k_tail(const TailArgs a) {
  __shared__ double red[17];
  ChainScalars &sc = *a.sc;
  const float ve0 = sc.ve;
  double s = 0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) s += a.e[i];
  s = block_sum(s, red);
  const float me = (float)(s / (double)a.n);                                        // e.mean()
  const double z = rng_normal(a.rng, RNG_GLOBAL_MARKER, a.iter, RNG_G_MU, 0);
  const float eM = (float)((double)me + (double)sqrtf(ve0 / a.n) * z);              // :620
  double ss = 0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) { const double v = a.e[i] - (double)eM; a.e[i] = v; ss = fma(v, v, ss); }
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    const float ssf = (float)ss;                                                    // e.squaredNorm()
    const float b2f = (float)sc.sum_b2;                                             // b.squaredNorm()
    float ve = ve0, vb = sc.vb, pi = sc.pi, Sb = sc.Sb;
    const float mu = sc.mu + eM;
    const double chi_e = rng_chisq(a.rng, (double)(a.n + a.df), RNG_GLOBAL_MARKER, a.iter, RNG_G_VE);
    switch (a.model) {
      case BWGR_BAYESA: case BWGR_BAYESB: case BWGR_BAYESDPI: case BWGR_BAYESL:
        ve = (float)((double)(ssf + sc.Se) / chi_e);
        break;
      case BWGR_BAYESRR: {
        ve = (float)((double)(ssf + sc.Se) / chi_e);
        const double chi_b = rng_chisq(a.rng, (double)(a.p + a.df), RNG_GLOBAL_MARKER, a.iter, RNG_G_VB);
        vb = (float)((double)(b2f + Sb) / chi_b);
        sc.lam = ve / vb;
      } break;
      case BWGR_BAYESC: case BWGR_BAYESCPI: {
        const double chi_b = rng_chisq(a.rng, (double)(a.df + a.p), RNG_GLOBAL_MARKER, a.iter, RNG_G_VB);
        vb = (float)((double)(b2f + Sb) / chi_b);
        ve = (float)((double)(ssf + sc.Se) / chi_e);
        sc.lam = ve / vb;
      } break;
    }
    if (a.model == BWGR_BAYESCPI) {
      pi = (float)(sc.sum_d / (double)a.p);
      Sb = a.df * (a.R2) * sc.vy / sc.MSx / (1 - pi);
    }
    if (a.model == BWGR_BAYESDPI) pi = (float)(sc.sum_d / (double)a.p);
    sc.ve = ve; sc.vb = vb; sc.pi = pi; sc.Sb = Sb; sc.mu = mu;
    sc.C = -0.5f / sqrtf(ve);
    sc.inc_rate = (float)(sc.sum_d / (double)a.p);
    sc.sum_d = 0.0; sc.sum_b2 = 0.0;
    if (a.accumulate) { sc.MU += mu; sc.VE += ve; sc.VBs += vb; sc.Pi += pi; }
  }
}
// per-marker part of the tail: lambda_j for the next sweep and the posterior sums
__global__ void k_marker_tail(const float *b, const float *d, const float *vb, float *lam, float *B, float *D, float *VB,
                              int p, int model, float Phi, int accumulate, const ChainScalars *sc) {
  const float ve = sc->ve;
  const bool per = (model == BWGR_BAYESA || model == BWGR_BAYESB || model == BWGR_BAYESDPI || model == BWGR_BAYESL);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p; j += gridDim.x * blockDim.x) {
    if (per) {
      const float v = vb[j];
      lam[j] = (model == BWGR_BAYESL) ? sqrtf(Phi * ve / v) : ve * (1.0f / v);
      if (accumulate) VB[j] += v;
    }
    if (accumulate) { B[j] += b[j]; D[j] += d[j]; }
  }
}

// ---- tail of the two-effect samplers (src/Rcpp20260726ai.cpp:1042-1047, :1138-1143, :1204-1210): one intercept and one
// residual variance from the shared residual; RR2 also draws the two common marker variances (second chi-square with
// the marker word RNG_GLOBAL_MARKER - 1).  Scalars are written to both chains' blocks; MU / VE accumulate in the first.
struct Tail2Args {
  double *e; int n, p1, p2; int rr; float df; int accumulate; uint32_t iter; Rng rng; ChainScalars *sc1, *sc2;
};
__global__ __launch_bounds__(1024) void k_tail2(const Tail2Args a) {
  __shared__ double red[17];
  ChainScalars &s1 = *a.sc1, &s2 = *a.sc2;
  const float ve0 = s1.ve;
  double s = 0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) s += a.e[i];
  s = block_sum(s, red);
  const float me = (float)(s / (double)a.n);
  const double z = rng_normal(a.rng, RNG_GLOBAL_MARKER, a.iter, RNG_G_MU, 0);
  const float eM = (float)((double)me + (double)sqrtf(ve0 / a.n) * z);
  double ss = 0;
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) { const double v = a.e[i] - (double)eM; a.e[i] = v; ss = fma(v, v, ss); }
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    const float ssf = (float)ss;
    const float mu = s1.mu + eM;
    const double chi_e = rng_chisq(a.rng, (double)(a.n + a.df), RNG_GLOBAL_MARKER, a.iter, RNG_G_VE);
    const float ve = (float)((double)(ssf + s1.Se) / chi_e);
    if (a.rr) {
      const double c1 = rng_chisq(a.rng, (double)(a.df + a.p1), RNG_GLOBAL_MARKER, a.iter, RNG_G_VB);
      const double c2 = rng_chisq(a.rng, (double)(a.df + a.p2), RNG_GLOBAL_MARKER - 1u, a.iter, RNG_G_VB);
      s1.vb = (float)((double)(s1.Sb + (float)s1.sum_b2) / c1);
      s2.vb = (float)((double)(s2.Sb + (float)s2.sum_b2) / c2);
      s1.lam = ve / s1.vb; s2.lam = ve / s2.vb;
    }
    const float Cn = -0.5f / sqrtf(ve);
    s1.ve = ve; s2.ve = ve; s1.mu = mu; s2.mu = mu; s1.C = Cn; s2.C = Cn;
    s1.inc_rate = (float)(s1.sum_d / (double)a.p1); s2.inc_rate = (float)(s2.sum_d / (double)a.p2);
    s1.sum_d = 0.0; s1.sum_b2 = 0.0; s2.sum_d = 0.0; s2.sum_b2 = 0.0;
    if (a.accumulate) { s1.MU += mu; s1.VE += ve; s1.VBs += s1.vb; s2.VBs += s2.vb; }
  }
}
__global__ void k_set_rr2_start(ChainScalars *sc) { sc->lam = sc->MSx; }   // BayesRR2 starts with Lmb = MSx (:1190)
__global__ void k_hat2(const float *h1, const float *h2, float MU, float *hat, int n) {   // fit = X1*B1 + X2*B2; fit += MU
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float f = h1[i] + h2[i]; hat[i] = f + MU; }
}

// ---- finalisation ----
__global__ void k_final_markers(float *B, float *D, float *VB, float *pval, int p, float MCMC, int per) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p; j += gridDim.x * blockDim.x) {
    B[j] /= MCMC; D[j] /= MCMC;
    if (per) VB[j] /= MCMC;
    if (pval) pval[j] = -1.0f * logf(1.0f - D[j]);
  }
}
// partial[c][i] = sum_{j in column chunk c} x_ij * coef_j   (fp64), 4 rows per thread
template <typename XT, typename CT>
__global__ __launch_bounds__(256) void k_gemv_part(const XT *X, int64_t ld, int R, int p, const CT *coef, int cols_per_chunk, double *part) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= ld) return;
  const int c = blockIdx.y;
  const int ja = c * cols_per_chunk, jb = min(p, ja + cols_per_chunk);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int j = ja; j < jb; ++j) {
    const double cj = (double)coef[j];
    if (cj == 0.0) continue;
    const XT *xp = X + xoff(i0, j, R, p);
    float x0, x1, x2, x3;
    if constexpr (sizeof(XT) == 1) {
      const uint32_t w = *reinterpret_cast<const uint32_t *>(xp);
      x0 = (float)(int8_t)(w & 0xFF); x1 = (float)(int8_t)((w >> 8) & 0xFF); x2 = (float)(int8_t)((w >> 16) & 0xFF); x3 = (float)(int8_t)(w >> 24);
    } else {
      const float4 v = *reinterpret_cast<const float4 *>(xp);
      x0 = v.x; x1 = v.y; x2 = v.z; x3 = v.w;
    }
    a0 = fma((double)x0, cj, a0); a1 = fma((double)x1, cj, a1); a2 = fma((double)x2, cj, a2); a3 = fma((double)x3, cj, a3);
  }
  double *o = part + (int64_t)c * ld + i0;
  o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
}
// int8 panels: 16 rows per thread (one 16-byte load per marker), four markers' loads in flight, no data-dependent branch --
// the 4-rows-per-thread loop above with its skip of zero coefficients waited for every load and ran at 1.1 TB/s
template <typename CT>
__global__ __launch_bounds__(256) void k_gemv_part_i8(const int8_t *X, int64_t ld, int R, int p, const CT *coef, int cols_per_chunk, double *part) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i0 >= ld) return;
  const int c = blockIdx.y;
  const int ja = c * cols_per_chunk, jb = min(p, ja + cols_per_chunk);
  const int8_t *base = X + xoff(i0, 0, R, p);          // marker j of this slab: base + j*R (R is a multiple of 128)
  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
#define GEMV_ADD(w_, cj_, k0_) { \
    acc[(k0_) + 0] = fma((double)(int)(int8_t)((w_) & 0xFF), cj_, acc[(k0_) + 0]); acc[(k0_) + 1] = fma((double)(int)(int8_t)(((w_) >> 8) & 0xFF), cj_, acc[(k0_) + 1]); \
    acc[(k0_) + 2] = fma((double)(int)(int8_t)(((w_) >> 16) & 0xFF), cj_, acc[(k0_) + 2]); acc[(k0_) + 3] = fma((double)((int)(w_) >> 24), cj_, acc[(k0_) + 3]); }
#define GEMV_COL(v_, cj_) { GEMV_ADD((v_).x, cj_, 0) GEMV_ADD((v_).y, cj_, 4) GEMV_ADD((v_).z, cj_, 8) GEMV_ADD((v_).w, cj_, 12) }
  int j = ja;
  for (; j + 4 <= jb; j += 4) {
    const uint4 v0 = *reinterpret_cast<const uint4 *>(base + (size_t)j * R), v1 = *reinterpret_cast<const uint4 *>(base + (size_t)(j + 1) * R);
    const uint4 v2 = *reinterpret_cast<const uint4 *>(base + (size_t)(j + 2) * R), v3 = *reinterpret_cast<const uint4 *>(base + (size_t)(j + 3) * R);
    const double c0 = (double)coef[j], c1 = (double)coef[j + 1], c2 = (double)coef[j + 2], c3 = (double)coef[j + 3];
    GEMV_COL(v0, c0) GEMV_COL(v1, c1) GEMV_COL(v2, c2) GEMV_COL(v3, c3)
  }
  for (; j < jb; ++j) { const uint4 v0 = *reinterpret_cast<const uint4 *>(base + (size_t)j * R); const double c0 = (double)coef[j]; GEMV_COL(v0, c0) }
#undef GEMV_COL
#undef GEMV_ADD
  double *o = part + (int64_t)c * ld + i0;
#pragma unroll
  for (int k = 0; k < 16; ++k) o[k] = acc[k];
}
__global__ void k_hat_finish(const double *part, int64_t ld, int nchunks, int n, float MU, float *hat, const double *cen_off = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = cen_off ? -*cen_off : 0.0;   // (implicitly centred columns: X_c B = X B - sum_j mean_j B_j)
  for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * ld + i];
  const float f = (float)s;
  hat[i] = f + MU;
}


// ---- wgr(): R-side (double) steps around the KMUP sweep, R/wgr.R:41-168 --------------------------------------------
struct WgrScalars {
  double mu, Ve, Va, Sb, Se, MSx, vy, bb, B0, VE, VA, sumD, cxx;
  double Vp, VP, Sk;   // polygenic term (eigK)
};
// per-column double statistics as R computes them: xx = crossprod, var = sum((x-mean)^2)/(n-1); one wave per column
template <typename XT>
__global__ void k_stats64(const XT *X, int R, int n, int p, double *xx, double *vx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= p) return;
  double s1 = 0, s2 = 0;
  for (int i = lane; i < n; i += 64) { const double v = (double)xval(X, xoff(i, j, R, p)); s1 += v; s2 = fma(v, v, s2); }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  s1 = __shfl(s1, 0, 64); s2 = __shfl(s2, 0, 64);
  const double mean = s1 / (double)n;
  double sv = 0;
  for (int i = lane; i < n; i += 64) { const double dev = (double)xval(X, xoff(i, j, R, p)) - mean; sv = fma(dev, dev, sv); }
  sv = wave_sum(sv);
  if (lane == 0) { xx[j] = s2; vx[j] = sv / (double)(n - 1); }
}
__global__ void k_dsum_stage1(const double *v, int64_t n, double *part, int square) {
  __shared__ double red[17];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += square ? v[i] * v[i] : v[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// setup: mu = mean(y), e = y - mu, vy = var(y), MSx, priors (R/wgr.R:49-59); part = 256 partial sums of column variances,
// part2 = 256 partial sums of xx
__global__ void k_wgr_init(const double *y, double *eR, int n, int64_t ld, const double *part, const double *part2, int p,
                           double df, double R2, WgrScalars *ws) {
  __shared__ double red[17];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += y[i];
  s = block_sum(s, red);
  const double mu = s / (double)n;
  double sv = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const double dev = y[i] - mu; sv += dev * dev; }
  sv = block_sum(sv, red);
  double ms = 0, sx = 0;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) { ms += part[i]; sx += part2[i]; }
  ms = block_sum(ms, red);
  sx = block_sum(sx, red);
  for (int64_t i = threadIdx.x; i < ld; i += blockDim.x) eR[i] = (i < n) ? (y[i] - mu) : 0.0;
  if (threadIdx.x == 0) {
    WgrScalars w; memset(&w, 0, sizeof(w));
    w.mu = mu; w.vy = sv / (double)(n - 1); w.MSx = ms; w.Ve = 1.0; w.Va = ms;
    w.Sb = (R2) * df * w.vy / ms; w.Se = (1 - R2) * df * w.vy; w.cxx = sx / (double)p;
    w.Sk = R2 * w.vy * (df + 2); w.Vp = 1.0;                                       // R/wgr.R:60, :30
    *ws = w;
  }
}
__global__ void k_wgr_marker_init(double *bR, double *dR, double *VbR, double *LR, double *B, double *D, double *VB, int p, const WgrScalars *ws) {
  const double Va = ws->Va, Ve = ws->Ve;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p; j += gridDim.x * blockDim.x) {
    bR[j] = 0; dR[j] = 1; VbR[j] = Va; LR[j] = Va / Ve; B[j] = 0; D[j] = 0; VB[j] = 0;   // L = Vb/Ve (sic), R/wgr.R:55
  }
}
// narrowing at the .Call boundary (src/RcppExports.cpp:20-27): double R vectors -> float KMUP arguments
__global__ void k_wgr_pre(const double *bR, const double *dR, const double *LR, const double *xx64, const double *eR, float *bf, float *df_,
                          float *Lf, float *xxf, double *e64, int p, int n, int64_t ld, float pi, const WgrScalars *ws, ChainScalars *sc) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = gid; j < p; j += gsz) { bf[j] = (float)bR[j]; df_[j] = (float)dR[j]; Lf[j] = (float)LR[j]; xxf[j] = (float)xx64[j]; }
  for (int64_t i = gid; i < ld; i += gsz) e64[i] = (i < n) ? (double)(float)eR[i] : 0.0;
  if (gid == 0) {
    ChainScalars c; memset(&c, 0, sizeof(c));
    const float Ve = (float)ws->Ve;
    c.ve = Ve; c.pi = pi; c.C = -0.5f / sqrtf(Ve); c.odds = pi / (1.0f - pi); c.dfp1 = 1.0f;
    c.inc_rate = 1.0f - pi;
    *sc = c;
  }
}
// widening of KMUP's outputs + marker-variance step (R/wgr.R:86-111)
__global__ void k_wgr_post(const float *bf, const float *df_, double *bR, double *dR, double *VbR, int p, int use_d, int iv, int de,
                           double dfv, uint32_t iter, Rng rng, const WgrScalars *ws) {
  const double Sb = ws->Sb, Ve = ws->Ve, MSx = ws->MSx;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p; j += gridDim.x * blockDim.x) {
    const double b = (double)bf[j];
    bR[j] = b;
    if (use_d) dR[j] = (double)df_[j];
    if (iv) VbR[j] = de ? sqrt(b * b * Ve / MSx) : (Sb + b * b) / rng_chisq(rng, dfv + 1.0, (uint32_t)j, iter, RNG_CHI);
  }
}
// Va (common variance) and Ve draws (R/wgr.R:113,121); e64 holds KMUP's residual (float values)
__global__ __launch_bounds__(1024) void k_wgr_scal(const double *e64, int n, double n_dof, int p, const double *bbpart, int iv, double dfv, uint32_t iter, Rng rng, WgrScalars *ws) {
  __shared__ double red[17];
  double ee = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const double v = (double)(float)e64[i]; ee += v * v; }
  ee = block_sum(ee, red);
  double bb = 0;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) bb += bbpart[i];
  bb = block_sum(bb, red);
  if (threadIdx.x == 0) {
    if (!iv) ws->Va = (bb + ws->Sb) / rng_chisq(rng, dfv + (double)p, RNG_GLOBAL_MARKER, iter, RNG_G_VB);
    ws->Ve = (ee + ws->Se) / rng_chisq(rng, n_dof + dfv, RNG_GLOBAL_MARKER, iter, RNG_G_VE);   // n*bag + df, R/wgr.R:121
    ws->bb = bb;
  }
}
// L = Ve/Vb (R/wgr.R:122) and posterior sums of the marker vectors (R/wgr.R:130-134)
__global__ void k_wgr_L(const double *bR, const double *dR, double *VbR, double *LR, double *B, double *D, double *VB, int p, int iv, int accumulate, const WgrScalars *ws) {
  const double Ve = ws->Ve, Va = ws->Va;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p; j += gridDim.x * blockDim.x) {
    if (!iv) VbR[j] = Va;
    LR[j] = Ve / VbR[j];
    if (accumulate) { B[j] += bR[j]; D[j] += dR[j]; if (iv) VB[j] += VbR[j]; }
  }
}
// e = y - mu - X b from the fp64 partial products (R/wgr.R:124)
__global__ void k_wgr_efinish(const double *part, int64_t ld, int nchunks, int n, const double *y, double *eR, const WgrScalars *ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * ld + i];
  eR[i] = y[i] - ws->mu - s;
}
// intercept (R/wgr.R:125-127) and scalar posterior sums
__global__ __launch_bounds__(1024) void k_wgr_mu(double *eR, int n, int iv, int accumulate, uint32_t iter, Rng rng, WgrScalars *ws) {
  __shared__ double red[17];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += eR[i];
  s = block_sum(s, red);
  const double mu0 = s / (double)n + (ws->Ve / (double)n) * rng_normal(rng, RNG_GLOBAL_MARKER, iter, RNG_G_MU, 0);   // sd = Ve/n (sic)
  for (int i = threadIdx.x; i < n; i += blockDim.x) eR[i] -= mu0;
  __syncthreads();
  if (threadIdx.x == 0) {
    ws->mu += mu0;
    if (accumulate) { ws->B0 += ws->mu; ws->VE += ws->Ve; if (!iv) ws->VA += ws->Va; }
  }
}
// ---- polygenic term: narrowing for KMUP(U,h,dh,xxK,e,Lk,Ve,0) (R/wgr.R:70-76), widening of its outputs ----
__global__ void k_wgr_pre_k(const double *hR, const double *Vd, const double *eR, float *hf, float *dhf, float *xxKf, float *Lkf, double *e64,
                            int pk, int n, int64_t ld, const WgrScalars *ws, ChainScalars *sc) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  const double Ve = ws->Ve, Vp = ws->Vp;
  for (int64_t k = gid; k < pk; k += gsz) { hf[k] = (float)hR[k]; dhf[k] = 0.0f; xxKf[k] = 1.0f; Lkf[k] = (float)(Ve / (Vd[k] * Vp)); }
  for (int64_t i = gid; i < ld; i += gsz) e64[i] = (i < n) ? (double)(float)eR[i] : 0.0;
  if (gid == 0) {
    ChainScalars c; memset(&c, 0, sizeof(c));
    const float Vef = (float)Ve;
    c.ve = Vef; c.pi = 0.0f; c.C = -0.5f / sqrtf(Vef); c.odds = 0.0f; c.dfp1 = 1.0f;
    *sc = c;
  }
}
__global__ void k_wgr_post_k(const float *hf, double *hR, const double *e64, double *eR, int pk, int n) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = gid; k < pk; k += gsz) hR[k] = (double)hf[k];
  for (int64_t i = gid; i < n; i += gsz) eR[i] = (double)(float)e64[i];   // KMUP returns e as float (src/Rcpp20260726ai.cpp:37)
}
// Vp = (sum(h^2/V) + Sk)/rchisq(1, df+pk)  (R/wgr.R:117)
__global__ __launch_bounds__(1024) void k_wgr_vp(const double *hR, const double *Vd, int pk, double dfv, uint32_t iter, Rng rng, WgrScalars *ws) {
  __shared__ double red[17];
  double s = 0;
  for (int k = threadIdx.x; k < pk; k += blockDim.x) s += hR[k] * hR[k] / Vd[k];
  s = block_sum(s, red);
  if (threadIdx.x == 0) ws->Vp = (s + ws->Sk) / rng_chisq(rng, dfv + (double)pk, RNG_GLOBAL_MARKER, iter, RNG_G_VK);
}
// out[i] (+)= sum_k U[i,k] * coef[k] * scale   (U %*% h in double, R/wgr.R:124,148)
__global__ void k_uh(const double *Ud, const double *coef, int n, int pk, double scale, double *out, int accumulate_into_neg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int k = 0; k < pk; ++k) s = fma(Ud[(size_t)k * n + i], coef[k] * scale, s);
  if (accumulate_into_neg) out[i] -= s; else out[i] = s;
}
__global__ void k_wgr_accum_k(const double *hR, double *H, int pk, WgrScalars *ws) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < pk; k += gridDim.x * blockDim.x) H[k] += hR[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) ws->VP += ws->Vp;
}
__global__ void k_add_vec(double *a, const double *b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
// ---- bagging (bag != 1): KMUP2 works on a row subsample (src/Rcpp20260726ai.cpp:49-57) ----
// rows Use[0..nb) of the base panel -> a panel of nb rows (both slab-major, each with its own slab height)
template <typename XT>
__global__ void k_gather_rows(const XT *Xb, int Rb, const int *use, int nb, XT *Xo, int Ro, int64_t ldo, int64_t p) {
  const int64_t total = p * ldo;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx / ldo, i = idx - j * ldo;
    XT v = (XT)0;
    if (i < nb) v = Xb[xoff(use[i], j, Rb, p)];
    Xo[xoff(i, j, Ro, p)] = v;
  }
}
// int8 panels: a workgroup stages whole columns in LDS with 16-byte loads, picks the subsample's bytes there and writes the
// bagged panel with 16-byte stores (the element-wise kernel above moved one byte per load and ran at 0.6 TB/s)
__global__ __launch_bounds__(256) void k_gather_rows_i8(const int8_t *Xb, int Rb, int64_t ldb, const int *use, int nb, int8_t *Xo, int Ro,
                                                         int64_t ldo, int64_t p, int mpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char col[];   // mpw columns of ldb bytes
  const int64_t j0 = (int64_t)blockIdx.x * mpw;
  const int nm = (int)min((int64_t)mpw, p - j0);
  const int cin = (int)(ldb / 16), cout = (int)(ldo / 16);
  for (int c = threadIdx.x; c < nm * cin; c += 256) {
    const int jj = c / cin, ci = c - jj * cin;
    const int64_t i = 16 * (int64_t)ci;
    reinterpret_cast<uint4 *>(col + (size_t)jj * ldb)[ci] = *reinterpret_cast<const uint4 *>(Xb + xoff(i, j0 + jj, Rb, p));
  }
  __syncthreads();
  for (int co = threadIdx.x; co < cout; co += 256) {   // the 16 row indices of an output chunk serve every column of the group
    int ui[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int i = 16 * co + k; ui[k] = (i < nb) ? use[i] : -1; }
    for (int jj = 0; jj < nm; ++jj) {
      const unsigned char *cj = col + (size_t)jj * ldb;
      uint32_t w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int u = ui[4 * q + k]; v |= ((u >= 0) ? (uint32_t)cj[u] : 0u) << (8 * k); }
        w[q] = v;
      }
      *reinterpret_cast<uint4 *>(Xo + xoff(16 * (int64_t)co, j0 + jj, Ro, p)) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}
__global__ void k_gather_e(const double *eR, const int *use, int nb, int64_t ldo, double *e64) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ldo; i += (int64_t)gridDim.x * blockDim.x)
    e64[i] = (i < nb) ? (double)(float)eR[use[i]] : 0.0;
}
__global__ void k_scale_d(double *v, int64_t n, double s) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] *= s;
}
__global__ void k_set_bg(ChainScalars *sc, float bg) { sc->bg = bg; }
// posterior means (R/wgr.R:141-145)
__global__ void k_wgr_final(double *B, double *D, double *VB, int p, double mc, const double *dpart, int iv, WgrScalars *ws) {
  __shared__ double red[17];
  double sd = 0;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) sd += dpart[i];
  sd = block_sum(sd, red);
  const double meanD = sd / mc / (double)p;
  for (int j = threadIdx.x; j < p; j += blockDim.x) { D[j] = D[j] / mc; B[j] = B[j] / mc / meanD; if (iv) VB[j] = VB[j] / mc; }
  if (threadIdx.x == 0) ws->sumD = sd;
}
__global__ void k_hat64_finish(const double *part, int64_t ld, int nchunks, int n, double B0, double *hat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * ld + i];
  hat[i] = B0 + s;
}

// ---- synthetic genotypes: 4 rows per thread ----
__global__ void k_synth(int8_t *X, int64_t ld, int n, int64_t p, int64_t col0, uint32_t k0, uint32_t k1, float *freq) {
  const int64_t quads = ld / 4;
  const int64_t total = p * quads;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx / quads, q = idx - j * quads;
    const uint32_t jg = (uint32_t)(col0 + j);
    const uint4 fj = philox4x32_10(jg, 0u, 33u, 0u, k0, k1);
    const float f = 0.05f + 0.45f * ((float)(fj.x >> 8) * (1.0f / 16777216.0f));
    const uint32_t thr = (uint32_t)(f * 16777216.0f);
    if (q == 0 && freq) freq[j] = f;
    const uint4 a = philox4x32_10((uint32_t)q, jg, 32u, 0u, k0, k1);
    const uint4 c = philox4x32_10((uint32_t)q, jg, 32u, 1u, k0, k1);
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
    uint32_t packed = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t g = ((w[2 * r] >> 8) < thr) + ((w[2 * r + 1] >> 8) < thr);
      if (q * 4 + r >= n) g = 0;
      packed |= g << (8 * r);
    }
    *reinterpret_cast<uint32_t *>(X + j * ld + q * 4) = packed;
  }
}

__global__ void k_debug_variates(Rng g, int kind, double nu, uint32_t marker0, uint32_t iter, uint32_t purpose, int count, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t mk = marker0 + (uint32_t)i;
  double v;
  if (kind == 0) v = rng_normal(g, mk, iter, purpose, 0);
  else if (kind == 1) v = rng_uniform(g, mk, iter, purpose, 0);
  else v = rng_chisq(g, nu, mk, iter, purpose);
  out[i] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host objects
// ---- a fixed-point sweep (k_sweep3 / k_sweep2w's fixed-point streamers) that leaves its range is redone on the fp64 residual:
// the state it starts from is kept (12 bytes per marker and the residual: 12 MB against a 10 GB read at C4), and when the range flag
// comes back the state is restored, the flag cleared and sc->redo set, which lets the fp64 launches queued behind (redo_only) run ----
namespace {
struct SnapArgs { double *e, *se; float *b, *d, *vb, *sb, *sd, *svb; int64_t ld; int j0, j1; ChainScalars *sc; };
__global__ void k_range_snapshot(const SnapArgs s) {
  const int64_t nt = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < s.ld; i += nt) s.se[i] = s.e[i];
  for (int64_t j = s.j0 + t0; j < s.j1; j += nt) { s.sb[j] = s.b[j]; s.sd[j] = s.d[j]; if (s.vb) s.svb[j] = s.vb[j]; }
  if (t0 == 0) { s.sc->snap_sum_d = s.sc->sum_d; s.sc->snap_sum_b2 = s.sc->sum_b2; }
}
__global__ void k_range_recover(const SnapArgs s) {
  if (s.sc->error != 2u) return;
  const int64_t nt = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < s.ld; i += nt) s.e[i] = s.se[i];
  for (int64_t j = s.j0 + t0; j < s.j1; j += nt) { s.b[j] = s.sb[j]; s.d[j] = s.sd[j]; if (s.vb) s.vb[j] = s.svb[j]; }
}
__global__ void k_range_flag(ChainScalars *sc) {   // (after k_range_recover: every thread of it has read the status)
  if (sc->error == 2u) { sc->error = 0u; sc->redo = 1u; sc->nredo += 1u; sc->sum_d = sc->snap_sum_d; sc->sum_b2 = sc->snap_sum_b2; }
}
__global__ void k_redo_clear(ChainScalars *sc) { sc->redo = 0u; }
}  // namespace

// ------------------------------------------------------------------------------------------------
struct bwgr_panel {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t n = 0, p = 0, ld = 0;
  int is_f32 = 0;
  int m = 0, K = 0, R = 0;
  int64_t nblocks = 0;
  void *X = nullptr, *gram = nullptr, *gramx = nullptr, *gramx2 = nullptr, *gramx3 = nullptr, *gramp = nullptr;
  double *xspec2 = nullptr, *xspec3 = nullptr;   // [nblocks][SW_MAXM]: speculative cross terms of the lag-3 / lag-4 pipelines (k_spec)
  bool lag4_ok = false;       // the lag-4 streamer (ring of four tiles) fits the LDS at this geometry
  int nfeed = 2;              // q feeder workgroups of k_sweep2 (one gather + sum of K KB takes about a block period at K = 40)
  uint16_t *gramp16 = nullptr, *gramx16 = nullptr;   // 16-bit copies for the sequencer (int8 panels)
  int *gram16_bad = nullptr;
  bool gram16 = false;        // the copies are exact: every entry in 0..65535
  int pstride = 0;
  size_t x_bytes = 0, gram_bytes = 0;
  float *xx = nullptr, *vx = nullptr, *msx_dev = nullptr;
  float MSx = 0;
  double *xpart = nullptr, *qpart = nullptr;
  unsigned long long *dgran = nullptr;
  uint32_t *xflags = nullptr;
  unsigned char *xchg = nullptr; size_t xchg_bytes = 0;   // xflags | dgran | qpart in one allocation: one memset per launch
  size_t lds_bytes = 0, lds2_bytes = 0;
  int sweep_version = 2;   // 2: streamer/sequencer pipeline (k_sweep2); 1: replicated recurrence (k_sweep)
  unsigned long long *stamps = nullptr;   // diagnostic build only
  PreStage ps = {};
  const void *ps_owner = nullptr; int ps_iter = -1;   // whose sweep constants the scratch holds (a chain pre-stages a whole iteration once)
  // k_draws: the next iteration's state-independent variates, drawn on a second (low-priority) stream beside this iteration's sweep
  double *draws = nullptr; hipStream_t draws_stream = nullptr; hipEvent_t draws_ready = nullptr, draws_free = nullptr;
  bool draws_valid = false; Rng draws_rng = {}; uint32_t draws_iter = 0, draws_marker0 = 0; int draws_flags = 0, draws_j0 = 0, draws_j1 = 0; const void *draws_sc = nullptr;
  int gram_maxdist = 3;           // panel_build_gram stops at this block distance (the EM scratch panel needs 1)
  bwgr_panel *parent = nullptr;   // a clone shares the parent's read-only arrays (X, Gram, xx, vx) and owns only the scratch
  int nclones = 0;
  int nchains = 0;                // live chains on this handle: panel_destroy refuses while any is alive
  int debug_withhold = 0;         // test hook: the next sweeps run with slab workgroup 0 missing (bwgr_debug_withhold)
  // k_sweep3 (selection models on int8 panels, sweep3.hip.h)
  bool want3 = false;             // build what k_sweep3 needs with the panel (off for the per-iteration scratch panels of bagging and the EM family)
  bool e3_ready = false;
  int e3_D = 0;                   // fold-in lag in blocks; cross Gram arrays reach D-1 blocks back
  int K3 = 0, R3 = 0, sub3 = 0;   // streamer workgroups, rows of each, streamers per slab
  bool solo3 = true;              // a chain alone on the GPU runs 128-row streamers (BWGR_SOLO3=0: never)
  unsigned char *gx12 = nullptr;  // 16-bit panels: an included marker's distance-1 and distance-2 rows side by side (k_near_rows); root panels own it
  void *g3x[S3_MAXD] = {};        // g3x[d-1]: cross Gram blocks of distance d in the element type k_sweep3 reads (aliases the older arrays where they fit)
  bool g3own[S3_MAXD] = {};       // allocated here (not an alias)
  int xmax = 0;                   // largest |x| of an int8 panel
  int *xmax_dev = nullptr;
  unsigned long long *qsum3 = nullptr, *lists3 = nullptr;   // per handle (clones have their own)
  uint32_t epoch3 = 0;
  size_t lds3_bytes = 0;
  double *snap_e = nullptr; float *snap_b = nullptr, *snap_d = nullptr, *snap_vb = nullptr;   // state before a fixed-point sweep (range recovery)
  // the affine models' block solve as a triangular product (sweep2w.hip.h)
  bool winv_on = true;            // BWGR_WINV=0: the serial recurrence of k_sweep2's sequencer instead
  double *winv = nullptr;         // per handle: [nblocks][S2W_WDOUBLES], written by k_affine_inv before every affine sweep
  unsigned char *gxt[S2W_MAXDIST] = {};   // the cross Gram blocks as the sequencer's MFMA operand (k_gx_planes); shared with clones
  unsigned long long *qsumw = nullptr;    // per handle: the fixed-point streamers' slab-dot sums [nblocks][SW_MAXM][2]
  int wpf = 4, wahead = 5, wnq = 0, wlag_cap = 4;   // (BWGR_WLAG=5|6: distances 4 / 5 through LDS planes -- measured slower: C4-shape BayesA 22.3 / 23.0 / 24.9 ms per sweep at depth 4 / 5 / 6)   // BWGR_WPF / BWGR_WAHEAD / BWGR_WNQ (0: by the streamer count) / BWGR_WLAG, read when the panel is made
  bool wfx_on = true;             // BWGR_WFX=0: k_sweep2's streamers under the product sequencer instead of the fixed-point ones
  int winv_nd = 0;                // distances built = the deepest lag the affine sweeps can run, minus one
  size_t ldsw_bytes = 0;
  std::vector<hipStream_t> pair_streams;   // root panel: the streams pairs of chains run on (bwgr_chain_run_pair); owned here, so that they outlive every clone
  bool force3 = false;            // a pair run (bwgr_chain_run_pair): every selection sweep is k_sweep3's, whatever the inclusion rate
  float eng3_thr = 0.03f;         // k_sweep3 takes the sweeps whose chains hold fewer than this share of markers in the model (BWGR_ENG3_THR);
                                  // measured crossover at n = 10 000: us per block at 1.4 / 3.7 / 5.8 / 10.9 % inclusion: k_sweep3 2.26 / 3.73 / 5.56 / 12.1, k_sweep2 3.07 / 3.29 / 3.60 / 4.69
  // implicitly centred columns (bwgr_panel_set_centred; int8 panels with k_sweep3): the column sums, the centred |x_j - mean_j|^2 as floats (what
  // a chain's xx is then), both owned by the root panel; per handle the running block sums of s_k * drej_k of the current iteration
  bool cen = false;
  int32_t *csum = nullptr; float *xxc = nullptr;
  double *cpre = nullptr;
  hipStream_t own_stream = nullptr;
  // occupancy guard: the compute units this handle's enqueued sweeps hold while they run, the stream they run on, and an event behind the last of them
  hipStream_t pre_pair_stream = nullptr; bool pre_pair_set = false;   // the stream this handle ran on before a pair run moved it (restored by its next sweep alone)
  hipEvent_t guard_ev = nullptr; int guard_cus = 0; hipStream_t guard_stream = nullptr; bool guard_listed = false;
};

static bool panel_cen(const bwgr_panel *P) { return (P->parent ? P->parent : P)->cen; }

struct bwgr_chain {
  bwgr_panel *P = nullptr;
  int model = 0, iit = 0, ibi = 0, done = 0, rng_mode = 0;
  float itf = 0, bif = 0, pi = 0, df = 0, R2 = 0, Phi = 0;
  uint64_t seed = 0;
  float *y = nullptr, *b = nullptr, *d = nullptr, *vb = nullptr, *lam = nullptr;
  double *e = nullptr;
  float *B = nullptr, *D = nullptr, *VB = nullptr;
  ChainScalars *sc = nullptr;
  int64_t marker0 = 0, p_total = 0;   // sharding: global id of local marker 0, markers over all ranks
  float MSx_eff = 0;                  // MSx over all ranks
  bool e_owned = true;
  double *e0 = nullptr;               // residual at the start of the current exchange round (sharded stepping)
  int flags_extra = 0;                // two-effect BayesB2: SWF_ALT_B2 (the likelihood comparison uses the drawn alternative)
  std::vector<hipEvent_t> ev;  // pairs around each sweep launch since the last query
  float ms_acc = 0; int launch_acc = 0;
  bool finalized = false;
};

// Concurrent chains (bwgr_panel_clone) want one hardware queue per stream; the runtime's default is four.  Set before the
// first HIP call of the process unless the user chose a value.
__attribute__((constructor)) static void bwgr_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// device -> host copy ordered on the panel's stream (which may be a non-blocking one: the null stream does not wait for it)
static hipError_t d2h(hipStream_t st, void *dst, const void *src, size_t bytes) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

static Rng make_rng(uint64_t seed, int mode) {
  Rng g; g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32); g.degenerate = (mode == BWGR_RNG_DEGENERATE); return g;
}

static int require_device(int device) {
  int c = 0; bwgr_device_count(&c);
  if (c <= 0) return fail(BWGR_ENODEV, "no HIP device visible: libbwgr_hip has no CPU fallback");
  if (device < 0 || device >= c) return fail(BWGR_EINVAL, "device %d out of range (%d visible)", device, c);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(BWGR_ENODEV, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  HIPCHK(hipSetDevice(device));
  return BWGR_OK;
}

template <typename XT> static int max_slab_rows(int m) {
  int best = 0;
  for (int R = 128; R <= 4096; R += 128)
    if (sweep_lds_bytes<XT>(m, R) <= (size_t)160 * 1024 && (size_t)m * R * sizeof(XT) <= (size_t)SW_TCH * 16 * (SW_THREADS - 64)) best = R;
  return best;
}

// the words the workgroups poll (flags, delta granules, q words and the feeders' sums) live in one allocation
static hipError_t alloc_exchange(bwgr_panel *P) {
  const size_t K = (size_t)P->K;
  const size_t fb = (sizeof(uint32_t) * (K + 1) * SW_FLAG_STRIDE + 255) & ~(size_t)255;
  const size_t gb = (sizeof(unsigned long long) * S2_NSLOT * SW_MAXM + 255) & ~(size_t)255;
  const size_t qb = sizeof(double) * S2_NSLOT * (K + 1) * SW_MAXM;
  P->xchg_bytes = fb + gb + qb;
  hipError_t e = hipMalloc(&P->xchg, P->xchg_bytes);
  if (e != hipSuccess) return e;
  P->xflags = reinterpret_cast<uint32_t *>(P->xchg);
  P->dgran = reinterpret_cast<unsigned long long *>(P->xchg + fb);
  P->qpart = reinterpret_cast<double *>(P->xchg + fb + gb);
  return hipSuccess;
}
// polled words are zeroed before every launch (epochs count within a launch)
// ---- occupancy guard -------------------------------------------------------------------------------------------------------------
// The sweep kernels' workgroups wait for one another (slab-dot exchanges, the sequencer's decisions), so every workgroup of a launch has
// to be resident at once -- beside the workgroups of whatever other handles' sweeps are in flight on the same device.  A launch that would
// not fit spins to its wall-clock bound and ends in BWGR_ETIMEOUT; the guard refuses it up front with BWGR_EINVAL instead.  The sweep's
// launch code runs twice: once "dry" (g_plan set: SPIN_LAUNCH records kernel, grid, threads and LDS instead of launching, and nothing
// else is enqueued or allocated), then for real.
struct SpinLaunch { const void *fn; int grid, threads; size_t lds; };
static thread_local std::vector<SpinLaunch> *g_plan = nullptr;
#define SWEEP_DRY (g_plan != nullptr)
// resident: the workgroups of the grid that stay for the sweep (L2 prefetch workgroups beyond the first few leave at once)
#define SPIN_LAUNCH_N(resident, kern, grid, blk, lds, stream, ...)                                                                   \
  do {                                                                                                                             \
    if (g_plan) g_plan->push_back(SpinLaunch{reinterpret_cast<const void *>(kern), (int)(resident), (int)dim3(blk).x, (size_t)(lds)}); \
    else hipLaunchKernelGGL(kern, grid, blk, lds, stream, __VA_ARGS__);                                                             \
  } while (0)
#define SPIN_LAUNCH(kern, grid, blk, lds, stream, ...) SPIN_LAUNCH_N(dim3(grid).x, kern, grid, blk, lds, stream, __VA_ARGS__)
static std::mutex g_guard_mu;
static std::vector<bwgr_panel *> g_guard_panels;   // handles with a guard event (any device)
// the arithmetic (also bwgr_debug_occupancy_fits, which the CPU tests call): a launch of `grid` workgroups, `per_cu` of which fit one
// compute unit, needs ceil(grid / per_cu) units; it fits when those and the `busy` units of other streams' sweeps are within `cus`
static int occupancy_fits(int grid, int per_cu, int cus, int busy, int *need) {
  if (need) *need = 0;
  if (grid < 1 || cus < 1 || busy < 0) return BWGR_EINVAL;
  if (per_cu < 1) return BWGR_EINVAL;
  const int nd = (grid + per_cu - 1) / per_cu;
  if (need) *need = nd;
  return (nd + busy <= cus) ? BWGR_OK : BWGR_EINVAL;
}
extern "C" int bwgr_debug_occupancy_fits(int grid, int per_cu, int cus, int busy, int *need) { return occupancy_fits(grid, per_cu, cus, busy, need); }
static bool guard_on() { const char *g = getenv("BWGR_OCC_GUARD"); return !(g && g[0] == '0'); }
static int guard_per_cu(const SpinLaunch &L) {
  static std::mutex mu; static std::vector<std::pair<SpinLaunch, int>> cache;
  std::lock_guard<std::mutex> lk(mu);
  for (auto &c : cache) if (c.first.fn == L.fn && c.first.threads == L.threads && c.first.lds == L.lds) return c.second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, L.fn, L.threads, L.lds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
  cache.push_back({L, nb});
  return nb;
}
static int device_cus(int device) {
  static std::mutex mu; static std::vector<int> cus;
  std::lock_guard<std::mutex> lk(mu);
  if ((int)cus.size() <= device) cus.resize(device + 1, 0);
  if (!cus[device]) { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus[device] = prop.multiProcessorCount; else (void)hipGetLastError(); }
  return cus[device];
}
// compute units the recorded launches hold: launches of one sweep follow one another on one stream, so the largest of them
static int plan_cus(const std::vector<SpinLaunch> &plan, int cus, int busy, int *need_out) {
  int need = 0;
  for (const SpinLaunch &L : plan) {
    const int per = guard_per_cu(L);
    int nd = 0;
    if (per < 1) return fail(BWGR_EINVAL, "occupancy guard: a sweep kernel (%d threads, %zu bytes of LDS) does not fit a compute unit", L.threads, L.lds);
    if (occupancy_fits(L.grid, per, cus, busy, &nd) != BWGR_OK)
      return fail(BWGR_EINVAL, "occupancy guard: a sweep launch of %d workgroups (%d per compute unit) needs %d compute units; %d of %d are held by other handles' sweeps "
                  "in flight (their workgroups wait for one another, so all must be resident at once: run fewer chains side by side -- bwgr_panel_max_concurrent -- or "
                  "wait for the others)", L.grid, per, nd, busy, cus);
    need = std::max(need, nd);
  }
  *need_out = need;
  return BWGR_OK;
}
// units held by sweeps in flight on other streams of P's device (handles whose event has completed drop out)
static int guard_busy(const bwgr_panel *P, hipStream_t mine, const bwgr_panel *partner = nullptr) {
  std::vector<std::pair<hipStream_t, int>> per_stream;
  for (bwgr_panel *Q : g_guard_panels) {
    if (Q == P || Q == partner || Q->device != P->device || Q->guard_cus == 0) continue;   // (a pair's stream waits for both handles' earlier sweeps)
    if (hipEventQuery(Q->guard_ev) == hipSuccess) { Q->guard_cus = 0; continue; }
    (void)hipGetLastError();   // (hipErrorNotReady)
    if (Q->guard_stream == mine) continue;   // the same stream: one after the other
    bool seen = false;
    for (auto &ps : per_stream) if (ps.first == Q->guard_stream) { ps.second = std::max(ps.second, Q->guard_cus); seen = true; }
    if (!seen) per_stream.push_back({Q->guard_stream, Q->guard_cus});
  }
  int busy = 0;
  for (auto &ps : per_stream) busy += ps.second;
  return busy;
}
// after the real launches: this handle holds `need` units until the event behind them completes
static void guard_mark(bwgr_panel *P, hipStream_t st, int need) {
  if (need <= 0) return;
  std::lock_guard<std::mutex> lk(g_guard_mu);
  if (!P->guard_ev) { if (hipEventCreateWithFlags(&P->guard_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); P->guard_ev = nullptr; return; } }
  if (!P->guard_listed) { g_guard_panels.push_back(P); P->guard_listed = true; }
  if (P->guard_cus > 0 && P->guard_stream == st && hipEventQuery(P->guard_ev) != hipSuccess) { (void)hipGetLastError(); need = std::max(need, P->guard_cus); }
  P->guard_cus = need; P->guard_stream = st;
  (void)hipEventRecord(P->guard_ev, st);
}
static void guard_forget(bwgr_panel *P) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  for (size_t i = 0; i < g_guard_panels.size(); ++i) if (g_guard_panels[i] == P) { g_guard_panels.erase(g_guard_panels.begin() + i); break; }
  if (P->guard_ev) { (void)hipEventDestroy(P->guard_ev); P->guard_ev = nullptr; }
  P->guard_listed = false; P->guard_cus = 0;
}

static int reset_exchange(bwgr_panel *P) {
  if (SWEEP_DRY) return BWGR_OK;
  if (P->sweep_version >= 2) {
    HIPCHK(hipMemsetAsync(P->xchg, 0, P->xchg_bytes, P->stream));
  } else if (P->K > 1) {
    HIPCHK(hipMemsetAsync(P->xflags, 0, sizeof(uint32_t) * ((size_t)P->K + 1) * SW_FLAG_STRIDE, P->stream));
  }
  return BWGR_OK;
}

static void launch_gramx_i8(bwgr_panel *P, int32_t *g, int dist);
// ---- k_sweep3 (sweep3.hip.h): which launches take it, and what it needs beside the panel ----
static int sweep3_lag(const bwgr_panel *P) { return P->e3_D; }
static bool use_sweep3(const bwgr_panel *P, int flags) {
  return P->sweep_version == 3 && P->e3_ready && (flags & SWF_SELECT) != 0 && (flags & SWF_EM_ANY) == 0;
}
// geometry, scratch and attributes (every handle: panels and clones)
static int sweep3_alloc_scratch(bwgr_panel *P) {
  HIPCHK(hipMalloc(&P->qsum3, sizeof(unsigned long long) * 2 * SW_MAXM * (size_t)P->nblocks));
  HIPCHK(hipMalloc(&P->lists3, sizeof(unsigned long long) * S3_LSTRIDE * (size_t)P->nblocks));
  HIPCHK(hipMemsetAsync(P->lists3, 0, sizeof(unsigned long long) * S3_LSTRIDE * (size_t)P->nblocks, P->stream));
  return BWGR_OK;
}
// the cross Gram arrays of distance 2 .. D-1 in the element type of the 16-bit (or, failing that, 32-bit) staging
static int sweep3_build(bwgr_panel *P) {
  P->e3_ready = false;
  if (P->is_f32 || !P->want3 || P->sweep_version != 3) return BWGR_OK;
  const int m = P->m;
  int R3 = (P->R % 256 == 0) ? 256 : 128;
  if (const char *rv = getenv("BWGR_R3")) { const int v = atoi(rv); if ((v == 64 || v == 128 || v == 256) && P->R % v == 0) { R3 = v; P->solo3 = false; } }   // (an explicit height holds for every launch)
  const int sub = P->R / R3, K3 = P->K * sub;
  int D = 12;   // (the streamers fold a list whose words they saw a step ahead: more lag than the fold itself needs -- C4: 12.45 ms at 8, 11.27 at 9, 10.78 at 10, 10.41 at 11, 10.37 at 12, 10.48 at 13)
  // (at least 2: a block's list leaves the sequencer while the next block is in its rounds)
  if (const char *dv = getenv("BWGR_D3")) { const int v = atoi(dv); if (v >= 2 && v <= S3_MAXD) D = v; }
  D = (int)std::min<int64_t>(D, std::max<int64_t>(2, P->nblocks));
  const size_t lds = std::max(std::max(s3_streamer_lds(R3), std::max(s3_streamer_dma_lds(128), R3 == 256 ? s3_streamer_dma_lds(256) : (size_t)0)), s3_seq_lds(D, P->gram16));
  // the slab dots are summed as integers: sum over all rows of |x| * 128 per digit, four digits of 8 bits, 8 bits of arrival count
  if (K3 > 255 || K3 + 1 > 256 || lds > (size_t)160 * 1024 || (int64_t)P->ld * std::max(P->xmax, 1) >= (1ll << 23) || (size_t)m * R3 > (size_t)4 * 16 * SW_THREADS) {
    P->sweep_version = 2;
    return BWGR_OK;
  }
  P->R3 = R3; P->sub3 = sub; P->K3 = K3; P->e3_D = D; P->lds3_bytes = lds;
  if (const char *sv3 = getenv("BWGR_SOLO3")) P->solo3 = sv3[0] != '0';
  if (const char *tv = getenv("BWGR_ENG3_THR")) { const float v = (float)atof(tv); if (v > 0.0f) P->eng3_thr = v; }
  const size_t blk_elems = (size_t)P->nblocks * m * m;
  const bool g16 = P->gram16;
  const int Dbuild = D;
  int32_t *tmp = nullptr;
  for (int d = 1; d < Dbuild; ++d) {
    if (P->nblocks <= d) { P->g3x[d - 1] = nullptr; continue; }
    if (d == 1) { P->g3x[0] = g16 ? (void *)P->gramx16 : P->gramx; continue; }
    if (!g16 && d == 2 && P->gramx2) { P->g3x[1] = P->gramx2; continue; }
    if (!g16 && d == 3 && P->gramx3) { P->g3x[2] = P->gramx3; continue; }
    void *arr = nullptr;
    HIPCHK(hipMalloc(&arr, blk_elems * (g16 ? 2 : 4)));
    P->g3x[d - 1] = arr; P->g3own[d - 1] = true;
    if (g16) {
      const int32_t *src;
      if (d == 2 && P->gramx2) src = (const int32_t *)P->gramx2;
      else if (d == 3 && P->gramx3) src = (const int32_t *)P->gramx3;
      else {
        if (!tmp) HIPCHK(hipMalloc(&tmp, blk_elems * 4));
        launch_gramx_i8(P, tmp, d);
        src = tmp;
      }
      hipLaunchKernelGGL(k_gram_narrow, dim3(2048), dim3(256), 0, P->stream, src + (size_t)d * m * m, (uint16_t *)arr + (size_t)d * m * m, (int64_t)(P->nblocks - d) * m * m, P->gram16_bad);
    } else launch_gramx_i8(P, (int32_t *)arr, d);
    HIPCHK(hipGetLastError());
  }
  int bad = 0;
  if (g16) HIPCHK(hipMemcpyAsync(&bad, P->gram16_bad, sizeof(int), hipMemcpyDeviceToHost, P->stream));
  HIPCHK(hipStreamSynchronize(P->stream));
  if (tmp) hipFree(tmp);
  if (bad) {   // an entry of a far block left the 16-bit range although the near blocks fit: rare; leave the panel to k_sweep2
    for (int d = 1; d < S3_MAXD; ++d) if (P->g3own[d - 1]) { hipFree(P->g3x[d - 1]); P->g3x[d - 1] = nullptr; P->g3own[d - 1] = false; }
    P->sweep_version = 2;
    return BWGR_OK;
  }
  {   // 16-bit panels: an included marker's distance-1 / 2 rows in one piece
    if (g16) {
      HIPCHK(hipMalloc(&P->gx12, (size_t)P->nblocks * m * 2 * m * 2));
      hipLaunchKernelGGL(k_near_rows, dim3(4096), dim3(256), 0, P->stream, (const uint16_t *)P->g3x[0], (const uint16_t *)(D >= 3 ? P->g3x[1] : nullptr), (uint16_t *)P->gx12, m, (int64_t)P->nblocks);
      HIPCHK(hipGetLastError());
    }
  }
  CHK(sweep3_alloc_scratch(P));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep3<uint16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep3<int32_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep3<uint16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep3<int32_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep3p<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep3p<int32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  P->e3_ready = true;
  return BWGR_OK;
}
// the DMA streamer's lane offsets are 32-bit: (columns of the launch) * (rows of a slab) bytes must stay below 4 GiB
static bool stream3_dma_fits(int64_t ncols, int64_t R) { return ncols >= 0 && R > 0 && (uint64_t)ncols * (uint64_t)R < (1ull << 32); }
extern "C" int bwgr_debug_stream3_dma(int64_t ncols, int64_t R) { return stream3_dma_fits(ncols, R) ? 1 : 0; }
// what one launch of k_sweep3 / k_sweep3p needs beside the sweep's own arguments; zeroes the launch's slab-dot sums, takes a new epoch
static void sweep3_args(bwgr_panel *P, const SweepArgs &a, Sweep3Args &A) {
  memset(&A, 0, sizeof(A));
  A.a = a;
  const bwgr_panel *root = P->parent ? P->parent : P;
  for (int d = 0; d < S3_MAXD; ++d) A.gx[d] = root->g3x[d];
  A.gp = root->gram16 ? (const void *)root->gramp16 : root->gramp;
  A.D = P->e3_D; A.K3 = P->K3; A.R3 = P->R3; A.sub = P->sub3; A.g16 = root->gram16 ? 1 : 0;
  A.qsum = P->qsum3; A.lists = P->lists3;
  A.gx12 = root->gram16 ? root->gx12 : nullptr;
  A.pf = -1; A.pf2 = -1;
#ifdef BWGR_EXPERIMENTS
  if (const char *dv = getenv("BWGR_DBG3")) A.dbg = atoi(dv);   // (timing switches, some of which break the chain: the experiment build only)
#endif
  {   // 128-row streamers land their tiles by LDS-DMA (s3_streamer_dma; C4 15.0 -> 13.65 ms per sweep); BWGR_STREAM3=reg: through registers, as the 256-row ones do
    // The DMA streamer forms a tile piece's source as a 32-bit lane offset from the launch's first column (no 64-bit vector arithmetic): only
    // launches whose column range spans less than 4 GiB of one slab take it (p * R < 2^32: 16.7 M markers at R = 256); wider ones keep the
    // register path, whose offsets are size_t.  bwgr_debug_stream3_dma() exposes the rule to the CPU tests.
    const char *sv = getenv("BWGR_STREAM3");
    const int64_t j_lo = (int64_t)a.blk_begin * a.m, j_hi = std::min<int64_t>(P->p, (int64_t)a.blk_end * a.m);
    const bool fits32 = stream3_dma_fits(j_hi - j_lo, P->R);
    if (!(sv && sv[0] == 'r') && fits32) A.dbg |= (1 << 22);
    if (sv && sv[0] == 'd' && fits32) A.dbg |= (1 << 23);   // (EXPERIMENT: the 256-row streamers too, three tile buffers)
  }
  if (SWEEP_DRY) return;
  P->epoch3 = (P->epoch3 + 1) & 0xFFFFFFu; if (P->epoch3 == 0) P->epoch3 = 1;
  A.epoch = P->epoch3;
  (void)hipMemsetAsync(P->qsum3 + (size_t)a.blk_begin * 2 * SW_MAXM, 0, sizeof(unsigned long long) * 2 * SW_MAXM * (size_t)(a.blk_end - a.blk_begin), P->stream);
}
static void launch_sweep3(bwgr_panel *P, const SweepArgs &a) {
  Sweep3Args A;
  sweep3_args(P, a, A);
  // one more workgroup, on the sequencer's XCD (workgroups with equal index mod 8 share an XCD), warms that XCD's L2 with what the
  // staging waves load; BWGR_PF3=1 switches it on
  // A chain that has the GPU to itself (a root panel without clones) runs 128-row streamers, two to a slab: 80 compute units instead
  // of 41, 15.98-16.18 against 16.49 ms per sweep at C4 (the same chain bit for bit: the slab dots are integer sums).  With clones
  // alive -- chains side by side, pairs -- every chain keeps the 256-row streamers the concurrency counts assume.  BWGR_SOLO3=0: never.
  const bool solo = P->solo3 && !P->parent && P->nclones == 0;
  if (solo && P->R3 == 256 && 2 * P->K3 + 1 <= 256) { A.R3 = 128; A.sub = P->R / 128; A.K3 = P->K * A.sub; }
  const char *pv = getenv("BWGR_PF3");
  // (on for a chain alone on the GPU: 15.61 -> 15.37 ms per sweep at C4 on the steadied kernel; beside other chains the workgroup is
  // not counted by bwgr_panel_max_concurrent, so it stays off there; BWGR_PF3=0|1 decides otherwise)
  const bool pf_on = (pv ? pv[0] == '1' : solo) && A.K3 + 2 <= 256;
  A.pf = pf_on ? ((A.K3 + 2 > 8) ? 8 : A.K3 + 1) : -1;
  A.qsplit = 1;
  { const char *sv = getenv("BWGR_SKIPVB"); A.skip_vb = ((a.flags & SWF_VB_VEC) && !(sv && sv[0] == '0')) ? 1 : 0; }
  const char *p2v = getenv("BWGR_PF3B");
  const bool pf2_on = pf_on && A.pf == 8 && A.gx12 && A.K3 + 3 > 16 && A.K3 + 3 <= 256 && !(p2v && p2v[0] == '0');
  A.pf2 = pf2_on ? 16 : -1;
  const dim3 grid(A.K3 + 1 + (pf_on ? 1 : 0) + (pf2_on ? 1 : 0)), blk(SW_THREADS);
  const bool cen = (a.flags & SWF_CENTRE) != 0;
  if (cen && !SWEEP_DRY) hipLaunchKernelGGL(k_cen_begin, dim3(1), dim3(1024), 0, P->stream, a, 0);
  if (A.g16) { if (cen) SPIN_LAUNCH((k_sweep3<uint16_t, true>), grid, blk, P->lds3_bytes, P->stream, A); else SPIN_LAUNCH((k_sweep3<uint16_t, false>), grid, blk, P->lds3_bytes, P->stream, A); }
  else { if (cen) SPIN_LAUNCH((k_sweep3<int32_t, true>), grid, blk, P->lds3_bytes, P->stream, A); else SPIN_LAUNCH((k_sweep3<int32_t, false>), grid, blk, P->lds3_bytes, P->stream, A); }
  if (cen && !SWEEP_DRY) hipLaunchKernelGGL(k_cen_end, dim3(64), dim3(256), 0, P->stream, a, 0);
  if (A.skip_vb && !SWEEP_DRY) {   // (every launch: idempotent -- after a range redo the fp64 engine has written the same values from the same expression)
    const int j0 = a.blk_begin * a.m, j1 = (int)std::min<int64_t>(P->p, (int64_t)a.blk_end * a.m);
    hipLaunchKernelGGL(k_vb_fill, dim3((unsigned)std::min<int64_t>(1024, (j1 - j0 + 255) / 256)), dim3(256), 0, P->stream, a, j0, j1);
  }
}

// The selection models' sweeps on a panel that has k_sweep3: the device picks the engine from the chain's current inclusion
// rate (ChainScalars::inc_rate against the panel's threshold), so both engines' kernels are enqueued and one side leaves at once
// (a few microseconds per iteration); a threshold >= 1 means k_sweep3 always and the other side is not enqueued at all.
static float sweep3_gate(const bwgr_panel *P, int flags) {
  if (!use_sweep3(P, flags)) return 0.0f;
  return (P->eng3_thr >= 1.0f || P->force3) ? INFINITY : P->eng3_thr;
}

// The affine sweeps of an int8 panel with 16-bit Gram staging run k_sweep2w: the block solve as a product with the inverse
// k_affine_inv forms before the sweep (sweep2w.hip.h).
static bool use_winv(const bwgr_panel *P, int flags) {
  if (!P->winv_on || P->sweep_version < 2 || P->is_f32 || !P->gramp || P->winv_nd < 1 || P->K > 2 * (S2W_QW + S2W_QX)) return false;
  if (flags & (SWF_SELECT | SWF_EM_ANY | SWF_SERIAL)) return false;
  return P->ldsw_bytes > 0 && P->ldsw_bytes <= (size_t)160 * 1024;
}
// ... with its own streamers (s2w_streamer_fx: 128 rows each, fixed-point residual) where the slab count allows
static bool use_wfx(const bwgr_panel *P) { return P->wfx_on && (P->R % S2W_FXR) == 0 && P->K * (P->R / S2W_FXR) <= 255; }
static int winv_alloc(bwgr_panel *P) {
  if (P->winv) return BWGR_OK;
  HIPCHK(hipMalloc(&P->winv, sizeof(double) * (size_t)S2W_WDOUBLES * (size_t)P->nblocks));
  HIPCHK(hipMalloc(&P->qsumw, sizeof(unsigned long long) * 4 * 2 * SW_MAXM * (size_t)P->nblocks));   // (up to four copies)
  return BWGR_OK;
}

static void launch_prestage(bwgr_panel *P, const SweepArgs &a_in) {
  SweepArgs a = a_in;
  a.gate3 = sweep3_gate(P, a.flags);
  const int j0 = a.blk_begin * a.m, j1 = (int)std::min<int64_t>(P->p, (int64_t)a.blk_end * a.m);
  const int64_t tasks = 4ll * (j1 - j0);
  const bool s3 = a.gate3 > 0.0f;
  const bool fxa = !s3 && use_winv(P, a.flags) && P->winv && use_wfx(P);   // an affine sweep on the fixed-point streamers
  int sh_add = 0;
  if (const char *dv = getenv("BWGR_DEBUG_SH_ADD")) sh_add = atoi(dv);   // test hook: less headroom, to leave the range on purpose
  if (s3 || fxa) hipLaunchKernelGGL(k_escale_reset, dim3(1), dim3(1), 0, P->stream, a.sc);
  // the variates drawn ahead (draws_ahead, below) when they are this very iteration's: same streams, same counters, same flags, this range inside theirs
  const int dflags = a.flags & (SWF_SELECT | SWF_VB_VEC);
  const bool have_draws = !SWEEP_DRY && P->draws_valid && P->draws_iter == a.iter && P->draws_marker0 == a.marker0 && P->draws_flags == dflags &&
                          P->draws_sc == (const void *)a.sc && P->draws_j0 <= j0 && P->draws_j1 >= j1 && memcmp(&P->draws_rng, &a.rng, sizeof(Rng)) == 0 &&
                          !(a.flags & (SWF_MH | SWF_EM_ANY));
  a.draws = nullptr;
  if (have_draws) { const hipError_t he = hipStreamWaitEvent(P->stream, P->draws_ready, 0); if (he == hipSuccess) a.draws = P->draws; else { fprintf(stderr, "bwgr: draws wait failed: %s\n", hipGetErrorString(he)); (void)hipGetLastError(); } }
  if (a.draws) {
    hipLaunchKernelGGL(k_prestage_fin, dim3((unsigned)std::min<int64_t>(2048, (j1 - j0 + 255) / 256)), dim3(256), 0, P->stream, a, j0, j1);
    { const hipError_t he = hipEventRecord(P->draws_free, P->stream); if (he != hipSuccess) { fprintf(stderr, "bwgr: draws_free record failed: %s\n", hipGetErrorString(he)); (void)hipGetLastError(); } }
    P->draws_valid = false;   // (consumed: the buffer is the next iteration's from here)
  } else hipLaunchKernelGGL(k_prestage, dim3((unsigned)std::min<int64_t>(4096, (tasks + 255) / 256)), dim3(256), 0, P->stream, a, j0, j1);
  if (s3) {   // the sweep's fixed-point scale, then the in-block speculative terms on that grid
    int xbits = 0; while ((1 << xbits) < std::max(1, (P->parent ? P->parent : P)->xmax)) ++xbits;
    hipLaunchKernelGGL(k_escale, dim3(1), dim3(1024), 0, P->stream, a.e, P->ld, a.sc, xbits, a.gate3, sh_add);
    if (a.flags & SWF_CENTRE) {   // the rejected steps' share of sum(e_stored), block by block (the whole panel: launch_prestage is called with every block)
      hipLaunchKernelGGL(k_cen_tot, dim3((unsigned)(a.blk_end - a.blk_begin)), dim3(128), 0, P->stream, a, a.blk_begin, 0);
      hipLaunchKernelGGL(k_cen_scan, dim3(1), dim3(1024), 0, P->stream, a, (int)P->nblocks, 0);
    }
    { const bwgr_panel *root = P->parent ? P->parent : P;
      hipLaunchKernelGGL(k_spec3, dim3((unsigned)(a.blk_end - a.blk_begin)), dim3(128), 0, P->stream, a, a.blk_begin, root->gram16 ? (const uint16_t *)root->gramp16 : (const uint16_t *)nullptr); }
    if (std::isinf(a.gate3)) return;
  }
  if (use_winv(P, a.flags) && P->winv) {
    if (use_wfx(P)) {   // the sweep's fixed-point scale: the residual, and what k_prestage knows of the steps (|b0|, the noise terms) times the largest |x|
      int xbits = 0; while ((1 << xbits) < std::max(1, (P->parent ? P->parent : P)->xmax)) ++xbits;
      hipLaunchKernelGGL(k_escale, dim3(1), dim3(1024), 0, P->stream, a.e, P->ld, a.sc, xbits, INFINITY, sh_add);
    }
    hipLaunchKernelGGL(k_affine_inv, dim3((unsigned)(a.blk_end - a.blk_begin)), dim3(512), S2W_INV_LDS, P->stream, a, P->winv, (a.flags & SWF_DELTA2) ? 2.0 : 1.0);
    return;
  }
  if (P->sweep_version >= 2) {
    const int sel = (a.flags & SWF_SELECT) ? 1 : 0;
    const unsigned nb = (unsigned)(a.blk_end - a.blk_begin);
    if ((a.flags & SWF_CENTRE) && sel) {   // the fp64 engine's share of an implicitly centred iteration (the float steps themselves; runs on k_sweep2's side of the gate)
      hipLaunchKernelGGL(k_cen_tot, dim3(nb), dim3(128), 0, P->stream, a, a.blk_begin, 1);
      hipLaunchKernelGGL(k_cen_scan, dim3(1), dim3(1024), 0, P->stream, a, (int)P->nblocks, 1);
    }
    if (P->is_f32) hipLaunchKernelGGL(k_spec<double>, dim3(nb), dim3(128), 0, P->stream, a, a.blk_begin, sel);
    else hipLaunchKernelGGL(k_spec<int32_t>, dim3(nb), dim3(128), 0, P->stream, a, a.blk_begin, sel);
  }
}

// The next iteration's variates, enqueued beside this iteration's sweep (see k_draws).  `a` = this iteration's arguments over the whole panel.  Selection
// models with the logistic step on panels whose sweeps leave most of the chip idle; BWGR_DRAWS=0 switches it off.  Failing to set it up is not an error:
// k_prestage draws for itself whenever the buffer is not this iteration's.
static void draws_ahead(bwgr_panel *P, const SweepArgs &a, hipEvent_t before_sweep) {
  if (SWEEP_DRY || !(a.flags & SWF_SELECT) || (a.flags & (SWF_MH | SWF_EM_ANY))) return;
  // only for a chain that has the GPU to itself (as the 128-row streamers and the prefetcher workgroups): beside other chains or shards the idle
  // compute units it would run on are theirs (five chains side by side 255 -> 226 chain-iter/s, three shards 163 -> 119 iter/s with it)
  if (!(P->solo3 && !P->parent && P->nclones == 0)) return;
  static const bool off = [] { const char *v = getenv("BWGR_DRAWS"); return v && v[0] == '0'; }();
  if (off) return;
  if (!P->draws) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // (lo: the numerically largest = the lowest priority)
    if (hipMalloc(&P->draws, sizeof(double) * 5 * (size_t)P->p) != hipSuccess || hipStreamCreateWithPriority(&P->draws_stream, hipStreamNonBlocking, lo) != hipSuccess ||
        hipEventCreateWithFlags(&P->draws_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&P->draws_free, hipEventDisableTiming) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_draws), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      hipFree(P->draws); P->draws = nullptr;
      return;
    }
    (void)hipEventRecord(P->draws_free, P->stream);
  }
  const int j0 = a.blk_begin * a.m, j1 = (int)std::min<int64_t>(P->p, (int64_t)a.blk_end * a.m);
  // after this iteration's k_prestage has read the buffer (draws_free) -- or, the first time, after what is enqueued so far; 160 workgroups with 96 KB of
  // LDS each: at most 160 compute units, none of them one that runs a workgroup of the sweep
  // (... and after the iteration's speculative terms, which are bandwidth-bound and would share the chip with it: the event in front of the sweep)
  { const hipError_t h1 = hipStreamWaitEvent(P->draws_stream, P->draws_free, 0), h2 = hipStreamWaitEvent(P->draws_stream, before_sweep, 0);
    if (h1 != hipSuccess || h2 != hipSuccess) { fprintf(stderr, "bwgr: draws_ahead waits failed: %s / %s\n", hipGetErrorString(h1), hipGetErrorString(h2)); (void)hipGetLastError(); return; } }
  hipLaunchKernelGGL(k_draws, dim3(160), dim3(1024), 96 * 1024, P->draws_stream, a.rng, a.marker0, a.iter + 1u, a.flags, (const ChainScalars *)a.sc, (int64_t)P->p, j0, j1, P->draws);
  { const hipError_t h1 = hipGetLastError(); const hipError_t h2 = (h1 == hipSuccess) ? hipEventRecord(P->draws_ready, P->draws_stream) : hipSuccess;
    if (h1 != hipSuccess || h2 != hipSuccess) { fprintf(stderr, "bwgr: k_draws launch / record failed: %s / %s\n", hipGetErrorString(h1), hipGetErrorString(h2)); (void)hipGetLastError(); P->draws_valid = false; return; } }
  P->draws_valid = true; P->draws_rng = a.rng; P->draws_iter = a.iter + 1u; P->draws_marker0 = a.marker0; P->draws_flags = a.flags & (SWF_SELECT | SWF_VB_VEC);
  P->draws_j0 = j0; P->draws_j1 = j1; P->draws_sc = (const void *)a.sc;
}

static void launch_sweep_kernel_inner(bwgr_panel *P, const SweepArgs &a_in, bool redo);
// The fixed-point engines between a snapshot of the state they start from and the fp64 engine that redoes the sweep if they left
// their range (the reference's update cannot fail, src/Rcpp20260726ai.cpp:681).  Off for the debug abort hook (its launches must time out).
static bool range_snapshot(bwgr_panel *P, const SweepArgs &a, SnapArgs &sn) {
  if (SWEEP_DRY) return true;
  const size_t p = (size_t)P->p;
  if (!P->snap_e) {
    if (hipMalloc(&P->snap_e, sizeof(double) * (size_t)P->ld) != hipSuccess || hipMalloc(&P->snap_b, sizeof(float) * p) != hipSuccess ||
        hipMalloc(&P->snap_d, sizeof(float) * p) != hipSuccess || hipMalloc(&P->snap_vb, sizeof(float) * p) != hipSuccess) { (void)hipGetLastError(); return false; }
  }
  sn.e = a.e; sn.se = P->snap_e; sn.b = a.b; sn.d = a.d; sn.vb = (a.flags & SWF_VB_VEC) ? a.vb : nullptr; sn.sb = P->snap_b; sn.sd = P->snap_d; sn.svb = P->snap_vb;
  sn.ld = P->ld; sn.j0 = a.blk_begin * a.m; sn.j1 = (int)std::min<int64_t>(P->p, (int64_t)a.blk_end * a.m); sn.sc = a.sc;
  hipLaunchKernelGGL(k_range_snapshot, dim3(256), dim3(256), 0, P->stream, sn);
  return true;
}
static void launch_sweep_kernel(bwgr_panel *P, const SweepArgs &a_in) {
  SweepArgs a = a_in;
  a.gate3 = sweep3_gate(P, a.flags);
  const bool fx = (a.gate3 > 0.0f) || (use_winv(P, a.flags) && P->winv && use_wfx(P));
  SnapArgs sn;
#ifdef BWGR_EXPERIMENTS
  static const bool no_recover = getenv("BWGR_NO_RECOVER") != nullptr;   // (timing experiments that break the chain on purpose: the experiment build only)
#else
  constexpr bool no_recover = false;
#endif
  const bool guarded = fx && !P->debug_withhold && !no_recover && range_snapshot(P, a, sn);
  launch_sweep_kernel_inner(P, a_in, false);
  if (guarded) {
    if (!SWEEP_DRY) {
      hipLaunchKernelGGL(k_range_recover, dim3(256), dim3(256), 0, P->stream, sn);
      hipLaunchKernelGGL(k_range_flag, dim3(1), dim3(1), 0, P->stream, a.sc);
    }
    (void)reset_exchange(P);
    launch_sweep_kernel_inner(P, a_in, true);
    if (!SWEEP_DRY) hipLaunchKernelGGL(k_redo_clear, dim3(1), dim3(1), 0, P->stream, a.sc);
  }
}
static void launch_sweep_kernel_inner(bwgr_panel *P, const SweepArgs &a_in, bool redo) {
  SweepArgs a = a_in;
  if (P->debug_withhold) a.flags |= SWF_DEBUG_WITHHOLD;
  a.gate3 = redo ? 0.0f : sweep3_gate(P, a.flags);
  a.redo_only = redo ? 1 : 0;
  if (redo && P->sweep_version >= 2 && !use_winv(P, a.flags)) {   // the fp64 engine's speculative terms (k_spec) of the state just restored
    const int sel = (a.flags & SWF_SELECT) ? 1 : 0;
    if (!SWEEP_DRY && (a.flags & SWF_CENTRE) && sel) {   // the running block sums on the float steps (the fixed-point launch left them on its grid): every block, then the scan
      SweepArgs all = a; all.blk_begin = 0; all.blk_end = (int)P->nblocks;
      hipLaunchKernelGGL(k_cen_tot, dim3((unsigned)P->nblocks), dim3(128), 0, P->stream, all, 0, 2);
      hipLaunchKernelGGL(k_cen_scan, dim3(1), dim3(1024), 0, P->stream, all, (int)P->nblocks, 2);
    }
    if (!SWEEP_DRY) hipLaunchKernelGGL(k_spec<int32_t>, dim3((unsigned)(a.blk_end - a.blk_begin)), dim3(128), 0, P->stream, a, a.blk_begin, sel);
  }
  if (a.gate3 > 0.0f) { launch_sweep3(P, a); if (std::isinf(a.gate3)) return; }
  const bool sel = (a.flags & SWF_SELECT) != 0;
  // streamers, sequencer, and for the selection models the q feeders (the affine recurrence is compute-bound: its
  // sequencer gathers q itself under the recurrence, and a feeder hop in its lag-2 chain measured 15 % slower)
  a.nfeed = (P->sweep_version >= 2 && sel) ? P->nfeed : 0;
  if (use_winv(P, a.flags) && P->winv) {
    S2WArgs A;
    memset(&A, 0, sizeof(A));
    A.winv = P->winv;
    A.fx = (use_wfx(P) && !redo) ? 1 : 0;
    if (!A.fx) a.lag = std::min(a.lag, 4);   // (k_sweep2's streamers -- the range-recovery launch, BWGR_WFX=0 -- hold four tiles)
    A.nd = std::min(a.lag - 1, (int)S2W_MAXDIST);
#ifdef BWGR_EXPERIMENTS
    if (const char *dv = getenv("BWGR_DBGW")) A.dbg = atoi(dv);
#endif
    for (int d = 0; d < S2W_MAXDIST; ++d) A.gxt[d] = P->gxt[d < P->winv_nd ? d : 0];
    A.npf = P->wpf;      // measured at C2: 0 -> 540, 2 -> 636, 4 -> 685 iter/s (6 and 8 no better)
    A.ahead = P->wahead;
    A.qsum = P->qsumw; A.sub = P->R / S2W_FXR; A.K3 = P->K * A.sub;
    A.nq = P->wnq ? P->wnq : (A.K3 > 48 ? 2 : 1);   // (C2, 40 streamers: one copy 1.10 ms, two 1.21; C4 shape, 80 streamers: 27.8 / 25.6 / 27.6 ms with 1 / 2 / 4)
    if (A.fx && !SWEEP_DRY) (void)hipMemsetAsync(P->qsumw + (size_t)a.blk_begin * A.nq * 2 * SW_MAXM, 0, sizeof(unsigned long long) * A.nq * 2 * SW_MAXM * (size_t)(a.blk_end - a.blk_begin), P->stream);
    // (of the 8 npf workgroups past the sequencer, the npf on its XCD prefetch; the others leave at once)
    if (A.fx) SPIN_LAUNCH_N(A.K3 + 1 + A.npf, k_sweep2w<true>, dim3(A.K3 + 1 + 8 * A.npf), dim3(S2W_THREADS), P->ldsw_bytes, P->stream, a, A);
    else SPIN_LAUNCH_N(P->K + 1 + A.npf, k_sweep2w<false>, dim3(P->K + 1 + 8 * A.npf), dim3(S2W_THREADS), P->ldsw_bytes, P->stream, a, A);
    return;
  }
  if (P->sweep_version >= 2) {
    const dim3 grid(P->K + 1 + a.nfeed), blk(SW_THREADS);
    const bool cen2 = (a.flags & SWF_CENTRE) && sel && !P->is_f32 && !SWEEP_DRY;
    struct CenEnd { bool on; hipStream_t st; SweepArgs a; int mode; ~CenEnd() { if (on) hipLaunchKernelGGL(k_cen_end, dim3(64), dim3(256), 0, st, a, mode); } } cen_end{cen2, P->stream, a, redo ? 2 : 1};
    if (cen2) hipLaunchKernelGGL(k_cen_begin, dim3(1), dim3(1024), 0, P->stream, a, redo ? 2 : 1);
    if (P->is_f32) {
      if (sel) SPIN_LAUNCH((k_sweep2<float, true>), grid, blk, P->lds2_bytes, P->stream, a);
      else SPIN_LAUNCH((k_sweep2<float, false>), grid, blk, P->lds2_bytes, P->stream, a);
    } else {
      if (P->gram16 && sel) {   // selection models: 16-bit staging and the single-barrier sequencer (the affine recurrence is
                                // compute-bound and measured faster on the 32-bit blocks: no conversion in its inner loop)
        SweepArgs a16 = a;
        a16.gramp = P->gramp16; a16.gramx = P->gramx16;
        SPIN_LAUNCH((k_sweep2<int8_t, true, uint16_t>), grid, blk, P->lds2_bytes, P->stream, a16);
      } else if (sel) SPIN_LAUNCH((k_sweep2<int8_t, true>), grid, blk, P->lds2_bytes, P->stream, a);
      else SPIN_LAUNCH((k_sweep2<int8_t, false>), grid, blk, P->lds2_bytes, P->stream, a);
    }
    return;
  }
  if (P->is_f32) {
    if (sel) SPIN_LAUNCH((k_sweep<float, true>), dim3(P->K), dim3(SW_THREADS), P->lds_bytes, P->stream, a);
    else SPIN_LAUNCH((k_sweep<float, false>), dim3(P->K), dim3(SW_THREADS), P->lds_bytes, P->stream, a);
  } else {
    if (sel) SPIN_LAUNCH((k_sweep<int8_t, true>), dim3(P->K), dim3(SW_THREADS), P->lds_bytes, P->stream, a);
    else SPIN_LAUNCH((k_sweep<int8_t, false>), dim3(P->K), dim3(SW_THREADS), P->lds_bytes, P->stream, a);
  }
}

// selection models run the deeper pipelines (their cross terms are sparse)
static bool use_winv(const bwgr_panel *P, int flags);
static void choose_lag(const bwgr_panel *P, SweepArgs &a) {
  const char *lv = getenv("BWGR_LAG");
  // Selection sweeps of k_sweep2: three blocks deep.  (The single-barrier sequencer also knows a fourth level, BWGR_LAG=4: it was
  // the default while k_sweep2 also ran the sparse chains; those are k_sweep3's now, and from 5 % of the markers in the model upwards
  // the third cross term's row fetches cost more than the depth gives -- C4-size BayesC at 5 / 19 / 36 % inclusion: 31.4 / 21.3 /
  // 14.5 iter/s at depth 3 against 30.9 / 19.4 / 9.4 at depth 4; BayesCpi at 51 %: 11.1 against 6.7.)  BWGR_LAG=2|3|4 sets it (A/B tests).
  const int cap = (lv && lv[0] >= '2' && lv[0] <= '4') ? lv[0] - '0' : 3;
  int lag = 2;
  if (P->sweep_version >= 2 && (a.flags & SWF_SELECT)) {
    // the generic sequencer (32-bit Gram entries, fp32 panels) reads a distance-2 row per accepted marker straight from global memory on
    // one wave: two blocks deep unless asked (us per block at n = 10 000, depth 2 / 3: 1.4 % inclusion 4.53 / 4.67, 10.9 % 5.29 / 14.5,
    // BayesCpi at 52 % 12.7 / 58.0); the 16-bit / single-barrier sequencer stages those rows and knows a third cross term as well
    if (P->gramx2 && lv) lag = 3;
    if (!P->is_f32 && P->gramx2 && P->gram16) lag = 3;
    if (!P->is_f32 && P->gramx3 && P->gram16 && P->lag4_ok) lag = 4;
  }
  a.lag = lag < cap ? lag : cap;
  if (use_winv(P, a.flags)) {   // the affine sweeps' product sequencer: as deep as the panel's cross Gram planes reach (BWGR_WLAG caps it)
    a.lag = std::min(P->winv_nd + 1, P->wlag_cap);
    if (!use_wfx(P)) a.lag = std::min(a.lag, 4);   // (k_sweep2's streamers hold four tiles)
#ifdef BWGR_EXPERIMENTS
    if (const char *tl = getenv("BWGR_WLAG_TIMING")) a.lag = atoi(tl);   // TIMING ONLY: deeper than the cross terms reach (wrong chain)
#endif
  }
}
// A handle that a pair run moved onto the pair's stream goes back to the stream it had (its own, or the caller's) when it next sweeps alone:
// one wait, on the handle's stream -- nothing is enqueued on the pair stream, which other pairs' hardware queue shares
static int leave_pair_stream(bwgr_panel *P) {
  if (!P->pre_pair_set) return BWGR_OK;
  hipEvent_t ev;
  HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t he = hipEventRecord(ev, P->stream);
  if (he == hipSuccess) he = hipStreamWaitEvent(P->pre_pair_stream, ev, 0);
  (void)hipEventDestroy(ev);
  if (he != hipSuccess) return fail(BWGR_EHIP, "leaving the pair stream: %s", hipGetErrorString(he));
  P->stream = P->pre_pair_stream; P->pre_pair_set = false;
  return BWGR_OK;
}
// A dry run of the sweep's launch code (nothing is enqueued) gives the compute units its kernels hold; refused with BWGR_EINVAL when they
// cannot be resident beside the sweeps other handles have in flight on this device.  BWGR_OCC_GUARD=0 switches the guard off.
static int sweep_guard(bwgr_panel *P, const SweepArgs &a, int *need) {
  *need = 0;
  if (!guard_on()) return BWGR_OK;
  std::vector<SpinLaunch> plan;
  g_plan = &plan; launch_sweep_kernel(P, a); g_plan = nullptr;
  const int cus = device_cus(P->device);
  if (cus < 1) return BWGR_OK;
  std::lock_guard<std::mutex> lk(g_guard_mu);
  return plan_cus(plan, cus, guard_busy(P, P->stream), need);
}
static int launch_sweep(bwgr_panel *P, SweepArgs &a) {
  CHK(leave_pair_stream(P));
  choose_lag(P, a);
  if (use_winv(P, a.flags)) CHK(winv_alloc(P));
  int need = 0;
  CHK(sweep_guard(P, a, &need));
  CHK(reset_exchange(P));
  P->ps_owner = nullptr;   // the scratch is about to hold this sweep's constants, nobody's iteration
  launch_prestage(P, a);
  launch_sweep_kernel(P, a);
  HIPCHK(hipGetLastError());
  guard_mark(P, P->stream, need);
  return BWGR_OK;
}

static void fill_panel_args(const bwgr_panel *P, SweepArgs &a) {
  a.X = P->X; a.ld = P->ld; a.gram = P->gram;
  a.n = (int)P->n; a.p = (int)P->p; a.m = P->m; a.K = P->K; a.R = P->R;
  a.blk_begin = 0; a.blk_end = (int)P->nblocks;
  a.xpart = P->xpart; a.xflags = P->xflags; a.stamps = P->stamps; a.ps = P->ps;
  a.gramx = P->gramx; a.gramx2 = P->gramx2; a.xspec2 = P->xspec2; a.gramx3 = P->gramx3; a.xspec3 = P->xspec3; a.lag = 2; a.nfeed = P->nfeed; a.gramp = P->gramp; a.pstride = P->pstride; a.qpart = P->qpart; a.dgran = P->dgran;
}

// ------------------------------------------------------------------------------------------------
// panel
// ------------------------------------------------------------------------------------------------
template <typename ST, typename XT>
static int upload(bwgr_panel *P, const void *X, int memloc, int64_t ldx) {
  const int64_t n = P->n, p = P->p;
  XT *dst = reinterpret_cast<XT *>(P->X);
  if (memloc == BWGR_DEVICE) {
    hipLaunchKernelGGL((k_convert<ST, XT>), dim3(4096), dim3(256), 0, P->stream, reinterpret_cast<const ST *>(X), ldx, dst, P->ld, (int)n, (int64_t)0, p, P->R, p);
    HIPCHK(hipGetLastError());
    return BWGR_OK;
  }
  // host source: stage column chunks of <= 256 MiB
  const int64_t col_bytes = ldx * (int64_t)sizeof(ST);
  int64_t cols = std::max<int64_t>(1, ((int64_t)256 << 20) / std::max<int64_t>(1, col_bytes));
  cols = std::min(cols, p);
  ST *stage = nullptr;
  HIPCHK(hipMalloc(&stage, (size_t)(cols * col_bytes)));
  for (int64_t j0 = 0; j0 < p; j0 += cols) {
    const int64_t nc = std::min(cols, p - j0);
    // the last column may be shorter than ldx in the caller's allocation: copy n rows of it separately
    const size_t bytes = (size_t)((nc - 1) * col_bytes + n * (int64_t)sizeof(ST));
    hipError_t e = hipMemcpyAsync(stage, reinterpret_cast<const ST *>(X) + j0 * ldx, bytes, hipMemcpyHostToDevice, P->stream);
    if (e != hipSuccess) { hipFree(stage); return fail(BWGR_EHIP, "upload memcpy failed: %s", hipGetErrorString(e)); }
    hipLaunchKernelGGL((k_convert<ST, XT>), dim3(2048), dim3(256), 0, P->stream, stage, ldx, dst, P->ld, (int)n, j0, nc, P->R, p);
    e = hipStreamSynchronize(P->stream);
    if (e != hipSuccess) { hipFree(stage); return fail(BWGR_EHIP, "upload convert failed: %s", hipGetErrorString(e)); }
  }
  HIPCHK(hipFree(stage));
  return BWGR_OK;
}

extern "C" int bwgr_panel_destroy(bwgr_panel *P) {
  if (!P) return BWGR_OK;
  if (P->nclones > 0) return fail(BWGR_EINVAL, "panel_destroy: %d clone(s) of this panel are still alive", P->nclones);
  if (P->nchains > 0) return fail(BWGR_EINVAL, "panel_destroy: %d chain(s) on this panel are still alive (destroy them first)", P->nchains);
  (void)hipSetDevice(P->device);
  if (!P->parent) {
    for (int d = 0; d < S3_MAXD; ++d) if (P->g3own[d]) hipFree(P->g3x[d]);
    hipFree(P->gx12);
    hipFree(P->xmax_dev);
    for (int d = 0; d < S2W_MAXDIST; ++d) hipFree(P->gxt[d]);
    hipFree(P->X); hipFree(P->gram); hipFree(P->gramx); hipFree(P->gramx2); hipFree(P->gramx3); hipFree(P->gramp16); hipFree(P->gramx16); hipFree(P->gram16_bad); hipFree(P->gramp); hipFree(P->xx); hipFree(P->vx); hipFree(P->msx_dev);
  } else {
    P->parent->nclones--;
  }
  // the scratch a sweep writes: speculative cross terms, pre-staged constants, exchange words
  hipFree(P->qsum3); hipFree(P->lists3); hipFree(P->winv); hipFree(P->qsumw);
  hipFree(P->snap_e); hipFree(P->snap_b); hipFree(P->snap_d); hipFree(P->snap_vb);
  hipFree(P->cpre); if (!P->parent) { hipFree(P->csum); hipFree(P->xxc); }
  hipFree(P->xspec2); hipFree(P->xspec3); hipFree(P->ps.spec); hipFree(P->ps.blocks); hipFree(P->ps.quick); hipFree(P->xpart); hipFree(P->xchg); hipFree(P->stamps);
  guard_forget(P);
  if (P->draws_stream) { (void)hipStreamSynchronize(P->draws_stream); hipStreamDestroy(P->draws_stream); }
  if (P->draws_ready) hipEventDestroy(P->draws_ready);
  if (P->draws_free) hipEventDestroy(P->draws_free);
  hipFree(P->draws);
  if (P->own_stream) hipStreamDestroy(P->own_stream);
  for (hipStream_t q : P->pair_streams) hipStreamDestroy(q);
  delete P;
  return BWGR_OK;
}

static int panel_build_gram(bwgr_panel *P);
static int panel_setup(bwgr_panel *P) {
  const int p = (int)P->p, n = (int)P->n;
  // a10: xx, vx, MSx
  const int wpb = 4;
  if (P->is_f32) hipLaunchKernelGGL(k_stats<float>, dim3((p + wpb - 1) / wpb), dim3(64 * wpb), 0, P->stream, (const float *)P->X, P->R, n, p, P->xx, P->vx);
  else hipLaunchKernelGGL(k_stats<int8_t>, dim3((p + wpb - 1) / wpb), dim3(64 * wpb), 0, P->stream, (const int8_t *)P->X, P->R, n, p, P->xx, P->vx);
  HIPCHK(hipGetLastError());
  {
    const int nparts = 256;
    double *part = nullptr;
    HIPCHK(hipMalloc(&part, sizeof(double) * nparts));
    hipLaunchKernelGGL(k_sum_stage1, dim3(nparts), dim3(256), 0, P->stream, P->vx, (int64_t)p, part);
    hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, P->stream, part, nparts, P->msx_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&P->MSx, P->msx_dev, sizeof(float), hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));
    HIPCHK(hipFree(part));
  }
  if (!P->is_f32) {
    if (!P->xmax_dev) HIPCHK(hipMalloc(&P->xmax_dev, sizeof(int)));
    HIPCHK(hipMemsetAsync(P->xmax_dev, 0, sizeof(int), P->stream));
    hipLaunchKernelGGL(k_absmax_i8, dim3(2048), dim3(256), 0, P->stream, (const int8_t *)P->X, P->x_bytes, P->xmax_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&P->xmax, P->xmax_dev, sizeof(int), hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));
  }
  CHK(panel_build_gram(P));
  // the trajectory engine for the selection models (int8 panels that asked for it; BWGR_SWEEP=2 keeps k_sweep2)
  const char *sv = getenv("BWGR_SWEEP");
  if (P->want3 && !P->is_f32 && P->sweep_version == 2 && !(sv && sv[0] == '2')) {
    P->sweep_version = 3;
    CHK(sweep3_build(P));
  }
  return BWGR_OK;
}

// cross Gram blocks X_{b-dist}' X_b of an int8 panel, b = dist .. nblocks-1, into g[b][m][m] (int32, exact)
static void launch_gramx_i8(bwgr_panel *P, int32_t *g, int dist) {
  const int p = (int)P->p, m = P->m, TJ = m / 16;
  const unsigned nbx = (unsigned)(P->nblocks - dist);
  const size_t lds = (size_t)2 * m * 33 * sizeof(int32_t);
  const int8_t *X = (const int8_t *)P->X;
  if (m == 128) hipLaunchKernelGGL(k_gram_mfma_i8, dim3(nbx), dim3(256), 0, P->stream, X, P->ld, P->R, p, g, dist);
  else switch (TJ) {
    case 1: hipLaunchKernelGGL(k_gramx_i8<1>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    case 2: hipLaunchKernelGGL(k_gramx_i8<2>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    case 3: hipLaunchKernelGGL(k_gramx_i8<3>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    case 4: hipLaunchKernelGGL(k_gramx_i8<4>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    case 5: hipLaunchKernelGGL(k_gramx_i8<5>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    case 6: hipLaunchKernelGGL(k_gramx_i8<6>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    case 7: hipLaunchKernelGGL(k_gramx_i8<7>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
    default: hipLaunchKernelGGL(k_gramx_i8<8>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
  }
}

// diagonal, off-diagonal and packed Gram blocks of the resident X
static int panel_build_gram(bwgr_panel *P) {
  const int p = (int)P->p;
  const int m = P->m, TJ = m / 16;
  if (P->is_f32) {
    const size_t lds = (size_t)m * 65 * sizeof(float);
    double *g = (double *)P->gram; const float *X = (const float *)P->X;
    switch (TJ) {
      case 1: hipLaunchKernelGGL(k_gram_f32<1>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 2: hipLaunchKernelGGL(k_gram_f32<2>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 3: hipLaunchKernelGGL(k_gram_f32<3>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      default: hipLaunchKernelGGL(k_gram_f32<4>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
    }
  } else {
    const size_t lds = (size_t)m * 33 * sizeof(int32_t);
    int32_t *g = (int32_t *)P->gram; const int8_t *X = (const int8_t *)P->X;
    if (m == 128) hipLaunchKernelGGL(k_gram_mfma_i8, dim3(P->nblocks), dim3(256), 0, P->stream, X, P->ld, P->R, p, g, 0);
    else switch (TJ) {
      case 1: hipLaunchKernelGGL(k_gram_i8<1>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 2: hipLaunchKernelGGL(k_gram_i8<2>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 3: hipLaunchKernelGGL(k_gram_i8<3>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 4: hipLaunchKernelGGL(k_gram_i8<4>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 5: hipLaunchKernelGGL(k_gram_i8<5>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 6: hipLaunchKernelGGL(k_gram_i8<6>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      case 7: hipLaunchKernelGGL(k_gram_i8<7>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
      default: hipLaunchKernelGGL(k_gram_i8<8>, dim3(P->nblocks), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g); break;
    }
  }
  HIPCHK(hipGetLastError());
  for (int dist = 1; dist <= 3; ++dist) {   // off-diagonal blocks (blk-dist, blk): the cross terms of the lag-2 / 3 / 4 pipelines
    if (P->nblocks <= dist || dist > P->gram_maxdist || (dist == 2 && !P->gramx2) || (dist == 3 && !P->gramx3)) continue;
    const unsigned nbx = (unsigned)(P->nblocks - dist);
    if (P->is_f32) {
      const size_t lds = (size_t)2 * m * 65 * sizeof(float);
      double *g = (double *)(dist == 1 ? P->gramx : dist == 2 ? P->gramx2 : P->gramx3); const float *X = (const float *)P->X;
      switch (TJ) {
        case 1: hipLaunchKernelGGL(k_gramx_f32<1>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
        case 2: hipLaunchKernelGGL(k_gramx_f32<2>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
        case 3: hipLaunchKernelGGL(k_gramx_f32<3>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
        default: hipLaunchKernelGGL(k_gramx_f32<4>, dim3(nbx), dim3(256), lds, P->stream, X, P->ld, P->R, p, m, g, dist); break;
      }
    } else {
      launch_gramx_i8(P, (int32_t *)(dist == 1 ? P->gramx : dist == 2 ? P->gramx2 : P->gramx3), dist);
    }
    HIPCHK(hipGetLastError());
  }
  if (P->is_f32) hipLaunchKernelGGL(k_gram_pack<double>, dim3((unsigned)P->nblocks), dim3(256), 0, P->stream, (const double *)P->gram, (double *)P->gramp, m, P->pstride, P->nblocks);
  else hipLaunchKernelGGL(k_gram_pack<int32_t>, dim3((unsigned)P->nblocks), dim3(256), 0, P->stream, (const int32_t *)P->gram, (int32_t *)P->gramp, m, P->pstride, P->nblocks);
  HIPCHK(hipGetLastError());
  P->gram16 = false;
  if (P->gramp16) {
    HIPCHK(hipMemsetAsync(P->gram16_bad, 0, sizeof(int), P->stream));
    hipLaunchKernelGGL(k_gram_narrow, dim3(2048), dim3(256), 0, P->stream, (const int32_t *)P->gramp, P->gramp16, (int64_t)P->nblocks * P->pstride, P->gram16_bad);
    if (P->nblocks > 1)
      hipLaunchKernelGGL(k_gram_narrow, dim3(2048), dim3(256), 0, P->stream, (const int32_t *)P->gramx + (size_t)m * m, P->gramx16 + (size_t)m * m, (int64_t)(P->nblocks - 1) * m * m, P->gram16_bad);
    HIPCHK(hipGetLastError());
    int bad = 1;
    HIPCHK(hipMemcpyAsync(&bad, P->gram16_bad, sizeof(int), hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));
    const char *gv = getenv("BWGR_GRAM16");   // BWGR_GRAM16=0 forces the 32-bit staging (A/B tests)
    P->gram16 = (bad == 0) && !(gv && gv[0] == '0');
  }
  // the affine sweeps' sequencer (sweep2w.hip.h) takes the cross blocks as biased byte planes: built where every entry fits 16 bits
  P->winv_nd = 0;
  if (!P->is_f32 && P->gram16 && P->winv_on && m <= SW_MAXM) {
    HIPCHK(hipMemsetAsync(P->gram16_bad, 0, sizeof(int), P->stream));
    int nd = 0;
    int32_t *tmpx = nullptr;   // distances 4 and 5 (pipelines five and six blocks deep; main panels only): built here, kept as planes only
    for (int dist = 1; dist <= S2W_MAXDIST; ++dist) {
      const int32_t *src = (const int32_t *)(dist == 1 ? P->gramx : dist == 2 ? P->gramx2 : dist == 3 ? P->gramx3 : nullptr);
      if (dist > S2W_NEARD) {
        if (!P->want3 || P->gram_maxdist < S2W_NEARD || P->nblocks <= dist || dist > P->wlag_cap - 1) break;
        if (!tmpx && hipMalloc(&tmpx, (size_t)P->nblocks * m * m * 4) != hipSuccess) { (void)hipGetLastError(); tmpx = nullptr; break; }
        launch_gramx_i8(P, tmpx, dist);
        src = tmpx;
      }
      if (P->nblocks <= dist || (dist <= S2W_NEARD && dist > P->gram_maxdist) || !src) break;
      if (!P->gxt[dist - 1]) HIPCHK(hipMalloc(&P->gxt[dist - 1], (size_t)P->nblocks * S2W_PBYTES));
      hipLaunchKernelGGL(k_gx_planes, dim3(4096), dim3(256), 0, P->stream, src, P->gxt[dist - 1], m, (int64_t)P->nblocks, dist, P->gram16_bad);
      HIPCHK(hipGetLastError());
      nd = dist;
    }
    int bad = 1;
    HIPCHK(hipMemcpyAsync(&bad, P->gram16_bad, sizeof(int), hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));
    if (tmpx) hipFree(tmpx);
    P->winv_nd = bad ? 0 : nd;
  }
  HIPCHK(hipStreamSynchronize(P->stream));
  return BWGR_OK;
}

// geometry + every device allocation of a panel of n rows x p markers (no data yet)
static int panel_alloc(bwgr_panel **out, int is_f32, int64_t n, int64_t p, int device, int block, int nwg) {
  *out = nullptr;
  if (n < 2 || p < 1) return fail(BWGR_EINVAL, "panel: need n >= 2, p >= 1 (n=%lld p=%lld)", (long long)n, (long long)p);
  if (n > 0x7FFFFF00ll || p > 0x7FFFFF00ll) return fail(BWGR_EINVAL, "panel: n and p must fit 31 bits");
  CHK(require_device(device));
  bwgr_panel *P = new bwgr_panel();
  P->device = device; P->n = n; P->p = p; P->is_f32 = is_f32;
  const int mmax = P->is_f32 ? 64 : SW_MAXM;
  int m = block > 0 ? block : mmax;
  if (m > mmax) { delete P; return fail(BWGR_EINVAL, "panel_create: block %d > %d (limit for this genotype type)", m, mmax); }
  m = (int)std::min<int64_t>(m, ((p + 15) / 16) * 16);
  m = ((m + 15) / 16) * 16;
  P->m = m;
  const int Rmax = P->is_f32 ? max_slab_rows<float>(m) : max_slab_rows<int8_t>(m);
  // the pipelined engine keeps three tiles per streamer, so it takes fewer rows per slab than k_sweep at small blocks:
  // prefer the largest slab it fits (unless that needs more workgroups than the chip has CUs, or k_sweep is forced)
  int Rpick = Rmax;
  {
    const char *sv = getenv("BWGR_SWEEP");
    int R2 = 0;
    for (int Rt = 128; Rt <= Rmax; Rt += 128)
      if ((P->is_f32 ? sweep2_lds_bytes<float>(m, Rt) : sweep2_lds_bytes<int8_t>(m, Rt)) <= (size_t)160 * 1024 &&
          (P->is_f32 || (size_t)m * Rt <= S2I_TILE_BYTES_MAX)) R2 = Rt;   // (an int8 tile must fit its movers' registers)
    if (!(sv && sv[0] == '1') && R2 > 0 && (n + R2 - 1) / R2 + 1 + 6 <= 256) Rpick = R2;
  }
  int K = nwg > 0 ? nwg : (int)((n + Rpick - 1) / Rpick);
  int R = (int)((((n + K - 1) / K) + 127) / 128) * 128;
  if (K > 256 || R > Rmax) {
    delete P;
    return fail(BWGR_EINVAL, "panel_create: n=%lld needs %d slab workgroups of %d rows (limits: 256 workgroups, %d rows)", (long long)n, K, R, Rmax);
  }
  P->K = K; P->R = R; P->ld = (int64_t)K * R;
  P->nblocks = (p + m - 1) / m;
  if (P->nblocks >= (1ll << 24)) { delete P; return fail(BWGR_EINVAL, "panel_create: %lld marker blocks; the delta granules carry a 24-bit block epoch", (long long)P->nblocks); }
  P->lds_bytes = P->is_f32 ? sweep_lds_bytes<float>(m, R) : sweep_lds_bytes<int8_t>(m, R);
  P->lds2_bytes = P->is_f32 ? sweep2_lds_bytes<float>(m, R) : sweep2_lds_bytes<int8_t>(m, R);
  if (!P->is_f32 && s2i_lds_bytes(m, R, 4) <= (size_t)160 * 1024) {
    P->lag4_ok = true;
    P->lds2_bytes = std::max(P->lds2_bytes, s2i_lds_bytes(m, R, 4));
  }
  {
    const char *sv = getenv("BWGR_SWEEP");   // A/B switch for tests and profiling
    P->sweep_version = (sv && sv[0] == '1') ? 1 : 2;
    P->nfeed = std::min(6, std::max(2, (K + 39) / 40 + 1));   // K = 40: 2, K = 79: 3, K >= 161: 6
    if (const char *nf = getenv("BWGR_NFEED")) { const int v = atoi(nf); if (v >= 1 && v <= 6) P->nfeed = v; }   // experiments
    if (P->lds2_bytes > (size_t)160 * 1024 || K + 1 + P->nfeed > 256) P->sweep_version = 1;
    if (!P->is_f32 && (size_t)m * R > S2I_TILE_BYTES_MAX) P->sweep_version = 1;
  }
  P->x_bytes = (size_t)P->ld * (size_t)p * (P->is_f32 ? 4 : 1);
  P->gram_bytes = (size_t)P->nblocks * m * m * (P->is_f32 ? 8 : 4);   // per Gram array (diagonal blocks; off-diagonal blocks)
  int rc = BWGR_OK;
  auto bail = [&](int code) { bwgr_panel_destroy(P); return code; };
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bail(fail(BWGR_EHIP, "%s failed: %s", #x, hipGetErrorString(e_))); } while (0)
  PCHK(hipMalloc(&P->X, P->x_bytes));
  PCHK(hipMalloc(&P->gram, P->gram_bytes));
  PCHK(hipMalloc(&P->gramx, P->gram_bytes));
  if (P->sweep_version >= 2 && P->nblocks > 2) {   // distance-2 blocks: the selection models' lag-3 pipeline
    PCHK(hipMalloc(&P->gramx2, P->gram_bytes));
    PCHK(hipMalloc(&P->xspec2, sizeof(double) * (size_t)P->nblocks * SW_MAXM));
  }
  const char *lagenv = getenv("BWGR_LAG");
  if (P->sweep_version >= 2 && !P->is_f32 && P->nblocks > 3 && P->lag4_ok && !(lagenv && (lagenv[0] == '2' || lagenv[0] == '3'))) {   // distance-3 blocks: the lag-4 pipeline
    PCHK(hipMalloc(&P->gramx3, P->gram_bytes));
    PCHK(hipMalloc(&P->xspec3, sizeof(double) * (size_t)P->nblocks * SW_MAXM));
  }
  P->pstride = ((m * (m - 1) / 2 + 7) / 8) * 8;
  PCHK(hipMalloc(&P->gramp, (size_t)P->nblocks * std::max(P->pstride, 8) * (P->is_f32 ? 8 : 4)));
  if (!P->is_f32 && P->sweep_version >= 2) {
    PCHK(hipMalloc(&P->gramp16, (size_t)P->nblocks * std::max(P->pstride, 8) * 2));
    PCHK(hipMalloc(&P->gramx16, (size_t)P->nblocks * m * m * 2));
    PCHK(hipMalloc(&P->gram16_bad, sizeof(int)));
  }
  PCHK(hipMalloc(&P->ps.spec, sizeof(SpecBuf) * (size_t)P->nblocks));
  if (P->sweep_version >= 2) PCHK(hipMalloc(&P->ps.quick, sizeof(QuickBuf) * (size_t)P->nblocks));
  PCHK(hipMalloc(&P->xx, sizeof(float) * p));
  PCHK(hipMalloc(&P->vx, sizeof(float) * p));
  PCHK(hipMalloc(&P->msx_dev, sizeof(float)));
  PCHK(hipMalloc(&P->xpart, sizeof(double) * 2 * (size_t)K * SW_MAXM));
  PCHK(alloc_exchange(P));
  PCHK(hipMalloc(&P->ps.blocks, sizeof(StageBuf) * (size_t)P->nblocks));
#if defined(BWGR_STAMPS) || defined(BWGR_EXPERIMENTS)
  PCHK(hipMalloc(&P->stamps, sizeof(unsigned long long) * 256));
  PCHK(hipMemset(P->stamps, 0, sizeof(unsigned long long) * 256));
#endif
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<int8_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<int8_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2<int8_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2<int8_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2<int8_t, true, uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2w<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep2w<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  PCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_affine_inv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef BWGR_EXPERIMENTS
  if (!P->is_f32) P->ldsw_bytes = s2w_lds_bytes(m, R, getenv("BWGR_WLAG_TIMING") ? atoi(getenv("BWGR_WLAG_TIMING")) : 6);
#else
  if (!P->is_f32) P->ldsw_bytes = s2w_lds_bytes(m, R, 6);
#endif   // (room for the deepest pipeline BWGR_WLAG can ask for)
  if (const char *wv = getenv("BWGR_WINV")) P->winv_on = !(wv[0] == '0');
  if (const char *wv = getenv("BWGR_WFX")) P->wfx_on = !(wv[0] == '0');
  if (const char *pv = getenv("BWGR_WPF")) P->wpf = std::max(0, std::min(8, atoi(pv)));
  if (const char *pv = getenv("BWGR_WAHEAD")) P->wahead = std::max(1, atoi(pv));
  if (const char *qv = getenv("BWGR_WNQ")) { const int v = atoi(qv); if (v == 1 || v == 2 || v == 4) P->wnq = v; }
  if (const char *wl = getenv("BWGR_WLAG")) if (wl[0] >= '2' && wl[0] <= '6') P->wlag_cap = wl[0] - '0';
#undef PCHK
  (void)rc;
  *out = P;
  return BWGR_OK;
}

extern "C" int bwgr_panel_create(bwgr_panel **out, const void *X, int xtype, int memloc, int64_t n, int64_t p,
                                 int64_t ldx, int device, int block, int nwg) {
  if (!out || !X) return fail(BWGR_EINVAL, "panel_create: null pointer");
  *out = nullptr;
  if (ldx < n) return fail(BWGR_EINVAL, "panel_create: need ldx >= n (n=%lld ldx=%lld)", (long long)n, (long long)ldx);
  if (xtype != BWGR_X_I8 && xtype != BWGR_X_F32 && xtype != BWGR_X_F64) return fail(BWGR_EINVAL, "panel_create: bad xtype %d", xtype);
  if (memloc != BWGR_HOST && memloc != BWGR_DEVICE) return fail(BWGR_EINVAL, "panel_create: bad memloc %d", memloc);
  bwgr_panel *P = nullptr;
  CHK(panel_alloc(&P, xtype != BWGR_X_I8, n, p, device, block, nwg));
  P->want3 = true;
  int rc;
  if (xtype == BWGR_X_I8) rc = upload<int8_t, int8_t>(P, X, memloc, ldx);
  else if (xtype == BWGR_X_F32) rc = upload<float, float>(P, X, memloc, ldx);
  else rc = upload<double, float>(P, X, memloc, ldx);
  if (rc == BWGR_OK) rc = panel_setup(P);
  if (rc != BWGR_OK) { bwgr_panel_destroy(P); return rc; }
  *out = P;
  return BWGR_OK;
}

#if defined(BWGR_STAMPS) || defined(BWGR_EXPERIMENTS)
// diagnostic builds only: cumulative per-phase s_memtime ticks of workgroup 0, event counters (not part of include/bwgr.h)
extern "C" int bwgr_debug_stamps(bwgr_panel *P, unsigned long long out[256]) {
  HIPCHK(d2h(P->stream, out, P->stamps, sizeof(unsigned long long) * 256));
  HIPCHK(hipMemset(P->stamps, 0, sizeof(unsigned long long) * 256));
  return BWGR_OK;
}
#endif

// A clone: the same genotypes and Gram arrays (shared, read-only during sweeps), its own sweep scratch and its own
// stream, so that chains on the parent and on its clones run concurrently on disjoint CUs.
extern "C" int bwgr_panel_clone(bwgr_panel **out, bwgr_panel *src) {
  if (!out || !src) return fail(BWGR_EINVAL, "panel_clone: null pointer");
  *out = nullptr;
  bwgr_panel *root = src->parent ? src->parent : src;
  HIPCHK(hipSetDevice(root->device));
  HIPCHK(hipStreamSynchronize(root->stream));   // the shared arrays are complete
  bwgr_panel *P = new bwgr_panel(*root);
  P->parent = root; P->nclones = 0; P->nchains = 0; P->own_stream = nullptr; P->stream = nullptr; P->ps_owner = nullptr; P->ps_iter = -1;
  P->pair_streams.clear();   // (the root's: a clone owns none)
  P->draws = nullptr; P->draws_stream = nullptr; P->draws_ready = nullptr; P->draws_free = nullptr; P->draws_valid = false;   // (its own, made on first use)
  P->pre_pair_stream = nullptr; P->pre_pair_set = false; P->guard_ev = nullptr; P->guard_cus = 0; P->guard_stream = nullptr; P->guard_listed = false;
  P->qsum3 = P->lists3 = nullptr; P->epoch3 = 0;
  P->snap_e = nullptr; P->snap_b = P->snap_d = P->snap_vb = nullptr; P->xmax_dev = nullptr; P->winv = nullptr; P->qsumw = nullptr;
  P->xspec2 = P->xspec3 = nullptr; P->ps = {}; P->xpart = P->qpart = nullptr; P->dgran = nullptr; P->xflags = nullptr; P->xchg = nullptr; P->stamps = nullptr;
  root->nclones++;
  auto bail = [&](int code) { bwgr_panel_destroy(P); return code; };
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bail(fail(BWGR_EHIP, "%s failed: %s", #x, hipGetErrorString(e_))); } while (0)
  const int K = P->K;
  if (root->xspec2) PCHK(hipMalloc(&P->xspec2, sizeof(double) * (size_t)P->nblocks * SW_MAXM));
  if (root->xspec3) PCHK(hipMalloc(&P->xspec3, sizeof(double) * (size_t)P->nblocks * SW_MAXM));
  PCHK(hipMalloc(&P->ps.spec, sizeof(SpecBuf) * (size_t)P->nblocks));
  if (P->sweep_version >= 2) PCHK(hipMalloc(&P->ps.quick, sizeof(QuickBuf) * (size_t)P->nblocks));
  PCHK(hipMalloc(&P->ps.blocks, sizeof(StageBuf) * (size_t)P->nblocks));
  PCHK(hipMalloc(&P->xpart, sizeof(double) * 2 * (size_t)K * SW_MAXM));
  PCHK(alloc_exchange(P));
#if defined(BWGR_STAMPS) || defined(BWGR_EXPERIMENTS)
  PCHK(hipMalloc(&P->stamps, sizeof(unsigned long long) * 256));
  PCHK(hipMemset(P->stamps, 0, sizeof(unsigned long long) * 256));
#endif
  PCHK(hipStreamCreateWithFlags(&P->own_stream, hipStreamNonBlocking));
#undef PCHK
  P->stream = P->own_stream;
  if (P->e3_ready) { int rc3 = sweep3_alloc_scratch(P); if (rc3 != BWGR_OK) return bail(rc3); HIPCHK(hipStreamSynchronize(P->stream)); }
  *out = P;
  return BWGR_OK;
}

// chains (one per panel or clone) whose sweep kernels fit the chip side by side: each takes nwg + 1 (+ feeders) CUs
extern "C" int bwgr_panel_max_concurrent(const bwgr_panel *P, int selection, int *count) {
  if (!P || !count) return fail(BWGR_EINVAL, "null pointer");
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, P->device));
  // selection on a panel with k_sweep3: the device sends a chain above the engine threshold to k_sweep2 (K + 1 + feeders), and a sweep that
  // leaves the fixed-point range is redone there: the larger of the two
  const int wgs2 = P->K + 1 + ((P->sweep_version >= 2 && selection) ? P->nfeed : 0);
  int wgs = (selection && P->sweep_version == 3 && P->e3_ready) ? std::max(P->K3 + 1, wgs2) : wgs2;
  if (!selection && use_winv(P, 0)) wgs = (use_wfx(P) ? P->K * (P->R / S2W_FXR) : P->K) + 1 + P->wpf;   // streamers, sequencer, L2 prefetchers (the launch's other workgroups leave at once)
  // one sweep workgroup per CU even where the LDS would admit two (small blocks): measured, sharing a CU costs more than it adds
  *count = std::max(1, prop.multiProcessorCount / wgs);
  if (const char *ov = getenv("BWGR_MAX_CONCURRENT")) { const int v = atoi(ov); if (v > 0) *count = v; }   // experiments
  return BWGR_OK;
}

// pairs of chains (bwgr_chain_run_pair) that fit side by side: a pair's launch holds K3 + 2 compute units, and about 40 stay free for the
// iterations' small kernels (six pairs at C4 measured slower than five); 0 on a panel without k_sweep3.  BWGR_MAX_PAIRS overrides.
extern "C" int bwgr_panel_max_pairs(const bwgr_panel *P, int *pairs) {
  if (!P || !pairs) return fail(BWGR_EINVAL, "null pointer");
  *pairs = 0;
  if (!(P->sweep_version == 3 && P->e3_ready) || s3p_streamer_lds(P->R3) > (size_t)160 * 1024) return BWGR_OK;
  const int cus = device_cus(P->device);
  if (cus < 1) return fail(BWGR_EHIP, "panel_max_pairs: no device properties");
  *pairs = std::max(1, (cus - 40) / (P->K3 + 2));
  if (const char *ov = getenv("BWGR_MAX_PAIRS")) { const int v = atoi(ov); if (v > 0) *pairs = v; }
  return BWGR_OK;
}

// Test hook for the abort path: while on != 0 every sweep launched on this panel runs with slab workgroup 0 absent, so the
// workgroups that wait for it spin to their wall-clock bound, raise the shared abort word and the launch reports BWGR_ETIMEOUT.
extern "C" int bwgr_debug_withhold(bwgr_panel *P, int on) {
  if (!P) return fail(BWGR_EINVAL, "null panel");
  P->debug_withhold = on ? 1 : 0;
  return BWGR_OK;
}

extern "C" int bwgr_panel_set_stream(bwgr_panel *P, void *hip_stream) {
  if (!P) return fail(BWGR_EINVAL, "null panel");
  P->stream = reinterpret_cast<hipStream_t>(hip_stream);
  P->pre_pair_set = false;   // (the caller's choice stands: no return to an earlier stream)
  return BWGR_OK;
}

extern "C" int bwgr_panel_info(const bwgr_panel *P, int64_t info[8]) {
  if (!P || !info) return fail(BWGR_EINVAL, "null pointer");
  info[0] = P->n; info[1] = P->p; info[2] = P->ld; info[3] = P->m; info[4] = P->K; info[5] = P->R;
  info[6] = (int64_t)P->x_bytes; info[7] = (int64_t)(2 * P->gram_bytes);
  return BWGR_OK;
}

extern "C" int bwgr_panel_pipeline(const bwgr_panel *P, int selection, int info[4]) {
  if (!P || !info) return fail(BWGR_EINVAL, "null pointer");
  SweepArgs a{};
  a.flags = selection ? SWF_SELECT : 0u;
  choose_lag(P, a);
  const bool s3 = selection && P->sweep_version == 3 && P->e3_ready;
  const bool w4 = !selection && use_winv(P, 0);
  info[0] = s3 ? 3 : w4 ? 4 : std::min(P->sweep_version, 2);
  info[1] = s3 ? P->e3_D : a.lag;
  info[2] = s3 ? 0 : ((P->sweep_version >= 2 && selection) ? P->nfeed : 0);
  info[3] = P->is_f32 ? 0 : (P->gram16 ? 16 : 32);
  return BWGR_OK;
}

extern "C" int bwgr_panel_stats(bwgr_panel *P, float *xx, float *vx, float *MSx) {
  if (!P) return fail(BWGR_EINVAL, "null panel");
  HIPCHK(hipSetDevice(P->device));
  if (xx) HIPCHK(d2h(P->stream, xx, panel_cen(P) ? (P->parent ? P->parent : P)->xxc : P->xx, sizeof(float) * P->p));   // (centred panel: the centred columns' norms)
  if (vx) HIPCHK(d2h(P->stream, vx, P->vx, sizeof(float) * P->p));
  if (MSx) *MSx = P->MSx;
  return BWGR_OK;
}

// ------------------------------------------------------------------------------------------------
// KMUP
// ------------------------------------------------------------------------------------------------
// device buffers of one call, released on every exit path
namespace {
struct DevBufs {
  std::vector<void *> v;
  ~DevBufs() { for (void *q : v) hipFree(q); }
  template <typename T> T *get(size_t count) {
    void *q = nullptr;
    if (hipMalloc(&q, sizeof(T) * (count ? count : 1)) != hipSuccess) return nullptr;
    v.push_back(q);
    return reinterpret_cast<T *>(q);
  }
};
}  // namespace

// row gather of the resident panel P into the subsample panel PB (rows use_d[0..nbag), device array); KMUP2's H = X(Use, j)
static void launch_gather_rows(bwgr_panel *P, bwgr_panel *PB, const int *use_d, int64_t nbag) {
  if (P->is_f32) hipLaunchKernelGGL(k_gather_rows<float>, dim3(4096), dim3(256), 0, P->stream, (const float *)P->X, P->R, use_d, (int)nbag, (float *)PB->X, PB->R, PB->ld, P->p);
  else if (P->ld <= 64 * 1024) {   // a column fits the LDS: stage, pick, write in 16-byte pieces
    const int mpw = (int)std::max<int64_t>(1, std::min<int64_t>(8, (32 * 1024) / P->ld));   // ~30 KB of LDS per workgroup: five of them per CU
    hipLaunchKernelGGL(k_gather_rows_i8, dim3((unsigned)((P->p + mpw - 1) / mpw)), dim3(256), (size_t)mpw * P->ld, P->stream, (const int8_t *)P->X, P->R, P->ld,
                       use_d, (int)nbag, (int8_t *)PB->X, PB->R, PB->ld, P->p, mpw);
  } else hipLaunchKernelGGL(k_gather_rows<int8_t>, dim3(4096), dim3(256), 0, P->stream, (const int8_t *)P->X, P->R, use_d, (int)nbag, (int8_t *)PB->X, PB->R, PB->ld, P->p);
}

// one sweep over panel PS with host-side b, d, xx, L and a device residual e64 (ld doubles, padding zero); KMUP and KMUP2
static int kmup_sweep(bwgr_panel *PS, float *b, float *d, const float *xx, const float *L, double *e64, float Ve, float pi, float bg,
                      int kmup2, uint64_t seed, uint32_t iter, int rng_mode, const char *who) {
  DevBufs bufs;
  const size_t p = (size_t)PS->p, pb = sizeof(float) * p;
  float *db = bufs.get<float>(p), *dd = bufs.get<float>(p), *dxx = bufs.get<float>(p), *dL = bufs.get<float>(p), *dvb = bufs.get<float>(p);
  ChainScalars *sc = bufs.get<ChainScalars>(1);
  if (!db || !dd || !dxx || !dL || !dvb || !sc) return fail(BWGR_ENOMEM, "%s: device allocation failed", who);
  HIPCHK(hipMemcpyAsync(db, b, pb, hipMemcpyHostToDevice, PS->stream));
  HIPCHK(hipMemcpyAsync(dd, d, pb, hipMemcpyHostToDevice, PS->stream));
  HIPCHK(hipMemcpyAsync(dxx, xx, pb, hipMemcpyHostToDevice, PS->stream));
  HIPCHK(hipMemcpyAsync(dL, L, pb, hipMemcpyHostToDevice, PS->stream));
  ChainScalars h; memset(&h, 0, sizeof(h));
  h.ve = Ve; h.pi = pi; h.C = -0.5f / sqrtf(Ve); h.odds = pi / (1.0f - pi); h.dfp1 = 1.0f; h.bg = bg;
  h.inc_rate = 1.0f - pi;
  HIPCHK(hipMemcpyAsync(sc, &h, sizeof(h), hipMemcpyHostToDevice, PS->stream));
  SweepArgs a; memset(&a, 0, sizeof(a));
  fill_panel_args(PS, a);
  a.flags = SWF_LAM_VEC | (pi > 0 ? (SWF_SELECT | SWF_ALT_B2) : 0) | (kmup2 ? SWF_KMUP2 : 0);
  a.e = e64; a.b = db; a.d = dd; a.vb = dvb; a.xx = dxx; a.lam = dL; a.sc = sc;
  a.iter = iter; a.rng = make_rng(seed, rng_mode);
  CHK(launch_sweep(PS, a));
  HIPCHK(hipMemcpyAsync(b, db, pb, hipMemcpyDeviceToHost, PS->stream));
  HIPCHK(hipMemcpyAsync(d, dd, pb, hipMemcpyDeviceToHost, PS->stream));
  HIPCHK(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, PS->stream));
  HIPCHK(hipStreamSynchronize(PS->stream));
  if (h.error) return sweep_error(h.error, who);
  return BWGR_OK;
}

extern "C" int bwgr_kmup(bwgr_panel *P, float *b, float *d, const float *xx, float *e, const float *L, float Ve,
                         float pi, uint64_t seed, uint32_t iter, int rng_mode) {
  if (P && panel_cen(P)) return fail(BWGR_EINVAL, "kmup: this panel sweeps implicitly centred columns (bwgr_panel_set_centred), which only the fused chains do; call bwgr_panel_set_centred(P, 0) first");
  if (!P || !b || !d || !xx || !e || !L) return fail(BWGR_EINVAL, "kmup: null pointer");
  HIPCHK(hipSetDevice(P->device));
  DevBufs bufs;
  float *de = bufs.get<float>((size_t)P->n);
  double *de64 = bufs.get<double>((size_t)P->ld);
  if (!de || !de64) return fail(BWGR_ENOMEM, "kmup: device allocation failed");
  HIPCHK(hipMemcpyAsync(de, e, sizeof(float) * P->n, hipMemcpyHostToDevice, P->stream));
  hipLaunchKernelGGL(k_f2d, dim3(64), dim3(256), 0, P->stream, de, de64, P->n, P->ld);
  CHK(kmup_sweep(P, b, d, xx, L, de64, Ve, pi, 0.0f, 0, seed, iter, rng_mode, "kmup"));
  hipLaunchKernelGGL(k_d2f, dim3(64), dim3(256), 0, P->stream, de64, de, P->n);
  HIPCHK(hipGetLastError());
  HIPCHK(d2h(P->stream, e, de, sizeof(float) * P->n));
  return BWGR_OK;
}

// KMUP2(X,Use,b,d,xx,E,L,Ve,pi), src/Rcpp20260726ai.cpp:41-77: the sweep on the row subsample Use (0-based, nuse entries, in
// the caller's order, repeats allowed) of the resident panel.  The rows are gathered into a subsample panel on the device, its
// Gram blocks built, and the sweep runs with KMUP2's conditional mean (numerator + b0, denominator xx*bg + L, bg = n0/nuse).
// e_out receives the nuse residuals of the subsample (:76); E (n0 entries) is not modified.
extern "C" int bwgr_kmup2(bwgr_panel *P, const int *Use, int64_t nuse, float *b, float *d, const float *xx, const float *E,
                          float *e_out, const float *L, float Ve, float pi, uint64_t seed, uint32_t iter, int rng_mode) {
  if (P && panel_cen(P)) return fail(BWGR_EINVAL, "kmup2: this panel sweeps implicitly centred columns (bwgr_panel_set_centred), which only the fused chains do; call bwgr_panel_set_centred(P, 0) first");
  if (!P || !Use || !b || !d || !xx || !E || !e_out || !L) return fail(BWGR_EINVAL, "kmup2: null pointer");
  if (nuse < 2 || nuse > 0x7FFFFF00ll) return fail(BWGR_EINVAL, "kmup2: need 2 <= length(Use) < 2^31 (got %lld)", (long long)nuse);
  for (int64_t k = 0; k < nuse; ++k)
    if (Use[k] < 0 || Use[k] >= P->n) return fail(BWGR_EINVAL, "kmup2: Use[%lld] = %d is outside 0..%lld", (long long)k, Use[k], (long long)P->n - 1);
  HIPCHK(hipSetDevice(P->device));
  bwgr_panel *PB = nullptr;
  CHK(panel_alloc(&PB, P->is_f32, nuse, P->p, P->device, P->m, 0));
  PB->stream = P->stream;
  struct Drop { bwgr_panel *q; ~Drop() { if (q) bwgr_panel_destroy(q); } } drop{PB};
  DevBufs bufs;
  int *use_d = bufs.get<int>((size_t)nuse);
  float *dE = bufs.get<float>((size_t)P->n), *deo = bufs.get<float>((size_t)nuse);
  double *dE64 = bufs.get<double>((size_t)P->n), *e64 = bufs.get<double>((size_t)PB->ld);
  if (!use_d || !dE || !deo || !dE64 || !e64) return fail(BWGR_ENOMEM, "kmup2: device allocation failed");
  HIPCHK(hipMemcpyAsync(use_d, Use, sizeof(int) * (size_t)nuse, hipMemcpyHostToDevice, P->stream));
  HIPCHK(hipMemcpyAsync(dE, E, sizeof(float) * P->n, hipMemcpyHostToDevice, P->stream));
  launch_gather_rows(P, PB, use_d, nuse);
  HIPCHK(hipGetLastError());
  CHK(panel_build_gram(PB));
  hipLaunchKernelGGL(k_f2d, dim3(64), dim3(256), 0, P->stream, dE, dE64, P->n, P->n);
  hipLaunchKernelGGL(k_gather_e, dim3(64), dim3(256), 0, P->stream, dE64, use_d, (int)nuse, PB->ld, e64);   // e0[k] = E[Use[k]], :49-53
  HIPCHK(hipGetLastError());
  CHK(kmup_sweep(PB, b, d, xx, L, e64, Ve, pi, (float)P->n / (float)nuse, 1, seed, iter, rng_mode, "kmup2"));
  hipLaunchKernelGGL(k_d2f, dim3(64), dim3(256), 0, P->stream, e64, deo, nuse);
  HIPCHK(hipGetLastError());
  HIPCHK(d2h(P->stream, e_out, deo, sizeof(float) * (size_t)nuse));
  return BWGR_OK;
}

// ------------------------------------------------------------------------------------------------
// fused chains
// ------------------------------------------------------------------------------------------------
static bool per_marker_vb(int model) { return model == BWGR_BAYESA || model == BWGR_BAYESB || model == BWGR_BAYESL || model == BWGR_BAYESDPI; }
static bool has_d(int model) { return model == BWGR_BAYESB || model == BWGR_BAYESC || model == BWGR_BAYESCPI || model == BWGR_BAYESDPI; }

extern "C" int bwgr_chain_destroy(bwgr_chain *C) {
  if (!C) return BWGR_OK;
  if (C->P && C->P->ps_owner == C) C->P->ps_owner = nullptr;   // (a later chain may be allocated at this address)
  if (C->P && C->P->draws_sc == (const void *)C->sc) {          // ... and so may its scalars: variates drawn ahead for this chain are nobody's now
    if (C->P->draws_stream) (void)hipStreamSynchronize(C->P->draws_stream);   // (k_draws reads the chain's df from them)
    C->P->draws_valid = false; C->P->draws_sc = nullptr;
  }
  if (C->P) { C->P->nchains--; (void)hipSetDevice(C->P->device); }
  for (hipEvent_t ev : C->ev) hipEventDestroy(ev);
  hipFree(C->e0); hipFree(C->y); if (C->e_owned) hipFree(C->e); hipFree(C->b); hipFree(C->d); hipFree(C->vb); hipFree(C->lam);
  hipFree(C->B); hipFree(C->D); hipFree(C->VB); hipFree(C->sc);
  delete C;
  return BWGR_OK;
}

extern "C" int bwgr_chain_create_sharded(bwgr_chain **out, bwgr_panel *P, int model, const float *y, int memloc, float it,
                                         float bi, float pi, float df, float R2, uint64_t seed, int rng_mode, int64_t marker0,
                                         int64_t p_total, float MSx_total, double *e_ext) {
  if (!out || !P || !y) return fail(BWGR_EINVAL, "chain_create: null pointer");
  *out = nullptr;
  if (model < BWGR_BAYESA || model > BWGR_BAYESDPI) return fail(BWGR_EINVAL, "chain_create: bad model %d", model);
  if (marker0 < 0 || p_total < marker0 + P->p || p_total > 0xFFFFFFF0ll) return fail(BWGR_EINVAL, "chain_create: bad shard [%lld,+%lld) of %lld", (long long)marker0, (long long)P->p, (long long)p_total);
  HIPCHK(hipSetDevice(P->device));
  if (panel_cen(P)) {
    if (!has_d(model) || !use_sweep3(P, SWF_SELECT))
      return fail(BWGR_EINVAL, "chain_create: an implicitly centred panel (bwgr_panel_set_centred) runs the selection models BayesB / C / Cpi / Dpi only");
    if (!P->cpre) HIPCHK(hipMalloc(&P->cpre, sizeof(double) * ((size_t)P->nblocks + 1)));
  }
  bwgr_chain *C = new bwgr_chain();
  C->P = P; P->nchains++; C->model = model; C->itf = it; C->bif = bi; C->iit = (int)it; C->ibi = (int)bi;
  C->pi = pi; C->df = df; C->R2 = R2; C->seed = seed; C->rng_mode = rng_mode;
  C->marker0 = marker0; C->p_total = p_total; C->MSx_eff = MSx_total;
  C->Phi = MSx_total * (1 - R2) / R2;
  const size_t pb = sizeof(float) * P->p;
  auto bail = [&](int code) { bwgr_chain_destroy(C); return code; };
#define CCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return bail(fail(BWGR_EHIP, "%s failed: %s", #x, hipGetErrorString(e_))); } while (0)
  CCHK(hipMalloc(&C->y, sizeof(float) * P->n));
  if (e_ext) { C->e = e_ext; C->e_owned = false; } else CCHK(hipMalloc(&C->e, sizeof(double) * P->ld));
  CCHK(hipMalloc(&C->b, pb)); CCHK(hipMalloc(&C->d, pb)); CCHK(hipMalloc(&C->vb, pb)); CCHK(hipMalloc(&C->lam, pb));
  CCHK(hipMalloc(&C->B, pb)); CCHK(hipMalloc(&C->D, pb)); CCHK(hipMalloc(&C->VB, pb)); CCHK(hipMalloc(&C->sc, sizeof(ChainScalars)));
  CCHK(hipMemcpyAsync(C->y, y, sizeof(float) * P->n, memloc == BWGR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, P->stream));
  InitArgs ia; ia.y = C->y; ia.e = C->e; ia.n = (int)P->n; ia.p = (int)P->p; ia.ld = P->ld; ia.model = model;
  ia.pi = pi; ia.df = df; ia.R2 = R2; ia.MSx = MSx_total; ia.sc = C->sc;
  hipLaunchKernelGGL(k_chain_init, dim3(1), dim3(1024), 0, P->stream, ia);
  hipLaunchKernelGGL(k_marker_init, dim3(1024), dim3(256), 0, P->stream, C->b, C->d, C->vb, C->lam, C->B, C->D, C->VB, (int)P->p, C->sc);
  CCHK(hipGetLastError());
  CCHK(hipStreamSynchronize(P->stream));
#undef CCHK
  *out = C;
  return BWGR_OK;
}

extern "C" int bwgr_chain_create(bwgr_chain **out, bwgr_panel *P, int model, const float *y, int memloc, float it,
                                 float bi, float pi, float df, float R2, uint64_t seed, int rng_mode) {
  if (!P) return fail(BWGR_EINVAL, "chain_create: null pointer");
  return bwgr_chain_create_sharded(out, P, model, y, memloc, it, bi, pi, df, R2, seed, rng_mode, 0, P->p, P->MSx, nullptr);
}

static void chain_args(const bwgr_chain *C, int blk_begin, int blk_end, SweepArgs &a) {
  const bwgr_panel *P = C->P;
  const int model = C->model;
  memset(&a, 0, sizeof(a));
  fill_panel_args(P, a);
  a.blk_begin = blk_begin; a.blk_end = blk_end;
  int fl = 0;
  if (has_d(model)) fl |= SWF_SELECT;
  if (model == BWGR_BAYESDPI) fl |= SWF_ALT_B2 | SWF_MH;
  if (per_marker_vb(model)) fl |= SWF_LAM_VEC | SWF_VB_VEC;
  a.flags = fl | C->flags_extra;
  a.e = C->e; a.b = C->b; a.d = C->d; a.vb = C->vb; a.xx = P->xx; a.lam = C->lam; a.sc = C->sc;
  if (panel_cen(P)) {   // implicitly centred columns: the centred squared norms, the column sums, this handle's running block sums
    const bwgr_panel *root = P->parent ? P->parent : P;
    a.flags |= SWF_CENTRE; a.xx = root->xxc; a.csum = root->csum; a.cpre = P->cpre; a.ninv = 1.0 / (double)P->n;
  }
  a.iter = (uint32_t)C->done; a.marker0 = (uint32_t)C->marker0; a.rng = make_rng(C->seed, C->rng_mode);
}

extern "C" int bwgr_chain_sweep_blocks(bwgr_chain *C, int blk_begin, int blk_end) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  bwgr_panel *P = C->P;
  if (blk_begin < 0 || blk_end > P->nblocks || blk_begin >= blk_end) return fail(BWGR_EINVAL, "sweep_blocks: bad range [%d,%d) of %lld", blk_begin, blk_end, (long long)P->nblocks);
  if (C->done >= C->iit) return fail(BWGR_EINVAL, "sweep_blocks: all %d iterations already run", C->iit);
  HIPCHK(hipSetDevice(P->device));
  CHK(leave_pair_stream(P));
  SweepArgs a;
  chain_args(C, blk_begin, blk_end, a);
  choose_lag(P, a);
  if (use_winv(P, a.flags)) CHK(winv_alloc(P));
  int need = 0;
  CHK(sweep_guard(P, a, &need));   // (before anything of this sweep is enqueued: a refused sweep leaves the chain as it was)
  CHK(reset_exchange(P));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(BWGR_EHIP, "sweep_blocks: hipEventCreate failed"); }
  // the per-marker constants and speculative terms of an iteration depend on the state at its start only (a block's b is
  // untouched until the block is swept), so a chain that sweeps its panel in several ranges -- the exchange rounds of the
  // marker-sharded sampler -- pre-stages all of them with the first range
  bool prestaged = false;
  if (P->ps_owner != C || P->ps_iter != C->done) {
    SweepArgs all = a; all.blk_begin = 0; all.blk_end = (int)P->nblocks;
    launch_prestage(P, all);
    P->ps_owner = C; P->ps_iter = C->done;
    prestaged = true;
  }
  hipError_t he = hipEventRecord(e0, P->stream);
  if (he == hipSuccess) { launch_sweep_kernel(P, a); he = hipGetLastError(); }
  if (he == hipSuccess && prestaged && C->done + 1 < C->iit) {   // the next iteration's variates, beside this sweep
    SweepArgs all = a; all.blk_begin = 0; all.blk_end = (int)P->nblocks;
    draws_ahead(P, all, e0);
  }
  if (he == hipSuccess) he = hipEventRecord(e1, P->stream);
  if (he != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail(BWGR_EHIP, "sweep_blocks: %s", hipGetErrorString(he)); }
  C->ev.push_back(e0); C->ev.push_back(e1);
  guard_mark(P, P->stream, need);
  if (C->ev.size() >= 4096) CHK(bwgr_chain_sweep_ms(C, nullptr, nullptr));   // bound the number of live events
  return BWGR_OK;
}

namespace {
__global__ void k_round_delta(const double *__restrict__ e, const double *__restrict__ e0, double *__restrict__ delta, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) delta[i] = e[i] - e0[i];
}
__global__ void k_round_apply(double *__restrict__ e, const double *__restrict__ e0, const double *__restrict__ delta, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) e[i] = e0[i] + delta[i];
}
}  // namespace

// One exchange round of the marker-sharded sampler in two calls (the same steps a caller can take with sweep_blocks and
// its own vector arithmetic; here they cost two small launches instead of three framework operations per round):
//   round_sweep: remember e, sweep the blocks [blk_begin, blk_end) (an empty range sweeps nothing), delta = e - e_before
//   <caller: all-reduce(sum) delta over the ranks>
//   round_apply: e = e_before + delta
// delta_dev: ld doubles on the chain's device (the panel's padded row count).
extern "C" int bwgr_chain_round_sweep(bwgr_chain *C, int blk_begin, int blk_end, double *delta_dev) {
  if (!C || !delta_dev) return fail(BWGR_EINVAL, "round_sweep: null pointer");
  bwgr_panel *P = C->P;
  HIPCHK(hipSetDevice(P->device));
  if (!C->e0) HIPCHK(hipMalloc(&C->e0, sizeof(double) * P->ld));
  HIPCHK(hipMemcpyAsync(C->e0, C->e, sizeof(double) * P->ld, hipMemcpyDeviceToDevice, P->stream));
  if (blk_begin < blk_end) CHK(bwgr_chain_sweep_blocks(C, blk_begin, blk_end));
  hipLaunchKernelGGL(k_round_delta, dim3(64), dim3(256), 0, P->stream, C->e, C->e0, delta_dev, P->ld);
  HIPCHK(hipGetLastError());
  return BWGR_OK;
}
extern "C" int bwgr_chain_round_apply(bwgr_chain *C, const double *delta_dev) {
  if (!C || !delta_dev) return fail(BWGR_EINVAL, "round_apply: null pointer");
  if (!C->e0) return fail(BWGR_EINVAL, "round_apply: no round_sweep before it");
  bwgr_panel *P = C->P;
  HIPCHK(hipSetDevice(P->device));
  hipLaunchKernelGGL(k_round_apply, dim3(64), dim3(256), 0, P->stream, C->e, C->e0, delta_dev, P->ld);
  HIPCHK(hipGetLastError());
  return BWGR_OK;
}

extern "C" int bwgr_chain_get_sums(bwgr_chain *C, double sums[2]) {
  if (!C || !sums) return fail(BWGR_EINVAL, "null pointer");
  HIPCHK(hipSetDevice(C->P->device));
  ChainScalars h;
  HIPCHK(hipMemcpyAsync(&h, C->sc, sizeof(h), hipMemcpyDeviceToHost, C->P->stream));
  HIPCHK(hipStreamSynchronize(C->P->stream));
  if (h.error) return sweep_error(h.error, "chain");
  sums[0] = h.sum_d; sums[1] = h.sum_b2;
  return BWGR_OK;
}

__global__ void k_set_sums(ChainScalars *sc, double sd, double sb2) { sc->sum_d = sd; sc->sum_b2 = sb2; }

extern "C" int bwgr_chain_end_iteration(bwgr_chain *C, const double sums_total[2]) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  if (C->done >= C->iit) return fail(BWGR_EINVAL, "end_iteration: all %d iterations already run", C->iit);
  bwgr_panel *P = C->P;
  HIPCHK(hipSetDevice(P->device));
  const int model = C->model, i = C->done;
  if (sums_total) hipLaunchKernelGGL(k_set_sums, dim3(1), dim3(1), 0, P->stream, C->sc, sums_total[0], sums_total[1]);
  const int accumulate = (i > C->ibi) ? 1 : 0;   // if(i>ibi), src/Rcpp20260726ai.cpp:624
  TailArgs t; t.e = C->e; t.n = (int)P->n; t.p = (int)C->p_total; t.model = model; t.df = C->df; t.R2 = C->R2; t.Phi = C->Phi;
  t.accumulate = accumulate; t.iter = (uint32_t)i; t.rng = make_rng(C->seed, C->rng_mode); t.sc = C->sc;
  hipLaunchKernelGGL(k_tail, dim3(1), dim3(1024), 0, P->stream, t);
  hipLaunchKernelGGL(k_marker_tail, dim3((unsigned)std::min<int64_t>(2048, (P->p + 255) / 256)), dim3(256), 0, P->stream,
                     C->b, C->d, C->vb, C->lam, C->B, C->D, C->VB, (int)P->p, model, C->Phi, accumulate, C->sc);
  HIPCHK(hipGetLastError());
  C->done++;
  return BWGR_OK;
}

namespace {
__global__ void k_get_sums_dev(const ChainScalars *sc, double *out2) { out2[0] = sc->sum_d; out2[1] = sc->sum_b2; }
__global__ void k_set_sums_dev(ChainScalars *sc, const double *in2) { sc->sum_d = in2[0]; sc->sum_b2 = in2[1]; }
}  // namespace
// device-side forms of get_sums / end_iteration(sums_total) for the sharded sampler: the two sums stay on the device (the
// caller all-reduces sums_dev, two doubles, in place), so an iteration needs no host round trip
extern "C" int bwgr_chain_get_sums_dev(bwgr_chain *C, double *sums_dev) {
  if (!C || !sums_dev) return fail(BWGR_EINVAL, "null pointer");
  HIPCHK(hipSetDevice(C->P->device));
  hipLaunchKernelGGL(k_get_sums_dev, dim3(1), dim3(1), 0, C->P->stream, C->sc, sums_dev);
  HIPCHK(hipGetLastError());
  return BWGR_OK;
}
extern "C" int bwgr_chain_end_iteration_dev(bwgr_chain *C, const double *sums_total_dev) {
  if (!C || !sums_total_dev) return fail(BWGR_EINVAL, "null pointer");
  HIPCHK(hipSetDevice(C->P->device));
  hipLaunchKernelGGL(k_set_sums_dev, dim3(1), dim3(1), 0, C->P->stream, C->sc, sums_total_dev);
  HIPCHK(hipGetLastError());
  return bwgr_chain_end_iteration(C, nullptr);
}

extern "C" int bwgr_chain_run(bwgr_chain *C, int iters) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  if (iters < 0 || C->done + iters > C->iit) return fail(BWGR_EINVAL, "chain_run: %d more iterations would exceed it=%d (done %d)", iters, C->iit, C->done);
  for (int k = 0; k < iters; ++k) {
    CHK(bwgr_chain_sweep_blocks(C, 0, (int)C->P->nblocks));
    CHK(bwgr_chain_end_iteration(C, nullptr));
  }
  return BWGR_OK;
}

// Two chains of the same resident panel (C1 on a clone of C0's panel, or the other way round), advanced in lockstep by k_sweep3p: one
// set of streamer workgroups -- one pass over the genotypes -- serves both, each chain keeps its own sequencer.  Selection models on
// panels that have k_sweep3; every sweep of the pair is k_sweep3's (no device-side choice of engine).  Each chain's results are
// bit-identical to a run of its own.  (No reference counterpart: the callers that fit many models on one X -- mcmcCV's loop,
// /root/reference/R/cv.R:113-216 -- are where it plugs in.)
extern "C" int bwgr_chain_run_pair(bwgr_chain *C0, bwgr_chain *C1, int iters) {
  if (C0 && C0->P && panel_cen(C0->P)) return fail(BWGR_EINVAL, "chain_run_pair: this panel sweeps implicitly centred columns (bwgr_panel_set_centred), which only the fused chains do; call bwgr_panel_set_centred(P, 0) first");
  if (!C0 || !C1 || C0 == C1) return fail(BWGR_EINVAL, "chain_run_pair: two distinct chains");
  bwgr_panel *P0 = C0->P, *P1 = C1->P;
  const bwgr_panel *r0 = P0->parent ? P0->parent : P0, *r1 = P1->parent ? P1->parent : P1;
  if (r0 != r1 || P0 == P1) return fail(BWGR_EINVAL, "chain_run_pair: the chains must sit on two handles (panel and clone) of one resident panel");
  if (iters < 0 || C0->done + iters > C0->iit || C1->done + iters > C1->iit) return fail(BWGR_EINVAL, "chain_run_pair: %d more iterations exceed it", iters);
  SweepArgs t0, t1;
  chain_args(C0, 0, (int)P0->nblocks, t0); chain_args(C1, 0, (int)P1->nblocks, t1);
  if (!use_sweep3(P0, t0.flags) || !use_sweep3(P1, t1.flags) || !P0->qsum3 || !P1->qsum3)
    return fail(BWGR_EINVAL, "chain_run_pair: both chains must be selection models on a panel with k_sweep3");
  if (s3p_streamer_lds(P0->R3) > (size_t)160 * 1024) return fail(BWGR_EINVAL, "chain_run_pair: the paired streamers' LDS does not fit");
  HIPCHK(hipSetDevice(P0->device));
  // everything of the pair runs on ONE stream, owned by the root panel (it outlives both handles), and both handles move onto it
  // for as long as they run in pairs -- one cross-stream wait each, the first time: a wait per call would sit in a hardware queue
  // that other pairs' streams share, and stall them.  A handle's next sweep alone takes it back (leave_pair_stream).
  bwgr_panel *root = P0->parent ? P0->parent : P0;
  auto is_pair_stream = [&](hipStream_t st) { for (hipStream_t q : root->pair_streams) if (q == st) return true; return false; };
  hipStream_t s0 = nullptr;
  if (is_pair_stream(P0->stream)) s0 = P0->stream;
  else if (is_pair_stream(P1->stream)) s0 = P1->stream;
  else { HIPCHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); root->pair_streams.push_back(s0); }
  for (bwgr_panel *PX : {P0, P1}) {
    if (PX->stream == s0) continue;
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, PX->stream)); HIPCHK(hipStreamWaitEvent(s0, ev, 0));
    HIPCHK(hipEventDestroy(ev));
    if (!PX->pre_pair_set) { PX->pre_pair_stream = PX->stream; PX->pre_pair_set = true; }   // (leave_pair_stream takes the handle back)
    PX->stream = s0;
  }
  int rc = BWGR_OK;
  const size_t lds = std::max(s3p_streamer_lds(P0->R3), s3_seq_lds(P0->e3_D, r0->gram16));
  int need = 0;
  if (guard_on() && device_cus(P0->device) > 0) {   // the pair's one launch: K3 streamers and two sequencers, resident beside the other streams' sweeps
    std::vector<SpinLaunch> plan;
    plan.push_back(SpinLaunch{r0->gram16 ? reinterpret_cast<const void *>(k_sweep3p<uint16_t>) : reinterpret_cast<const void *>(k_sweep3p<int32_t>), P0->K3 + 2, SW_THREADS, lds});
    std::lock_guard<std::mutex> lk(g_guard_mu);
    CHK(plan_cus(plan, device_cus(P0->device), guard_busy(P0, s0, P1), &need));
  }
  P0->force3 = P1->force3 = true;
  for (int k = 0; k < iters && rc == BWGR_OK; ++k) {
    SweepArgs a0, a1;
    chain_args(C0, 0, (int)P0->nblocks, a0); chain_args(C1, 0, (int)P1->nblocks, a1);
    choose_lag(P0, a0); choose_lag(P1, a1);
    if ((rc = reset_exchange(P0)) != BWGR_OK || (rc = reset_exchange(P1)) != BWGR_OK) break;
    launch_prestage(P0, a0); launch_prestage(P1, a1);
    P0->ps_owner = C0; P0->ps_iter = C0->done; P1->ps_owner = C1; P1->ps_iter = C1->done;
    a0.gate3 = a1.gate3 = INFINITY;
    if (P0->debug_withhold || P1->debug_withhold) { a0.flags |= SWF_DEBUG_WITHHOLD; a1.flags |= SWF_DEBUG_WITHHOLD; }
    Sweep3Args A0, A1;
    sweep3_args(P0, a0, A0); sweep3_args(P1, a1, A1);
    const dim3 grid(P0->K3 + 2), blk(SW_THREADS);
    hipEvent_t evs[4] = {nullptr, nullptr, nullptr, nullptr};   // both chains time the launch they share
    bool ev_ok = true;
    for (int q = 0; q < 4 && ev_ok; ++q) ev_ok = hipEventCreate(&evs[q]) == hipSuccess;
    if (!ev_ok) { for (hipEvent_t q : evs) if (q) (void)hipEventDestroy(q); rc = fail(BWGR_EHIP, "chain_run_pair: hipEventCreate failed"); break; }
    (void)hipEventRecord(evs[0], s0); (void)hipEventRecord(evs[2], s0);
    if (A0.g16) hipLaunchKernelGGL(k_sweep3p<uint16_t>, grid, blk, lds, s0, A0, A1);
    else hipLaunchKernelGGL(k_sweep3p<int32_t>, grid, blk, lds, s0, A0, A1);
    (void)hipEventRecord(evs[1], s0); (void)hipEventRecord(evs[3], s0);
    C0->ev.push_back(evs[0]); C0->ev.push_back(evs[1]); C1->ev.push_back(evs[2]); C1->ev.push_back(evs[3]);
    guard_mark(P0, s0, need);
    if (hipGetLastError() != hipSuccess) { rc = fail(BWGR_EHIP, "chain_run_pair: launch failed"); break; }
    if ((rc = bwgr_chain_end_iteration(C0, nullptr)) != BWGR_OK) break;
    rc = bwgr_chain_end_iteration(C1, nullptr);
    if (C0->ev.size() >= 4096) (void)bwgr_chain_sweep_ms(C0, nullptr, nullptr);
    if (C1->ev.size() >= 4096) (void)bwgr_chain_sweep_ms(C1, nullptr, nullptr);
  }
  P0->force3 = P1->force3 = false;
  return rc;
}

extern "C" int bwgr_chain_sync(bwgr_chain *C) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  HIPCHK(hipSetDevice(C->P->device));
  HIPCHK(hipStreamSynchronize(C->P->stream));
  ChainScalars h;
  HIPCHK(d2h(C->P->stream, &h, C->sc, sizeof(h)));
  if (h.error) return sweep_error(h.error, "chain");
  return BWGR_OK;
}

extern "C" int bwgr_chain_iterations(const bwgr_chain *C, int *done) {
  if (!C || !done) return fail(BWGR_EINVAL, "null pointer");
  *done = C->done;
  return BWGR_OK;
}

extern "C" int bwgr_chain_sweep_ms(bwgr_chain *C, float *avg_ms, int *launches) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  HIPCHK(hipSetDevice(C->P->device));
  HIPCHK(hipStreamSynchronize(C->P->stream));
  float total = 0; int nl = 0;
  for (size_t k = 0; k + 1 < C->ev.size(); k += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, C->ev[k], C->ev[k + 1]));
    total += ms; nl++;
    hipEventDestroy(C->ev[k]); hipEventDestroy(C->ev[k + 1]);
  }
  C->ev.clear();
  C->ms_acc += total; C->launch_acc += nl;
  if (avg_ms && launches) {
    *launches = C->launch_acc;
    *avg_ms = C->launch_acc ? C->ms_acc / C->launch_acc : 0.0f;
    C->ms_acc = 0; C->launch_acc = 0;
  }
  return BWGR_OK;
}

extern "C" int bwgr_chain_redo_count(bwgr_chain *C, int *count) {
  if (!C || !count) return fail(BWGR_EINVAL, "null pointer");
  CHK(bwgr_chain_sync(C));
  ChainScalars h;
  HIPCHK(d2h(C->P->stream, &h, C->sc, sizeof(h)));
  *count = (int)h.nredo;
  return BWGR_OK;
}

extern "C" int bwgr_chain_state(bwgr_chain *C, float *b, float *d, float *e, float *vb, float *scal) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  CHK(bwgr_chain_sync(C));
  bwgr_panel *P = C->P;
  const size_t pb = sizeof(float) * P->p;
  ChainScalars h;
  HIPCHK(d2h(P->stream, &h, C->sc, sizeof(h)));
  if (b) HIPCHK(d2h(P->stream, b, C->b, pb));
  if (d) HIPCHK(d2h(P->stream, d, C->d, pb));
  if (e) {
    float *ef = nullptr;
    HIPCHK(hipMalloc(&ef, sizeof(float) * P->n));
    hipLaunchKernelGGL(k_d2f, dim3(64), dim3(256), 0, P->stream, C->e, ef, P->n);
    HIPCHK(hipMemcpyAsync(e, ef, sizeof(float) * P->n, hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));
    hipFree(ef);
  }
  if (vb) {
    if (per_marker_vb(C->model)) HIPCHK(d2h(P->stream, vb, C->vb, pb));
    else for (int64_t j = 0; j < P->p; ++j) vb[j] = h.vb;
  }
  if (scal) { scal[0] = h.mu; scal[1] = h.ve; scal[2] = h.vb; scal[3] = h.pi; }
  return BWGR_OK;
}

// column chunks of the two-stage GEMV: enough workgroups to fill the chip at 16 rows per thread (int8) or 4 (float)
static int gemv_chunks(const bwgr_panel *P) { return (int)std::min<int64_t>(P->is_f32 ? 64 : 512, std::max<int64_t>(1, P->p / 512)); }
template <typename CT>
static void gemv_launch(bwgr_panel *P, const CT *coef_dev, int nchunks, int cpc, double *part) {
  if (P->is_f32) {
    dim3 grid((unsigned)((P->ld / 4 + 255) / 256), (unsigned)nchunks);
    hipLaunchKernelGGL((k_gemv_part<float, CT>), grid, dim3(256), 0, P->stream, (const float *)P->X, P->ld, P->R, (int)P->p, coef_dev, cpc, part);
  } else {
    dim3 grid((unsigned)((P->ld / 16 + 255) / 256), (unsigned)nchunks);
    hipLaunchKernelGGL((k_gemv_part_i8<CT>), grid, dim3(256), 0, P->stream, (const int8_t *)P->X, P->ld, P->R, (int)P->p, coef_dev, cpc, part);
  }
}

// hat = X*B + MU   (src/Rcpp20260726ai.cpp:629-630), fp64 accumulation, deterministic two-stage
template <typename CT>
static int gemv_hat(bwgr_panel *P, const CT *coef_dev, float MU, float *hat_dev, bool centred = false) {
  const int nchunks = gemv_chunks(P);
  const int cpc = (int)((P->p + nchunks - 1) / nchunks);
  double *part = nullptr;
  HIPCHK(hipMalloc(&part, sizeof(double) * ((size_t)nchunks * P->ld + 1)));
  gemv_launch<CT>(P, coef_dev, nchunks, cpc, part);
  double *cen_off = nullptr;
  if constexpr (std::is_same<CT, float>::value) {
    if (centred) {   // X_c B = X B - sum_j mean_j B_j
      cen_off = part + (size_t)nchunks * P->ld;
      hipLaunchKernelGGL(k_cen_dot, dim3(1), dim3(1024), 0, P->stream, (P->parent ? P->parent : P)->csum, coef_dev, P->p, 1.0 / (double)P->n, cen_off);
    }
  }
  hipLaunchKernelGGL(k_hat_finish, dim3((unsigned)((P->n + 255) / 256)), dim3(256), 0, P->stream, part, P->ld, nchunks, (int)P->n, MU, hat_dev, (const double *)cen_off);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(P->stream));
  HIPCHK(hipFree(part));
  return BWGR_OK;
}

extern "C" int bwgr_chain_result(bwgr_chain *C, float *mu, float *b, float *d, float *hat, float *vb, float *ve,
                                 float *h2, float *MSx, float *pi, float *pval) {
  if (!C) return fail(BWGR_EINVAL, "null chain");
  if (C->done != C->iit) return fail(BWGR_EINVAL, "chain_result: %d of %d iterations run", C->done, C->iit);
  CHK(bwgr_chain_sync(C));
  bwgr_panel *P = C->P;
  const size_t pb = sizeof(float) * P->p;
  const bool per = per_marker_vb(C->model);
  const float MCMC = C->itf - C->bif;                                              // :626
  DevBufs bufs;
  float *pval_dev = nullptr;
  if (pval && !(pval_dev = bufs.get<float>((size_t)P->p))) return fail(BWGR_ENOMEM, "chain_result: device allocation failed");
  if (!C->finalized) {
    hipLaunchKernelGGL(k_final_markers, dim3(1024), dim3(256), 0, P->stream, C->B, C->D, C->VB, pval_dev, (int)P->p, MCMC, per ? 1 : 0);
    HIPCHK(hipGetLastError());
    C->finalized = true;
  } else if (pval_dev) {
    hipLaunchKernelGGL(k_final_markers, dim3(1024), dim3(256), 0, P->stream, C->B, C->D, C->VB, pval_dev, (int)P->p, 1.0f, per ? 1 : 0);
  }
  ChainScalars h;
  HIPCHK(hipMemcpyAsync(&h, C->sc, sizeof(h), hipMemcpyDeviceToHost, P->stream));
  HIPCHK(hipStreamSynchronize(P->stream));
  const float MU = h.MU / MCMC, VE = h.VE / MCMC;
  float VBs = h.VBs / MCMC, Pi = 0, vg;
  if (C->model == BWGR_BAYESCPI || C->model == BWGR_BAYESDPI) Pi = 1 - h.Pi / MCMC;   // :911
  if (per) {
    // vg = VB.sum()
    double *part = nullptr; float *sdev = nullptr;
    HIPCHK(hipMalloc(&part, sizeof(double) * 256)); HIPCHK(hipMalloc(&sdev, sizeof(float)));
    hipLaunchKernelGGL(k_sum_stage1, dim3(256), dim3(256), 0, P->stream, C->VB, (int64_t)P->p, part);
    hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, P->stream, part, 256, sdev);
    HIPCHK(d2h(P->stream, &vg, sdev, sizeof(float)));
    hipFree(part); hipFree(sdev);
  } else {
    vg = VBs * C->MSx_eff;
    if (C->model == BWGR_BAYESCPI) vg = VBs * C->MSx_eff / Pi;                         // :913
  }
  if (mu) *mu = MU;
  if (ve) *ve = VE;
  if (h2) *h2 = vg / (vg + VE);
  if (MSx) *MSx = C->MSx_eff;
  if (pi) *pi = Pi;
  if (b) HIPCHK(d2h(P->stream, b, C->B, pb));
  if (d) HIPCHK(d2h(P->stream, d, C->D, pb));
  if (vb) { if (per) HIPCHK(d2h(P->stream, vb, C->VB, pb)); else vb[0] = VBs; }
  if (pval) HIPCHK(d2h(P->stream, pval, pval_dev, pb));
  if (hat) {
    float *hat_dev = nullptr;
    HIPCHK(hipMalloc(&hat_dev, sizeof(float) * P->n));
    int rc = gemv_hat<float>(P, C->B, MU, hat_dev, panel_cen(P));
    if (rc == BWGR_OK) HIPCHK(d2h(P->stream, hat, hat_dev, sizeof(float) * P->n));
    hipFree(hat_dev);
    CHK(rc);
  }
  return BWGR_OK;
}

extern "C" int bwgr_bayes(bwgr_panel *P, int model, const float *y, float it, float bi, float pi, float df, float R2,
                          uint64_t seed, int rng_mode, float *mu, float *b, float *d, float *hat, float *vb, float *ve,
                          float *h2, float *MSx, float *pi_out, float *pval) {
  bwgr_chain *C = nullptr;
  CHK(bwgr_chain_create(&C, P, model, y, BWGR_HOST, it, bi, pi, df, R2, seed, rng_mode));
  int rc = bwgr_chain_run(C, (int)it);
  if (rc == BWGR_OK) rc = bwgr_chain_result(C, mu, b, d, hat, vb, ve, h2, MSx, pi_out, pval);
  bwgr_chain_destroy(C);
  return rc;
}

// BayesA2 / BayesB2 / BayesRR2, src/Rcpp20260726ai.cpp:990-1218: two chains over two panels sharing the residual
extern "C" int bwgr_bayes2(bwgr_panel *P1, bwgr_panel *P2, int base_model, const float *y, float it, float bi, float pi, float df,
                           float R2, uint64_t seed, int rng_mode, float *mu, float *b1, float *d1, float *vb1, float *b2,
                           float *d2, float *vb2, float *ve, float *hat, float *h2) {
  if ((P1 && panel_cen(P1)) || (P2 && panel_cen(P2))) return fail(BWGR_EINVAL, "bayes2: this panel sweeps implicitly centred columns (bwgr_panel_set_centred), which only the fused chains do; call bwgr_panel_set_centred(P, 0) first");
  if (!P1 || !P2 || !y) return fail(BWGR_EINVAL, "bayes2: null pointer");
  if (base_model != BWGR_BAYESA && base_model != BWGR_BAYESB && base_model != BWGR_BAYESRR)
    return fail(BWGR_EINVAL, "bayes2: base model must be BayesA, BayesB or BayesRR (got %d)", base_model);
  if (P1->device != P2->device || P1->n != P2->n || P1->ld != P2->ld || P1->K != P2->K || P1->R != P2->R)
    return fail(BWGR_EINVAL, "bayes2: the two panels must share device, rows and slab geometry (n %lld/%lld, %d x %d vs %d x %d rows)",
                (long long)P1->n, (long long)P2->n, P1->K, P1->R, P2->K, P2->R);
  const int64_t p1 = P1->p, p2 = P2->p, n = P1->n;
  if (p1 + p2 > 0xFFFFFFF0ll - 2) return fail(BWGR_EINVAL, "bayes2: p1 + p2 too large");
  HIPCHK(hipSetDevice(P1->device));
  hipStream_t s2_saved = P2->stream;
  P2->stream = P1->stream;   // one stream orders the two chains' kernels
  bwgr_chain *C1 = nullptr, *C2 = nullptr;
  float *h1d = nullptr, *h2d = nullptr, *hatd = nullptr, *sdev = nullptr; double *part = nullptr;
  int rc = bwgr_chain_create_sharded(&C1, P1, base_model, y, BWGR_HOST, it, bi, pi, df, R2, seed, rng_mode, 0, p1 + p2, P1->MSx, nullptr);
  if (rc == BWGR_OK) rc = bwgr_chain_create_sharded(&C2, P2, base_model, y, BWGR_HOST, it, bi, pi, df, R2, seed, rng_mode, p1, p1 + p2, P2->MSx, C1->e);
  auto done = [&](int code) {
    if (C2) bwgr_chain_destroy(C2);
    if (C1) bwgr_chain_destroy(C1);
    hipFree(h1d); hipFree(h2d); hipFree(hatd); hipFree(sdev); hipFree(part);
    P2->stream = s2_saved;
    return code;
  };
  if (rc != BWGR_OK) return done(rc);
#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return done(fail(BWGR_EHIP, "%s failed: %s", #x, hipGetErrorString(e_))); } while (0)
  const bool rr = (base_model == BWGR_BAYESRR), per = !rr;
  if (base_model == BWGR_BAYESB) { C1->flags_extra = SWF_ALT_B2; C2->flags_extra = SWF_ALT_B2; }
  if (rr) {
    hipLaunchKernelGGL(k_set_rr2_start, dim3(1), dim3(1), 0, P1->stream, C1->sc);
    hipLaunchKernelGGL(k_set_rr2_start, dim3(1), dim3(1), 0, P1->stream, C2->sc);
  }
  const int iit = (int)it, ibi = (int)bi;
  for (int i = 0; i < iit; ++i) {
    rc = bwgr_chain_sweep_blocks(C1, 0, (int)P1->nblocks);
    if (rc == BWGR_OK) rc = bwgr_chain_sweep_blocks(C2, 0, (int)P2->nblocks);
    if (rc != BWGR_OK) return done(rc);
    const int accumulate = (i > ibi) ? 1 : 0;                                        // if(i>ibi), :1047
    Tail2Args t; t.e = C1->e; t.n = (int)n; t.p1 = (int)p1; t.p2 = (int)p2; t.rr = rr ? 1 : 0; t.df = df;
    t.accumulate = accumulate; t.iter = (uint32_t)i; t.rng = make_rng(seed, rng_mode); t.sc1 = C1->sc; t.sc2 = C2->sc;
    hipLaunchKernelGGL(k_tail2, dim3(1), dim3(1024), 0, P1->stream, t);
    hipLaunchKernelGGL(k_marker_tail, dim3((unsigned)std::min<int64_t>(2048, (p1 + 255) / 256)), dim3(256), 0, P1->stream,
                       C1->b, C1->d, C1->vb, C1->lam, C1->B, C1->D, C1->VB, (int)p1, base_model, 0.0f, accumulate, C1->sc);
    hipLaunchKernelGGL(k_marker_tail, dim3((unsigned)std::min<int64_t>(2048, (p2 + 255) / 256)), dim3(256), 0, P1->stream,
                       C2->b, C2->d, C2->vb, C2->lam, C2->B, C2->D, C2->VB, (int)p2, base_model, 0.0f, accumulate, C2->sc);
    BCHK(hipGetLastError());
    C1->done++; C2->done++;
  }
  rc = bwgr_chain_sync(C1);
  if (rc == BWGR_OK) rc = bwgr_chain_sync(C2);
  if (rc != BWGR_OK) return done(rc);
  const float MCMC = it - bi;                                                        // :1049
  hipLaunchKernelGGL(k_final_markers, dim3(1024), dim3(256), 0, P1->stream, C1->B, C1->D, C1->VB, (float *)nullptr, (int)p1, MCMC, per ? 1 : 0);
  hipLaunchKernelGGL(k_final_markers, dim3(1024), dim3(256), 0, P1->stream, C2->B, C2->D, C2->VB, (float *)nullptr, (int)p2, MCMC, per ? 1 : 0);
  BCHK(hipGetLastError());
  ChainScalars g1, g2;
  BCHK(hipMemcpyAsync(&g1, C1->sc, sizeof(g1), hipMemcpyDeviceToHost, P1->stream));
  BCHK(hipMemcpyAsync(&g2, C2->sc, sizeof(g2), hipMemcpyDeviceToHost, P1->stream));
  BCHK(hipStreamSynchronize(P1->stream));
  const float MU = g1.MU / MCMC, VE = g1.VE / MCMC, VB1s = g1.VBs / MCMC, VB2s = g2.VBs / MCMC;
  float vg;
  if (per) {                                                                         // vg = VB1.sum() + VB2.sum(), :1051
    float v1 = 0, v2 = 0;
    BCHK(hipMalloc(&part, sizeof(double) * 256)); BCHK(hipMalloc(&sdev, sizeof(float)));
    hipLaunchKernelGGL(k_sum_stage1, dim3(256), dim3(256), 0, P1->stream, C1->VB, (int64_t)p1, part);
    hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, P1->stream, part, 256, sdev);
    BCHK(d2h(P1->stream, &v1, sdev, sizeof(float)));
    hipLaunchKernelGGL(k_sum_stage1, dim3(256), dim3(256), 0, P1->stream, C2->VB, (int64_t)p2, part);
    hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, P1->stream, part, 256, sdev);
    BCHK(d2h(P1->stream, &v2, sdev, sizeof(float)));
    vg = v1 + v2;
  } else vg = VB1s * P1->MSx + VB2s * P2->MSx;                                       // :1213
  if (mu) *mu = MU;
  if (ve) *ve = VE;
  if (h2) *h2 = vg / (vg + VE);
  if (b1) BCHK(d2h(P1->stream, b1, C1->B, sizeof(float) * p1));
  if (b2) BCHK(d2h(P1->stream, b2, C2->B, sizeof(float) * p2));
  if (d1) BCHK(d2h(P1->stream, d1, C1->D, sizeof(float) * p1));
  if (d2) BCHK(d2h(P1->stream, d2, C2->D, sizeof(float) * p2));
  if (vb1) { if (per) BCHK(d2h(P1->stream, vb1, C1->VB, sizeof(float) * p1)); else vb1[0] = VB1s; }
  if (vb2) { if (per) BCHK(d2h(P1->stream, vb2, C2->VB, sizeof(float) * p2)); else vb2[0] = VB2s; }
  if (hat) {                                                                         // fit = X1*B1 + X2*B2; fit += MU, :1052-1053
    BCHK(hipMalloc(&h1d, sizeof(float) * n)); BCHK(hipMalloc(&h2d, sizeof(float) * n)); BCHK(hipMalloc(&hatd, sizeof(float) * n));
    rc = gemv_hat<float>(P1, C1->B, 0.0f, h1d);
    if (rc == BWGR_OK) rc = gemv_hat<float>(P2, C2->B, 0.0f, h2d);
    if (rc != BWGR_OK) return done(rc);
    hipLaunchKernelGGL(k_hat2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, P1->stream, h1d, h2d, MU, hatd, (int)n);
    BCHK(hipGetLastError());
    BCHK(d2h(P1->stream, hat, hatd, sizeof(float) * n));
  }
#undef BCHK
  return done(BWGR_OK);
}

// host side of the RNG contract for wgr's row resampling, R/wgr.R:68: Use = sort(sample(n, n*bag, rp)) - 1
static double host_uniform(uint64_t seed, uint32_t marker, uint32_t iter, uint32_t purpose, uint32_t k) {
  uint32_t c0 = marker, c1 = iter, c2 = purpose, c3 = k, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6) + 0.5) / 9007199254740992.0;
}
static void bag_rows(uint64_t seed, uint32_t iter, int64_t n, int64_t k, int rp, std::vector<int> &use) {
  use.resize((size_t)k);
  if (rp) {
    for (int64_t t = 0; t < k; ++t) { int r = (int)(host_uniform(seed, (uint32_t)t, iter, RNG_BAG, 1) * (double)n); use[(size_t)t] = r >= n ? (int)n - 1 : r; }
  } else {
    std::vector<std::pair<double, int>> kv((size_t)n);
    for (int64_t i = 0; i < n; ++i) kv[(size_t)i] = std::make_pair(host_uniform(seed, (uint32_t)i, iter, RNG_BAG, 0), (int)i);
    std::sort(kv.begin(), kv.end());
    for (int64_t t = 0; t < k; ++t) use[(size_t)t] = kv[(size_t)t].second;
  }
  std::sort(use.begin(), use.end());
}

extern "C" int bwgr_sample_rows(uint64_t seed, uint32_t iter, int64_t n, int64_t k, int rp, int *rows) {
  if (!rows || n < 1 || k < 0 || (!rp && k > n) || n > 0x7FFFFFFFll) return fail(BWGR_EINVAL, "sample_rows: bad arguments (n=%lld, k=%lld, rp=%d)", (long long)n, (long long)k, rp);
  std::vector<int> use;
  bag_rows(seed, iter, n, k, rp, use);
  for (int64_t t = 0; t < k; ++t) rows[t] = use[(size_t)t];
  return BWGR_OK;
}

// X * coef (fp64 partial products per column chunk); caller finishes.  Returns nchunks and the device buffer.
template <typename CT>
static int gemv_parts(bwgr_panel *P, const CT *coef_dev, double **part_out, int *nchunks_out) {
  const int nchunks = gemv_chunks(P);
  const int cpc = (int)((P->p + nchunks - 1) / nchunks);
  double *part = *part_out;
  if (!part) HIPCHK(hipMalloc(&part, sizeof(double) * (size_t)nchunks * P->ld));
  gemv_launch<CT>(P, coef_dev, nchunks, cpc, part);
  HIPCHK(hipGetLastError());
  *part_out = part; *nchunks_out = nchunks;
  return BWGR_OK;
}

extern "C" int bwgr_wgr(bwgr_panel *P, const double *y, int it, int bi, int th, int iv, int de, double pi, double df, double R2,
                        uint64_t seed, int rng_mode, double *mu, double *b, double *Vb, double *d, double *Ve, double *hat,
                        double *cxx) {
  return bwgr_wgr_ex(P, y, it, bi, th, iv, de, pi, df, R2, seed, rng_mode, nullptr, nullptr, 0, 1.0, 0, mu, b, Vb, d, Ve, hat, cxx, nullptr, nullptr);
}

extern "C" int bwgr_wgr_ex(bwgr_panel *P, const double *y, int it, int bi, int th, int iv, int de, double pi, double df, double R2,
                           uint64_t seed, int rng_mode, const double *U, const double *V, int64_t pk, double bag, int rp,
                           double *mu, double *b, double *Vb, double *d, double *Ve, double *hat, double *cxx, double *u, double *Vk) {
  if (P && panel_cen(P)) return fail(BWGR_EINVAL, "wgr: this panel sweeps implicitly centred columns (bwgr_panel_set_centred), which only the fused chains do; call bwgr_panel_set_centred(P, 0) first");
  if (!P || !y) return fail(BWGR_EINVAL, "wgr: null pointer");
  if (!U || pk <= 0) { U = nullptr; pk = 0; }
  if (U && !V) return fail(BWGR_EINVAL, "wgr: eigenvalues missing");
  const bool bagging = (bag != 1.0);
  if (bagging && U) return fail(BWGR_EINVAL, "wgr: bag != 1 with eigK is undefined in the reference (R/wgr.R:73-79 index a subsampled e with full row ids)");
  if (bagging && !(bag > 0.0)) return fail(BWGR_EINVAL, "wgr: bag must be > 0");
  const int64_t nbag = bagging ? (int64_t)((double)P->n * bag) : P->n;
  if (bagging && nbag < 2) return fail(BWGR_EINVAL, "wgr: n*bag < 2");
  // sample(n, n*bag, FALSE) cannot take more than the population (R errors out, R/wgr.R:68)
  if (bagging && !rp && nbag > P->n) return fail(BWGR_EINVAL, "wgr: bag > 1 needs rp = TRUE (cannot take %lld of %lld rows without replacement)", (long long)nbag, (long long)P->n);
  if (bagging) df = df / (bag * bag);                                              // R/wgr.R:20
  if (it < 1 || bi < 0 || th < 1) return fail(BWGR_EINVAL, "wgr: need it >= 1, bi >= 0, th >= 1");
  if (de) iv = 1;                                                                  // R/wgr.R:9
  HIPCHK(hipSetDevice(P->device));
  const int p = (int)P->p, n = (int)P->n;
  const size_t pd = sizeof(double) * p, pf = sizeof(float) * p;
  const Rng rng = make_rng(seed, rng_mode);
  int mc = 0; for (int q = bi; q <= it; q += th) mc++;                             // post = seq(bi,it,th)
  if (mc < 1) return fail(BWGR_EINVAL, "wgr: seq(bi,it,th) is empty");
  std::vector<void *> owned;
  auto dalloc = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, bytes) != hipSuccess) return nullptr; owned.push_back(q); return q; };
  auto cleanup = [&]() { for (void *q : owned) hipFree(q); };
  bwgr_panel *PU = nullptr;   // eigenvectors as an fp32 panel (KMUP narrows U to float like any other X)
  if (U) {
    int rcu = bwgr_panel_create(&PU, U, BWGR_X_F64, BWGR_HOST, n, pk, n, P->device, 0, 0);
    if (rcu != BWGR_OK) return rcu;
    PU->stream = P->stream;
  }
  bwgr_panel *PB = nullptr;   // the row subsample of this iteration (bag != 1): same markers, nbag rows
  int *use_d = nullptr;
  std::vector<int> use_h;
  auto drop_panels = [&]() { if (PU) bwgr_panel_destroy(PU); if (PB) bwgr_panel_destroy(PB); hipFree(use_d); PU = PB = nullptr; use_d = nullptr; };
  if (bagging) {
    int rcb = panel_alloc(&PB, P->is_f32, nbag, P->p, P->device, P->m, 0);
    if (rcb != BWGR_OK) { drop_panels(); return rcb; }
    PB->stream = P->stream;
    if (hipMalloc(&use_d, sizeof(int) * (size_t)nbag) != hipSuccess) { drop_panels(); return fail(BWGR_ENOMEM, "wgr: device allocation failed"); }
  }
  const int64_t ldmax = std::max<int64_t>(std::max<int64_t>(P->ld, PU ? PU->ld : 0), PB ? PB->ld : 0);
  const size_t kd = sizeof(double) * (size_t)std::max<int64_t>(pk, 1), kf = sizeof(float) * (size_t)std::max<int64_t>(pk, 1);
  double *yd = (double *)dalloc(sizeof(double) * n), *eR = (double *)dalloc(sizeof(double) * ldmax), *e64 = (double *)dalloc(sizeof(double) * ldmax);
  double *Ud = (double *)dalloc(sizeof(double) * (size_t)std::max<int64_t>(n * pk, 1)), *Vd = (double *)dalloc(kd), *hR = (double *)dalloc(kd), *Hk = (double *)dalloc(kd), *uhd = (double *)dalloc(sizeof(double) * n);
  float *hf = (float *)dalloc(kf), *dhf = (float *)dalloc(kf), *xxKf = (float *)dalloc(kf), *Lkf = (float *)dalloc(kf), *vbk = (float *)dalloc(kf);
  ChainScalars *sck = (ChainScalars *)dalloc(sizeof(ChainScalars));
  double *xx64 = (double *)dalloc(pd), *vx64 = (double *)dalloc(pd), *bR = (double *)dalloc(pd), *dR = (double *)dalloc(pd);
  double *VbR = (double *)dalloc(pd), *LR = (double *)dalloc(pd), *B = (double *)dalloc(pd), *D = (double *)dalloc(pd), *VB = (double *)dalloc(pd);
  float *bf = (float *)dalloc(pf), *dfl = (float *)dalloc(pf), *Lf = (float *)dalloc(pf), *xxf = (float *)dalloc(pf), *vbf = (float *)dalloc(pf);
  double *part1 = (double *)dalloc(sizeof(double) * 256), *part2 = (double *)dalloc(sizeof(double) * 256), *hatd = (double *)dalloc(sizeof(double) * n);
  WgrScalars *ws = (WgrScalars *)dalloc(sizeof(WgrScalars));
  ChainScalars *sc = (ChainScalars *)dalloc(sizeof(ChainScalars));
  if (!Ud || !Vd || !hR || !Hk || !uhd || !hf || !dhf || !xxKf || !Lkf || !vbk || !sck ||
      !yd || !eR || !e64 || !xx64 || !vx64 || !bR || !dR || !VbR || !LR || !B || !D || !VB || !bf || !dfl || !Lf || !xxf || !vbf || !part1 || !part2 || !hatd || !ws || !sc) {
    cleanup(); drop_panels(); return fail(BWGR_ENOMEM, "wgr: device allocation failed");
  }
  int rc = BWGR_OK;
  double *gpart = nullptr; int nchunks = 0;
#define WCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = fail(BWGR_EHIP, "%s failed: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
  {
    WCHK(hipMemcpyAsync(yd, y, sizeof(double) * n, hipMemcpyHostToDevice, P->stream));
    if (pk > 0) {
      WCHK(hipMemcpyAsync(Ud, U, sizeof(double) * (size_t)n * pk, hipMemcpyHostToDevice, P->stream));
      WCHK(hipMemcpyAsync(Vd, V, kd, hipMemcpyHostToDevice, P->stream));
      WCHK(hipMemsetAsync(hR, 0, kd, P->stream)); WCHK(hipMemsetAsync(Hk, 0, kd, P->stream));
    }
    const int wpb = 4;
    if (P->is_f32) hipLaunchKernelGGL(k_stats64<float>, dim3((p + wpb - 1) / wpb), dim3(64 * wpb), 0, P->stream, (const float *)P->X, P->R, n, p, xx64, vx64);
    else hipLaunchKernelGGL(k_stats64<int8_t>, dim3((p + wpb - 1) / wpb), dim3(64 * wpb), 0, P->stream, (const int8_t *)P->X, P->R, n, p, xx64, vx64);
    hipLaunchKernelGGL(k_dsum_stage1, dim3(256), dim3(256), 0, P->stream, vx64, (int64_t)p, part1, 0);
    hipLaunchKernelGGL(k_dsum_stage1, dim3(256), dim3(256), 0, P->stream, xx64, (int64_t)p, part2, 0);
    hipLaunchKernelGGL(k_wgr_init, dim3(1), dim3(1024), 0, P->stream, yd, eR, n, ldmax, part1, part2, p, df, R2, ws);
    hipLaunchKernelGGL(k_wgr_marker_init, dim3(1024), dim3(256), 0, P->stream, bR, dR, VbR, LR, B, D, VB, p, ws);
    if (bagging) hipLaunchKernelGGL(k_scale_d, dim3(256), dim3(256), 0, P->stream, xx64, (int64_t)p, bag);     // xx = crossprod * bag, R/wgr.R:46
    WCHK(hipGetLastError());
    const unsigned pg = (unsigned)std::min<int64_t>(2048, (P->p + 255) / 256);
    for (int i = 1; i <= it; ++i) {                                                // R/wgr.R:66
      const uint32_t itx = (uint32_t)(i - 1);
      const int accumulate = (i >= bi && ((i - bi) % th) == 0) ? 1 : 0;            // i %in% post
      if (pk > 0) {                                                                // R/wgr.R:70-76
        hipLaunchKernelGGL(k_wgr_pre_k, dim3(64), dim3(256), 0, P->stream, hR, Vd, eR, hf, dhf, xxKf, Lkf, e64, (int)pk, n, ldmax, ws, sck);
        SweepArgs ak; memset(&ak, 0, sizeof(ak));
        fill_panel_args(PU, ak);
        ak.flags = SWF_LAM_VEC;
        ak.e = e64; ak.b = hf; ak.d = dhf; ak.vb = vbk; ak.xx = xxKf; ak.lam = Lkf; ak.sc = sck; ak.iter = itx; ak.marker0 = 0x80000000u; ak.rng = rng;
        rc = launch_sweep(PU, ak);
        if (rc != BWGR_OK) goto done;
        hipLaunchKernelGGL(k_wgr_post_k, dim3(64), dim3(256), 0, P->stream, hf, hR, e64, eR, (int)pk, n);
      }
      hipLaunchKernelGGL(k_wgr_pre, dim3(pg), dim3(256), 0, P->stream, bR, dR, LR, xx64, eR, bf, dfl, Lf, xxf, e64, p, n, ldmax, (float)pi, ws, sc);
      SweepArgs a; memset(&a, 0, sizeof(a));
      bwgr_panel *PS = bagging ? PB : P;                                           // the panel this iteration sweeps
      if (bagging) {                                                               // R/wgr.R:68 + KMUP2's gathers
        bag_rows(seed, itx, n, nbag, rp, use_h);
        WCHK(hipMemcpyAsync(use_d, use_h.data(), sizeof(int) * (size_t)nbag, hipMemcpyHostToDevice, P->stream));
        launch_gather_rows(P, PB, use_d, nbag);
        rc = panel_build_gram(PB);                                                  // syncs the stream (use_h stays valid)
        if (rc != BWGR_OK) goto done;
        hipLaunchKernelGGL(k_gather_e, dim3(64), dim3(256), 0, P->stream, eR, use_d, (int)nbag, ldmax, e64);
        hipLaunchKernelGGL(k_set_bg, dim3(1), dim3(1), 0, P->stream, sc, (float)n / (float)nbag);
      }
      fill_panel_args(PS, a);
      a.flags = SWF_LAM_VEC | (pi > 0 ? (SWF_SELECT | SWF_ALT_B2) : 0) | (bagging ? SWF_KMUP2 : 0) | (de ? SWF_SERIAL : 0);
      a.e = e64; a.b = bf; a.d = dfl; a.vb = vbf; a.xx = xxf; a.lam = Lf; a.sc = sc; a.iter = itx; a.rng = rng;
      rc = launch_sweep(PS, a);                                                    // KMUP / KMUP2, R/wgr.R:85
      if (rc != BWGR_OK) goto done;
      hipLaunchKernelGGL(k_wgr_post, dim3(pg), dim3(256), 0, P->stream, bf, dfl, bR, dR, VbR, p, pi > 0 ? 1 : 0, iv, de, df, itx, rng, ws);
      hipLaunchKernelGGL(k_dsum_stage1, dim3(256), dim3(256), 0, P->stream, bR, (int64_t)p, part1, 1);
      if (pk > 0) hipLaunchKernelGGL(k_wgr_vp, dim3(1), dim3(1024), 0, P->stream, hR, Vd, (int)pk, df, itx, rng, ws);   // R/wgr.R:116-119
      hipLaunchKernelGGL(k_wgr_scal, dim3(1), dim3(1024), 0, P->stream, e64, (int)nbag, (double)n * bag, p, part1, iv, df, itx, rng, ws);
      hipLaunchKernelGGL(k_wgr_L, dim3(pg), dim3(256), 0, P->stream, bR, dR, VbR, LR, B, D, VB, p, iv, accumulate, ws);
      rc = gemv_parts<double>(P, bR, &gpart, &nchunks);
      if (rc != BWGR_OK) goto done;
      hipLaunchKernelGGL(k_wgr_efinish, dim3((n + 255) / 256), dim3(256), 0, P->stream, gpart, P->ld, nchunks, n, yd, eR, ws);
      if (pk > 0) hipLaunchKernelGGL(k_uh, dim3((n + 255) / 256), dim3(256), 0, P->stream, Ud, hR, n, (int)pk, 1.0, eR, 1);   // - U %*% h
      hipLaunchKernelGGL(k_wgr_mu, dim3(1), dim3(1024), 0, P->stream, eR, n, iv, accumulate, itx, rng, ws);
      if (pk > 0 && accumulate) hipLaunchKernelGGL(k_wgr_accum_k, dim3(8), dim3(256), 0, P->stream, hR, Hk, (int)pk, ws);
      WCHK(hipGetLastError());
      if ((i & 63) == 0) WCHK(hipStreamSynchronize(P->stream));                    // bound the launch queue
    }
    hipLaunchKernelGGL(k_dsum_stage1, dim3(256), dim3(256), 0, P->stream, D, (int64_t)p, part1, 0);
    hipLaunchKernelGGL(k_wgr_final, dim3(1), dim3(1024), 0, P->stream, B, D, VB, p, (double)mc, part1, iv, ws);
    WgrScalars h; ChainScalars hc;
    WCHK(hipMemcpyAsync(&h, ws, sizeof(h), hipMemcpyDeviceToHost, P->stream));
    WCHK(hipMemcpyAsync(&hc, sc, sizeof(hc), hipMemcpyDeviceToHost, P->stream));
    WCHK(hipStreamSynchronize(P->stream));
    if (hc.error) { rc = sweep_error(hc.error, "wgr"); goto done; }
    const double B0 = h.B0 / mc;
    rc = gemv_parts<double>(P, B, &gpart, &nchunks);                               // HAT = B0 + gen0 %*% B, R/wgr.R:152
    if (rc != BWGR_OK) goto done;
    hipLaunchKernelGGL(k_hat64_finish, dim3((n + 255) / 256), dim3(256), 0, P->stream, gpart, P->ld, nchunks, n, B0, hatd);
    if (pk > 0) {                                                                  // poly = U0 %*% H; HAT += poly, R/wgr.R:148-150
      hipLaunchKernelGGL(k_uh, dim3((n + 255) / 256), dim3(256), 0, P->stream, Ud, Hk, n, (int)pk, 1.0 / (double)mc, uhd, 0);
      hipLaunchKernelGGL(k_add_vec, dim3((n + 255) / 256), dim3(256), 0, P->stream, hatd, uhd, n);
    }
    WCHK(hipGetLastError());
    if (mu) *mu = B0;
    if (Ve) *Ve = h.VE / mc;
    if (cxx) *cxx = h.cxx * bag;                                                   // mean(xx), xx = crossprod * bag
    if (b) WCHK(d2h(P->stream, b, B, pd));
    if (d) WCHK(d2h(P->stream, d, D, pd));
    if (Vb) { if (iv) WCHK(d2h(P->stream, Vb, VB, pd)); else Vb[0] = h.VA / mc; }
    if (hat) WCHK(d2h(P->stream, hat, hatd, sizeof(double) * n));
    if (pk > 0 && u) WCHK(d2h(P->stream, u, uhd, sizeof(double) * n));
    if (pk > 0 && Vk) *Vk = h.VP / mc;
  }
done:
#undef WCHK
  (void)hipStreamSynchronize(P->stream);
  if (gpart) hipFree(gpart);
  cleanup();
  if (PU) bwgr_panel_destroy(PU);
  if (PB) bwgr_panel_destroy(PB);
  if (use_d) hipFree(use_d);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU inside the library (row e): bwgr_group_* -- one marker shard per device of this process, the residual replicated,
// ONE host thread, one stream per device, RCCL all-reduces of the residual delta (n fp64) on those streams at the exchange
// rounds.  This is what an R process (the reference's only host, R/wgr.R:2) needs in order to use more than one GPU through a
// .Call; bwgr_amd/dist.py remains the torchrun driver of the benchmark (one process per GPU).  For G > 1 this is the
// partitioned sampler of DESIGN.md section 8 (statistical parity); G = 1 is the plain exact chain.
// RCCL is loaded with dlopen on first use, so single-GPU users never touch it.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
namespace {
typedef struct ncclComm *bwgr_ncclComm_t;
struct RcclApi {
  void *h = nullptr;
  int (*CommInitAll)(bwgr_ncclComm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(bwgr_ncclComm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, bwgr_ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
  if (g_rccl.h) return BWGR_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(BWGR_EHIP, "group: cannot load librccl.so (%s)", dlerror());
  RcclApi a; a.h = h;
  a.CommInitAll = (int (*)(bwgr_ncclComm_t *, int, const int *))dlsym(h, "ncclCommInitAll");
  a.CommDestroy = (int (*)(bwgr_ncclComm_t))dlsym(h, "ncclCommDestroy");
  a.AllReduce = (int (*)(const void *, void *, size_t, int, int, bwgr_ncclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
  a.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
  a.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
  a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  if (!a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd) return fail(BWGR_EHIP, "group: librccl.so lacks an expected symbol");
  g_rccl = a;
  return BWGR_OK;
}
constexpr int BWGR_NCCL_FLOAT64 = 8, BWGR_NCCL_SUM = 0;   // ncclFloat64, ncclSum (rccl.h)
}  // namespace

struct bwgr_group {
  int G = 0;
  int model = 0;
  int64_t n = 0, p = 0;
  int block = 0, bps = 1, rounds = 1;
  std::vector<int> dev;
  std::vector<int64_t> lo, hi;
  std::vector<bwgr_panel *> P;
  std::vector<bwgr_chain *> C;
  std::vector<double *> delta, sums;
  std::vector<bwgr_ncclComm_t> comm;
  bool use_comm = false;
  bool centred = true;        // every shard's columns are centred (bwgr_panel_centred): what makes G > 1 statistically sound
  float MSx_total = 0;
  // several shards on ONE device (every entry of `devices` equal): the shards' sweeps run side by side on their own streams, each on its own
  // compute units, and an exchange round is a sum kernel between events -- no RCCL.  One exact chain is a latency-bound pipeline that fills a
  // third of the chip; the partitioned sampler's shards fill the rest.
  bool same_dev = false;
  std::vector<hipStream_t> streams;      // owned
  std::vector<hipEvent_t> ev_sweep;      // [g]: shard g's round_sweep (or sums) is enqueued up to here
  hipEvent_t ev_sum[2] = {nullptr, nullptr};
  double *total[2] = {nullptr, nullptr}; // the summed residual deltas of a round, by round parity (ld doubles each); total_sums likewise (2 doubles)
  double *total_sums[2] = {nullptr, nullptr};
  uint64_t round_no = 0;
};

extern "C" int bwgr_group_destroy(bwgr_group *Gp) {
  if (!Gp) return BWGR_OK;
  for (size_t g = 0; g < Gp->C.size(); ++g) if (Gp->C[g]) bwgr_chain_destroy(Gp->C[g]);
  for (size_t g = 0; g < Gp->P.size(); ++g) {
    if (g < Gp->dev.size()) (void)hipSetDevice(Gp->dev[g]);
    if (g < Gp->delta.size()) hipFree(Gp->delta[g]);
    if (g < Gp->sums.size()) hipFree(Gp->sums[g]);
    if (Gp->P[g]) bwgr_panel_destroy(Gp->P[g]);
  }
  if (Gp->use_comm) for (bwgr_ncclComm_t c : Gp->comm) if (c) g_rccl.CommDestroy(c);
  if (Gp->same_dev) {
    for (hipEvent_t e : Gp->ev_sweep) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) { if (Gp->ev_sum[k]) (void)hipEventDestroy(Gp->ev_sum[k]); hipFree(Gp->total[k]); hipFree(Gp->total_sums[k]); }
    for (hipStream_t q : Gp->streams) if (q) (void)hipStreamDestroy(q);
  }
  delete Gp;
  return BWGR_OK;
}

// X: HOST matrix, column-major n x p (ldx >= n), any bwgr_xtype; y: n host floats.  Device g of `devices` stages the
// block-aligned column shard [lo_g, hi_g) (as bwgr_amd/dist.py::shard_bounds) and runs the chain of that shard.
// markers_per_sync: markers swept per device between two residual all-reduces (0: 131072 / ndev, the benchmark's default; 131072 for shards
// side by side on one device).
// ---- are a panel's columns centred?  The marker-sharded partitioned sampler is statistically sound only then (DESIGN.md section 8:
// uncentred genotypes are all collinear through the mean direction, every shard corrects the same stale residual mean and the summed
// corrections overshoot; on centred columns 2 / 4 / 8 shards follow the exact chain: tools/centred_shard_probe.py).  From the panel's
// own statistics: mean_j^2 = (xx_j - (n - 1) vx_j) / n; a column counts as centred when |mean_j| <= 1e-3 sd_j. ----
namespace {
__global__ void k_uncentred(const float *xx, const float *vx, int64_t p, double n, int *flag) {
  int any = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < p; j += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)vx[j], m2 = fmax(0.0, ((double)xx[j] - (n - 1.0) * v) / n);
    any |= (m2 > 1e-6 * v + 1e-30);
  }
  if (any) *flag = 1;
}
}  // namespace
// Sweep the IMPLICITLY centred columns x_j - mean(x_j) of an int8 panel from now on (on != 0) or the raw columns again (on == 0): nothing is
// converted or copied -- the genotypes stay int8, the streamers, Gram arrays and MFMA tiles stay those of the raw columns, and the sequencer of
// k_sweep3 carries the scalar terms (sweep3.hip.h, "implicitly centred sweeps").  What changes for the callers: bwgr_panel_stats returns the centred
// columns' squared norms, the fused chains (bwgr_chain_*, bwgr_bayes, bwgr_group_*) run the reference's sweep on the centred columns
// (src/Rcpp20260726ai.cpp:668-682 with X_j - mean_j for X_j; selection models on k_sweep3 only), hat = X_c B + mu, and bwgr_panel_centred answers 1 -- which is
// what makes the marker-sharded sampler of several devices sound without a float copy of the panel.  Refused while chains are alive.
extern "C" int bwgr_panel_set_centred(bwgr_panel *P, int on) {
  if (!P) return fail(BWGR_EINVAL, "null panel");
  if (P->parent) return fail(BWGR_EINVAL, "panel_set_centred: set it on the root panel (clones follow it)");
  if (P->nchains > 0) return fail(BWGR_EINVAL, "panel_set_centred: %d chains are alive on this panel", P->nchains);
  HIPCHK(hipSetDevice(P->device));
  if (!on) { P->cen = false; return BWGR_OK; }
  if (P->is_f32) return fail(BWGR_EINVAL, "panel_set_centred: float panels are swept as given (centre the columns before the upload)");
  if (!use_sweep3(P, SWF_SELECT)) return fail(BWGR_EINVAL, "panel_set_centred: this panel has no k_sweep3 (geometry or Gram range): the implicitly centred sweep is k_sweep3's");
  if (!P->csum) {
    HIPCHK(hipMalloc(&P->csum, sizeof(int32_t) * (size_t)P->p));
    HIPCHK(hipMalloc(&P->xxc, sizeof(float) * (size_t)P->p));
    const int wpb = 4;
    hipLaunchKernelGGL(k_colsum_i8, dim3((unsigned)((P->p + wpb - 1) / wpb)), dim3(64 * wpb), 0, P->stream, (const int8_t *)P->X, P->R, (int)P->n, (int)P->p, P->csum, P->xxc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(P->stream));
  }
  P->cen = true;
  return BWGR_OK;
}

extern "C" int bwgr_panel_centred(bwgr_panel *P, int *centred) {
  if (!P || !centred) return fail(BWGR_EINVAL, "null pointer");
  if (panel_cen(P)) { *centred = 1; return BWGR_OK; }   // implicitly centred: exactly
  HIPCHK(hipSetDevice(P->device));
  int *flag = nullptr, h = 0;
  HIPCHK(hipMalloc(&flag, sizeof(int)));
  HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), P->stream));
  hipLaunchKernelGGL(k_uncentred, dim3(256), dim3(256), 0, P->stream, P->xx, P->vx, P->p, (double)P->n, flag);
  HIPCHK(d2h(P->stream, &h, flag, sizeof(int)));
  HIPCHK(hipFree(flag));
  *centred = h ? 0 : 1;
  return BWGR_OK;
}
extern "C" int bwgr_group_sound(const bwgr_group *Gp, int *sound);

static int group_create_impl(bwgr_group **out, int ndev, const int *devices, const void *X, int xtype, int64_t n, int64_t p,
                             int64_t ldx, int block, const float *y, int model, float it, float bi, float pi, float df, float R2,
                             uint64_t seed, int rng_mode, int64_t markers_per_sync, int centre, int memloc = BWGR_HOST) {
  if (!out || !devices || !X || !y) return fail(BWGR_EINVAL, "group_create: null pointer");
  if (memloc != BWGR_HOST && memloc != BWGR_DEVICE) return fail(BWGR_EINVAL, "group_create: bad memloc %d", memloc);
  if (memloc == BWGR_DEVICE && ndev > 1 && !std::all_of(devices, devices + ndev, [&](int d) { return d == devices[0]; }))
    return fail(BWGR_EINVAL, "group_create: a device-resident X serves shards of that one device only");
  if (centre && xtype != BWGR_X_I8) return fail(BWGR_EINVAL, "group_create_centred: implicit centring is for int8 genotypes (centre float columns before the call)");
  *out = nullptr;
  if (ndev < 1 || ndev > 64) return fail(BWGR_EINVAL, "group_create: ndev = %d", ndev);
  for (int g = 1; g < ndev; ++g) for (int h = 0; h < g; ++h)
    if (devices[g] == devices[h] && !std::all_of(devices, devices + ndev, [&](int d) { return d == devices[0]; }))
      return fail(BWGR_EINVAL, "group_create: a device may appear once, or every shard sits on the same device (shards side by side on one GPU)");
  if (xtype != BWGR_X_I8 && xtype != BWGR_X_F32 && xtype != BWGR_X_F64) return fail(BWGR_EINVAL, "group_create: bad xtype %d", xtype);
  const int mmax = (xtype == BWGR_X_I8) ? SW_MAXM : 64;
  const int m = block > 0 ? block : mmax;
  const int64_t nblk = (p + m - 1) / m, per = (nblk + ndev - 1) / ndev;
  if ((int64_t)(ndev - 1) * per * m >= p) return fail(BWGR_EINVAL, "group_create: p = %lld has only %lld blocks of %d markers: too few for %d devices", (long long)p, (long long)nblk, m, ndev);
  bwgr_group *Gp = new bwgr_group();
  Gp->G = ndev; Gp->model = model; Gp->n = n; Gp->p = p; Gp->block = m;
  Gp->dev.assign(devices, devices + ndev);
  Gp->same_dev = ndev > 1 && std::all_of(devices, devices + ndev, [&](int d) { return d == devices[0]; });
  Gp->P.assign(ndev, nullptr); Gp->C.assign(ndev, nullptr); Gp->delta.assign(ndev, nullptr); Gp->sums.assign(ndev, nullptr);
  auto bail = [&](int code) { bwgr_group_destroy(Gp); return code; };
  const size_t esz = (xtype == BWGR_X_I8) ? 1 : (xtype == BWGR_X_F32 ? 4 : 8);
  double msx = 0.0;
  for (int g = 0; g < ndev; ++g) {
    const int64_t lo = std::min<int64_t>(p, (int64_t)g * per * m), hi = std::min<int64_t>(p, (int64_t)(g + 1) * per * m);
    Gp->lo.push_back(lo); Gp->hi.push_back(hi);
    int rc = bwgr_panel_create(&Gp->P[g], reinterpret_cast<const unsigned char *>(X) + (size_t)lo * (size_t)ldx * esz, xtype, memloc, n, hi - lo, ldx, devices[g], m, 0);
    if (rc != BWGR_OK) return bail(rc);
    if (Gp->same_dev) {   // shards of one device: each on a stream of its own; from three shards on, 256-row streamers (K3 + 1 units a shard instead of 2 K3 + 2)
      hipStream_t q = nullptr;
      if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return bail(fail(BWGR_EHIP, "group_create: hipStreamCreate failed"));
      Gp->streams.push_back(q);
      Gp->P[g]->stream = q;
      if (ndev > 2 && !getenv("BWGR_SOLO3")) Gp->P[g]->solo3 = false;   // (an explicit BWGR_SOLO3 decides otherwise: experiments)
    }
    if (centre) { rc = bwgr_panel_set_centred(Gp->P[g], 1); if (rc != BWGR_OK) return bail(rc); }   // the shard's own column means (rows are not sharded)
    msx += (double)Gp->P[g]->MSx;
    int cen = 1;
    rc = bwgr_panel_centred(Gp->P[g], &cen);
    if (rc != BWGR_OK) return bail(rc);
    if (!cen) Gp->centred = false;
  }
  Gp->MSx_total = (float)msx;
  if (ndev > 1 && !Gp->centred) {
    const char *ok = getenv("BWGR_GROUP_ALLOW_UNCENTRED");
    if (!(ok && ok[0] == '1'))
      return bail(fail(BWGR_EINVAL, "group_create: the columns of X are not centred, and on uncentred columns the marker-sharded sampler of %d devices is "
                                    "statistically unsound (every shard corrects the same stale residual mean: DESIGN.md section 8).  Pass centred columns "
                                    "(x_j - mean(x_j), float: the posterior of b and hat is the same under the sampler's flat intercept prior), use one "
                                    "device, or set BWGR_GROUP_ALLOW_UNCENTRED=1 to run it knowingly", ndev));
  }
  for (int g = 0; g < ndev; ++g) {
    int rc = bwgr_chain_create_sharded(&Gp->C[g], Gp->P[g], model, y, BWGR_HOST, it, bi, pi, df, R2, seed, rng_mode, Gp->lo[g], p, Gp->MSx_total, nullptr);
    if (rc != BWGR_OK) return bail(rc);
    if (hipSetDevice(devices[g]) != hipSuccess || hipMalloc(&Gp->delta[g], sizeof(double) * (size_t)Gp->P[g]->ld) != hipSuccess ||
        hipMalloc(&Gp->sums[g], sizeof(double) * 2) != hipSuccess) return bail(fail(BWGR_ENOMEM, "group_create: device allocation failed"));
    if (Gp->P[g]->ld != Gp->P[0]->ld) return bail(fail(BWGR_EINVAL, "group_create: shards disagree on the padded row count"));
  }
  // (shards side by side exchange through one kernel on the same card, not a ring over xGMI, but every exchange is a launch boundary for all of
  // them: 131072 markers per shard between two exchanges there)
  const int64_t mps = markers_per_sync > 0 ? markers_per_sync : std::max<int64_t>(m, Gp->same_dev ? 131072 : 131072 / ndev);
  Gp->bps = (int)std::max<int64_t>(1, mps / m);
  int64_t nbmax = 0;
  for (int g = 0; g < ndev; ++g) nbmax = std::max<int64_t>(nbmax, Gp->P[g]->nblocks);
  Gp->rounds = (int)((nbmax + Gp->bps - 1) / Gp->bps);
  const char *fc = getenv("BWGR_GROUP_FORCE_COMM");   // (tests: exercise the RCCL path with a single device)
  Gp->use_comm = (ndev > 1 && !Gp->same_dev) || (fc && fc[0] == '1' && !Gp->same_dev);
  if (Gp->same_dev) {
    if (hipSetDevice(devices[0]) != hipSuccess) return bail(fail(BWGR_EHIP, "group_create: hipSetDevice failed"));
    Gp->ev_sweep.assign(ndev, nullptr);
    for (int g = 0; g < ndev; ++g) if (hipEventCreateWithFlags(&Gp->ev_sweep[g], hipEventDisableTiming) != hipSuccess) return bail(fail(BWGR_EHIP, "group_create: hipEventCreate failed"));
    for (int k = 0; k < 2; ++k) {
      if (hipEventCreateWithFlags(&Gp->ev_sum[k], hipEventDisableTiming) != hipSuccess || hipMalloc(&Gp->total[k], sizeof(double) * (size_t)Gp->P[0]->ld) != hipSuccess ||
          hipMalloc(&Gp->total_sums[k], sizeof(double) * 2) != hipSuccess) return bail(fail(BWGR_ENOMEM, "group_create: device allocation failed"));
    }
  }
  if (Gp->use_comm) {
    int rc = rccl_load();
    if (rc != BWGR_OK) return bail(rc);
    Gp->comm.assign(ndev, nullptr);
    const int nr = g_rccl.CommInitAll(Gp->comm.data(), ndev, devices);
    if (nr != 0) return bail(fail(BWGR_EHIP, "group_create: ncclCommInitAll failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?"));
  }
  *out = Gp;
  return BWGR_OK;
}

extern "C" int bwgr_group_create(bwgr_group **out, int ndev, const int *devices, const void *X, int xtype, int64_t n, int64_t p,
                                 int64_t ldx, int block, const float *y, int model, float it, float bi, float pi, float df, float R2,
                                 uint64_t seed, int rng_mode, int64_t markers_per_sync) {
  return group_create_impl(out, ndev, devices, X, xtype, n, p, ldx, block, y, model, it, bi, pi, df, R2, seed, rng_mode, markers_per_sync, 0);
}
// ... on the implicitly centred columns of an int8 matrix (bwgr_panel_set_centred on every shard): sound with several devices, int8 in HBM
extern "C" int bwgr_group_create_centred(bwgr_group **out, int ndev, const int *devices, const void *X, int xtype, int64_t n, int64_t p,
                                         int64_t ldx, int block, const float *y, int model, float it, float bi, float pi, float df, float R2,
                                         uint64_t seed, int rng_mode, int64_t markers_per_sync, int memloc) {
  return group_create_impl(out, ndev, devices, X, xtype, n, p, ldx, block, y, model, it, bi, pi, df, R2, seed, rng_mode, markers_per_sync, 1, memloc);
}
static int group_allreduce(bwgr_group *Gp, std::vector<double *> &buf, size_t count) {
  int nr = g_rccl.GroupStart();
  for (int g = 0; g < Gp->G && nr == 0; ++g)
    nr = g_rccl.AllReduce(buf[g], buf[g], count, BWGR_NCCL_FLOAT64, BWGR_NCCL_SUM, Gp->comm[g], Gp->P[g]->stream);
  const int ne = g_rccl.GroupEnd();
  if (nr == 0) nr = ne;
  if (nr != 0) return fail(BWGR_EHIP, "group: ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?");
  return BWGR_OK;
}

extern "C" int bwgr_group_sound(const bwgr_group *Gp, int *sound) {
  if (!Gp || !sound) return fail(BWGR_EINVAL, "null pointer");
  *sound = (Gp->G == 1 || Gp->centred) ? 1 : 0;   // one device: the exact chain; more: sound on centred columns only
  return BWGR_OK;
}

namespace {
struct SumPtrs { const double *src[64]; };
__global__ void k_group_sum(const SumPtrs s, int G, double *out, int64_t count) {   // out = sum over the shards, in shard order (the same bits on every run)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    double t = 0.0;
    for (int g = 0; g < G; ++g) t += s.src[g][i];
    out[i] = t;
  }
}
}  // namespace
// one exchange among the shards of one device: every shard's stream has produced buf[g]; stream 0 sums them into `total` (by round parity, so
// that the readers of the previous round are never overwritten: a buffer's next writer waits for events that every reader's stream records later)
static int group_local_sum(bwgr_group *Gp, std::vector<double *> &buf, size_t count, double *const total[2], double **out) {
  const int k = (int)(Gp->round_no++ & 1u);
  SumPtrs sp;
  for (int g = 0; g < Gp->G; ++g) { sp.src[g] = buf[g]; HIPCHK(hipEventRecord(Gp->ev_sweep[g], Gp->streams[g])); }
  for (int g = 1; g < Gp->G; ++g) HIPCHK(hipStreamWaitEvent(Gp->streams[0], Gp->ev_sweep[g], 0));
  hipLaunchKernelGGL(k_group_sum, dim3((unsigned)std::min<size_t>(64, (count + 255) / 256)), dim3(256), 0, Gp->streams[0], sp, Gp->G, total[k], (int64_t)count);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(Gp->ev_sum[k], Gp->streams[0]));
  for (int g = 1; g < Gp->G; ++g) HIPCHK(hipStreamWaitEvent(Gp->streams[g], Gp->ev_sum[k], 0));
  *out = total[k];
  return BWGR_OK;
}
static int group_run_same_device(bwgr_group *Gp, int iters) {
  HIPCHK(hipSetDevice(Gp->dev[0]));
  for (int k = 0; k < iters; ++k) {
    for (int r = 0; r < Gp->rounds; ++r) {
      for (int g = 0; g < Gp->G; ++g) {
        const int nb = (int)Gp->P[g]->nblocks;
        const int lo = std::min(nb, r * Gp->bps), hi = std::min(nb, (r + 1) * Gp->bps);
        CHK(bwgr_chain_round_sweep(Gp->C[g], lo, hi, Gp->delta[g]));
      }
      double *tot = nullptr;
      CHK(group_local_sum(Gp, Gp->delta, (size_t)Gp->P[0]->ld, Gp->total, &tot));
      for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_round_apply(Gp->C[g], tot));
    }
    for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_get_sums_dev(Gp->C[g], Gp->sums[g]));
    double *tot2 = nullptr;
    CHK(group_local_sum(Gp, Gp->sums, 2, Gp->total_sums, &tot2));
    for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_end_iteration_dev(Gp->C[g], tot2));
  }
  return BWGR_OK;
}

extern "C" int bwgr_group_run(bwgr_group *Gp, int iters) {
  if (!Gp) return fail(BWGR_EINVAL, "null group");
  if (iters < 0) return fail(BWGR_EINVAL, "group_run: iters < 0");
  if (Gp->same_dev) return group_run_same_device(Gp, iters);
  if (!Gp->use_comm) return bwgr_chain_run(Gp->C[0], iters);   // one device: the plain exact chain
  for (int k = 0; k < iters; ++k) {
    for (int r = 0; r < Gp->rounds; ++r) {
      for (int g = 0; g < Gp->G; ++g) {
        const int nb = (int)Gp->P[g]->nblocks;
        const int lo = std::min(nb, r * Gp->bps), hi = std::min(nb, (r + 1) * Gp->bps);   // (lo == hi: a device that has run out of blocks still takes part)
        CHK(bwgr_chain_round_sweep(Gp->C[g], lo, hi, Gp->delta[g]));
      }
      CHK(group_allreduce(Gp, Gp->delta, (size_t)Gp->P[0]->ld));
      for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_round_apply(Gp->C[g], Gp->delta[g]));
    }
    for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_get_sums_dev(Gp->C[g], Gp->sums[g]));
    CHK(group_allreduce(Gp, Gp->sums, 2));
    for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_end_iteration_dev(Gp->C[g], Gp->sums[g]));
  }
  return BWGR_OK;
}

extern "C" int bwgr_group_sync(bwgr_group *Gp) {
  if (!Gp) return fail(BWGR_EINVAL, "null group");
  for (int g = 0; g < Gp->G; ++g) CHK(bwgr_chain_sync(Gp->C[g]));
  return BWGR_OK;
}

extern "C" int bwgr_group_info(const bwgr_group *Gp, int64_t info[4]) {
  if (!Gp || !info) return fail(BWGR_EINVAL, "null pointer");
  info[0] = Gp->G; info[1] = Gp->rounds; info[2] = (int64_t)Gp->bps * Gp->block; info[3] = Gp->use_comm ? 1 : 0;
  return BWGR_OK;
}

// the reference's return list over the whole panel (host outputs; any may be NULL): b, d, pval: p floats; vb: p floats for the
// per-marker-variance models, else 1; hat: n floats
extern "C" int bwgr_group_result(bwgr_group *Gp, float *mu, float *b, float *d, float *hat, float *vb, float *ve, float *h2,
                                 float *MSx, float *pi_out, float *pval) {
  if (!Gp) return fail(BWGR_EINVAL, "null group");
  const bool per = per_marker_vb(Gp->model);
  std::vector<float> hg(hat ? (size_t)Gp->n : 0), vbl;
  float mu0 = 0, ve0 = 0, h20 = 0, msx0 = 0, pi0 = 0, vbs = 0;
  double vg = 0.0;
  if (hat) for (int64_t i = 0; i < Gp->n; ++i) hat[i] = 0.0f;
  for (int g = 0; g < Gp->G; ++g) {
    const int64_t lo = Gp->lo[g], pg = Gp->hi[g] - lo;
    float mug, veg, h2g, msxg, pig;
    vbl.assign(per ? (size_t)pg : 1, 0.0f);
    CHK(bwgr_chain_result(Gp->C[g], &mug, b ? b + lo : nullptr, d ? d + lo : nullptr, hat ? hg.data() : nullptr, vbl.data(), &veg, &h2g,
                          &msxg, &pig, pval ? pval + lo : nullptr));
    if (g == 0) { mu0 = mug; ve0 = veg; h20 = h2g; msx0 = msxg; pi0 = pig; vbs = vbl[0]; }
    if (per) { for (int64_t j = 0; j < pg; ++j) { vg += (double)vbl[(size_t)j]; if (vb) vb[lo + j] = vbl[(size_t)j]; } }
    if (hat) for (int64_t i = 0; i < Gp->n; ++i) hat[i] += hg[(size_t)i] - mug;   // X_g B_g
  }
  if (hat) for (int64_t i = 0; i < Gp->n; ++i) hat[i] += mu0;
  if (!per && vb) vb[0] = vbs;
  if (mu) *mu = mu0;
  if (ve) *ve = ve0;
  if (h2) *h2 = per ? (float)(vg / (vg + (double)ve0)) : h20;   // (the common-variance models form vg from MSx over all shards already)
  if (MSx) *MSx = msx0;
  if (pi_out) *pi_out = pi0;
  return BWGR_OK;
}

// ------------------------------------------------------------------------------------------------
// f4: EM / Gauss-Seidel family (emRR, emBA, emBB, emBC, emBCpi, emDE, emBL, emEN, emML), src/Rcpp20260726ai.cpp:80-521, :1502-1545
//
// Deterministic coordinate updates -- b_j = (X_j.e + xx_j b_j)/(xx_j + lambda_j), i.e. the affine sweep with the variates
// switched off, or the member's soft-selection / soft-threshold update (lane_em) -- in a marker order that the reference re-shuffles before every sweep (std::shuffle with
// std::mt19937(i), :103 ...).  The exact blocked sweep needs the Gram blocks of consecutive markers, so every sweep
//   (1) shuffles the order on the host with the very library call the reference makes,
//   (2) gathers the columns of the resident panel into a scratch panel in that order (one pass over X),
//   (3) rebuilds the scratch panel's diagonal and distance-1 Gram blocks,
//   (4) runs the affine sweep kernel on it (b, xx, lambda gathered; b scattered back),
//   (5) runs the model's tail (variance components, lambda, intercept) in one workgroup.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void k_permute_cols(const uint4 *__restrict__ X, uint4 *__restrict__ Xp, const int32_t *__restrict__ order,
                               int64_t p, int K, int cps) {
  // slab-major layout: (slab s, marker j) is one segment of R elements = cps 16-byte chunks
  const int64_t total = (int64_t)K * p * cps;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t seg = c / cps;
    const int off = (int)(c - seg * cps);
    const int64_t sl = seg / p, jj = seg - sl * p;
    Xp[c] = X[(sl * p + order[jj]) * cps + off];
  }
}

struct EmState {
  float mu, ve, vb, Lmb, cnv, Sb, Se, Rho, cxx, df, vy, MSx;
  float va, Sa, Pi, Pi0, PriorPi, sumvx, R2, alpha, Lmb1, Lmb2, Sy, trAC22;
};

// per-marker inputs of one sweep, in sweep order: b, xx, lambda
__global__ void k_em_stage(const int32_t *__restrict__ order, int64_t p, int model, int weighted, const float *__restrict__ b,
                           const float *__restrict__ xx, const float *__restrict__ lam, const float *__restrict__ D,
                           const EmState *__restrict__ st, float *__restrict__ bq, float *__restrict__ xxq, float *__restrict__ lamq) {
  const float Lmb = st->Lmb, Lmb2 = st->Lmb2;
  for (int64_t jj = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; jj < p; jj += (int64_t)gridDim.x * blockDim.x) {
    const int j = order[jj];
    bq[jj] = b[j]; xxq[jj] = xx[j];
    float l;
    if (model == BWGR_EM_BA || model == BWGR_EM_DE || model == BWGR_EM_BB) l = lam[j];
    else if (model == BWGR_EM_BL || model == BWGR_EM_EN) l = Lmb2;                   // denominators Lmb2 + xx, :382, :434
    else if (model == BWGR_EM_LASSO) l = 0.0f;                                       // denominator xx, :1480
    else if (weighted) l = Lmb / D[j];                                               // :496
    else l = Lmb;
    lamq[jj] = l;
  }
}
__global__ void k_em_unstage(const int32_t *__restrict__ order, int64_t p, const float *__restrict__ bq, float *__restrict__ b,
                             const float *__restrict__ dq, float *__restrict__ d) {
  for (int64_t jj = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; jj < p; jj += (int64_t)gridDim.x * blockDim.x) {
    const int j = order[jj];
    b[j] = bq[jj];
    if (d) d[j] = dq[jj];
  }
}
__global__ void k_em_fix_xx(float *xx, int64_t p) {                                   // if(xx[k]==0) xx[k]=0.1f, :261
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < p; j += (int64_t)gridDim.x * blockDim.x) if (xx[j] == 0.0f) xx[j] = 0.1f;
}
__global__ void k_em_init(const float *__restrict__ y, double *__restrict__ e, int n, int64_t ld, EmState *st) {
  // mu = y.mean(); e = y.array()-mu  (float), :98-99; padding rows of e stay 0
  __shared__ double sh[1024];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)y[i];
  sh[threadIdx.x] = s; __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  const float mu = (float)(sh[0] / (double)n);
  for (int64_t i = threadIdx.x; i < ld; i += 1024) e[i] = i < n ? (double)(y[i] - mu) : 0.0;
  if (threadIdx.x == 0) st->mu = mu;
}

struct EmTailArgs {
  int model, n, conv; int64_t p;
  double *e; const float *y; const float *b; const float *bc; const float *d; float *lam; float *vbv; const float *xx;
  EmState *st; ChainScalars *sc;
};

__device__ double em_block_sum(double v, double *sh) {
  __syncthreads();
  sh[threadIdx.x] = v; __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  return sh[0];
}

// what follows a sweep, one workgroup: the model's variance components and lambda, then eM = e.mean(); mu += eM; e -= eM;
// last, the scalars the next sweep's kernels read (C, Pi0, Lmb1) are refreshed in the ChainScalars block
__global__ __launch_bounds__(1024) void k_em_tail(const EmTailArgs a) {
  __shared__ double sh[1024];
  EmState st = *a.st;
  const int n = a.n; const int64_t p = a.p; const int tid = threadIdx.x; const int model = a.model;
  const float df = st.df;
  float b2n = 0, e2n = 0, dmean = 0;
  if (model == BWGR_EM_RR || model == BWGR_EM_BC || model == BWGR_EM_BCPI || model == BWGR_EM_EN) {
    double s = 0; for (int64_t j = tid; j < p; j += 1024) s = fma((double)a.b[j], (double)a.b[j], s);
    b2n = (float)em_block_sum(s, sh);                                                // b.squaredNorm()
  }
  if (model == BWGR_EM_BC || model == BWGR_EM_BCPI) {
    double s = 0; for (int64_t j = tid; j < p; j += 1024) s += (double)a.d[j];
    dmean = (float)(em_block_sum(s, sh) / (double)p);                                // d.mean()
  }
  if (model == BWGR_EM_BA || model == BWGR_EM_BB || model == BWGR_EM_RR || model == BWGR_EM_BC || model == BWGR_EM_BCPI) {
    double s = 0; for (int i = tid; i < n; i += 1024) s = fma(a.e[i], a.e[i], s);
    e2n = (float)em_block_sum(s, sh);                                                // e.squaredNorm() (before centring)
  }
  if (model == BWGR_EM_BA || model == BWGR_EM_BB) {
    st.ve = (e2n + st.Se) / ((float)n + df);                                         // :113, :170
    for (int64_t j = tid; j < p; j += 1024) {
      const float bj = a.b[j];
      const float vbj = (st.Sb + bj * bj) / (df + 1);                                // :110, :167
      a.vbv[j] = vbj;
      a.lam[j] = st.ve * (1.0f / vbj);                                               // Lmb = ve * vb.cwiseInverse(), :114, :171
    }
  } else if (model == BWGR_EM_RR) {
    st.vb = (b2n + st.Sb) / ((float)p + df);                                         // :338
    st.ve = (e2n + st.Se) / ((float)n + df);                                         // :339
    st.Lmb = sqrtf(st.Rho * st.ve / st.vb);                                          // :340
  } else if (model == BWGR_EM_BC) {
    st.ve = (e2n + st.Se) / ((float)n + df);                                         // :229
    st.va = (b2n + st.Sa) / ((float)p + df) / (dmean - st.Pi);                       // :230
    st.Lmb = st.ve / st.va;                                                          // :231
  } else if (model == BWGR_EM_BCPI) {
    st.Pi = ((1.0f - dmean) * (float)p + st.PriorPi * df) / ((float)p + df);         // :1533
    st.Pi0 = (1.0f - st.Pi) / st.Pi;                                                 // :1534
    st.MSx = st.sumvx * st.Pi * (1.0f - st.Pi);                                      // :1535
    st.Sa = st.R2 * (df + 2) * st.vy / st.MSx;                                       // :1536
    st.ve = (e2n + st.Se) / ((float)n + df);                                         // :1538
    st.va = (b2n + st.Sa) / ((float)p + df) / (dmean - st.Pi);                       // :1539
    st.Lmb = st.ve / st.va;                                                          // :1540
  }
  {
    double s = 0; for (int i = tid; i < n; i += 1024) s += a.e[i];
    const float eM = (float)(em_block_sum(s, sh) / (double)n);                       // :115-117
    st.mu += eM;
    for (int i = tid; i < n; i += 1024) a.e[i] = a.e[i] - (double)eM;
    __syncthreads();
  }
  if (model == BWGR_EM_DE || model == BWGR_EM_EN) {
    double s = 0; for (int i = tid; i < n; i += 1024) s = fma(a.e[i], (double)a.y[i], s);
    st.ve = (float)em_block_sum(s, sh) / (float)(n - 1);                             // Ve = e.dot(y)/(n-1), :289, :445
  }
  if (model == BWGR_EM_DE) {
    for (int64_t j = tid; j < p; j += 1024) {
      const float bj = a.b[j];
      const float vbj = bj * bj + st.ve / (a.xx[j] + a.lam[j] + 0.0001f);            // :290
      a.vbv[j] = vbj;
      a.lam[j] = sqrtf(st.cxx * st.ve / vbj);                                        // :292
    }
  } else if (model == BWGR_EM_EN) {
    st.va = (b2n + st.trAC22 * st.ve) / (float)p;                                    // :446
    st.Lmb = st.ve / st.va;                                                          // :447
    st.Lmb1 = 0.5f * st.Lmb * st.alpha * st.Sy;                                      // :448
    st.Lmb2 = st.Lmb * (1 - st.alpha);                                               // :449
  } else if (model == BWGR_EM_ML) {
    double s1 = 0, s2 = 0;
    for (int i = tid; i < n; i += 1024) {
      const float ym = a.y[i] - st.mu;
      s1 = fma((double)ym, a.e[i], s1);                                              // :505
      s2 = fma((double)ym, (double)ym - a.e[i], s2);                                 // :506
    }
    const float d1 = (float)em_block_sum(s1, sh), d2 = (float)em_block_sum(s2, sh);
    st.ve = d1 / (float)n;
    st.vb = d2 / (float)((float)n * st.MSx);
    st.Lmb = st.ve / st.vb;                                                          // :507
  }
  if (a.conv) {
    double s = 0; for (int64_t j = tid; j < p; j += 1024) s += (double)fabsf(a.bc[j] - a.b[j]);
    st.cnv = (float)em_block_sum(s, sh);                                             // :295, :451, :509
  }
  if (tid == 0) {
    *a.st = st;
    a.sc->C = -0.5f / sqrtf(st.ve);                                                  // C of the next sweep, :157, :216, :1522
    a.sc->odds = st.Pi0;
    a.sc->lam = st.Lmb1;
  }
}

__global__ void k_em_fit_ml(const float *__restrict__ y, const double *__restrict__ e, float *__restrict__ hat, int n) {   // fit = y - e, :512
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hat[i] = (float)((double)y[i] - e[i]);
}

}  // namespace

// the marker order of sweep `upto` (0-based): identity shuffled with std::mt19937(0), (1), ..., (upto) -- the library's own
// std::shuffle, so that the oracle's restatement of it can be pinned (tests/test_em_order.py)
extern "C" int bwgr_em_order(int64_t p, int upto, int32_t *order) {
  if (!order || p < 1 || p > 0x7FFFFF00ll) return fail(BWGR_EINVAL, "em_order: bad arguments");
  std::vector<int> ord((size_t)p);
  for (int64_t j = 0; j < p; ++j) ord[(size_t)j] = (int)j;
  for (int i = 0; i <= upto; ++i) std::shuffle(ord.begin(), ord.end(), std::mt19937(i));
  for (int64_t j = 0; j < p; ++j) order[j] = ord[(size_t)j];
  return BWGR_OK;
}

extern "C" int bwgr_em(bwgr_panel *P, int model, const float *y, float df, float R2, float par, const float *D, int maxit_in,
                       float *mu, float *b, float *d, float *hat, float *vbvec, float *scal, int *iters) {
  if (P && panel_cen(P)) return fail(BWGR_EINVAL, "em: this panel sweeps implicitly centred columns (bwgr_panel_set_centred), which only the fused chains do; call bwgr_panel_set_centred(P, 0) first");
  if (!P || !y || !b || !scal) return fail(BWGR_EINVAL, "em: null pointer");
  if (model < BWGR_EM_RR || model > BWGR_EM_LASSO) return fail(BWGR_EINVAL, "em: bad model %d", model);
  if (D && model != BWGR_EM_ML) return fail(BWGR_EINVAL, "em: marker weights D belong to emML only");
  const bool soft = (model == BWGR_EM_BB || model == BWGR_EM_BC || model == BWGR_EM_BCPI);
  const bool lasso = (model == BWGR_EM_LASSO);
  const bool nonaffine = soft || lasso || model == BWGR_EM_BL || model == BWGR_EM_EN;
  if (nonaffine && P->sweep_version < 2) return fail(BWGR_EINVAL, "em: this member needs the pipelined sweep engine (k_sweep2), which this panel's geometry does not fit");
  HIPCHK(hipSetDevice(P->device));
  const int64_t p = P->p, n = P->n;
  const bool conv = (model == BWGR_EM_DE || model == BWGR_EM_ML || model == BWGR_EM_EN || lasso);
  const bool shuffled = (model != BWGR_EM_BCPI && !lasso);                            // emBCpi and lasso sweep in natural order, :1523, :1476
  const int maxit = maxit_in > 0 ? maxit_in : (conv ? 300 : 200);                     // :81, :251, :309, :401, :465
  const float tol = (model == BWGR_EM_DE) ? 10e-6f : (model == BWGR_EM_EN) ? 10e-11f : 10e-8f;   // :252, :402, :466
  hipStream_t st = P->stream;
  // scratch panel: same geometry, its own X and Gram; only the diagonal and distance-1 blocks are ever built (lag 2)
  bwgr_panel *Q = nullptr;
  if (shuffled) {
    CHK(panel_alloc(&Q, P->is_f32, n, p, P->device, P->m, P->K));
    Q->stream = st; Q->gram_maxdist = 1;
    hipFree(Q->gramx2); hipFree(Q->gramx3); hipFree(Q->xspec2); hipFree(Q->xspec3); hipFree(Q->gramp16); hipFree(Q->gramx16);
    Q->gramx2 = Q->gramx3 = nullptr; Q->xspec2 = Q->xspec3 = nullptr; Q->gramp16 = Q->gramx16 = nullptr;
  }
  std::vector<void *> owned;
  int rc = BWGR_OK;
  auto done = [&](int code) { (void)hipStreamSynchronize(st); for (void *q : owned) hipFree(q); if (Q) bwgr_panel_destroy(Q); return code; };
#define ECHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return done(fail(BWGR_EHIP, "%s failed: %s", #x, hipGetErrorString(e_))); } while (0)
  if (Q && (Q->K != P->K || Q->R != P->R || Q->m != P->m || std::min(Q->sweep_version, 2) != std::min(P->sweep_version, 2))) return done(fail(BWGR_EINVAL, "em: scratch panel geometry differs"));
  bwgr_panel *S = Q ? Q : P;                                                          // the panel the sweeps run on
  const size_t pb = sizeof(float) * (size_t)p;
  float *yd = nullptr, *bd = nullptr, *bcd = nullptr, *dd = nullptr, *lamd = nullptr, *vbd = nullptr, *xxd = nullptr, *Dd = nullptr;
  float *bq = nullptr, *xxq = nullptr, *lamq = nullptr, *dq = nullptr, *vq = nullptr, *hatd = nullptr;
  double *ed = nullptr; int32_t *ordd = nullptr; EmState *std_ = nullptr; ChainScalars *sc = nullptr;
  auto dmalloc = [&](void **q, size_t bytes) { hipError_t e_ = hipMalloc(q, bytes); if (e_ == hipSuccess) owned.push_back(*q); return e_; };
  ECHK(dmalloc((void **)&yd, sizeof(float) * n)); ECHK(dmalloc((void **)&bd, pb)); ECHK(dmalloc((void **)&bcd, pb)); ECHK(dmalloc((void **)&dd, pb));
  ECHK(dmalloc((void **)&lamd, pb)); ECHK(dmalloc((void **)&vbd, pb)); ECHK(dmalloc((void **)&xxd, pb));
  ECHK(dmalloc((void **)&bq, pb)); ECHK(dmalloc((void **)&xxq, pb)); ECHK(dmalloc((void **)&lamq, pb)); ECHK(dmalloc((void **)&dq, pb)); ECHK(dmalloc((void **)&vq, pb));
  ECHK(dmalloc((void **)&ed, sizeof(double) * P->ld)); ECHK(dmalloc((void **)&ordd, sizeof(int32_t) * p));
  ECHK(dmalloc((void **)&std_, sizeof(EmState))); ECHK(dmalloc((void **)&sc, sizeof(ChainScalars)));
  ECHK(dmalloc((void **)&hatd, sizeof(float) * n));
  if (D) { ECHK(dmalloc((void **)&Dd, pb)); ECHK(hipMemcpyAsync(Dd, D, pb, hipMemcpyHostToDevice, st)); }
  ECHK(hipMemcpyAsync(yd, y, sizeof(float) * n, hipMemcpyHostToDevice, st));
  ECHK(hipMemsetAsync(bd, 0, pb, st)); ECHK(hipMemsetAsync(dd, 0, pb, st));
  ECHK(hipMemcpyAsync(xxd, P->xx, pb, hipMemcpyDeviceToDevice, st));
  // vy = fvar(y) with the library's reduction (float result of fp64 sums, like the fused samplers' setup)
  float vy = 0;
  {
    InitArgs ia; memset(&ia, 0, sizeof(ia));
    ChainScalars h0; memset(&h0, 0, sizeof(h0));
    ECHK(hipMemcpyAsync(sc, &h0, sizeof(h0), hipMemcpyHostToDevice, st));
    ia.y = yd; ia.e = ed; ia.n = (int)n; ia.p = (int)p; ia.ld = P->ld; ia.model = BWGR_BAYESRR; ia.pi = 0; ia.df = df; ia.R2 = R2; ia.MSx = P->MSx; ia.sc = sc;
    hipLaunchKernelGGL(k_chain_init, dim3(1), dim3(1024), 0, st, ia);
    ECHK(hipGetLastError());
    ECHK(d2h(st, &h0, sc, sizeof(h0)));
    vy = h0.vy;
  }
  const float sumvx = P->MSx;                                                        // vx.sum()
  EmState h; memset(&h, 0, sizeof(h));
  h.df = df; h.vy = vy; h.MSx = sumvx; h.sumvx = sumvx; h.R2 = R2; h.ve = 1.0f;
  std::vector<float> hostv, xxh, yxh, bh;
  auto fill = [&](float *dst, float v) { hostv.assign((size_t)p, v); hipError_t e_ = hipMemcpyAsync(dst, hostv.data(), pb, hipMemcpyHostToDevice, st); return e_ != hipSuccess ? e_ : hipStreamSynchronize(st); };
  if (model == BWGR_EM_BA || model == BWGR_EM_BB) {
    h.ve = 1;                                                                        // :84, :135
    if (model == BWGR_EM_BB) {
      float Pi = par; if (Pi > 0.5f) Pi = 1 - Pi;                                    // :141
      h.Pi = Pi; h.MSx = sumvx * Pi;                                                 // :147
      h.Pi0 = (1 - Pi) / Pi;                                                         // :154
    }
    h.Sb = R2 * (df + 2) * vy / h.MSx;                                               // :96, :148
    h.Se = (1 - R2) * (df + 2) * vy;                                                 // :97, :149
    ECHK(fill(lamd, 1.0f)); ECHK(fill(vbd, 1.0f));                                   // vb = 1, Lmb = ve * vb^-1 = 1, :87-88
  } else if (model == BWGR_EM_RR) {
    h.Lmb = sumvx;                                                                   // :319
    h.Rho = sumvx * (1 - R2) / R2;                                                   // :320
    h.ve = 0.5f * vy;                                                                // :322
    h.vb = h.ve / sumvx;                                                             // :323
    h.Se = (1 - R2) * (df + 2) * vy;                                                 // :324
    h.Sb = R2 * (df + 2) * vy / sumvx;                                               // :325
  } else if (model == BWGR_EM_DE) {
    hipLaunchKernelGGL(k_em_fix_xx, dim3(1024), dim3(256), 0, st, xxd, p);           // :261
    h.cxx = sumvx * (1 - R2) / R2;                                                   // :265
    ECHK(fill(lamd, (float)p + h.cxx));                                              // :269
  } else if (model == BWGR_EM_ML) {
    h.Lmb = sumvx;                                                                   // :486
  } else if (model == BWGR_EM_BC || model == BWGR_EM_BCPI) {
    float Pi = par; if (Pi > 0.5f) Pi = 1 - Pi;                                      // :197, :1508
    h.Pi = Pi; h.PriorPi = Pi;                                                       // :1511
    h.MSx = sumvx * Pi * (1 - Pi);                                                   // :203, :1512
    h.Sa = R2 * (df + 2) * vy / h.MSx;                                               // :204
    h.Se = (1 - R2) * (df + 2) * vy;                                                 // :205
    h.ve = h.Sa; h.va = h.Se; h.Lmb = h.ve / h.va;                                   // :209-211 (sic)
    h.Pi0 = (1 - Pi) / Pi;                                                           // :213
  } else if (model == BWGR_EM_BL || model == BWGR_EM_EN || lasso) {
    xxh.resize((size_t)p);
    ECHK(d2h(st, xxh.data(), xxd, pb));
    h.alpha = par;
    if (lasso) {
      double sx = 0; for (int64_t j = 0; j < p; ++j) sx += (double)xxh[(size_t)j];
      h.Lmb1 = (float)(sx / (double)p) / (float)p;                                   // Lmb = xx.mean()/p, :1472
    } else if (model == BWGR_EM_BL) {
      double sx = 0; for (int64_t j = 0; j < p; ++j) sx += (double)xxh[(size_t)j];
      h.cxx = (float)(sx / (double)p);                                               // xx.mean(), :368
      const float hh = R2;                                                           // h2 = R2, :359
      h.Lmb1 = h.cxx * ((1 - hh) / hh) * h.alpha * 0.5f;                             // :369
      h.Lmb2 = h.cxx * ((1 - hh) / hh) * (1 - h.alpha);                              // :370
    } else {
      h.cxx = sumvx * (1 - R2) / R2;                                                 // :412
      h.Sy = sqrtf(vy);                                                              // :414
      h.Lmb = h.cxx;                                                                 // :415
      h.Lmb1 = 0.5f * h.Lmb * h.alpha * h.Sy;                                        // :416
      h.Lmb2 = h.Lmb * (1 - h.alpha);                                                // :417
      float tr = 0; for (int64_t k = 0; k < p; ++k) tr += 1.0f / (xxh[(size_t)k] + h.Lmb);   // the reference's own float loop, :418-419
      h.trAC22 = tr;
    }
  }
  ECHK(hipMemcpyAsync(std_, &h, sizeof(h), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_em_init, dim3(1), dim3(1024), 0, st, yd, ed, (int)n, P->ld, std_);   // mu, e (overwrites k_chain_init's e)
  ECHK(hipGetLastError());
  {
    ChainScalars h0; memset(&h0, 0, sizeof(h0));
    h0.ve = 1.0f; h0.pi = 0.0f; h0.dfp1 = 1.0f;                                      // the sweep's variates are switched off
    h0.C = -0.5f / sqrtf(h.ve);                                                      // :157, :216, :1522
    h0.odds = h.Pi0; h0.lam = h.Lmb1; h0.Sb = h.cxx;                                 // Pi0; Lmb1; emBL's cxx (k_prestage)
    ECHK(hipMemcpyAsync(sc, &h0, sizeof(h0), hipMemcpyHostToDevice, st));
  }
  std::vector<int> order((size_t)p), order_next;
  for (int64_t j = 0; j < p; ++j) order[(size_t)j] = (int)j;
  if (shuffled) std::shuffle(order.begin(), order.end(), std::mt19937(0));            // sweep 0's order
  if (!shuffled) ECHK(hipMemcpyAsync(ordd, order.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice, st));
  const int cps = (int)((size_t)P->R * (P->is_f32 ? 4 : 1) / 16);
  uint32_t flags = SWF_LAM_VEC;
  if (model == BWGR_EM_BA) flags |= SWF_DELTA2;
  if (soft) flags |= SWF_EM_SEL;
  if (model == BWGR_EM_EN) flags |= SWF_EM_EN;
  if (model == BWGR_EM_BL) flags |= SWF_EM_BL;
  if (lasso) flags |= SWF_EM_LASSO;
  int numit = 0;
  const bool emdbg = getenv("BWGR_EM_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int i = 0; i < maxit; ++i) {
    const double t_0 = now();
    if (shuffled) {
      // order = sweep i's marker order.  std::shuffle(order.begin(), order.end(), std::mt19937(i)) -- :103, :277, :331, :491 ...,
      // the reference's own call -- was made for sweep 0 before the loop and is made for sweep i+1 below, while the GPU
      // runs sweep i (10-15 ms of host time per sweep at p = 10^6)
      ECHK(hipMemcpyAsync(ordd, order.data(), sizeof(int32_t) * p, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_permute_cols, dim3(8192), dim3(256), 0, st, (const uint4 *)P->X, (uint4 *)Q->X, ordd, p, P->K, cps);
    }
    if (conv) ECHK(hipMemcpyAsync(bcd, bd, pb, hipMemcpyDeviceToDevice, st));        // bc = b
    hipLaunchKernelGGL(k_em_stage, dim3(1024), dim3(256), 0, st, ordd, p, model, D ? 1 : 0, bd, xxd, lamd, Dd, std_, bq, xxq, lamq);
    ECHK(hipGetLastError());
    if (shuffled) {
      rc = panel_build_gram(Q);
      if (rc != BWGR_OK) return done(rc);
    }
    SweepArgs a; memset(&a, 0, sizeof(a));
    fill_panel_args(S, a);
    a.flags = flags;
    a.e = ed; a.b = bq; a.d = dq; a.vb = vq; a.xx = xxq; a.lam = lamq; a.sc = sc;
    a.iter = (uint32_t)i; a.rng = make_rng(0, BWGR_RNG_DEGENERATE);
    rc = launch_sweep(S, a);
    if (rc != BWGR_OK) return done(rc);
    hipLaunchKernelGGL(k_em_unstage, dim3(1024), dim3(256), 0, st, ordd, p, bq, bd, (soft || lasso) ? dq : nullptr, (soft || lasso) ? dd : nullptr);
    EmTailArgs t; t.model = model; t.n = (int)n; t.conv = conv ? 1 : 0; t.p = p; t.e = ed; t.y = yd; t.b = bd; t.bc = bcd; t.d = dd;
    t.lam = lamd; t.vbv = vbd; t.xx = xxd; t.st = std_; t.sc = sc;
    hipLaunchKernelGGL(k_em_tail, dim3(1), dim3(1024), 0, st, t);
    ECHK(hipGetLastError());
    ++numit;
    const double t_1 = now();
    if (shuffled && i + 1 < maxit) { order_next = order; std::shuffle(order_next.begin(), order_next.end(), std::mt19937(i + 1)); }
    const double t_2 = now();
    // the convergence test needs cnv (and the sweep's status word)
    ChainScalars hc;
    ECHK(hipMemcpyAsync(&h, std_, sizeof(h), hipMemcpyDeviceToHost, st));
    ECHK(hipMemcpyAsync(&hc, sc, sizeof(hc), hipMemcpyDeviceToHost, st));
    ECHK(hipStreamSynchronize(st));
    if (emdbg) fprintf(stderr, "em sweep %d: launches %.2f ms, host shuffle %.2f ms, wait %.2f ms\n", i, t_1 - t_0, t_2 - t_1, now() - t_2);
    if (hc.error) return done(sweep_error(hc.error, "em"));
    if (lasso) {   // Lmb from the sweep's yx and b: the reference's own sequential float loop, :1487-1490
      yxh.resize((size_t)p); bh.resize((size_t)p);
      ECHK(d2h(st, yxh.data(), dd, pb)); ECHK(d2h(st, bh.data(), bd, pb));
      float tmp = 0.0f;
      for (int64_t j = 0; j < p; ++j) tmp += fabsf(yxh[(size_t)j]) - fabsf(bh[(size_t)j] * xxh[(size_t)j]);
      float L = 2.0f * tmp / (float)p;
      L = 2.0f * sqrtf(fabsf(L));
      h.Lmb1 = L;
      ECHK(hipMemcpyAsync(std_, &h, sizeof(h), hipMemcpyHostToDevice, st));
      ECHK(hipMemcpyAsync(&sc->lam, &h.Lmb1, sizeof(float), hipMemcpyHostToDevice, st));
      ECHK(hipStreamSynchronize(st));
    }
    if (conv && h.cnv < tol) break;                                                  // :296, :452, :510, :1492
    if (shuffled) order.swap(order_next);
  }
  float h2;
  if (model == BWGR_EM_ML) {
    hipLaunchKernelGGL(k_em_fit_ml, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, yd, ed, hatd, (int)n);
    h2 = h.vb * h.MSx / (h.vb * h.MSx + h.ve);                                       // :513
  } else if (lasso) {
    hipLaunchKernelGGL(k_em_fit_ml, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, yd, ed, hatd, (int)n);   // fit = y - e, :1493
    std::vector<double> eh((size_t)n);
    ECHK(d2h(st, eh.data(), ed, sizeof(double) * n));
    double s = 0; for (int64_t k = 0; k < n; ++k) s = fma(eh[(size_t)k], (double)y[k], s);
    h2 = 1.0f - ((float)s / (float)(n - 1)) / vy;                                    // :1494
  } else {
    rc = gemv_hat<float>(P, bd, h.mu, hatd);                                         // fit = gen*b + mu, :120-121
    if (rc != BWGR_OK) return done(rc);
    if (model == BWGR_EM_DE) {                                                       // h2 = Vb.sum()/(Vb.sum()+Ve), :304
      double *part = nullptr; float *sdev = nullptr; float sv = 0;
      ECHK(dmalloc((void **)&part, sizeof(double) * 256)); ECHK(dmalloc((void **)&sdev, sizeof(float)));
      hipLaunchKernelGGL(k_sum_stage1, dim3(256), dim3(256), 0, st, vbd, p, part);
      hipLaunchKernelGGL(k_sum_stage2, dim3(1), dim3(256), 0, st, part, 256, sdev);
      ECHK(d2h(st, &sv, sdev, sizeof(float)));
      h2 = sv / (sv + h.ve);
    } else if (model == BWGR_EM_BL) {                                                // h2 = 1 - fvar(e)/fvar(y), :396
      std::vector<double> eh((size_t)n);
      ECHK(d2h(st, eh.data(), ed, sizeof(double) * n));
      double s = 0; for (int64_t k = 0; k < n; ++k) s += (double)(float)eh[(size_t)k];
      const float m = (float)(s / (double)n);
      double sv = 0; for (int64_t k = 0; k < n; ++k) { const float dev = (float)eh[(size_t)k] - m; const float sq = dev * dev; sv += (double)sq; }
      h2 = 1 - (float)(sv / (double)(float)(n - 1)) / vy;
    } else if (model == BWGR_EM_EN) h2 = h.va * h.cxx / (h.va * h.cxx + h.ve);       // :459
    else h2 = 1 - h.ve / vy;                                                         // :119, :178, :237, :344, :1542
  }
  ECHK(hipGetLastError());
  if (mu) *mu = h.mu;
  ECHK(d2h(st, b, bd, pb));
  if (d && soft) ECHK(d2h(st, d, dd, pb));
  if (hat) ECHK(d2h(st, hat, hatd, sizeof(float) * n));
  if (vbvec && (model == BWGR_EM_BA || model == BWGR_EM_DE || model == BWGR_EM_BB)) ECHK(d2h(st, vbvec, vbd, pb));
  for (int k = 0; k < 6; ++k) scal[k] = 0.0f;
  scal[1] = h.ve; scal[2] = h2;
  if (model == BWGR_EM_RR || model == BWGR_EM_ML) scal[0] = h.vb;
  if (model == BWGR_EM_ML) scal[3] = h.vb * h.MSx;                                   // Va = vb*MSx, :519
  if (model == BWGR_EM_BC || model == BWGR_EM_BCPI) { scal[0] = h.va; scal[3] = h.va * h.MSx; }   // Va, Vg = va*MSx, :243, :1547
  if (model == BWGR_EM_BCPI) scal[4] = h.Pi;
  if (model == BWGR_EM_EN) scal[0] = h.va * h.cxx;                                   // :457
  if (model == BWGR_EM_BL) scal[1] = 0.0f;
  if (lasso) { scal[0] = h.Lmb1; scal[1] = 0.0f; }                                   // Lmb, :1497
  if (iters) *iters = numit;
#undef ECHK
  return done(BWGR_OK);
}

// ------------------------------------------------------------------------------------------------
// synthetic data and test hooks
// ------------------------------------------------------------------------------------------------
extern "C" int bwgr_synth_genotypes(void *Xdev, int64_t n, int64_t p, int64_t ldx, int64_t col0, uint64_t seed,
                                    float *freq_dev, int device, void *hip_stream) {
  if (!Xdev || n < 1 || p < 1 || ldx < n || (ldx & 3)) return fail(BWGR_EINVAL, "synth: need ldx >= n and ldx %% 4 == 0");
  CHK(require_device(device));
  hipLaunchKernelGGL(k_synth, dim3(8192), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), (int8_t *)Xdev, ldx, (int)n, p, col0,
                     (uint32_t)seed, (uint32_t)(seed >> 32), freq_dev);
  HIPCHK(hipGetLastError());
  return BWGR_OK;
}

extern "C" int bwgr_debug_variates(int device, uint64_t seed, int kind, double nu, uint32_t marker0, uint32_t iter,
                                   uint32_t purpose, int count, double *out_host) {
  if (!out_host || count < 1) return fail(BWGR_EINVAL, "debug_variates: bad arguments");
  CHK(require_device(device));
  double *dev = nullptr;
  HIPCHK(hipMalloc(&dev, sizeof(double) * count));
  hipLaunchKernelGGL(k_debug_variates, dim3((count + 255) / 256), dim3(256), 0, 0, make_rng(seed, 0), kind, nu, marker0, iter, purpose, count, dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out_host, dev, sizeof(double) * count, hipMemcpyDeviceToHost));
  HIPCHK(hipFree(dev));
  return BWGR_OK;
}
