// Microbenchmark (gfx950): what one compute unit sustains for the request kinds the k_sweep3 sequencer's helper waves use -- per wave and
// per CU -- from an L2-resident source: LDS-DMA 16 B / lane, LDS-DMA 4 B / lane, register loads 16 B / lane (+ an LDS store), register loads 4 B.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dma_rate_probe tools/dma_rate_probe.hip && gpurun_out/dma_rate_probe
// Prints ns per request (one wave instruction = 64 lanes) for W = 1, 2, 4, 8 requesting waves in ONE workgroup of 512 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbase), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ void dma4(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(gbase), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ u4 ld16(const unsigned char *gbase, uint32_t voff) {
  u4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t ld4(const unsigned char *gbase, uint32_t voff) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(gbase) : "memory");
  return v;
}
#pragma clang diagnostic pop

// mode 0: LDS-DMA x4; 1: LDS-DMA dword; 2: register load x4 + ds_write_b128; 3: register load dword + ds_write_b32; 4: register load x4, no LDS store
template <int MODE, int K>
__global__ __launch_bounds__(512) void k_probe(const unsigned char *src, size_t span, int W, int iters, unsigned long long *out, uint32_t *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= W) return;
  const uint32_t la0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem) + (uint32_t)wave * (K * 1024u);
  uint32_t acc = 0;
  const size_t wspan = span / 8;   // each wave walks its own part
  const unsigned char *wsrc = src + (size_t)wave * wspan;
  const unsigned long long t0 = wall_clock64();
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned char *b = wsrc + off;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) dma16(b, (uint32_t)(k * 1024 + lane * 16), la0 + k * 1024u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) dma4(b, (uint32_t)(k * 256 + lane * 4), la0 + k * 256u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 2 || MODE == 4) {
      u4 v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = ld16(b, (uint32_t)(k * 1024 + lane * 16));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (MODE == 2) {
#pragma unroll
        for (int k = 0; k < K; ++k) reinterpret_cast<u4 *>(smem + (size_t)wave * K * 1024 + k * 1024)[lane] = v[k];
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k].x;
      }
    } else {
      uint32_t v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = ld4(b, (uint32_t)(k * 256 + lane * 4));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < K; ++k) reinterpret_cast<uint32_t *>(smem + (size_t)wave * K * 1024 + k * 256)[lane] = v[k];
    }
    off += (MODE == 1 || MODE == 3) ? K * 256 : K * 1024;
    if (off + K * 1024 > wspan) off = 0;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  acc ^= reinterpret_cast<uint32_t *>(smem)[threadIdx.x];
  if (lane == 0) out[wave] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE, int K>
static void run(const char *name, const unsigned char *src, size_t span, unsigned long long *out, uint32_t *sink, double tick_ns) {
  const int iters = 4000;
  for (int W : {1, 2, 4, 8}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k_probe<MODE, K>), dim3(1), dim3(512), 8 * K * 1024, 0, src, span, W, iters, out, sink);
      hipDeviceSynchronize();
    }
    unsigned long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < W; ++w) mx = h[w] > mx ? h[w] : mx;
    const double ns = mx * tick_ns;
    const double bytes = (double)W * iters * K * ((MODE == 1 || MODE == 3) ? 256.0 : 1024.0);
    printf("%-44s K=%2d W=%d  %8.1f ns per burst  %6.1f ns per request and wave  %6.1f ns per request (CU)  %6.1f GB/s\n", name, K, W, ns / iters, ns / iters / K,
           ns / iters / K / W, bytes / ns);
  }
}

int main() {
  const size_t span = 8u << 20;   // 8 MiB: L2-resident (MALL at worst) after the first pass
  unsigned char *src; unsigned long long *out; uint32_t *sink;
  hipMalloc(&src, span + (1 << 20)); hipMemset(src, 1, span + (1 << 20));
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  int rate_khz = 0;
  hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  const double tick_ns = 1e6 / (double)rate_khz;
  printf("wall clock %d kHz\n", rate_khz);
  run<0, 12>("LDS-DMA 16 B/lane (global_load_lds_dwordx4)", src, span, out, sink, tick_ns);
  run<1, 12>("LDS-DMA 4 B/lane (global_load_lds_dword)", src, span, out, sink, tick_ns);
  run<2, 12>("register load 16 B/lane + ds_write_b128", src, span, out, sink, tick_ns);
  run<4, 12>("register load 16 B/lane, no LDS store", src, span, out, sink, tick_ns);
  run<3, 12>("register load 4 B/lane + ds_write_b32", src, span, out, sink, tick_ns);
  run<0, 4>("LDS-DMA 16 B/lane (global_load_lds_dwordx4)", src, span, out, sink, tick_ns);
  run<2, 4>("register load 16 B/lane + ds_write_b128", src, span, out, sink, tick_ns);
  // small footprint: the same 48 KiB per wave over and over (L2 hits for certain)
  run<0, 12>("LDS-DMA 16 B/lane, 64 KiB footprint", src, 8 * 65536, out, sink, tick_ns);
  run<2, 12>("register load 16 B/lane + store, 64 KiB fp", src, 8 * 65536, out, sink, tick_ns);
  return 0;
}
