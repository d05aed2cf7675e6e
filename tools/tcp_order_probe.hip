// Microbenchmark (gfx950): does a miss of one wave delay the L2 hits of another wave of the same compute unit?  One workgroup; wave 1 times bursts of
// four LDS-DMA requests that hit L2 (a 64 KiB buffer read over and over) while wave 0 (a) idles, (b) streams LDS-DMA requests that miss to HBM
// (a 2 GiB buffer, 1 KiB per request, never re-read), one outstanding burst at a time.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/tcp_order_probe tools/tcp_order_probe.hip && tools/_bin/tcp_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const unsigned char *gbase, uint32_t voff, uint32_t la) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gbase), "s"(la) : "memory", "m0");
}
#pragma clang diagnostic pop
__global__ __launch_bounds__(512) void k_probe(const unsigned char *hot, const unsigned char *cold, size_t cold_bytes, int misses_per_iter, int iters, unsigned long long *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t la0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)smem) + (uint32_t)wave * 8192u;
  if (wave == 1) {
    unsigned long long worst = 0, total = 0;
    size_t off = 0;
    for (int it = 0; it < iters; ++it) {
      const unsigned long long t0 = wall_clock64();
      for (int k = 0; k < 4; ++k) dma16(hot + off, (uint32_t)(k * 1024 + lane * 16), la0 + k * 1024u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long dt = wall_clock64() - t0;
      total += dt; worst = dt > worst ? dt : worst;
      off = (off + 4096) & 65535;
      __builtin_amdgcn_s_sleep(8);
    }
    if (lane == 0) { out[0] = total; out[1] = worst; }
  } else if (wave == 0 && misses_per_iter > 0) {
    size_t off = 0;
    for (int it = 0; it < 8 * iters; ++it) {
      for (int k = 0; k < misses_per_iter; ++k) { dma16(cold + off, (uint32_t)(lane * 16), la0 + (k & 7) * 1024u); off += 1 << 20; if (off + (1 << 20) > cold_bytes) off = (size_t)(it & 1023) * 1024; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (*(volatile unsigned long long *)(out + 2)) break;   // wave 1 is done
    }
  }
  __syncthreads();
  if (threadIdx.x == 64) out[2] = 1;
}
int main() {
  const size_t cold_bytes = (size_t)2 << 30;
  unsigned char *hot, *cold; unsigned long long *out;
  hipMalloc(&hot, 1 << 20); hipMemset(hot, 1, 1 << 20);
  hipMalloc(&cold, cold_bytes); hipMemset(cold, 2, cold_bytes);
  hipMalloc(&out, 64);
  int rate_khz = 0; hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  const double tick_ns = 1e6 / (double)rate_khz;
  const int iters = 20000;
  for (int m : {0, 1, 2, 8}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, 64);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(512), 65536, 0, hot, cold, cold_bytes, m, iters, out);
      hipDeviceSynchronize();
    }
    unsigned long long h[3]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave 0 keeps %d HBM-miss request(s) in flight: wave 1's burst of four L2-hit requests takes %7.1f ns on average, %7.1f ns at worst\n", m, h[0] * tick_ns / iters, h[1] * tick_ns);
  }
  return 0;
}
