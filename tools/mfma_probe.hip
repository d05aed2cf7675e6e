// Diagnostic only (not part of the product): checks the lane maps of v_mfma_i32_16x16x64_i8 with exact integer data
// and times v_dot4_i32_i8 / the MFMA back to back on one wave.   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const int8_t *A, const int8_t *B, int *D) {   // A[16][64] row-major, B[64][16] row-major
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  int8_t a[16], b[16];
  for (int j = 0; j < 16; ++j) { a[j] = A[r * 64 + 16 * g + j]; b[j] = B[(16 * g + j) * 16 + r]; }
  v4i av, bv, c = {0, 0, 0, 0};
  __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];      // assumed: col = lane&15, row = 4*(lane>>4)+reg
}
__global__ void k_time(long long *out, int *sink) {
  const int l = threadIdx.x;
  int a0 = l, a1 = l * 3, a2 = l * 5, a3 = l * 7, a4 = 1, a5 = 2, a6 = 3, a7 = 4; const int x = 0x01020304 + l, y = 0x04030201;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 1000; ++i) {
    a0 = __builtin_amdgcn_sdot4(x, y, a0, false); a1 = __builtin_amdgcn_sdot4(x, y, a1, false);
    a2 = __builtin_amdgcn_sdot4(x, y, a2, false); a3 = __builtin_amdgcn_sdot4(x, y, a3, false);
    a4 = __builtin_amdgcn_sdot4(x, y, a4, false); a5 = __builtin_amdgcn_sdot4(x, y, a5, false);
    a6 = __builtin_amdgcn_sdot4(x, y, a6, false); a7 = __builtin_amdgcn_sdot4(x, y, a7, false);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  v4i av = {x, y, x, y}, bv = {y, x, y, x}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < 1000; ++i) {
    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c3, 0, 0, 0);
  }
  long long t2 = __builtin_amdgcn_s_memtime();
  double d0 = l, d1 = 1.5, d2 = 2.5, d3 = 3.5; const double m = 1.0000001;
  for (int i = 0; i < 1000; ++i) { d0 = fma(d0, m, d1); d1 = fma(d1, m, d2); d2 = fma(d2, m, d3); d3 = fma(d3, m, d0);
    asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); }
  long long t3 = __builtin_amdgcn_s_memtime();
  if (l == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; }
  sink[l] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3] + (int)(d0 + d1 + d2 + d3);
}
int main() {
  int8_t hA[16 * 64], hB[64 * 16]; int hD[256], ref[256];
  srand(7);
  for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int k = 0; k < 64; ++k) s += (int)hA[i * 64 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  int8_t *dA, *dB; int *dD, *sink; long long *dT;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024); hipMalloc(&sink, 256); hipMalloc(&dT, 64);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("mfma_i32_16x16x64_i8 with the assumed maps: %d of 256 outputs differ\n", bad);
  long long hT[3];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, dT, sink); hipMemcpy(hT, dT, 24, hipMemcpyDeviceToHost); }
  printf("s_memtime ticks: 8000 v_dot4 = %lld (%.2f/instr), 4000 mfma = %lld (%.2f/instr), 4000 dfma = %lld (%.2f/instr)\n", hT[0], hT[0] / 8000.0, hT[1], hT[1] / 4000.0, hT[2], hT[2] / 4000.0);
  return bad != 0;
}
