#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// LDS image: byte at address a holds (a & 0xFF) pattern with row/col encoded: we fill lds[a] = a for a < 4096 (mod 256) plus a second array for high bits
__global__ void k(uint32_t *out, int stride, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned char)(i & 0xFF);
  __syncthreads();
  const int lane = threadIdx.x;
  // address pattern A (guess): group of 16 lanes; lane 2q+p supplies row q (q = 0..7), columns 8p..8p+7 of a block with row stride `stride`
  const int g = lane >> 4, l = lane & 15;
  uint32_t addr;
  if (mode == 0) addr = (uint32_t)(g * 16 + (l >> 1) * stride + (l & 1) * 8);
  else addr = (uint32_t)(g * 16 + (l & 7) * stride + (l >> 3) * 8);
  const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + addr;
  typedef int v2i __attribute__((ext_vector_type(2)));
  v2i r;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(la) : "memory");
  out[2 * lane] = (uint32_t)r[0]; out[2 * lane + 1] = (uint32_t)r[1];
}
int main() {
  uint32_t *d; hipMalloc(&d, 64 * 8);
  uint32_t h[128];
  for (int mode = 0; mode < 2; ++mode) {
    const int stride = 256;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d stride %d (byte value = address & 255; row r -> + r*256 wraps to same low byte, so value = column byte offset)\n", mode, stride);
    for (int l = 0; l < 20; ++l) printf(" lane %2d: %08x %08x\n", l, h[2 * l], h[2 * l + 1]);
  }
  // second experiment: stride 64 so that row and column are both visible in the low byte
  for (int mode = 0; mode < 2; ++mode) {
    const int stride = 32;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d stride %d (value = 32*row + col [+16 per lane group])\n", mode, stride);
    for (int l = 0; l < 18; ++l) printf(" lane %2d: %08x %08x\n", l, h[2 * l], h[2 * l + 1]);
  }
  return 0;
}
