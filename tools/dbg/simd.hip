#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  extern __shared__ char sm[];
  if ((threadIdx.x & 63) == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, 32 bits
    out[blockIdx.x * 16 + (threadIdx.x >> 6)] = (int)hw;
  }
  sm[threadIdx.x] = 1;
  for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  int* d; hipMalloc(&d, 64 * 16 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int threads : {512, 256}) {
    hipMemset(d, 0, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(threads), 95 * 1024, 0, d);
    int h[64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
      printf("threads %d block %d: simd of waves:", threads, b);
      for (int w = 0; w < threads / 64; ++w) printf(" %d", (h[b * 16 + w] >> 4) & 3);
      printf("   (cu %d)\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
