#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
  extern __shared__ char sm[];
  if (threadIdx.x == 0) {
    sm[0] = 1;
    out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
  }
  // keep the block alive a little so that residency matters
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  const int n = 64;
  int* d; hipMalloc(&d, n * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(k, dim3(n), dim3(512), 95 * 1024, 0, d);
  std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%d%c", h[i], (i % 16 == 15) ? '\n' : ' ');
  return 0;
}
