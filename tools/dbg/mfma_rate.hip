#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void k(float* out, long long* cyc, float a0, float b0) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 256; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int c = 0; c < CH; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH> void run(int waves_per_cu, const char* tag) {
  float* d; long long* c; hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&c, 1024 * 8);
  int threads = 64 * waves_per_cu;
  hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, d, c, 1.f, 2.f);
  hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, d, c, 1.f, 2.f);
  long long h[256]; hipMemcpy(h, c, 256 * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%s chains=%d waves/CU=%d: %.1f cycles per MFMA (per wave)\n", tag, CH, waves_per_cu, m / (256.0 * CH));
  hipFree(d); hipFree(c);
}
int main() {
  run<1>(4, "32x32x2f32"); run<2>(4, "32x32x2f32"); run<4>(4, "32x32x2f32");
  run<1>(8, "32x32x2f32"); run<2>(8, "32x32x2f32");
  run<1>(1, "32x32x2f32"); run<2>(1, "32x32x2f32");
  return 0;
}
