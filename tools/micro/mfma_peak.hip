// Micro-benchmark: what v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 sustain on this box, for the occupancies and
// chain shapes the dense kernels use.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL>
__global__ __launch_bounds__(256, 2) void k32(float* out, int iters, float a0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x, b = a0 * 0.5f;
  int s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < FILL; ++f) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float r = (float)s;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) r += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k64(double* out, int iters, double a0) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  double a = a0 + threadIdx.x, b = a0 * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double r = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 4; ++j) r += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <typename F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  void* out; hipMalloc(&out, 2048 * 256 * 8);
  const int iters = 4000;
  for (int wgs : {256, 512, 1024}) {
    double f32 = 4096.0 * 4 * iters * 4.0 * wgs;  // 4 waves per workgroup
#define RUN32(N, FL) { double ms = time_ms([&] { k32<N, FL><<<wgs, 256>>>((float*)out, iters / N, 1.f); }); \
      printf("f32 32x32x2  wgs %4d  chains %d  fill %2d scalar/MFMA : %.3f ms  %.1f TFLOP/s\n", wgs, N, FL, ms, f32 / ms / 1e9); }
    RUN32(1, 0) RUN32(2, 0) RUN32(4, 0) RUN32(1, 6) RUN32(1, 12) RUN32(1, 20)
    double f64 = 2048.0 * 4 * iters * 4.0 * wgs;
#define RUN64(N) { double ms = time_ms([&] { k64<N><<<wgs, 256>>>((double*)out, iters / N, 1.0); }); \
      printf("f64 16x16x4  wgs %4d  chains %d : %.3f ms  %.1f TFLOP/s\n", wgs, N, ms, f64 / ms / 1e9); }
    RUN64(1) RUN64(2) RUN64(4)
  }
  return 0;
}
