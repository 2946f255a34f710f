// Micro-benchmark: can a wave issue other instructions in the shadow of its own v_mfma_f32_32x32x2_f32?
// One dependent MFMA chain per wave; between two MFMAs FILL instructions of one kind: dependent scalar adds (the chain
// mfma_peak.hip uses), INDEPENDENT scalar adds, independent vector FMAs, LDS reads.  1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_fill.hip -o mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int FILL>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0) {
  __shared__ float lds[1024];
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  float a = a0 + threadIdx.x, b = a0 * 0.5f;
  int s[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
  lds[threadIdx.x] = a;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 255);
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) {
        if (KIND == 0) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s[0]));
        if (KIND == 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s[f & 7]));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[f & 7]) : "v"(b));
        if (KIND == 3) { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"((unsigned)(size_t)lp), "n"(0)); sink += 0.f * t; }
        if (KIND == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<double*>(&v[2 * (f & 3)])) : "v"(*reinterpret_cast<double*>(&v[0])));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = sink;
  for (int j = 0; j < 8; ++j) r += (float)s[j] + v[j];
  for (int j = 0; j < 16; ++j) r += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
// the same question for the two neighbours of the f32 instruction: v_mfma_f32_32x32x16_bf16 (the matrix cores proper, 16x the
// rate) and v_mfma_f64_16x16x4_f64 (like f32, the rate of the vector ALU)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MF, int FILL>
__global__ __launch_bounds__(256, 2) void k2(float* out, int iters, float a0) {
  f32x16 acc;
  f64x4 acc64 = {0, 0, 0, 0};
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a0 + j); bb[j] = (__bf16)(0.5f * a0); }
  double ad = a0 + threadIdx.x, bd = a0 * 0.5;
  float b = a0 * 0.5f;
  float v[8] = {a0, b, a0 + 1, b + 1, a0 + 2, b + 2, a0 + 3, b + 3};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MF == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
      if (MF == 2) acc64 = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, acc64, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[f & 7]) : "v"(b));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0.f;
  for (int j = 0; j < 8; ++j) r += v[j];
  for (int j = 0; j < 16; ++j) r += acc[j];
  for (int j = 0; j < 4; ++j) r += (float)acc64[j];
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
  const char* kinds[] = {"dependent s_add", "independent s_add", "independent v_fma_f32", "ds_read_b32", "independent v_pk_fma_f32"};
  for (int wgs : {256, 512}) {
    const double fl = 4096.0 * 4 * iters * 4.0 * wgs;
#define RUN(KD, FL) { double ms = time_ms([&] { k<KD, FL><<<wgs, 256>>>((float*)out, iters, 1.f); }); \
      printf("waves/SIMD %d  %-26s x %2d per MFMA : %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", wgs / 256, kinds[KD], FL, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * iters)); }
    RUN(0, 0) RUN(0, 8) RUN(1, 8) RUN(1, 16) RUN(2, 4) RUN(2, 8) RUN(2, 12) RUN(2, 16) RUN(4, 8) RUN(3, 4) RUN(3, 8)
#define RUN2(MF, NAME, FL) { double ms = time_ms([&] { k2<MF, FL><<<wgs, 256>>>((float*)out, iters, 1.f); }); \
      printf("waves/SIMD %d  %-22s + %2d v_fma_f32 per MFMA : %.3f ms  (%.1f cycles per MFMA at 2.4 GHz)\n", wgs / 256, NAME, FL, ms, ms * 1e-3 * 2.4e9 / (4.0 * iters)); }
    RUN2(1, "mfma_f32_32x32x16_bf16", 0) RUN2(1, "mfma_f32_32x32x16_bf16", 4) RUN2(1, "mfma_f32_32x32x16_bf16", 8)
    RUN2(2, "mfma_f64_16x16x4_f64", 0) RUN2(2, "mfma_f64_16x16x4_f64", 4) RUN2(2, "mfma_f64_16x16x4_f64", 8)
  }
  return 0;
}
