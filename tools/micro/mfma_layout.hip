// mfma_layout.hip -- check the operand / result lane maps of v_mfma_f32_16x16x4_f32 and the semantics of v_permlane32_swap that
// multi_mfma_kernels.hpp relies on.  build: hipcc -O2 --offload-arch=gfx950 -o mfma_layout mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, unsigned* sw) {
  const int l = threadIdx.x;
  // A[i][k] = 1 + i + 100 k (lane l: i = l & 15, k = l >> 4); B[k][j] = (k == 1) ? 1 + j * 0.001f : 0 -> D[i][j] = A[i][1] * B[1][j]
  const float a = 1.f + (l & 15) + 100.f * (l >> 4);
  const float b = ((l >> 4) == 1) ? 1.f + (l & 15) * 0.001f : 0.f;
  f4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
  const u2 s = __builtin_amdgcn_permlane32_swap((unsigned)l, 1000u + (unsigned)l, false, false);
  sw[l * 2] = s[0];
  sw[l * 2 + 1] = s[1];
}
int main() {
  float* out; unsigned* sw;
  hipMalloc(&out, 256 * 4); hipMalloc(&sw, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sw);
  float h[256]; unsigned hs[128];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, sw, sizeof(hs), hipMemcpyDeviceToHost);
  // expected under "lane (j = l & 15, q = l >> 4), reg r <-> D[i = 4 q + r][j]": D[i][j] = (101 + i) * (1 + 0.001 j)
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r, j = l & 15;
    const float e = (101.f + i) * (1.f + 0.001f * j);
    if (fabsf(h[l * 4 + r] - e) > 1e-3f) ++bad;
  }
  printf("D layout row = 4 (lane >> 4) + reg, col = lane & 15: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  for (int l : {0, 1, 17, 33, 63}) printf("lane %2d: d = %.3f %.3f %.3f %.3f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  for (int l : {0, 5, 31, 32, 37, 63}) printf("permlane32_swap(first = lane, second = 1000 + lane): lane %2d -> result0 %u result1 %u\n", l, hs[l * 2], hs[l * 2 + 1]);
  return 0;
}
