// shared_ring.hip -- the core loop a split-bf16 dense kernel would need ("the split-bf16 product"): the four waves
// of a workgroup consume the SAME operand positions (3 KB each: the three bf16 planes of a 32 x 16 block of A) from ONE LDS ring, each
// against its own B operand (its own panel of 16 baselines) and its own accumulators -- 64 columns per operand byte instead of 16.
//   * a group = 4 positions (12 KB); wave w requests position w of every group with three LDS-DMA loads (global_load_lds_dwordx4);
//   * per group: wait for the own loads of group g (counted vmcnt), s_barrier -> every position of group g has landed AND every wave
//     has finished group g - 1, so the slots of group g - 1 are free: request group g - 1 + RING into them; then consume group g
//     (per position: three ds_read_b128, six v_mfma_f32_32x32x16_bf16).
// Measured: useful fp32-product TFLOP/s over the chip with the operands (a) 0.4 MB per workgroup set, L2-resident, (b) streaming
// through 1.5 x 68 MB as the HERA-350 basis would (every block read by many workgroups of one XCD).
// build: hipcc -O3 --offload-arch=gfx950 -o shared_ring shared_ring.hip ; run: ./shared_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lds_dma(unsigned lds_addr, const void* sbase, unsigned voff) {
  unsigned keep;
  const unsigned long long v = reinterpret_cast<unsigned long long>(sbase);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  const void* sb = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(sb)
               : "memory");
}
#define WAIT_VM(N) do { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// ops: [nblocks][npos][3][64 lanes][16 B]; workgroup b streams block (b % nblocks), `sweeps` times
template <int RING>  // groups in the ring
__global__ __launch_bounds__(256, 2) void ring_kernel(const char* __restrict__ ops, int nblocks, int npos, int sweeps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [RING groups][4 positions][3 KB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* blk = ops + (size_t)(blockIdx.x % nblocks) * npos * 3072;
  const int ngroups = npos / 4;
  const int total = ngroups * sweeps;
  const unsigned ring_lds = (unsigned)reinterpret_cast<unsigned long long>(smem);
  const unsigned voff = (unsigned)lane * 16u;
  auto request = [&](int g) {  // this wave's position of group g (clamped past the end: the count of loads in flight stays what the waits assume)
    const int gg = g < total ? g % ngroups : (total - 1) % ngroups;
    const unsigned slot = ring_lds + (unsigned)((g % RING) * 4 + wave) * 3072u;
    const unsigned off = (unsigned)(gg * 4 + wave) * 3072u;
    lds_dma(slot, blk, voff + off);
    lds_dma(slot + 1024u, blk, voff + off + 1024u);
    lds_dma(slot + 2048u, blk, voff + off + 2048u);
  };
  bf16x8 c1, c2, c3;  // this wave's B operand: its own
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    c1[j] = (__bf16)(1.0f + 0.01f * (lane + wave + j));
    c2[j] = (__bf16)(0.01f * (lane - j));
    c3[j] = (__bf16)(0.0001f * (wave + j));
  }
  f32x16 acc = {0}, lo = {0};
  for (int g = 0; g < RING - 1; ++g) request(g);
  for (int g = 0; g < total; ++g) {
    // own loads of group g: all but the 3 (RING - 2) younger ones have landed
    WAIT_VM(3 * (RING - 2));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    request(g + RING - 1);  // into the slots of group g - 1, which every wave has left
    __builtin_amdgcn_sched_barrier(0);
    const f32x4* rd = reinterpret_cast<const f32x4*>(smem + (size_t)(g % RING) * 4 * 3072) + lane;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f32x4 r1 = rd[(p * 3 + 0) * 64], r2 = rd[(p * 3 + 1) * 64], r3 = rd[(p * 3 + 2) * 64];
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, r1), a2 = __builtin_bit_cast(bf16x8, r2), a3 = __builtin_bit_cast(bf16x8, r3);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c1, acc, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c2, lo, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, c1, acc, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, c3, lo, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, c2, acc, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, c1, lo, 0, 0, 0);
    }
  }
  WAIT_VM(0);
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r] + lo[r];
  if (s == 1.2345e-30f) sink[0] = s;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  float* sink;
  CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = p.multiProcessorCount * 2;
  // a block of the HERA-350 basis: ~100 vectors x 1024 channels -> forward 7 K-steps + adjoint 8 K-steps per 32-channel block, x 32 blocks
  const int npos = 480;  // positions per block (1.44 MB of bf16 planes)
  struct Case { const char* name; int nblocks; int sweeps; };
  const Case cases[] = {{"operands L2-resident (8 blocks, 11.5 MB)", 8, 8}, {"120 blocks (173 MB: HBM / Infinity Cache / L2 as a real pass)", 120, 8}};
  for (const Case& c : cases) {
    char* ops;
    const size_t bytes = (size_t)c.nblocks * npos * 3072;
    CK(hipMalloc(&ops, bytes));
    CK(hipMemset(ops, 0x3c, bytes));  // bf16 0x3c3c = 0.0115: finite operands
    auto run = [&](auto kernel, int ring, const char* label) {
      const size_t lds = (size_t)ring * 4 * 3072;
      float best = 1e9f;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, ops, c.nblocks, npos, c.sweeps, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r) best = std::min(best, ms);
      }
      if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
      const double useful = 2.0 * 32 * 32 * 16 * (double)npos * c.sweeps * grid * 4;  // fp32-product flops: 4 waves x positions
      const double opbytes = (double)npos * 3072 * c.sweeps * grid;                     // L2 -> LDS
      printf("  %-22s %8.3f ms  %6.1f useful TFLOP/s  (operand stream %.1f TB/s into LDS)\n", label, best, useful / (best * 1e-3) / 1e12,
             opbytes / (best * 1e-3) / 1e12);
    };
    printf("%s\n", c.name);
    run(ring_kernel<3>, 3, "ring of 3 groups");
    run(ring_kernel<4>, 4, "ring of 4 groups");
    run(ring_kernel<6>, 6, "ring of 6 groups");
    CK(hipFree(ops));
  }
  return 0;
}
