// tr_image.hip -- groundwork for a successor of the split-bf16 dense kernel ("what a successor would change"):
// ONE operand image per unit of channels, read row-wise by the forward product (ds_read_b128: 8 consecutive vectors of a channel) and
// TRANSPOSED by the adjoint (ds_read_b64_tr_b16: 4 consecutive channels of a vector), instead of two packed streams.
//   * image of a unit = [3 bf16 planes][RC channels][NVP vectors], rows of 256 bytes per 128-vector tile, 16-byte chunks XOR-swizzled
//     (cdna_hip_programming.md T10, image (b)): off(row, chunk) = 256 row + 16 (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
//     the global copy IS that byte image, so LDS-DMA lands it verbatim; RC x NVP x 6 B = 48 KB (RC = 64 at NVP = 128, 32 at 256), two buffers;
//   * a workgroup = 4 waves = 4 panels of 32 columns (16 baselines x re | im) of one basis block; per unit: wait + ONE barrier (the unit has
//     landed, the other buffer is free -> request the next unit), F (6 MFMAs per 16-vector step and channel block, coefficient operand from
//     REGISTERS, split into planes on the fly), a stand-in element stage (gbar_v = v: the accumulator registers 8 s .. 8 s + 7 are the adjoint's
//     B operand of channel step s, exactly the K order the transposed read delivers), B (6 MFMAs per 32-vector tile and 16-channel step);
//   * mode "check": one item against a double-precision host product (validates both address maps); mode "time": N items per class.
// No samples, gains or gbar_G here: this measures the operand path only (compare: the shipped kernel's skeleton rows of DESIGN 3.1e's table).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tr_image tr_image.hip ; run: ./tr_image
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kUnitBytes = 48 * 1024;  // 3 planes x RC x NVP x 2
constexpr int kPlaneBytes = 16 * 1024;

__host__ __device__ inline unsigned img_off(int row, int chunk) { return 256u * row + 16u * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
// byte offset of element (row, col) of one plane of a unit: 128-vector tiles of [RC][128] side by side
__host__ __device__ inline unsigned img_elem(int RC, int row, int col) { return (col >> 7) * (RC * 256) + img_off(row, (col & 127) >> 3) + 2 * (col & 7); }

__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma3(unsigned lds, const void* base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ bf16x8 lds_row(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ u32x2 lds_tr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void split3(const float* x, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h1 = (__bf16)x[j];
    const float r1 = x[j] - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    p1[j] = h1;
    p2[j] = h2;
    p3[j] = (__bf16)(r1 - (float)h2);
  }
}
// the compiler does not know that a ds_read's destination is not valid yet: the wait names the registers it makes valid (and is
// volatile: the reads of the NEXT step, issued in front of it, stay in front of it)
#define WAIT_LGKM(N, R0, R1, R2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R0), "+v"(R1), "+v"(R2)::"memory")
#define WAIT_LGKM6(N, R0, R1, R2, R3, R4, R5) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R0), "+v"(R1), "+v"(R2), "+v"(R3), "+v"(R4), "+v"(R5)::"memory")
#define MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
#define SIX(A1, A2, A3, B1, B2, B3, ACC) do { MFMA(A1, B1, ACC); MFMA(A1, B2, ACC); MFMA(A2, B1, ACC); MFMA(A1, B3, ACC); MFMA(A2, B2, ACC); MFMA(A3, B1, ACC); } while (0)

struct Args {
  const unsigned char* images;  // [nblocks][NU units][48 KB]
  const float* coef;            // [nitems][4 waves][NV][32]
  float* grad;                  // [nitems][4 waves][NV][32]
  const int* item_block;        // [grid] -> block of the item (-1: none)
  int nu;                       // units per item (fpad / RC)
};

// NVP: 128 | 256 (row pitch of the image); NV: vectors really used (multiple of 32)
template <int NVP, int NV>
__global__ __launch_bounds__(256, 1) void v2_skeleton(const Args A) {
  constexpr int RC = NVP == 128 ? 64 : 32;
  constexpr int NCB = RC / 32, NSTEP = NV / 16, NT = NV / 32;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int blk = A.item_block[blockIdx.x];
  if (blk < 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i32 = lane & 31, kg = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned char* img = A.images + (size_t)blk * A.nu * kUnitBytes;
  const void* gbase = uniform_ptr(img);
  // this wave's quarter of a unit: 12 KB = four requests of 3 KB
  auto request = [&](int u, int buf) {
    const unsigned voff = (unsigned)u * kUnitBytes + wave * 12288u + lane * 16u;
#pragma unroll
    for (int r = 0; r < 4; ++r) dma3(__builtin_amdgcn_readfirstlane(lds0 + buf * kUnitBytes + wave * 12288u + r * 3072u), gbase, voff + r * 3072u);
  };
  // coefficient operand of the wave's panel in registers: lane (col = i32, kg): C[16 s + 8 kg + j][col]
  float creg[NSTEP][8];
  const float* cw = A.coef + ((size_t)blockIdx.x * 4 + wave) * NV * 32;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) creg[s][j] = cw[(16 * s + 8 * kg + j) * 32 + i32];
  f32x16 dC[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dC[t][r] = 0.f;
  // lane addresses (bytes inside a plane of a unit).  Row read: row cb * 32 + i32, chunk (2 s + kg) & 15 of tile (16 s + 8 kg) >> 7.
  // Transposed read: lane 4 q + p of a 16-lane group supplies row R + q, columns Cb + 4 p ..: off(R + q, c0 + (p >> 1)) + 8 (p & 1).
  const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  request(0, 0);
  for (int u = 0; u < A.nu; ++u) {
    const int buf = u & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (u + 1 < A.nu) request(u + 1, buf ^ 1);
    const unsigned ub = lds0 + buf * kUnitBytes;
    f32x16 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    // ---- F: the operand of position n + 1 is requested before position n is waited for (position = (step, channel block))
    {
      auto addr = [&](int n) { const int s = n / NCB, cb = n % NCB; return ub + ((16 * s) >> 7) * (RC * 256) + img_off(cb * 32 + i32, (2 * s + kg) & 15); };
      bf16x8 ar[2][3];
      ar[0][0] = lds_row(addr(0));
      ar[0][1] = lds_row(addr(0) + kPlaneBytes);
      ar[0][2] = lds_row(addr(0) + 2 * kPlaneBytes);
      bf16x8 c1, c2, c3;
#pragma unroll
      for (int n = 0; n < NSTEP * NCB; ++n) {
        const int s = n / NCB, cb = n % NCB;
        if (cb == 0) split3(creg[s], c1, c2, c3);
        if (n + 1 < NSTEP * NCB) {
          ar[(n + 1) & 1][0] = lds_row(addr(n + 1));
          ar[(n + 1) & 1][1] = lds_row(addr(n + 1) + kPlaneBytes);
          ar[(n + 1) & 1][2] = lds_row(addr(n + 1) + 2 * kPlaneBytes);
          WAIT_LGKM(3, ar[n & 1][0], ar[n & 1][1], ar[n & 1][2]);
        } else {
          WAIT_LGKM(0, ar[n & 1][0], ar[n & 1][1], ar[n & 1][2]);
        }
        SIX(ar[n & 1][0], ar[n & 1][1], ar[n & 1][2], c1, c2, c3, acc[cb]);
      }
    }
    // ---- E (stand-in) + B: gbar_v = v; registers 8 cs .. 8 cs + 7 of acc[cb] are the B operand of channel step cs.
    // position n = ((cb, cs), t): six transposed reads (two per plane), requested one position ahead
    {
      constexpr int NP = NCB * 2 * NT;
      auto issue = [&](int n, u32x2 (&lo)[3], u32x2 (&hi)[3]) {
        const int t = n % NT, cc = n / NT, cb = cc >> 1, cs = cc & 1;
        const int R = cb * 32 + 16 * cs + 4 * kg;
        const int c0 = (4 * t + 2 * g16) & 15;
        const unsigned tb = ub + ((32 * t) >> 7) * (RC * 256) + 8 * (p4 & 1);
        const unsigned ad0 = tb + img_off(R + q4, c0 + (p4 >> 1)), ad1 = tb + img_off(R + 8 + q4, c0 + (p4 >> 1));
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          lo[pl] = lds_tr(ad0 + pl * kPlaneBytes);
          hi[pl] = lds_tr(ad1 + pl * kPlaneBytes);
        }
      };
      u32x2 lo[2][3], hi[2][3];
      issue(0, lo[0], hi[0]);
      bf16x8 g1, g2, g3;
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        const int t = n % NT, cc = n / NT, cb = cc >> 1, cs = cc & 1;
        if (t == 0) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = acc[cb][8 * cs + j];
          split3(x, g1, g2, g3);
        }
        const int k = n & 1;
        if (n + 1 < NP) {
          issue(n + 1, lo[k ^ 1], hi[k ^ 1]);
          WAIT_LGKM6(6, lo[k][0], lo[k][1], lo[k][2], hi[k][0], hi[k][1], hi[k][2]);
        } else {
          WAIT_LGKM6(0, lo[k][0], lo[k][1], lo[k][2], hi[k][0], hi[k][1], hi[k][2]);
        }
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, u32x4{lo[k][pl][0], lo[k][pl][1], hi[k][pl][0], hi[k][pl][1]});
        SIX(a[0], a[1], a[2], g1, g2, g3, dC[t]);
      }
    }
  }
  // dC[t][r] of lane (col, half): vector 32 t + (r & 3) + 8 (r >> 2) + 4 half
  float* gw = A.grad + ((size_t)blockIdx.x * 4 + wave) * NV * 32;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) gw[(32 * t + (r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i32] = dC[t][r];
}

static unsigned short bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_val(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int NVP, int NV>
static int run(int nitems, int nblocks, bool check, int reps) {
  constexpr int RC = NVP == 128 ? 64 : 32;
  const int fpad = 1024, nu = fpad / RC;
  const size_t img_bytes = (size_t)nblocks * nu * kUnitBytes;
  std::vector<unsigned char> h_img(img_bytes, 0);
  std::vector<float> h_a;  // block 0 only, for the check: [fpad][NV]
  srand(7);
  for (int b = 0; b < nblocks; ++b)
    for (int ch = 0; ch < fpad; ++ch)
      for (int k = 0; k < NV; ++k) {
        const float v = (float)((rand() % 20001) - 10000) / 10000.f / 32.f;
        if (b == 0) h_a.push_back(v);
        const unsigned short p1 = bf16_rne(v);
        const float r1 = v - bf16_val(p1);
        const unsigned short p2 = bf16_rne(r1);
        const unsigned short p3 = bf16_rne(r1 - bf16_val(p2));
        unsigned char* un = h_img.data() + ((size_t)b * nu + ch / RC) * kUnitBytes + img_elem(RC, ch % RC, k);
        memcpy(un, &p1, 2);
        memcpy(un + kPlaneBytes, &p2, 2);
        memcpy(un + 2 * kPlaneBytes, &p3, 2);
      }
  const int grid = (nitems + 7) / 8 * 8;
  std::vector<int> h_map(grid, -1);
  {  // blocks b and b + 8 share an XCD: give every XCD whole basis blocks (8 items each, as HERA-350 has)
    const int per = grid / 8;
    for (int x = 0; x < 8; ++x)
      for (int j = 0; j < per; ++j) {
        const int item = x * per + j;
        if (item < nitems) h_map[j * 8 + x] = (item / 8) % nblocks;
      }
    if (check) h_map[0] = 0;
  }
  std::vector<float> h_c((size_t)grid * 4 * NV * 32);
  for (auto& v : h_c) v = (float)((rand() % 20001) - 10000) / 10000.f;
  unsigned char* d_img;
  float *d_c, *d_g;
  int* d_map;
  CK(hipMalloc(&d_img, img_bytes));
  CK(hipMalloc(&d_c, h_c.size() * 4));
  CK(hipMalloc(&d_g, h_c.size() * 4));
  CK(hipMalloc(&d_map, grid * 4));
  CK(hipMemcpy(d_img, h_img.data(), img_bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_c, h_c.data(), h_c.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_map, h_map.data(), grid * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_g, 0, h_c.size() * 4));
  Args a{d_img, d_c, d_g, d_map, nu};
  const int lds = 2 * kUnitBytes;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&v2_skeleton<NVP, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((v2_skeleton<NVP, NV>), dim3(grid), dim3(256), lds, 0, a);
  CK(hipDeviceSynchronize());
  if (check) {
    // workgroup 0 runs block 0: grad[vec][col] = sum_ch A[ch][vec] * (sum_k A[ch][k] C[k][col]), wave 0..3
    std::vector<float> got((size_t)4 * NV * 32);
    CK(hipMemcpy(got.data(), d_g, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int w = 0; w < 4; ++w) {
      std::vector<double> v((size_t)fpad * 32, 0.0);
      for (int ch = 0; ch < fpad; ++ch)
        for (int k = 0; k < NV; ++k) {
          const double av = h_a[(size_t)ch * NV + k];
          for (int c = 0; c < 32; ++c) v[(size_t)ch * 32 + c] += av * h_c[((size_t)w * NV + k) * 32 + c];
        }
      for (int k = 0; k < NV; ++k)
        for (int c = 0; c < 32; ++c) {
          double ref = 0;
          for (int ch = 0; ch < fpad; ++ch) ref += (double)h_a[(size_t)ch * NV + k] * v[(size_t)ch * 32 + c];
          worst = fmax(worst, fabs(ref - got[((size_t)w * NV + k) * 32 + c]));
          scale = fmax(scale, fabs(ref));
        }
    }
    printf("check NVP %d NV %d: max |error| %.3e of max |value| %.3e -> %s\n", NVP, NV, worst, scale, worst <= 2e-5 * scale ? "OK" : "WRONG");
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((v2_skeleton<NVP, NV>), dim3(grid), dim3(256), lds, 0, a);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((v2_skeleton<NVP, NV>), dim3(grid), dim3(256), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = 2.0 * 2.0 * fpad * NV * 32.0 * 4 * nitems;  // useful fp32-product flops: forward + adjoint
  printf("NVP %3d NV %3d: %4d items (%d blocks, %.0f MB of images)  %.4f ms per launch = %.2f us per item-slot  %.1f useful TF\n", NVP, NV, nitems, nblocks,
         img_bytes / 1e6, ms, ms * 1e3 / ((nitems + 255) / 256), flops / ms / 1e9);
  CK(hipFree(d_img));
  CK(hipFree(d_c));
  CK(hipFree(d_g));
  CK(hipFree(d_map));
  return 0;
}

int main(int argc, char** argv) {
  // correctness of both address maps first (one item each)
  if (run<128, 64>(8, 1, true, 1)) return 1;
  if (run<128, 128>(8, 1, true, 1)) return 1;
  if (run<256, 224>(8, 1, true, 1)) return 1;
  // a pass over 955 items of ONE class each, 120 basis blocks (8 items per block): what the whole HERA-350 pass would cost were every block that wide
  const int N = 960, B = 120, reps = 10;
  if (run<128, 64>(N, B, false, reps)) return 1;
  if (run<128, 96>(N, B, false, reps)) return 1;
  if (run<128, 128>(N, B, false, reps)) return 1;
  if (run<256, 160>(N, B, false, reps)) return 1;
  if (run<256, 192>(N, B, false, reps)) return 1;
  if (run<256, 224>(N, B, false, reps)) return 1;
  return 0;
}
