// gain_reduce.hip -- what does the per-antenna reduction of gbar_G (gain_grad_kernel) cost, and what would reading every row of
// gbar_G ONCE buy?  The production kernel gives a workgroup an (antenna, channel block) and walks the antenna's list of baselines:
// every row of gbar_G is read twice (once per antenna of the baseline).  Variants timed here on the same tables:
//   v0  the production kernel (included from fit_kernels.hpp)
//   v1  antenna-tile kernel: a wave owns (tile of AB x AB antennas, channel block), walks the tile's baselines sorted by (ant0, ant1)
//       once, keeps the ant0-side sum in registers and the ant1-side sums in LDS, writes 2 AB partial rows per tile; a second kernel
//       sums an antenna's partial rows in fixed order
//   v3  one wave per (antenna, channel block), no combine through LDS
//   v0x the production reduction with channel block = blockIdx % 8 (one channel block per XCD: both reads of a row through one L2)
// Measured (MI355X, profiles/r04_micro_gain_reduce.log): none of them beats the production kernel on both shapes -- the reduction runs
// at ~5.4 TB/s of combined row reads whether the second read comes from HBM, the Infinity Cache or never happens (v1).
// Two table shapes: "full" = the 61 075 baselines of 350 antennas (one slice, shuffled row order); "share" = 8 slices x every 8th
// baseline (what a rank of the 8-GPU job holds).  A writer kernel fills gbar_G before every timed launch (as the fused kernel does).
// build: hipcc -O3 --offload-arch=gfx950 -I../../calamity_amd/csrc -o gain_reduce gain_reduce.hip ; run: ./gain_reduce
#include "fit_kernels.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace calk;

__global__ void writer(float4* q, size_t n, float s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = s * (float)((i * 2654435761u) & 1023) * (1.0f / 1024.0f) - 0.5f * s;
    q[i] = make_float4(v, -v, 0.5f * v, 0.25f * v);
  }
}

// ---- v1: tile kernel.  ent = (bl, il | jl << 8 | first-of-i-run << 16); tile t: entries [tptr[t], tptr[t + 1]), antennas I0[t] + il, J0[t] + jl
struct Tile { int e0, e1, i0, j0; };
template <int AB>
__global__ __launch_bounds__(64) void tile_kernel(const float2* __restrict__ q0, const float2* __restrict__ gains, const Tile* __restrict__ tiles,
                                                  const int2* __restrict__ ent, float2* __restrict__ part, int ntiles, int fpad) {
  typedef float vec_t __attribute__((ext_vector_type(4)));
  __shared__ vec_t s_acc[AB][64];
  const int t = blockIdx.x % ntiles, cb = blockIdx.x / ntiles;
  const int lane = threadIdx.x;
  const int f = (cb * 64 + lane) * 2;
  if (f >= fpad) return;
  const Tile T = tiles[t];
#pragma unroll
  for (int j = 0; j < AB; ++j) s_acc[j][lane] = vec_t{0, 0, 0, 0};
  vec_t ai = {0, 0, 0, 0}, gi = {0, 0, 0, 0};
  int cur_i = -1;
  float2* prow = part + (size_t)t * 2 * AB * fpad;
#pragma unroll 8
  for (int e = T.e0; e < T.e1; ++e) {
    const int2 en = ent[e];
    const int il = en.y & 255, jl = (en.y >> 8) & 255;
    const vec_t q = *reinterpret_cast<const vec_t*>(q0 + (long long)en.x * fpad + f);
    const vec_t gj = *reinterpret_cast<const vec_t*>(gains + (long long)(T.j0 + jl) * fpad + f);
    if (il != cur_i) {
      if (cur_i >= 0) *reinterpret_cast<vec_t*>(prow + (size_t)cur_i * fpad + f) = ai;
      ai = vec_t{0, 0, 0, 0};
      cur_i = il;
      gi = *reinterpret_cast<const vec_t*>(gains + (long long)(T.i0 + il) * fpad + f);
    }
    vec_t aj = s_acc[jl][lane];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float qr = q[2 * c], qi = q[2 * c + 1];
      ai[2 * c] = fmaf(-qi, gj[2 * c + 1], fmaf(qr, gj[2 * c], ai[2 * c]));
      ai[2 * c + 1] = fmaf(qi, gj[2 * c], fmaf(qr, gj[2 * c + 1], ai[2 * c + 1]));
      aj[2 * c] = fmaf(qi, gi[2 * c + 1], fmaf(qr, gi[2 * c], aj[2 * c]));
      aj[2 * c + 1] = fmaf(-qi, gi[2 * c], fmaf(qr, gi[2 * c + 1], aj[2 * c + 1]));
    }
    s_acc[jl][lane] = aj;
  }
  if (cur_i >= 0) *reinterpret_cast<vec_t*>(prow + (size_t)cur_i * fpad + f) = ai;
#pragma unroll
  for (int j = 0; j < AB; ++j) *reinterpret_cast<vec_t*>(prow + (size_t)(AB + j) * fpad + f) = s_acc[j][lane];
}
// stage 2: antenna a sums its partial rows (CSR aptr / arow: row indices into `part`), one wave per (antenna, channel block)
__global__ __launch_bounds__(64) void gather_kernel(const float2* __restrict__ part, const int* __restrict__ aptr, const int* __restrict__ arow,
                                                    float2* __restrict__ r0, int nants, int fpad) {
  typedef float vec_t __attribute__((ext_vector_type(4)));
  const int cb = blockIdx.x / nants, a = blockIdx.x - cb * nants;
  const int f = (cb * 64 + threadIdx.x) * 2;
  if (f >= fpad) return;
  vec_t s = {0, 0, 0, 0};
#pragma unroll 8
  for (int e = aptr[a]; e < aptr[a + 1]; ++e) s += *reinterpret_cast<const vec_t*>(part + (size_t)arow[e] * fpad + f);
  *reinterpret_cast<vec_t*>(r0 + (size_t)a * fpad + f) = s;
}

// ---- v3: one wave per (antenna, channel block)
template <int U>
__global__ __launch_bounds__(256) void wave_kernel(const float2* __restrict__ q0, const float2* __restrict__ gains, const int* __restrict__ ant_ptr,
                                                   const int2* __restrict__ ant_ent, float2* __restrict__ r0, int nants, int fpad, int nunits) {
  typedef float vec_t __attribute__((ext_vector_type(4)));
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (unit >= nunits) return;
  const int cb = unit / nants, a = unit - cb * nants;
  const int lane = threadIdx.x & 63;
  const int f = (cb * 64 + lane) * 2;
  if (f >= fpad) return;
  vec_t s = {0, 0, 0, 0};
  const int e0 = ant_ptr[a], e1 = ant_ptr[a + 1];
#pragma unroll U
  for (int e = e0; e < e1; ++e) {
    const int2 ent = ant_ent[e];
    const int bl = ent.x >> 1, role = ent.x & 1;
    const vec_t q = *reinterpret_cast<const vec_t*>(q0 + (long long)bl * fpad + f);
    const vec_t go = *reinterpret_cast<const vec_t*>(gains + (long long)ent.y * fpad + f);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float qr = q[2 * c], qi = role ? -q[2 * c + 1] : q[2 * c + 1];
      s[2 * c] = fmaf(-qi, go[2 * c + 1], fmaf(qr, go[2 * c], s[2 * c]));
      s[2 * c + 1] = fmaf(qi, go[2 * c], fmaf(qr, go[2 * c + 1], s[2 * c + 1]));
    }
  }
  *reinterpret_cast<vec_t*>(r0 + (size_t)a * fpad + f) = s;
}

// ---- v0x: the production reduction (antenna_sums) under another block -> (antenna, channel block) mapping: channel block = blockIdx % cblocks,
// i.e. with 8 channel blocks every XCD (workgroup b runs on XCD b % 8) owns ONE channel block and both reads of a row piece go through the same L2
template <int MAP>
__global__ __launch_bounds__(256) void mapped_kernel(const float2* __restrict__ q0, const float2* __restrict__ gains, const int* __restrict__ ant_ptr,
                                                     const int2* __restrict__ ant_ent, float2* __restrict__ r0, int nants, int fpad) {
  typedef float vec_t __attribute__((ext_vector_type(4)));
  __shared__ float s_part[3][3][64][4];
  const int cblocks = (fpad + 127) / 128;
  int cb, a;
  if (MAP == 0) { cb = blockIdx.x / nants; a = blockIdx.x - cb * nants; }
  else { cb = blockIdx.x % cblocks; a = blockIdx.x / cblocks; }
  const int lane = threadIdx.x & 63;
  const int f = (cb * 64 + lane) * 2;
  float s0[4], s1[4], s2[4];
  const bool mine = antenna_sums<float, false>(q0, q0, gains, ant_ptr, ant_ent, a, f, fpad, s_part, s0, s1, s2);
  if (mine) *reinterpret_cast<vec_t*>(r0 + (long long)a * fpad + f) = vec_t{s0[0], s0[1], s0[2], s0[3]};
}

struct Tables {
  int nants, nbls;
  std::vector<int> a0, a1;
};
static Tables make_tables(const char* shape) {
  Tables t;
  std::vector<std::pair<int, int>> pairs;
  for (int i = 0; i < 350; ++i)
    for (int j = i + 1; j < 350; ++j) pairs.push_back({i, j});
  std::mt19937 rng(7);
  if (!strstr(shape, "ordered")) std::shuffle(pairs.begin(), pairs.end(), rng);
  if (!strncmp(shape, "full", 4)) {
    t.nants = 350;
    for (auto& p : pairs) { t.a0.push_back(p.first); t.a1.push_back(p.second); }
  } else {
    t.nants = 8 * 350;
    for (int s = 0; s < 8; ++s)
      for (size_t k = 0; k < pairs.size(); k += 8) { t.a0.push_back(pairs[k].first + 350 * s); t.a1.push_back(pairs[k].second + 350 * s); }
  }
  t.nbls = (int)t.a0.size();
  return t;
}

template <typename F> static float time_it(const char* name, float2* q, size_t qn, F launch, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f, tot = 0;
  const int reps = 8;
  for (int r = 0; r < reps + 2; ++r) {
    hipLaunchKernelGGL(writer, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4*>(q), qn / 2, 1.0f + 0.01f * r);
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) { best = std::min(best, ms); tot += ms; }
  }
  CK(hipGetLastError());
  printf("  %-34s best %.4f ms  avg %.4f ms  (one read of gbar_G at that rate: %.2f TB/s)\n", name, best, tot / reps, bytes / (best * 1e-3) / 1e12);
  return best;
}

template <int AB> static void run_tiles(const Tables& t, float2* q, size_t qn, float2* gains, float2* r0, std::vector<float>& ref, int fpad, double bytes) {
  // tiles over global antenna blocks
  const int nb = (t.nants + AB - 1) / AB;
  std::vector<int> order(t.nbls);
  std::iota(order.begin(), order.end(), 0);
  auto key = [&](int b) { return ((long long)(t.a0[b] / AB) * nb + t.a1[b] / AB); };
  std::sort(order.begin(), order.end(), [&](int x, int y) {
    const long long kx = key(x), ky = key(y);
    if (kx != ky) return kx < ky;
    if (t.a0[x] != t.a0[y]) return t.a0[x] < t.a0[y];
    return t.a1[x] < t.a1[y];
  });
  std::vector<Tile> tiles;
  std::vector<int2> ent(t.nbls);
  long long prev = -1;
  for (int k = 0; k < t.nbls; ++k) {
    const int b = order[k];
    if (key(b) != prev) {
      if (!tiles.empty()) tiles.back().e1 = k;
      tiles.push_back({k, k, (t.a0[b] / AB) * AB, (t.a1[b] / AB) * AB});
      prev = key(b);
    }
    ent[k] = make_int2(b, (t.a0[b] % AB) | ((t.a1[b] % AB) << 8));
  }
  tiles.back().e1 = t.nbls;
  const int ntiles = (int)tiles.size();
  // heaviest tiles first
  std::stable_sort(tiles.begin(), tiles.end(), [](const Tile& x, const Tile& y) { return x.e1 - x.e0 > y.e1 - y.e0; });
  // stage-2 lists: antenna a <- (tile, slot) rows that received something
  std::vector<std::vector<int>> rows(t.nants);
  for (int ti = 0; ti < ntiles; ++ti) {
    std::vector<char> si(AB, 0), sj(AB, 0);
    for (int e = tiles[ti].e0; e < tiles[ti].e1; ++e) { si[ent[e].y & 255] = 1; sj[(ent[e].y >> 8) & 255] = 1; }
    for (int k = 0; k < AB; ++k) {
      if (si[k]) rows[tiles[ti].i0 + k].push_back(ti * 2 * AB + k);
      if (tiles[ti].j0 + k < t.nants) rows[tiles[ti].j0 + k].push_back(ti * 2 * AB + AB + k);  // (zero rows where nothing arrived)
    }
  }
  std::vector<int> aptr(t.nants + 1, 0), arow;
  for (int a = 0; a < t.nants; ++a) { aptr[a + 1] = aptr[a] + (int)rows[a].size(); arow.insert(arow.end(), rows[a].begin(), rows[a].end()); }
  Tile* d_tiles; int2* d_ent; int *d_aptr, *d_arow; float2* d_part;
  const size_t pbytes = (size_t)ntiles * 2 * AB * fpad * sizeof(float2);
  CK(hipMalloc(&d_tiles, ntiles * sizeof(Tile)));
  CK(hipMalloc(&d_ent, ent.size() * sizeof(int2)));
  CK(hipMalloc(&d_aptr, aptr.size() * sizeof(int)));
  CK(hipMalloc(&d_arow, arow.size() * sizeof(int)));
  CK(hipMalloc(&d_part, pbytes));
  CK(hipMemset(d_part, 0, pbytes));
  CK(hipMemcpy(d_tiles, tiles.data(), ntiles * sizeof(Tile), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ent, ent.data(), ent.size() * sizeof(int2), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_aptr, aptr.data(), aptr.size() * sizeof(int), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_arow, arow.data(), arow.size() * sizeof(int), hipMemcpyHostToDevice));
  const int cblocks = (fpad + 127) / 128;
  char name[96];
  snprintf(name, sizeof name, "v1 tiles AB=%d (%d tiles, %.0f MB partial)", AB, ntiles, pbytes / 1e6);
  CK(hipMemset(r0, 0, (size_t)t.nants * fpad * sizeof(float2)));
  time_it(name, q, qn, [&] {
    hipLaunchKernelGGL(tile_kernel<AB>, dim3(ntiles * cblocks), dim3(64), 0, 0, q, gains, d_tiles, d_ent, d_part, ntiles, fpad);
    hipLaunchKernelGGL(gather_kernel, dim3(t.nants * cblocks), dim3(64), 0, 0, d_part, d_aptr, d_arow, r0, t.nants, fpad);
  }, bytes);
  snprintf(name, sizeof name, "   (its first kernel alone)");
  time_it(name, q, qn, [&] { hipLaunchKernelGGL(tile_kernel<AB>, dim3(ntiles * cblocks), dim3(64), 0, 0, q, gains, d_tiles, d_ent, d_part, ntiles, fpad); }, bytes);
  std::vector<float> out((size_t)t.nants * fpad * 2);
  CK(hipMemcpy(out.data(), r0, out.size() * sizeof(float), hipMemcpyDeviceToHost));
  double err = 0, nrm = 0;
  for (size_t k = 0; k < out.size(); ++k) { err += (double)(out[k] - ref[k]) * (out[k] - ref[k]); nrm += (double)ref[k] * ref[k]; }
  printf("     relative difference from v0: %.2e\n", std::sqrt(err / nrm));
  CK(hipFree(d_tiles)); CK(hipFree(d_ent)); CK(hipFree(d_aptr)); CK(hipFree(d_arow)); CK(hipFree(d_part));
}

int main() {
  const int fpad = 1024;
  for (const char* shape : {"full", "full-ordered", "share"}) {
    Tables t = make_tables(shape);
    printf("%s: %d antennas, %d baselines, %d channels\n", shape, t.nants, t.nbls, fpad);
    const size_t qn = (size_t)t.nbls * fpad;  // float2 elements
    const double bytes = (double)qn * sizeof(float2);
    float2 *q, *gains, *r0;
    CK(hipMalloc(&q, qn * sizeof(float2)));
    CK(hipMalloc(&gains, (size_t)t.nants * fpad * sizeof(float2)));
    CK(hipMalloc(&r0, 3 * (size_t)t.nants * fpad * sizeof(float2)));
    hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, 0, reinterpret_cast<float4*>(gains), (size_t)t.nants * fpad / 2, 2.0f);
    std::vector<int> ptr(t.nants + 1, 0);
    for (int b = 0; b < t.nbls; ++b) { ptr[t.a0[b] + 1]++; ptr[t.a1[b] + 1]++; }
    for (int a = 0; a < t.nants; ++a) ptr[a + 1] += ptr[a];
    std::vector<int2> ent(2 * (size_t)t.nbls);
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int b = 0; b < t.nbls; ++b) {
      ent[fill[t.a0[b]]++] = make_int2(b * 2 + 0, t.a1[b]);
      ent[fill[t.a1[b]]++] = make_int2(b * 2 + 1, t.a0[b]);
    }
    int* d_ptr; int2* d_ent; DevState* d_st; double *d_part, *d_scal;
    CK(hipMalloc(&d_ptr, ptr.size() * sizeof(int)));
    CK(hipMalloc(&d_ent, ent.size() * sizeof(int2)));
    CK(hipMalloc(&d_st, 8 * sizeof(DevState)));
    CK(hipMalloc(&d_part, 4096));
    CK(hipMalloc(&d_scal, 4096));
    CK(hipMemset(d_st, 0, 8 * sizeof(DevState)));
    CK(hipMemset(d_part, 0, 4096));
    CK(hipMemcpy(d_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ent, ent.data(), ent.size() * sizeof(int2), hipMemcpyHostToDevice));
    SliceMap M{nullptr, nullptr, nullptr, 1, t.nants};
    const int cblocks = (fpad + 127) / 128;
    time_it("v0 production gain_grad_kernel", q, qn, [&] {
      hipLaunchKernelGGL((gain_grad_kernel<float, false>), dim3(t.nants * cblocks + 1), dim3(256), 0, 0, q, q, gains, d_ptr, d_ent, r0, r0, r0, t.nants, fpad,
                         d_part, 1, d_scal, d_st, M);
    }, bytes);
    std::vector<float> ref((size_t)t.nants * fpad * 2);
    CK(hipMemcpy(ref.data(), r0, ref.size() * sizeof(float), hipMemcpyDeviceToHost));
    time_it("v0  same reduction, cb = b / nants", q, qn, [&] {
      hipLaunchKernelGGL(mapped_kernel<0>, dim3(t.nants * cblocks), dim3(256), 0, 0, q, gains, d_ptr, d_ent, r0, t.nants, fpad);
    }, bytes);
    time_it("v0x same reduction, cb = b % cblocks", q, qn, [&] {
      hipLaunchKernelGGL(mapped_kernel<1>, dim3(t.nants * cblocks), dim3(256), 0, 0, q, gains, d_ptr, d_ent, r0, t.nants, fpad);
    }, bytes);
    const int nunits = t.nants * cblocks;
    time_it("v3 one wave per unit, unroll 8", q, qn, [&] {
      hipLaunchKernelGGL(wave_kernel<8>, dim3((nunits + 3) / 4), dim3(256), 0, 0, q, gains, d_ptr, d_ent, r0, t.nants, fpad, nunits);
    }, bytes);
    time_it("v3 one wave per unit, unroll 16", q, qn, [&] {
      hipLaunchKernelGGL(wave_kernel<16>, dim3((nunits + 3) / 4), dim3(256), 0, 0, q, gains, d_ptr, d_ent, r0, t.nants, fpad, nunits);
    }, bytes);
    run_tiles<8>(t, q, qn, gains, r0, ref, fpad, bytes);
    run_tiles<16>(t, q, qn, gains, r0, ref, fpad, bytes);
    run_tiles<32>(t, q, qn, gains, r0, ref, fpad, bytes);
    CK(hipFree(q)); CK(hipFree(gains)); CK(hipFree(r0)); CK(hipFree(d_ptr)); CK(hipFree(d_ent)); CK(hipFree(d_st)); CK(hipFree(d_part)); CK(hipFree(d_scal));
  }
  return 0;
}
