// mfma_kernels.hpp -- dense (matrix-core) formulation of the same step for the SHARED layout, fp32.
//
// When every baseline of a distinct delay shares ONE basis block A_u (the operator cache of
// /root/reference/calamity/modeling.py:291-301; 120 blocks / 68 MB at HERA-350), fg_model (calibration.py:1587-1590) over
// all those baselines is a real dense contraction:   V^T (2 nb x F) = C^T (2 nb x nvec) @ A_u^T (nvec x F)
// and its adjoint                                     dC^T (2 nb x nvec) = Gbar^T (2 nb x F) @ A_u (F x nvec)
// (BASELINE config 5).  Both run here on v_mfma_f32_32x32x2_f32 -- exact fp32, bit-for-bit a
// k-ordered fmaf chain -- with everything between them (gains, residual, chi^2, gbar_v, gbar_G; calibration.py:1593-1609
// and their adjoints) fused in registers, in the MFMA accumulator layout.
//
// Work item = a PANEL of 16 baselines with the same basis: 32 MFMA rows = (baseline, re|im).  Channels are the MFMA
// column (lane) dimension, so data / weights / gains loads and the gbar_G store are coalesced along frequency.
// Per 128-channel chunk: forward (each wave 32 channels, K = nvec) -> element-wise -> Gbar chunk to LDS -> adjoint
// (K = the chunk's channels; waves split the vector tiles, or the K range when there are fewer tiles than waves).
#pragma once
#include "fit_kernels.hpp"

namespace calk {

constexpr int kPanel = 16;        // baselines per panel
constexpr int kChunk = 128;       // channels per chunk (4 waves x 32)
constexpr int kMaxNT = 8;         // vector tiles of 32 the adjoint supports (nvec <= 256)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// which solver dtypes have a dense kernel, and the widest basis block it takes
template <typename T> struct DenseCfg { static constexpr int max_nvec = 0; };
template <> struct DenseCfg<float> { static constexpr int max_nvec = 32 * kMaxNT; };

struct PanelItem {
  int bl[kPanel];     // baseline ids (-1: padding slot)
  int nvec, nvp2, nvp32;
  int pad;
  long long a_kf4;    // packed forward operand  [F/32][nvp8/8][64 lanes][4]: lane (col, half), u -> A[32 fb + col][8 g + 2 u + half]
  long long a_fk4;    // packed adjoint operand  [F/8][nvp32/32][64 lanes][4]: lane (col, half), u -> A[8 cg + 2 u + half][32 t + col]
};

struct MfmaArgs {
  const float* a_kf4;
  const float* a_fk4;
  const PanelItem* panels;
  const int2* bl_ant;
  const int* bl_coff;          // coefficient offset of each baseline's group
  const float* data_r;         // [nbls][fpad]
  const float* data_i;
  const float* wgts;
  const float2* gains;         // [nants][fpad]
  const float* c_r;
  const float* c_i;
  float2* q0;                  // [nbls][fpad]
  float* gc_r;                 // [ncoef] final coefficient gradient (every baseline owns its coefficients)
  float* gc_i;
  double* part;                // [npanels][4]
  const DevState* state;
  int fpad;
  int use_alpha;               // "sum" regulariser, second pass: e = -2 w r + alpha w with alpha = 2 (S - P) read from state
  int nbls;                    // row nbls of data_r / data_i / wgts / q0 is an all-zero spare row for padding slots
};

// ----------------------------------------------------------------------------------------------------------------
// Wave specialisation.  A wave's vector-memory operations retire in issue order, so a wave
// that mixes L2-served MFMA operand loads with HBM loads (data, weights) stalls its matrix pipe on the slow ones.
// Here a 512-thread workgroup has 4 MATRIX waves (waves 0-3: forward GEMM, adjoint GEMM; they only ever load basis
// operands, which live in L2) and 4 ELEMENT waves (waves 4-7, one per SIMD beside a matrix wave: HBM loads of data /
// weights / gains one chunk ahead, residual, chi^2, gbar_v, gbar_G store).  They meet in LDS:
//   tick k:  matrix waves   F(k): V chunk k -> s_v[k&1]          and  B(k-2): adjoint with s_g[(k-2)&1]
//            element waves  E(k-1): s_v[(k-1)&1] -> s_g[(k-1)&1], then request chunk k's inputs
//   one workgroup barrier per tick; VALU work of the element wave overlaps the MFMAs of the matrix wave on its SIMD.
#ifndef CAL_WS_THREADS
#define CAL_WS_THREADS 512
#endif
constexpr int kWsThreads = CAL_WS_THREADS;  // 256 = timing experiment without element waves (results are wrong)
#ifdef CAL_WS_STAMP
// diagnostic build only: per-wave cycle stamps of a few workgroups (never read by the kernels)
__device__ long long g_ws_stamps[8][8][16][4];  // [block slot][wave][tick][stamp]
#define WS_STAMP(blk, tick, idx)                                                                                        \
  do {                                                                                                                  \
    if ((blk) >= 0 && lane == 0 && (tick) < 16) g_ws_stamps[blk][wave][tick][idx] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define WS_STAMP(blk, tick, idx)
#endif
constexpr int kSW = 36;  // LDS row stride of the V / gbar chunk buffers [channel][32 rows]: 16-B aligned rows, and 36 f mod 64
                         // walks the 16 bank quads, so ds_read/write_b128 by 16 consecutive channels are conflict-free


// ---- hand-scheduled operand stream of the matrix waves.
// hipcc places the s_waitcnt for a load in front of its first use, and around loop back-edges it falls back to draining the
// queue (vmcnt(0)) -- measured: every 4-MFMA group then exposes a full L2 round trip (~500-900 cycles) and the matrix pipe
// idles half of the time.  The basis operands are therefore requested with inline-asm loads the compiler does not track, a
// fixed number of groups ahead (ring of kRing register quads), and consumed behind explicit counted waits.
constexpr int kRing = 6;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ring_load(f32x4_t& dst, const f32x4_t* sbase, unsigned voff_bytes) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
#define RING_WAIT()                                                  \
  do {                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRing - 1) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
// Every phase ends with kRing requests whose data nobody reads (they keep the count in RING_WAIT constant).  They must have
// landed before the next phase requests into the same registers: two loads in flight to ONE register are not guaranteed
// to write it in issue order (the compiler never creates that situation: it waits on such a write-after-write), and a
// late discarded load overwrote fresh operands -- gradients that differed from run to run, but only once the element
// waves stopped being the slow side of every barrier, which until then had given those requests time to land.
#define RING_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// A ring register whose last request nobody reads is, to the compiler, free from that request on -- but the data lands
// later, into whatever the register has been given to meanwhile.  RING_KEEP (placed behind a drain) reads all six, so they
// stay allocated until their last request has landed.
#define RING_KEEP() asm volatile("" ::"v"(R0), "v"(R1), "v"(R2), "v"(R3), "v"(R4), "v"(R5))

template <bool GRAD>
__global__ __launch_bounds__(kWsThreads) void fused_mfma_ws_kernel(const MfmaArgs A) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (A.state->done | A.state->done_after) return;
  // XCD-aware block -> panel map (see the host): the 8 XCDs each walk their own contiguous list of panels
  const int per_xcd = gridDim.x >> 3;
  const PanelItem& P = A.panels[(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)];
  if (P.nvec == 0) {  // padding panel of a short XCD list
    if (threadIdx.x == 0) A.part[(size_t)blockIdx.x * 4] = A.part[(size_t)blockIdx.x * 4 + 1] = A.part[(size_t)blockIdx.x * 4 + 2] = 0.0;
    return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave-uniform quantities live in SGPRs: every operand address below is (scalar base) + (lane offset)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_matrix = wave < 4;
  const int w4 = wave & 3;     // channel quarter of the chunk handled by this wave (both roles)
  const int col = lane & 31;
  const int half = lane >> 5;
  const int nvec = P.nvec, nvp32 = P.nvp32;
  const int NT = nvp32 / 32;
  const int ngk = (nvec + 7) / 8;  // forward k-groups of 8 vectors (4 k-steps)

  float* s_c = reinterpret_cast<float*>(smem_raw);          // [ngk][64 lanes][4] packed coefficient operand
  float* s_v = s_c + (size_t)ngk * 256;                     // [2][kChunk][kSW]  model visibilities, rows = (slot, re|im)
  float* s_g = s_v + 2 * (size_t)kChunk * kSW;              // [2][kChunk][kSW]  gbar_v
  int* s_bl = reinterpret_cast<int*>(s_g + 2 * (size_t)kChunk * kSW);  // [16] baseline, [16] ant0, [16] ant1
  double* s_red = reinterpret_cast<double*>(s_bl + 48);     // [12]: loss, S_r, S_i partials of the 4 element waves

  if (tid < kPanel) {
    const int b = P.bl[tid];
    s_bl[tid] = b >= 0 ? b : A.nbls;  // padding slots use the all-zero extra row (weight 0) of the sample arrays
    const int2 ant = b >= 0 ? A.bl_ant[b] : make_int2(0, 0);
    s_bl[16 + tid] = ant.x;
    s_bl[32 + tid] = ant.y;
  }
  {
    // coefficient panel in the packed A-operand layout: s_c[(g * 64 + lane) * 4 + u] = c_part[row = lane & 31][k = 8 g + 2 u + (lane >> 5)],
    // rows 0-15 re, 16-31 im of the panel slots; zero beyond nvec and for padding slots
    const int n = ngk * 256;
    for (int i = tid; i < n; i += kWsThreads) {
      const int u = i & 3, l = (i >> 2) & 63, g = i >> 8;
      const int row = l & 31, k = 8 * g + 2 * u + (l >> 5);
      const int b = P.bl[row & 15];
      float v = 0.f;
      if (b >= 0 && k < nvec) v = (row < 16 ? A.c_r : A.c_i)[A.bl_coff[b] + k];
      s_c[i] = v;
    }
  }
  __syncthreads();

#ifdef CAL_MF_DEBUG_CHUNKS
  const int nchunks = CAL_MF_DEBUG_CHUNKS;  // timing experiment only (results are wrong)
#else
  const int nchunks = A.fpad / kChunk;
#endif
  const int nticks = nchunks + (GRAD ? 2 : 1);
#ifdef CAL_WS_STAMP
  const int sblk = (blockIdx.x >= 2048 && blockIdx.x < 2056) ? (int)blockIdx.x - 2048 : -1;
#endif

  if (is_matrix) {
    // ================================================= matrix waves ==============================================
    __builtin_amdgcn_s_setprio(2);  // the matrix wave's issue slots come first on the SIMD it shares with an element wave
    int t0, t1 = -1, kq = 0, nkq = 1;
    if (NT == 1) { t0 = 0; kq = w4; nkq = 4; }
    else if (NT == 2) { t0 = w4 & 1; kq = w4 >> 1; nkq = 2; }
    else if (NT <= 4) { t0 = w4 < NT ? w4 : -1; }
    else { t0 = w4; t1 = w4 + 4 < NT ? w4 + 4 : -1; }
    f32x16 gacc0, gacc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) gacc0[j] = gacc1[j] = 0.f;
    const f32x4* akf4 = reinterpret_cast<const f32x4*>(A.a_kf4 + P.a_kf4);   // scalar bases
    const f32x4* afk4 = reinterpret_cast<const f32x4*>(A.a_fk4 + P.a_fk4);
    const f32x4* sc4 = reinterpret_cast<const f32x4*>(s_c);
    const unsigned voff = (unsigned)lane * 16u;
    f32x4 R0, R1, R2, R3, R4, R5;  // the operand ring (kRing = 6 register quads)
    R0 = R1 = R2 = R3 = R4 = R5 = f32x4{0.f, 0.f, 0.f, 0.f};
    // items of the adjoint phase of this wave (wave constants) and the operand address of an item of either phase; both
    // clamp the item index, so an address always lies inside this panel's packed operand blocks
    const int ng_per = (kChunk / 8) / nkq;  // channel groups (8 channels = 4 k-steps) of this wave: 16, 8 or 4
    const int cg0 = kq * ng_per;
    const int two = t1 >= 0 ? 1 : 0;
    const int nitem = ng_per << two;
    const bool has_b = GRAD && t0 >= 0;
    auto f_addr = [&](int kk, int g) { return akf4 + ((size_t)(kk * 4 + w4) * ngk + (g < ngk ? g : ngk - 1)) * 64; };
    auto b_addr = [&](int ci, int i) {
      i = i < nitem ? i : nitem - 1;
      const int cg = cg0 + (i >> two), t = (i & two) ? t1 : t0;
      return afk4 + ((size_t)(ci * (kChunk / 8) + cg) * NT + t) * 64;
    };
    // Phase chaining: the last kRing requests of a phase (which used to fetch data nobody reads, only to keep the count of
    // RING_WAIT constant) fetch the FIRST kRing items of the wave's next phase -- across the barrier too, the packed basis
    // does not depend on the element waves -- so that phase starts with its operands in flight instead of an exposed L2
    // round trip.  No register ever has two loads in flight.  nx_kind: 0 none, 1 forward of chunk nx_k, 2 adjoint of nx_k.
    auto first_phase_of_tick = [&](int kk, int& kind, int& kx) {
      kind = 0;
      kx = 0;
      if (!GRAD) return;  // the forward-only pass is bound by its element waves: chaining only adds scalar work there
      if (kk >= nticks) return;
      if (kk < nchunks) { kind = 1; kx = kk; }
      else if (has_b && kk >= 2) { kind = 2; kx = kk - 2; }
    };
    bool primed = false;  // the ring already holds (in flight) the first kRing items of the phase about to start
    for (int k = 0; k < nticks; ++k) {
      WS_STAMP(sblk, k, 0);
      if (k < nchunks) {
        // ---- F(k): rows = (slot, re|im), columns = this wave's 32 channels, K = vectors: item = k-group g (4 MFMAs)
        f32x16 acc, acc2;  // two accumulator chains
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = acc2[j] = 0.f;
        const f32x4* tile = akf4 + ((size_t)(k * 4 + w4) * ngk) * 64;  // scalar base; item g at + 64 g
        auto src = [&](int g) { return tile + (g < ngk ? g : ngk - 1) * 64; };
        int nx_kind, nx_k;
        if (has_b && k >= 2) { nx_kind = 2; nx_k = k - 2; }
        else first_phase_of_tick(k + 1, nx_kind, nx_k);
        auto nxt = [&](int j) { return nx_kind == 1 ? f_addr(nx_k, j) : nx_kind == 2 ? b_addr(nx_k, j) : src(ngk - 1); };
        if (!primed) {
          RING_DRAIN();
          RING_KEEP();
          ring_load(R0, src(0), voff);
          ring_load(R1, src(1), voff);
          ring_load(R2, src(2), voff);
          ring_load(R3, src(3), voff);
          ring_load(R4, src(4), voff);
          ring_load(R5, src(5), voff);
        }
        primed = nx_kind != 0;
        f32x4 a_cur = (sc4 + 0)[lane], a_nxt;
#define F_STEP(RJ, J, NEXTSRC)                                                                   \
  {                                                                                              \
    const int g = g0 + (J);                                                                      \
    a_nxt = (sc4 + (g + 1 < ngk ? g + 1 : ngk - 1) * 64)[lane];                                  \
    RING_WAIT();                                                                                 \
    if (g < ngk) {                                                                               \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], RJ[0], acc, 0, 0, 0);                 \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], RJ[1], acc2, 0, 0, 0);               \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], RJ[2], acc, 0, 0, 0);                 \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], RJ[3], acc2, 0, 0, 0);               \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    ring_load(RJ, NEXTSRC, voff);                                                                \
    a_cur = a_nxt;                                                                               \
  }
        int g0 = 0;
        for (; g0 + kRing < ngk; g0 += kRing) {
          F_STEP(R0, 0, src(g + kRing))
          F_STEP(R1, 1, src(g + kRing))
          F_STEP(R2, 2, src(g + kRing))
          F_STEP(R3, 3, src(g + kRing))
          F_STEP(R4, 4, src(g + kRing))
          F_STEP(R5, 5, src(g + kRing))
        }
        // last group of steps: its requests belong to the next phase
        F_STEP(R0, 0, nxt(0))
        F_STEP(R1, 1, nxt(1))
        F_STEP(R2, 2, nxt(2))
        F_STEP(R3, 3, nxt(3))
        F_STEP(R4, 4, nxt(4))
        F_STEP(R5, 5, nxt(5))
#undef F_STEP
        // accumulator regs 4 q .. 4 q + 3 of this lane = rows 8 q + 4 half + 0..3 of column (channel) w4*32 + col
        f32x4* vout = reinterpret_cast<f32x4*>(s_v + (size_t)(k & 1) * kChunk * kSW + (w4 * 32 + col) * kSW + 4 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t;
          t[0] = acc[4 * q] + acc2[4 * q];
          t[1] = acc[4 * q + 1] + acc2[4 * q + 1];
          t[2] = acc[4 * q + 2] + acc2[4 * q + 2];
          t[3] = acc[4 * q + 3] + acc2[4 * q + 3];
          vout[2 * q] = t;  // + 8 rows
        }
      }
      WS_STAMP(sblk, k, 1);
      if (GRAD && k >= 2 && t0 >= 0) {
        // ---- B(k-2): rows = (slot, re|im), columns = vectors of tile t, K = this wave's slice of the chunk's channels.
        // item i = (channel group cg0 + i / ntl, tile i % ntl), ntl = 1 or 2 vector tiles per wave; 4 MFMAs per item
        const int ci = k - 2;
        const float* gp = s_g + (size_t)(ci & 1) * kChunk * kSW + half * kSW + col;       // + (8 cg + 2 v) kSW
        const f32x4* bp = afk4 + ((size_t)(ci * (kChunk / 8)) * NT) * 64;                  // scalar; + (cg NT + t) 64
        auto src = [&](int i) {
          i = i < nitem ? i : nitem - 1;
          const int cg = cg0 + (i >> two), t = (i & two) ? t1 : t0;
          return bp + ((size_t)cg * NT + t) * 64;
        };
        auto lds_a = [&](int i, float (&a)[4]) {
          i = i < nitem ? i : nitem - 1;
          const int cg = cg0 + (i >> two);
#pragma unroll
          for (int v = 0; v < 4; ++v) a[v] = gp[(8 * cg + 2 * v) * kSW];
        };
        int nx_kind, nx_k;
        first_phase_of_tick(k + 1, nx_kind, nx_k);
        auto nxt = [&](int j) { return nx_kind == 1 ? f_addr(nx_k, j) : nx_kind == 2 ? b_addr(nx_k, j) : src(nitem - 1); };
        if (!primed) {
          RING_DRAIN();
          RING_KEEP();
          ring_load(R0, src(0), voff);
          ring_load(R1, src(1), voff);
          ring_load(R2, src(2), voff);
          ring_load(R3, src(3), voff);
          ring_load(R4, src(4), voff);
          ring_load(R5, src(5), voff);
        }
        primed = nx_kind != 0;
        float a_cur[4], a_nxt[4];
        lds_a(0, a_cur);
        // even items accumulate into gacc0, odd items into gacc1: with two tiles that is tile t0 / t1, with one tile two
        // independent chains of the same tile (kRing is even, so the parity of an item is the parity of its ring slot)
#define B_STEP(RJ, J, GACC, NEXTSRC)                                                             \
  {                                                                                              \
    const int it = i0 + (J);                                                                     \
    lds_a(it + 1, a_nxt);                                                                        \
    RING_WAIT();                                                                                 \
    if (it < nitem) {                                                                            \
      GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], RJ[0], GACC, 0, 0, 0);               \
      GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], RJ[1], GACC, 0, 0, 0);               \
      GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], RJ[2], GACC, 0, 0, 0);               \
      GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], RJ[3], GACC, 0, 0, 0);               \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    ring_load(RJ, NEXTSRC, voff);                                                                \
    a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; a_cur[2] = a_nxt[2]; a_cur[3] = a_nxt[3];          \
  }
        int i0 = 0;
        for (; i0 + kRing < nitem; i0 += kRing) {
          B_STEP(R0, 0, gacc0, src(it + kRing))
          B_STEP(R1, 1, gacc1, src(it + kRing))
          B_STEP(R2, 2, gacc0, src(it + kRing))
          B_STEP(R3, 3, gacc1, src(it + kRing))
          B_STEP(R4, 4, gacc0, src(it + kRing))
          B_STEP(R5, 5, gacc1, src(it + kRing))
        }
        // last group of steps: its requests belong to the next phase
        B_STEP(R0, 0, gacc0, nxt(0))
        B_STEP(R1, 1, gacc1, nxt(1))
        B_STEP(R2, 2, gacc0, nxt(2))
        B_STEP(R3, 3, gacc1, nxt(3))
        B_STEP(R4, 4, gacc0, nxt(4))
        B_STEP(R5, 5, gacc1, nxt(5))
#undef B_STEP
      }
      WS_STAMP(sblk, k, 2);
      __syncthreads();
      WS_STAMP(sblk, k, 3);
#ifdef CAL_WS_STAMP
      if (sblk >= 0 && lane == 0 && (k == 0 || k == nticks - 1)) { g_ws_stamps[sblk][wave][k == 0 ? 14 : 15][0] = (long long)__builtin_amdgcn_s_memrealtime(); g_ws_stamps[sblk][wave][14][1] = nvec; g_ws_stamps[sblk][wave][14][2] = NT; g_ws_stamps[sblk][wave][14][3] = t0 * 100 + t1 * 10 + nkq; }
#endif
    }
    RING_DRAIN();  // retire the last phase's trailing requests
    RING_KEEP();
    if (!GRAD) return;
    if (t1 < 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) gacc0[j] += gacc1[j];
    }
    // ---- coefficient gradients: sum the K-slices of different matrix waves through LDS, then store
    if (nkq > 1) {
      float* s_x = s_v;  // [4 waves][16 regs][64 lanes] = 16 KB, inside the (now idle) s_v double buffer
#pragma unroll
      for (int j = 0; j < 16; ++j) s_x[(w4 * 16 + j) * 64 + lane] = gacc0[j];
    }
    __syncthreads();  // pairs with the element waves' barrier after their tick loop
    if (nkq > 1 && kq == 0) {
      const float* s_x = s_v;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = gacc0[j];
        for (int q = 1; q < nkq; ++q) v += s_x[((w4 + q * (NT == 1 ? 1 : 2)) * 16 + j) * 64 + lane];
        gacc0[j] = v;
      }
    }
    if (kq == 0 && t0 >= 0) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = tt == 0 ? t0 : t1;
        if (t < 0) continue;
        const int n = t * 32 + col;
        if (n < nvec) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int slot = (j & 3) + 8 * (j >> 2) + 4 * half;
            const int b = P.bl[slot];
            if (b >= 0) {
              const int coff = A.bl_coff[b];
              A.gc_r[coff + n] = tt == 0 ? gacc0[j] : gacc1[j];
              A.gc_i[coff + n] = tt == 0 ? gacc0[j + 8] : gacc1[j + 8];
            }
          }
        }
      }
    }
  } else {
    // ================================================= element waves =============================================
    // lane = (channel w4*32 + col of the chunk, slot group half): slots half*8 .. half*8 + 7.  No branches: padding slots
    // point at the all-zero extra row.  Sample offsets are unsigned 32-bit lane offsets from scalar bases.
    struct ElemIn {
      float dr[8], di[8], w[8];
      float2 g0[8], g1[8];
    };
    ElemIn X, Y;
    unsigned ob[8], og0[8], og1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int slot = half * 8 + j;
      // BYTE offsets (of a float for ob, of a float2 for og0 / og1): scalar base + unsigned 32-bit byte offset is the form
      // hipcc turns into `global_load v, v_off, s[base]`; with element offsets it builds a 64-bit address pair per load
      // (207 v_lshl_add_u64 and 40 registers spilled to scratch in this branch before)
      ob[j] = ((unsigned)s_bl[slot] * (unsigned)A.fpad + (unsigned)(w4 * 32 + col)) * 4u;
      og0[j] = ((unsigned)s_bl[16 + slot] * (unsigned)A.fpad + (unsigned)(w4 * 32 + col)) * 8u;
      og1[j] = ((unsigned)s_bl[32 + slot] * (unsigned)A.fpad + (unsigned)(w4 * 32 + col)) * 8u;
    }
    auto request = [&](int ci, ElemIn& E) {
      // address = array base (kernel argument, scalar) + 32-bit byte offset (lane offset + chunk offset), added per
      // load.  Written with the chunk folded into the base instead, loop-invariant code motion turned every load address
      // into a loop-invariant 64-bit VGPR pair: 80 registers, spilled to scratch and reloaded behind a full vmcnt(0)
      // wait in front of every load -- the element waves ran their loads one at a time.
      const char* dr = reinterpret_cast<const char*>(A.data_r);
      const char* di = reinterpret_cast<const char*>(A.data_i);
      const char* wg = reinterpret_cast<const char*>(A.wgts);
      const char* gn = reinterpret_cast<const char*>(A.gains);
      const unsigned c4 = (unsigned)ci * (kChunk * 4u), c8 = (unsigned)ci * (kChunk * 8u);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        E.dr[j] = *reinterpret_cast<const float*>(dr + (ob[j] + c4));
        E.di[j] = *reinterpret_cast<const float*>(di + (ob[j] + c4));
        E.w[j] = *reinterpret_cast<const float*>(wg + (ob[j] + c4));
        E.g0[j] = *reinterpret_cast<const float2*>(gn + (og0[j] + c8));
        E.g1[j] = *reinterpret_cast<const float2*>(gn + (og1[j] + c8));
      }
    };
    double loss_acc = 0.0, sr_acc = 0.0, si_acc = 0.0;
    const float al_r = A.use_alpha ? (float)A.state->alpha_r : 0.f, al_i = A.use_alpha ? (float)A.state->alpha_i : 0.f;
    auto process = [&](int ci, const ElemIn& E) {
      const f32x4* vin = reinterpret_cast<const f32x4*>(s_v + (size_t)(ci & 1) * kChunk * kSW + (w4 * 32 + col) * kSW + half * 8);
      f32x4* gout = reinterpret_cast<f32x4*>(s_g + (size_t)(ci & 1) * kChunk * kSW + (w4 * 32 + col) * kSW + half * 8);
      char* qo = reinterpret_cast<char*>(A.q0);
      const unsigned qc8 = (unsigned)ci * (kChunk * 8u);
      float lt = 0.f, st_r = 0.f, st_i = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 vr4 = vin[h], vi4 = vin[4 + h];  // rows half*8 + 4 h + 0..3 (re) and +16 (im)
        f32x4 gr4, gi4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = 4 * h + u;
          const float vr = vr4[u], vi = vi4[u];
          const float d_r = E.dr[j], d_i = E.di[j], w = E.w[j];
          const float2 g0 = E.g0[j], g1 = E.g1[j];
          const float G_r = g0.x * g1.x + g0.y * g1.y;
          const float G_i = g0.y * g1.x - g0.x * g1.y;
          const float m_r = G_r * vr - G_i * vi;
          const float m_i = G_i * vr + G_r * vi;
          const float r_r = d_r - m_r, r_i = d_i - m_i;
          lt += w * (r_r * r_r + r_i * r_i);
          st_r += w * m_r;  // S = sum w m of the "sum" regulariser (calibration.py:1648-1649)
          st_i += w * m_i;
          if (GRAD) {
            const float e_r = -2.f * w * r_r + al_r * w, e_i = -2.f * w * r_i + al_i * w;
            gr4[u] = G_r * e_r + G_i * e_i;
            gi4[u] = G_r * e_i - G_i * e_r;
            float2 q;
            q.x = vr * e_r + vi * e_i;
            q.y = vr * e_i - vi * e_r;
            *reinterpret_cast<float2*>(qo + (2u * ob[j] + qc8)) = q;
          }
        }
        if (GRAD) {
          gout[h] = gr4;
          gout[4 + h] = gi4;
        }
      }
      loss_acc += (double)lt;
      sr_acc += (double)st_r;
      si_acc += (double)st_i;
    };
    request(0, X);
    if (nchunks > 1) request(1, Y);
    __syncthreads();  // tick 0: nothing to consume yet
    for (int k = 1; k < nticks; k += 2) {
      WS_STAMP(sblk, k, 0);
#ifndef CAL_WS_X_NOELEM
      if (k <= nchunks) {
        process(k - 1, X);
        WS_STAMP(sblk, k, 1);
        if (k + 1 < nchunks) request(k + 1, X);
      }
#endif
      WS_STAMP(sblk, k, 2);
      __syncthreads();
      WS_STAMP(sblk, k, 3);
      if (k + 1 < nticks) {
#ifndef CAL_WS_X_NOELEM
        WS_STAMP(sblk, k + 1, 0);
        if (k + 1 <= nchunks) {
          process(k, Y);
          WS_STAMP(sblk, k + 1, 1);
          if (k + 2 < nchunks) request(k + 2, Y);
        }
#endif
        WS_STAMP(sblk, k + 1, 2);
        __syncthreads();
        WS_STAMP(sblk, k + 1, 3);
      }
    }
    {
      const double l = ldsum(loss_acc), sr = ldsum(sr_acc), si = ldsum(si_acc);
      if (lane == 0) {
        s_red[w4] = l;
        s_red[4 + w4] = sr;
        s_red[8 + w4] = si;
      }
    }
    __syncthreads();  // GRAD: pairs with the matrix waves' epilogue barrier; otherwise the matrix waves have exited and only
                      // the element waves are counted (ended waves leave the barrier)
    if (wave == 4 && lane == 0) {
      A.part[(size_t)blockIdx.x * 4 + 0] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
      A.part[(size_t)blockIdx.x * 4 + 1] = s_red[4] + s_red[5] + s_red[6] + s_red[7];
      A.part[(size_t)blockIdx.x * 4 + 2] = s_red[8] + s_red[9] + s_red[10] + s_red[11];
    }
  }
}

inline size_t mfma_ws_lds_bytes(int nvec_max) {
  return ((size_t)((nvec_max + 7) / 8) * 256 + 4 * (size_t)kChunk * kSW) * sizeof(float) + 48 * sizeof(int) + 12 * sizeof(double) + 64;
}

// packed MFMA-native operand layouts of the wave-specialised kernel (see PanelItem)
__global__ void mfma_pack_kernel(const float* __restrict__ src, float* __restrict__ a_kf4, float* __restrict__ a_fk4, int nfreqs, int fpad,
                                 int nvec, int nvp32) {
  const int ngk = (nvec + 7) / 8, NT = nvp32 / 32;
  const long long n1 = (long long)(fpad / 32) * ngk * 256, n2 = (long long)(fpad / 8) * NT * 256;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int u = (int)(i & 3), l = (int)((i >> 2) & 63);
      const long long r = i >> 8;
      const int g = (int)(r % ngk), fb = (int)(r / ngk);
      const int f = fb * 32 + (l & 31), k = 8 * g + 2 * u + (l >> 5);
      a_kf4[i] = (f < nfreqs && k < nvec) ? src[(long long)f * nvec + k] : 0.f;
    } else {
      const long long q = i - n1;
      const int u = (int)(q & 3), l = (int)((q >> 2) & 63);
      const long long r = q >> 8;
      const int t = (int)(r % NT), cg = (int)(r / NT);
      const int f = 8 * cg + 2 * u + (l >> 5), n = 32 * t + (l & 31);
      a_fk4[q] = (f < nfreqs && n < nvec) ? src[(long long)f * nvec + n] : 0.f;
    }
  }
}

}  // namespace calk
