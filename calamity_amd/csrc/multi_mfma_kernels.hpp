// multi_mfma_kernels.hpp -- time slices of one solver that share basis tiles (cal_problem_desc::bl_alias) on the matrix
// cores: the 8 member baselines of a head item are 16 right-hand sides (re | im) of ONE tile, which makes the skinny
// complex GEMV of /root/reference/calamity/calibration.py:1587-1590 a real (if narrow) dense contraction --
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 with 16 columns -- while the tile still streams from HBM exactly once per
// pass (fused_multi_kernel of fit_kernels.hpp does the same job in the vector ALU and is bound by its lane reductions; it keeps
// the wide blocks).  The text below describes float32; what double precision (--precision 64, calibration.py:1857) changes is
// listed at MmT<double>.
//
// One 4-wave workgroup per head item.  Wave w owns the 16-channel strips j = w, w + 4, ... of the item's channels and runs, per
// strip ("job"), with no data exchanged between waves (one barrier per job only keeps them on neighbouring strips):
//   F  forward   V^T[col][ch] = sum_k C[k][col] T[k][ch]: M = 16 columns (member, re | im; the coefficient operand, from LDS),
//                N = the strip's 16 channels, K = basis vectors.  The tile operand of a k-step is ONE dword per lane,
//                T[4 s + (lane >> 4)][ch0 + (lane & 15)], straight from a buffer load (64-byte row pieces; the four waves of the
//                workgroup read the four neighbouring pieces); the same register is stored to the wave's LDS strip for the adjoint;
//   E  element-wise stage in the accumulator registers: lane (ch, q) holds ONE channel and the four columns 4 q + r, i.e. the
//                real (q < 2) or imaginary parts of four members; v_permlane32_swap hands the lane 32 on the missing parts, so that
//                each lane evaluates two members at its channel -- all sample / gain loads and the gbar_G store of a member are
//                16 adjacent lanes on 16 adjacent channels.  (With columns on the lanes -- the layout that lets the accumulator
//                double as the adjoint's operand, as the dense kernels do -- every lane of a sample load went to another row:
//                630 L1 accesses per job instead of 120.)  gbar_v goes through a 16 x 16 per-wave LDS buffer to turn it round;
//   B  adjoint   GC[k][col] += sum_ch T[k][ch] gbar_v[ch][col]: M = 16 basis vectors, K = the strip's channels; the A operand is
//                the transposed tile, one ds_read_b128 per 16 vectors from the LDS strip (XOR-swizzled by 8-row group: conflict-free
//                for the 4 x 16-lane groups of that instruction), the B operand one ds_read_b128 of the gbar_v buffer.
//                Gradient tiles stay in registers for the whole item; the four waves' parts meet in LDS once per item.
// Every loop over vectors has a compile-time trip count: the item body is instantiated per NT = ceil(nvec / 16) (14 classes, one
// wave-uniform dispatch per workgroup).  (With run-time counts -- a guard per k-step, or a switch with fall-through -- hipcc
// kept dozens of lane masks in spilled SGPRs and waited for ALL loads in front of the first k-step.)
// Rows past nvec: the k-steps run to 16 NT and simply read on into the next tile (finite basis values; the buffer carries a
// zeroed pad behind its last tile) against zero coefficients; gradient rows past nvec are never stored.
// Where the time goes (per-rank job of an 8-GPU HERA-350 run, 7 634 baselines x 8 slices, gradient pass 1.33 ms): the pass
// moves 3.1 GB of tiles, 0.75 GB of samples, 1.0 GB of gain rows (23 MB of distinct data, but 8 slices x 350 antennas do not
// stay in a 4 MB L2) and writes 0.5 GB of gbar_G: 5.35 GB at 4 TB/s.  Tiles alone: 0.66 ms (4.7 TB/s); matrix pipe busy 27 %.
#pragma once
#include "fit_kernels.hpp"

namespace calk {

typedef float mm_f32x4 __attribute__((ext_vector_type(4)));
typedef float mm_f32x2 __attribute__((ext_vector_type(2)));
typedef double mm_f64x4 __attribute__((ext_vector_type(4)));
typedef double mm_f64x2 __attribute__((ext_vector_type(2)));

#ifndef CAL_MM_NT
#define CAL_MM_NT 0  // cache policy of the tile loads (2 = non-temporal: measured 20 % slower -- the two halves of a 128-byte line are read by neighbouring waves, and the second one should still find it in L2)
#endif
constexpr int kMmStrip = 16;               // channels of a wave's job
constexpr int kMmMaxVec = 224;             // widest block this kernel takes (wider ones: fused_multi_kernel)
constexpr int kMmTiles = kMmMaxVec / 16;   // gradient tiles of 16 vectors = classes of the item body
constexpr int kMmMembers = 8;              // members of a head item: 16 MFMA columns (both precisions; fused_multi_kernel: MultiCfg<T>::nb_max)
constexpr int kMmTilePadElems = 2048;      // zeroed elements the tile buffers carry behind their last tile (the read-on of the last k-steps: < 16 rows of 128 channels)
constexpr int kMmHeader = 256 + 768 + 128; // members, loss partials, alphas

// What the precision decides.
//   float:  v_mfma_f32_16x16x4_f32 (32 cycles); register r of the accumulator of lane (j, q) is row 4 q + r; two workgroups per CU
//           (247 registers); a lane holds the real OR the imaginary parts of four members and trades two with the lane 32 on.
//   double: v_mfma_f64_16x16x4_f64 (64 cycles); register r is row q + 4 r -- so lane (ch, q) of F's accumulator holds columns
//           q, q + 4, q + 8, q + 12 = (re, re, im, im) of members q and q + 4: E needs no lane exchange at all; tile loads are
//           8 bytes per lane (a k-step = four 128-byte row pieces), which doubles the bytes a wave keeps in flight at the same
//           number of loads; twice the registers per value: ONE workgroup per CU (512 registers, up to 150 KB of LDS).
//           LDS strip rows are 128 bytes: a row's 16-byte chunks are XOR-swizzled with (row >> 1) & 7, which makes the B phase's
//           ds_read_b128 (16 lanes on 16 rows per pass) hit 16 distinct bank groups.
template <typename T> struct MmT;
template <> struct MmT<float> {
  typedef mm_f32x4 v4;
  typedef mm_f32x2 v2;
  static constexpr int kGvPitch = 20;  // words per column of the per-wave gbar_v buffer (16 channels + 4: 16-byte aligned, rows on distinct bank groups)
  static constexpr int kWgPerCu = 2;
  static __device__ __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t rs, unsigned lo, unsigned so) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lo, so, CAL_MM_NT));
  }
  static __device__ __forceinline__ constexpr int row_of(int q, int r) { return 4 * q + r; }  // row of accumulator register r of a lane in quarter q
};
template <> struct MmT<double> {
  typedef mm_f64x4 v4;
  typedef mm_f64x2 v2;
  static constexpr int kGvPitch = 18;  // doubles per column: 144 bytes, 16 columns on 16 distinct bank groups
  static constexpr int kWgPerCu = 1;
  static __device__ __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t rs, unsigned lo, unsigned so) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lo, so, CAL_MM_NT));
  }
  static __device__ __forceinline__ constexpr int row_of(int q, int r) { return q + 4 * r; }
};

template <typename T> inline size_t multi_mfma_lds_bytes(int nvec_max) {
  const size_t nt = (nvec_max + 15) / 16;
  return kMmHeader + (nt * 4 * 64 + 4 * nt * 16 * 16 + 2 * 4 * 16 * MmT<T>::kGvPitch) * sizeof(T);  // coefficient operand, four strips, 2 x four gbar_v buffers
}
// widest block of the one-pass regularised form (REG == 2): two gradient sets in registers -- in double the allocator needs 40 registers
// per tile for them and the tile ring: 10 tiles fit the 512 (232 + 256), 12 spill
template <typename T> constexpr int kMmMaxVecOnePass = sizeof(T) == 8 ? 160 : kMmMaxVec;

// REG, the "sum" regulariser (calibration.py:1623-1656):
//   2  ONE pass, the form of fused_basis_kernel: the gradient pass also sums S = sum w m of every member and carries a second
//      adjoint set for the w part (gbar_v' = conj(G) w, gbar_G' = conj(v) w -> q1, second coefficient gradients -> gcp1); the
//      host's combine kernels fold alpha = 2 (S - P) in afterwards.  The tiles stream once per STEP, one exchange per step.
//      Twice the gradient accumulators: one workgroup per CU in float too; double: blocks of at most kMmMaxVecOnePass (160) vectors, one
//      sample set (LATE) in every class.
//   1  two passes, like the dense kernels (kept for solvers that also hold heads of fused_multi_kernel, and for wider double blocks):
//      the loss pass also sums S; once the slices' alpha are known (alpha_kernel) the gradient pass applies e = -2 w r + alpha w
//      with the alpha of each member's own slice -- no second adjoint set.  The gradient pass then leaves the loss partials of the
//      loss pass in place.
template <typename T, int MODE, int NT, int DEPTH, int REG>
__device__ __forceinline__ void multi_mfma_item(const FusedArgs<T>& A, const Item& it, int item_idx, unsigned char* smem) {
  using X = MmT<T>;
  typedef typename X::v4 v4;
  typedef typename X::v2 v2;
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int ES = (int)sizeof(T);
  constexpr int kGvPitch = X::kGvPitch;
  constexpr bool GRAD = MODE == MODE_GRAD;
  constexpr int NS = 4 * NT;   // forward k-steps (four vectors each), run as two interleaved accumulator chains
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;   // index inside a row of 16 lanes: the CHANNEL of the strip (tile operand, accumulator, E), the MFMA column (coefficient operand, B)
  const int kq = lane >> 4;    // the k of a step this lane feeds; the quarter of the accumulator's rows it holds (MmT::row_of)
  const int nvec = it.nvec;
  const int NB = it.role_n >> 2;
  const int fb_log2 = it.fb_log2;
  const unsigned FB = 1u << fb_log2;

  Member* s_mem = reinterpret_cast<Member*>(smem);
  double* s_red = reinterpret_cast<double*>(smem + 256);  // [loss, S_r, S_i][4 waves][8 members]
  T* s_al = reinterpret_cast<T*>(smem + 1024);                        // REG, gradient pass: [8 members](alpha_r, alpha_i) of the member's slice
  T* s_c = reinterpret_cast<T*>(smem + kMmHeader);                    // [NS steps][64 lanes]: C[4 s + (lane >> 4)][lane & 15]
  T* s_strips = s_c + NS * 64;                                        // [4 waves][16 NT rows][16 channels], swizzled
  T* s_strip = s_strips + wave * NT * 256;
  T* s_gv = s_strips + 4 * NT * 256 + wave * (16 * kGvPitch);          // [16 columns][16 channels] gbar_v of the current job
  T* s_gv2 = s_gv + 4 * 16 * kGvPitch;                                 // REG == 2: the same of the w part

  if (tid < NB * (int)(sizeof(Member) / 4)) reinterpret_cast<int*>(s_mem)[tid] = reinterpret_cast<const int*>(A.members + it.member0)[tid];
  __syncthreads();
  if (REG == 1 && GRAD && tid < 8) {
    const DevState* sst = A.state + (tid < NB ? s_mem[tid].slice : 0);
    s_al[2 * tid] = tid < NB ? (T)sst->alpha_r : (T)0;
    s_al[2 * tid + 1] = tid < NB ? (T)sst->alpha_i : (T)0;
  }
  for (int n = tid; n < NS * 64; n += kThreads) {
    const int k = n >> 4, j = n & 15, m = j & 7;  // column j: member j & 7, real part for j < 8
    T v = 0;
    if (k < nvec && m < NB) v = (j < 8 ? A.c_r : A.c_i)[s_mem[m].coff + k];
    s_c[n] = v;
  }
  // E works in the accumulator layout of F (see there).  float: lane (ch = lane & 15, q = lane >> 4) holds, for ONE channel of the
  // strip, the four columns 4 q .. 4 q + 3 = the real parts (q < 2) or imaginary parts (q >= 2) of members 4 (q & 1) .. + 3.  The
  // partner 32 lanes on holds the other part of the same four members; of those four the lower lane evaluates the first two, the
  // upper lane the last two.  double: the lane holds both parts of members q and q + 4 and evaluates those.
  // Sample rows and antenna pairs of this lane's two members as 32-bit element offsets (+ its channel):
  const int half = kq >> 1;
  const int m_first = F64 ? kq : 4 * (kq & 1) + 2 * half;  // member of i = 0
  constexpr int m_step = F64 ? 4 : 1;                      // ... and i = 1 is m_first + m_step
  unsigned so[2], g0o[2], g1o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_first + i * m_step;
    const bool active = m < NB;
    const unsigned row = active ? (unsigned)s_mem[m].bl : (unsigned)A.nbls;  // the spare all-zero row
    so[i] = row * (unsigned)A.fpad + (unsigned)col;
    g0o[i] = (active ? (unsigned)s_mem[m].ant0 : 0u) * (unsigned)A.fpad + (unsigned)col;
    g1o[i] = (active ? (unsigned)s_mem[m].ant1 : 0u) * (unsigned)A.fpad + (unsigned)col;
  }
  __syncthreads();

  const T* s_cl = s_c + lane;
  constexpr bool CREG = NT <= 7;  // the coefficient operand of all k-steps in registers
  T creg[CREG ? NS : 1];
  if (CREG) {
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) creg[s2] = s_cl[s2 * 64];
  }
  // LDS strip, element address of (row, ch):
  //   float:  row * 16 + (((ch >> 2) ^ (2 * ((row >> 3) & 1))) << 2) + (ch & 3)
  //   double: row * 16 + (((ch >> 1) ^ f(row & 15)) << 1) + (ch & 1),  f(r) = (r >> 1) ^ (2 * (((r >> 2) ^ (r >> 3)) & 1))
  // (double: a row is 128 bytes = one half of the 64 banks, by row parity; B reads 16-byte chunk 2 kq (then 2 kq + 1) of row
  // 16 t + col, and a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): the 16 rows of
  // a group come with kq = a for rows 0-3 / 12-15 and a ^ 1 for rows 4-11, so f folds that bit in and the eight even (odd) rows of
  // a group land on eight different chunks.  Round 5's first form, f(r) = r >> 1, assumed contiguous groups: 33 % conflict cycles.)
  // store of step s: row 4 s + kq, channel col.  float: two lane pointers, by (s & 2) (rows 8..15 of a 16-row group);
  // double: f(4 s + kq) = (kq >> 1) ^ {0, 0, 6, 6}[s & 3]: four lane pointers, by s & 3.
  T* s_w[4];
  if (F64) {
#pragma unroll
    for (int v = 0; v < 4; ++v) s_w[v] = s_strip + kq * 16 + ((((col >> 1) ^ (kq >> 1) ^ (v >= 2 ? 6 : 0)) << 1) | (col & 1));
  } else {
    s_w[0] = s_w[1] = s_strip + kq * 16 + col;
    s_w[2] = s_w[3] = s_strip + kq * 16 + ((((col >> 2) ^ 2) << 2) | (col & 3));
  }
  // B reads row 16 t + col, channels 4 kq .. 4 kq + 3: one ds_read_b128 (float), two (double: 16-byte chunks 2 kq and 2 kq + 1)
  const v4* s_rd = reinterpret_cast<const v4*>(s_strip + col * 16 + ((kq ^ (2 * (col >> 3))) << 2));  // float; + t * 256 elements
  const int f_col = (col >> 1) ^ (2 * (((col >> 2) ^ (col >> 3)) & 1));
  const v2* s_rd0 = reinterpret_cast<const v2*>(s_strip + col * 16 + (((2 * kq) ^ f_col) << 1));  // double
  const v2* s_rd1 = reinterpret_cast<const v2*>(s_strip + col * 16 + (((2 * kq + 1) ^ f_col) << 1));
  auto strip_row = [&](int t) -> v4 {
    if constexpr (F64) {
      const v2 lo2 = s_rd0[t * 128], hi2 = s_rd1[t * 128];
      return v4{lo2[0], lo2[1], hi2[0], hi2[1]};
    } else {
      return s_rd[t * 64];
    }
  };

  T* s_gw = s_gv + m_first * kGvPitch + col;                                  // E: column m_first (+ i m_step, + 8), channel col
  const T* s_gr = s_gv + col * kGvPitch + 4 * kq;                             // B: column col, channels 4 kq ..
  T* s_gw2 = s_gv2 + m_first * kGvPitch + col;
  const T* s_gr2 = s_gv2 + col * kGvPitch + 4 * kq;

  const int njobs = A.fpad >> 6;  // strips per wave: even, the host gives this kernel only rows padded to a multiple of 128 channels
  // The tile loads are buffer loads: resource = this item's tiles, per-lane offset `lo` (one VGPR for the whole kernel), the
  // k-step's offset in an SGPR -- no vector-ALU address arithmetic (with flat pointers hipcc spent a 64-bit vector add per load).
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A.tiles + it.tile_first), 0, 0x7fffffff, 0x00020000);
  const unsigned tile_bytes = ((unsigned)nvec << fb_log2) * ES;
  const unsigned lo = ((unsigned)kq * FB + (unsigned)col) * ES;  // byte offset of this lane's element inside a k-step
  const unsigned step_bytes = FB * 4u * ES;                      // four rows
  // Software pipeline.  A wave keeps 28 to 56 tile loads (7 to 14 KB; twice that in double) in flight at all times: the k-step registers form a ring that
  // is DEPTH jobs deep (4 up to 48 vectors, 2 up to 112, else 1: 28 to 56 loads), and F re-issues each register -- for the job DEPTH ahead --
  // right behind the MFMA that consumed it, so loads are issued in the order they are consumed and the compiler's counted waits
  // never drain the queue.  The samples of job n + 1 are requested at the START of job n (two register sets, the job loop is
  // unrolled by two), i.e. ahead of the tile loads F issues: waiting for them in E never waits for younger tile loads.
  // The loop body has NO branch: behind a conditional prefetch the counted waits must hold for the path that issued nothing,
  // and every wait for the samples also waited for the whole next tile.  Past the last job the loads therefore still run, with
  // a zero k-step stride on the last job's first rows (one cached kilobyte), and the last job's samples are requested again.
  // job n of wave w -> strip w + 4 n: the four waves of the workgroup read the four neighbouring 64-byte pieces of the same rows.
  // (Giving one wave the two halves of a 128-byte line in consecutive jobs raised the L2 hits from 15 M to 52 M per launch and
  // changed neither the fabric traffic nor the time for the better: 1.36 against 1.30 ms.)
  auto strip_of = [&](int n) -> unsigned { return (unsigned)(wave + 4 * n); };
  T treg[DEPTH][NS];
  auto job_off = [&](int n) -> unsigned {
    const unsigned ch = strip_of(n < njobs ? n : njobs - 1) * kMmStrip;
    return (ch >> fb_log2) * tile_bytes + (ch & (FB - 1)) * ES;
  };
  struct Samples { T dr[2], di[2], w[2]; v2 g0[2], g1[2]; };  // this lane's two members at its channel
  auto issue_samples = [&](int n, Samples& S) {
    const unsigned ch = strip_of(n < njobs ? n : njobs - 1) * kMmStrip;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      S.dr[i] = A.data_r[so[i] + ch];
      S.di[i] = A.data_i[so[i] + ch];
      S.w[i] = A.wgts[so[i] + ch];
      S.g0[i] = *reinterpret_cast<const v2*>(A.gains + g0o[i] + ch);
      S.g1[i] = *reinterpret_cast<const v2*>(A.gains + g1o[i] + ch);
    }
  };

  constexpr bool R2 = REG == 2 && GRAD;
  v4 dC[NT], dC2[R2 ? NT : 1];
#pragma unroll
  for (int t = 0; t < NT; ++t) dC[t] = v4{0, 0, 0, 0};
  if (R2) {
#pragma unroll
    for (int t = 0; t < NT; ++t) dC2[R2 ? t : 0] = v4{0, 0, 0, 0};
  }
  double loss_acc[2] = {0.0, 0.0};  // of this lane's two members: each member's loss goes to its OWN slot (its time slice's sum)
  double sr_acc[2] = {0.0, 0.0}, si_acc[2] = {0.0, 0.0};  // REG, loss pass: S = sum w m of the two members

#ifdef CAL_MM_STAMP
  long long t_bar = 0, t_f = 0, t_e = 0, t_b = 0, t_prev = 0;
#define MM_STAMP(ACC) { const long long t_now = (long long)__builtin_amdgcn_s_memtime(); ACC += t_now - t_prev; t_prev = t_now; }
#else
#define MM_STAMP(ACC)
#endif
  // one job: samples S (requested one job ago), ring set D; requests the next job's samples into Sn
  // LATE (double, more than 176 vectors: 16 registers per tile, and 512 are all there is): ONE sample set, requested again between
  // E and B -- the adjoint's MFMAs of such a block (3 600 cycles) cover the latency -- and the coefficient operand two groups ahead.
  constexpr bool LATE = F64 && (NT > 11 || R2);
  auto job = [&](int n, T (&tr)[NS], const Samples& S, Samples& Sn) {
    MM_STAMP(t_b)
    __syncthreads();  // keeps the four waves on neighbouring strips of the same rows (3-4 % faster than letting them drift)
    MM_STAMP(t_bar)
    if (!LATE) issue_samples(n + 1, Sn);
    __builtin_amdgcn_sched_barrier(0);
    // ---- F: two accumulator chains (40-cycle dependent latency against a 32-cycle issue); each register goes back out for job n + DEPTH
    const unsigned nxt_off = job_off(n + DEPTH);
    const unsigned nxt_step = n + DEPTH < njobs ? step_bytes : 0u;
    v4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    // One fenced group per pair of k-steps: two MFMAs, their strip store, the two re-issued loads.  (Left alone -- also with
    // sched_group_barrier hints -- the scheduler holds the loads back and sends them in a burst at the end of F: the queue ran
    // down to a third of its depth in every job.)  The coefficient operand is the same for all jobs of an item: up to 112
    // vectors it lives in registers (creg); wider blocks read it from LDS four groups ahead -- LDS operations complete in order,
    // so a wait for the read of group p also waits for the strip stores and reads issued since, and a two-group lead left about
    // a hundred cycles of LDS latency exposed in every group.
    constexpr int kLead = LATE ? 2 : 4;
    T bq[kLead + 1][2];
    if (!CREG) {
#pragma unroll
      for (int g = 0; g < kLead; ++g)
        if (2 * g < NS) {
          bq[g][0] = s_cl[(2 * g) * 64];
          bq[g][1] = s_cl[(2 * g + 1) * 64];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; s += 2) {
      const int p = s >> 1;
      T c0, c1;
      if (CREG) {
        c0 = creg[s];
        c1 = creg[s + 1];
      } else {
        if (s + 2 * kLead < NS) {
          bq[(p + kLead) % (kLead + 1)][0] = s_cl[(s + 2 * kLead) * 64];
          bq[(p + kLead) % (kLead + 1)][1] = s_cl[(s + 2 * kLead + 1) * 64];
        }
        c0 = bq[p % (kLead + 1)][0];
        c1 = bq[p % (kLead + 1)][1];
      }
      acc0 = X::mfma(c0, tr[s], acc0);
      acc1 = X::mfma(c1, tr[s + 1], acc1);
      if (GRAD) {  // steps s, s + 1 (s even) lie in the same 8-row group (float) / pair of 2-row groups that differ in the lane part only (double)
        T* w = s_w[s & 3];
        T* w1 = s_w[(s + 1) & 3];
        w[s * 64] = tr[s];
        w1[(s + 1) * 64] = tr[s + 1];
      }
      tr[s] = X::load(rs, lo, nxt_off + (unsigned)s * nxt_step);
      tr[s + 1] = X::load(rs, lo, nxt_off + (unsigned)(s + 1) * nxt_step);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    v4 acc = acc0 + acc1;            // acc[r] of lane (ch, q) = v(channel ch) of column MmT::row_of(q, r)
    MM_STAMP(t_f)
    // ---- E.  float: v_permlane32_swap hands each lane the missing part of the two members it evaluates (the lower lane's acc[i]
    // and the upper lane's acc[i + 2] stay, the other two registers cross): (re, im) of member m_first + i on both lanes, no
    // selects.  double: acc[i] and acc[i + 2] ARE (re, im) of member q + 4 i.
    const unsigned ch0 = strip_of(n) * kMmStrip;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      T vr, vi;
      if constexpr (F64) {
        vr = acc[i];
        vi = acc[i + 2];
      } else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const float mine_lo = acc[i], mine_hi = acc[i + 2];  // (scalars first: __builtin_bit_cast applied to a vector ELEMENT read element 0)
        const u2 sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mine_lo), __builtin_bit_cast(unsigned, mine_hi), false, false);
        const unsigned sw0 = sw[0], sw1 = sw[1];
        vr = __builtin_bit_cast(float, sw0);
        vi = __builtin_bit_cast(float, sw1);
      }
      const T g0x = S.g0[i][0], g0y = S.g0[i][1], g1x = S.g1[i][0], g1y = S.g1[i][1];
      const T d_r = S.dr[i], d_i = S.di[i], w = S.w[i];
      // G = g0 conj(g1)   (calibration.py:1598-1601: grgr + gigi, gigr - grgi)
      const T G_r = g0x * g1x + g0y * g1y;
      const T G_i = g0y * g1x - g0x * g1y;
      const T m_r = G_r * vr - G_i * vi;
      const T m_i = G_i * vr + G_r * vi;
      const T r_r = d_r - m_r, r_i = d_i - m_i;
      if (!(REG == 1 && GRAD)) loss_acc[i] += (double)(w * (r_r * r_r + r_i * r_i));
      if ((REG == 1 && !GRAD) || REG == 2) {
        sr_acc[i] += (double)(w * m_r);
        si_acc[i] += (double)(w * m_i);
      }
      if (GRAD) {
        T e_r = (T)-2 * w * r_r, e_i = (T)-2 * w * r_i;
        if (REG == 1) {
          e_r += s_al[2 * (m_first + i * m_step)] * w;
          e_i += s_al[2 * (m_first + i * m_step) + 1] * w;
        }
        // gbar_v = conj(G) e -> the wave's [column][channel] buffer, from where B takes it as its operand
        s_gw[(i * m_step) * kGvPitch] = G_r * e_r + G_i * e_i;
        s_gw[(8 + i * m_step) * kGvPitch] = G_r * e_i - G_i * e_r;
        v2 q;  // gbar_G = conj(v) e; members the set does not have: zeros to the spare row
        q[0] = vr * e_r + vi * e_i;
        q[1] = vr * e_i - vi * e_r;
        *reinterpret_cast<v2*>(A.q0 + so[i] + ch0) = q;
        if (R2) {  // the w part: gbar_v' = conj(G) w, gbar_G' = conj(v) w (fit_kernels.hpp, process_group_item)
          s_gw2[(i * m_step) * kGvPitch] = G_r * w;
          s_gw2[(8 + i * m_step) * kGvPitch] = -G_i * w;
          v2 qw;
          qw[0] = vr * w;
          qw[1] = -vi * w;
          *reinterpret_cast<v2*>(A.q1 + so[i] + ch0) = qw;
        }
      }
    }
    MM_STAMP(t_e)
    if (LATE) {
      __builtin_amdgcn_sched_barrier(0);
      issue_samples(n + 1, Sn);  // (Sn IS S here: SB below is SA)
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- B: gradient tiles in pairs (independent accumulators back to back)
    if (GRAD) {
      __builtin_amdgcn_wave_barrier();
      v4 gvb, gvb2 = {0, 0, 0, 0};  // lane (column, k): gbar_v of channels 4 k .. 4 k + 3
      if constexpr (F64) {
        const v2 g_lo = *reinterpret_cast<const v2*>(s_gr), g_hi = *reinterpret_cast<const v2*>(s_gr + 2);
        gvb = v4{g_lo[0], g_lo[1], g_hi[0], g_hi[1]};
        if (R2) {
          const v2 h_lo = *reinterpret_cast<const v2*>(s_gr2), h_hi = *reinterpret_cast<const v2*>(s_gr2 + 2);
          gvb2 = v4{h_lo[0], h_lo[1], h_hi[0], h_hi[1]};
        }
      } else {
        gvb = *reinterpret_cast<const v4*>(s_gr);
        if (R2) gvb2 = *reinterpret_cast<const v4*>(s_gr2);
      }
#pragma unroll
      for (int t = 0; t + 1 < NT; t += 2) {
        const v4 a0 = strip_row(t);
        const v4 a1 = strip_row(t + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dC[t] = X::mfma(a0[r], gvb[r], dC[t]);
          dC[t + 1] = X::mfma(a1[r], gvb[r], dC[t + 1]);
          if (R2) {
            dC2[R2 ? t : 0] = X::mfma(a0[r], gvb2[r], dC2[R2 ? t : 0]);
            dC2[R2 ? t + 1 : 0] = X::mfma(a1[r], gvb2[r], dC2[R2 ? t + 1 : 0]);
          }
        }
        if (F64 && NT > 8 && (t & 2)) __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler reads all 14 strip rows -- 112 registers -- ahead of the first MFMA: 56 spilled)
      }
      if (NT & 1) {
        const v4 a0 = strip_row(NT - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dC[NT - 1] = X::mfma(a0[r], gvb[r], dC[NT - 1]);
          if (R2) dC2[R2 ? NT - 1 : 0] = X::mfma(a0[r], gvb2[r], dC2[R2 ? NT - 1 : 0]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  };

  // (the prologue issues in the order of the steady state -- samples, then tiles: at the loop head the compiler's counted waits
  // must cover the entry path too, and samples requested BEHIND the tiles there made every iteration wait for all but five loads)
  Samples SA, SB_;
  Samples& SB = LATE ? SA : SB_;
  issue_samples(0, SA);
  __builtin_amdgcn_sched_barrier(0);  // (the scheduler is free to reorder independent loads: pin the order the waits are counted against)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const unsigned off = job_off(d);
#pragma unroll
    for (int s = 0; s < NS; ++s) treg[d][s] = X::load(rs, lo, off + (unsigned)s * step_bytes);
  }
  __builtin_amdgcn_sched_barrier(0);
#ifdef CAL_MM_STAMP
  t_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
  if (DEPTH == 4) {
    for (int n = 0; n < njobs; n += 4) {  // njobs is a multiple of 4 (the dispatch picks DEPTH 2 otherwise)
      job(n, treg[0], SA, SB);
      job(n + 1, treg[1 % DEPTH], SB, SA);
      job(n + 2, treg[2 % DEPTH], SA, SB);
      job(n + 3, treg[3 % DEPTH], SB, SA);
    }
  } else if (LATE && DEPTH == 1) {
    for (int n = 0; n < njobs; ++n) job(n, treg[0], SA, SA);  // (not unrolled by two: the body's registers are the point of LATE)
  } else {
    for (int n = 0; n < njobs; n += 2) {  // njobs is even
      job(n, treg[0], SA, SB);
      job(n + 1, treg[DEPTH - 1], SB, SA);
    }
  }

#ifdef CAL_MM_STAMP
  {
    MM_STAMP(t_b)
    if (GRAD && (blockIdx.x % 997) == 13 && lane == 0)
      printf("stamp item %d NT %d wave %d jobs %d: barrier %lld F %lld E %lld B %lld (s_memtime ticks, 100 MHz)\n", item_idx, NT, wave, njobs, t_bar, t_f, t_e, t_b);
  }
#endif
  // ---- epilogue: loss partial of every member (into the member's own slot: members may belong to different time slices),
  // coefficient gradients of every member.  Lane (col, kq) holds its two members at its channel: sum over the 16 lanes of
  // the row, then over the four waves in order.
  if (!(REG == 1 && GRAD)) {
    constexpr int NQ = REG ? 3 : 1;  // loss (, S_r, S_i): s_red [quantity][4 waves][8 members]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int qn = 0; qn < NQ; ++qn) {
        double v = qn == 0 ? loss_acc[i] : (qn == 1 ? sr_acc[i] : si_acc[i]);
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) v += __shfl_xor(v, sft, 64);
        if (col == 0) s_red[qn * 32 + wave * 8 + m_first + i * m_step] = v;
      }
    }
  }
  __syncthreads();  // also: every wave has left its strip
  if (!(REG == 1 && GRAD) && tid < NB) {
    const size_t slot = (size_t)s_mem[tid].item * 4;
    A.part[slot + 0] = ((s_red[tid] + s_red[8 + tid]) + s_red[16 + tid]) + s_red[24 + tid];
    A.part[slot + 1] = REG ? ((s_red[32 + tid] + s_red[40 + tid]) + s_red[48 + tid]) + s_red[56 + tid] : 0.0;
    A.part[slot + 2] = REG ? ((s_red[64 + tid] + s_red[72 + tid]) + s_red[80 + tid]) + s_red[88 + tid] : 0.0;
  }
  if (!GRAD) return;
  // dC[t][r] of lane (col, kq) of wave w = that wave's part of GC[16 t + row_of(kq, r)][col]; the four parts meet in the strip area
  // (+ the gbar_v buffers behind it): s_x[((w * NT + t) * 4 + r) * kXP + lane], rows of 64 lanes at a pitch of 72 words.
  // Reader: lane l of wave w sums the four parts of column j = (l & 7) + 8 (l >> 5), register r = (l >> 3) & 3, quarter kq = w:
  // the 32 lanes of a half read words 72 r + (l & 7) + const = 32 different banks.  (Round 4's reader -- thread (column tid >> 4,
  // vector tid & 15) on rows of 64 words -- was 8-way conflicted: 12 M of the gradient pass's 13.5 M conflict cycles.)
  constexpr int kXP = 72;
  static_assert(4 * NT * 4 * kXP <= 4 * NT * 256 + 2 * 4 * 16 * kGvPitch, "the exchange area fits the strips and gbar_v buffers");
  T* s_x = s_strips;
#pragma unroll
  for (int set = 0; set < (R2 ? 2 : 1); ++set) {
    if (set) __syncthreads();  // the sums of the first set have been read
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s_x[((wave * NT + t) * 4 + r) * kXP + lane] = set ? dC2[R2 ? t : 0][r] : dC[t][r];
    }
    __syncthreads();
    const int j = (lane & 7) + 8 * (lane >> 5), r = (lane >> 3) & 3, m = j & 7;
    if (m < NB) {
      T* gc = (set ? (j < 8 ? A.gcp1_r : A.gcp1_i) : (j < 8 ? A.gcp0_r : A.gcp0_i)) + s_mem[m].goff;
      const int kk = X::row_of(wave, r);
      for (int t = 0; t < NT; ++t) {
        const int k = 16 * t + kk;
        const T* px = s_x + (t * 4 + r) * kXP + j + 16 * wave;
        const T v = ((px[0] + px[NT * 4 * kXP]) + px[2 * NT * 4 * kXP]) + px[3 * NT * 4 * kXP];
        if (k < nvec) gc[k] = v;
      }
    }
  }
}

template <typename T, int MODE, int REG>
__global__ __launch_bounds__(kThreads, (REG == 2 ? 1 : MmT<T>::kWgPerCu)) void fused_multi_mfma_kernel(const FusedArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int item_idx = A.heads[blockIdx.x];
  if (item_idx < 0) return;  // an empty slot of the XCD-affine head list
  const Item it = A.items[item_idx];
  // (members of several time slices: the item runs while any of them does; a stopped member's outputs are not consumed)
  if (A.nslices == 1 && (A.state->done | A.state->done_after)) return;
  // the ring of tile registers is 4 jobs deep up to 48 vectors, 2 up to 112, else 1: 28 to 56 loads (7 to 14 KB; double: 14 to 28) in flight per wave
  const bool quad = ((A.fpad >> 6) & 3) == 0;  // a wave's jobs come in fours
  constexpr bool kWide = !(REG == 2 && sizeof(T) == 8);  // classes 11 .. 14 exist (the host: kMmMaxVecOnePass)
#define MM_CLASS(NT_, D_) multi_mfma_item<T, MODE, NT_, D_, REG>(A, it, item_idx, smem)
  switch ((it.nvec + 15) >> 4) {  // wave-uniform
    case 1: if (quad) MM_CLASS(1, 4); else MM_CLASS(1, 2); break;
    case 2: if (quad) MM_CLASS(2, 4); else MM_CLASS(2, 2); break;
    case 3: if (quad) MM_CLASS(3, 4); else MM_CLASS(3, 2); break;
    case 4: MM_CLASS(4, 2); break;
    case 5: MM_CLASS(5, 2); break;
    case 6: MM_CLASS(6, 2); break;
    case 7: MM_CLASS(7, 2); break;
    case 8: MM_CLASS(8, 1); break;
    case 9: MM_CLASS(9, 1); break;
    case 10: MM_CLASS(10, 1); break;
    case 11: if constexpr (kWide) MM_CLASS(11, 1); break;
    case 12: if constexpr (kWide) MM_CLASS(12, 1); break;
    case 13: if constexpr (kWide) MM_CLASS(13, 1); break;
    case 14: if constexpr (kWide) MM_CLASS(14, 1); break;
    default: break;  // the host gives this kernel no wider block
  }
#undef MM_CLASS
}
static_assert(kMmTiles == 14, "the dispatch above lists 14 classes");

}  // namespace calk
