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
  // element offsets into MfmaArgs::ops (the whole buffer stays below 4 GB: the kernel adds 32-bit byte offsets to ONE base)
  long long a_kf4;    // packed forward operand  [F/32][nvp8/8][64 lanes][4]: lane (col, half), u -> A[32 fb + col][8 g + 2 u + half]
  long long a_fk4;    // packed adjoint operand  [F/8][nvp32/32][64 lanes][4]: lane (col, half), u -> A[8 cg + 2 u + half][32 t + col]
};

struct MfmaArgs {
  const float* ops;            // packed operands of every basis block
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
constexpr int kWsThreads = 512;  // 4 matrix waves + 4 element waves
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


// ---- operand stream of the matrix waves: an LDS ring filled by direct-to-LDS loads.
// hipcc places the s_waitcnt for a load in front of its first use, and around loop back-edges it falls back to draining the
// queue (vmcnt(0)) -- measured: every 4-MFMA group then exposes a full L2 round trip (~500-900 cycles) and the matrix pipe
// idles half of the time.  So the packed basis operands are requested a fixed number of positions ahead (kRing) with
// loads the compiler does not track, and consumed behind explicit counted waits.  The loads are LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 B = one 1-KB ring slot per instruction, no register destination): an inline-asm
// load INTO REGISTERS is only safe while hipcc never copies or re-assigns the destination before the data has landed, and
// nothing guarantees that (a register ring worked for one shape of these loops and silently broke -- renamed slots, moves
// of registers with a load in flight -- when the loops were restructured).  With the data in LDS every register the
// compiler sees is written by an instruction it tracks (ds_read_b128 of the slot, behind the counted wait).
constexpr int kRing = 4;                    // (4 and 8 slots measured the same: the stream is not latency-bound) slots (1 KB each) per matrix wave; vmcnt counts the wave's DMAs in issue order
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ const f32x4_t* uniform_ptr(const f32x4_t* p) {  // tell the compiler the pointer is wave-uniform ("s" operands)
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const f32x4_t*>(((unsigned long long)hi << 32) | lo);
}
// slot at LDS byte address lds_slot (wave-uniform) <- 64 x 16 B at sbase + lane_bytes + item_bytes.  M0 carries the LDS
// address of an LDS-DMA and is not preserved by the compiler around an asm: saved, set and restored in ONE statement.
__device__ __forceinline__ void ring_issue(unsigned lds_slot, const f32x4_t* sbase, unsigned lane_bytes, unsigned item_bytes) {
  const unsigned vo = lane_bytes + item_bytes;
  sbase = uniform_ptr(sbase);  // under scalar-register pressure hipcc parks a base in vector registers and would hand THOSE to the asm
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_slot), "v"(vo), "s"(sbase)
               : "memory");
}
#define RING_WAIT(N)                                            \
  do {                                                          \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");    \
    __builtin_amdgcn_sched_barrier(0);                          \
  } while (0)

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

  unsigned char* s_ring = smem_raw;                         // [4 matrix waves][kRing][1 KB] operand ring, lowest LDS addresses
  float* s_c = reinterpret_cast<float*>(smem_raw + 4 * kRing * 1024);  // [ngk][64 lanes][4] packed coefficient operand
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

  const int nchunks = A.fpad / kChunk;
  const int nticks = nchunks + (GRAD ? 2 : 1);
#ifdef CAL_WS_STAMP
  const int sblk = (blockIdx.x >= 2048 && blockIdx.x < 2056) ? (int)blockIdx.x - 2048 : -1;
#endif

  if (is_matrix) {
    // ================================================= matrix waves ==============================================
    __builtin_amdgcn_s_setprio(2);  // the matrix wave's issue slots come first on the SIMD it shares with an element wave
    // Adjoint work of a chunk = NT vector tiles x 16 channel groups (1 item = 8 channels x 32 vectors = 4 MFMAs), split
    // evenly whatever NT is.  Tiles come in groups of four, one WHOLE tile per wave (segments 0 and 1: tiles w4, w4 + 4).
    // The r = NT % 4 tiles left over are split along K: their 16 r items, flattened tile-major, are cut into four runs of
    // 4 r, one per wave -- a run touches one or two tiles (segments 2 and 3).  Every wave runs exactly 4 NT items per chunk,
    // owns one accumulator per segment (at most four) and every segment length is a multiple of 4, which keeps the
    // accumulator of each step static (a segment = whole half-rounds of the operand ring).  The partial sums of the
    // K-split tiles meet in LDS once, at the end of the panel.
    const int nfull4 = NT >> 2;                 // whole tiles of this wave: 0, 1 or 2
    const int nrem = NT & 3;
    const int nfullt = nfull4 * 4;
    const int n0 = nfull4 > 0 ? 16 : 0, n1 = nfull4 > 1 ? 16 : 0;
    const int rs = 4 * nrem * w4;               // this wave's run of the flattened (tile, channel group) list of the K-split tiles
    const int t2 = nfullt + (rs >> 4), c2 = rs & 15;
    const int n2 = nrem == 0 ? 0 : (16 - c2 < 4 * nrem ? 16 - c2 : 4 * nrem);
    const int n3 = 4 * nrem - n2;               // continues in the next tile at channel group 0
    const int e0 = n0, e1 = e0 + n1, e2 = e1 + n2;
    const int nitem = e2 + n3;                  // = 4 NT
    f32x16 gacc0, gacc1, gacc2, gacc3;
#pragma unroll
    for (int j = 0; j < 16; ++j) gacc0[j] = gacc1[j] = gacc2[j] = gacc3[j] = 0.f;
    const f32x4* ops = reinterpret_cast<const f32x4*>(A.ops);  // ONE scalar base for every operand request
    const unsigned fblk = (unsigned)P.a_kf4 * 4u, bblk = (unsigned)P.a_fk4 * 4u;  // byte offsets of this panel's two packed blocks
    const f32x4* sc4 = reinterpret_cast<const f32x4*>(s_c);
    const unsigned voff = (unsigned)lane * 16u;
    // this wave's operand ring: LDS byte address of slot 0 (for the DMA) and this lane's quad of a slot (for the read back)
    const unsigned ring_lds = (unsigned)reinterpret_cast<unsigned long long>(s_ring + w4 * kRing * 1024);
    const f32x4* ring_rd = reinterpret_cast<const f32x4*>(s_ring + w4 * kRing * 1024) + lane;
    // What a step needs per position -- the operand's byte offset and, in the adjoint phase, the LDS offset of its gbar rows
    // -- is wave-uniform but awkward to compute (segments, clamps); done with scalar code in every step it cost more than
    // the step's MFMAs hide.  Each lane computes the entry of ONE position once per panel (vector compares and selects)
    // and a step fetches its entry with v_readlane:
    //   lanes 0-31  : adjoint position p = lane     -> (cg NT + t) KB, offset of its operand inside an adjoint chunk
    //   lanes 32-63 : adjoint position p = lane-32  -> byte offset of channel group cg inside a gbar chunk buffer
    unsigned tabB;
    {
      int p = lane & 31;
      p = p < nitem ? p : nitem - 1;
      int t, cg;
      if (p < e0) { t = w4; cg = p; }
      else if (p < e1) { t = w4 + 4; cg = p - e0; }
      else if (p < e2) { t = t2; cg = c2 + (p - e1); }
      else { t = t2 + 1; cg = p - e2; }
      tabB = lane < 32 ? (unsigned)(cg * NT + t) * 1024u : (unsigned)(8 * cg * kSW) * 4u;
    }
    auto tab = [&](int i) { return (unsigned)__builtin_amdgcn_readlane((int)tabB, i); };
    // byte offset (from `ops`) of forward chunk kk of this wave / of adjoint chunk ci
    auto f_base = [&](int kk) { return fblk + (unsigned)((kk * 4 + w4) * ngk) * 1024u; };
    auto b_base = [&](int ci) { return bblk + (unsigned)(ci * (kChunk / 8) * NT) * 1024u; };
    // offset of position q of a phase of `kind` (1 forward, 2 adjoint) whose chunk starts at `base`; q < 32, clamped
    auto pos_off = [&](int kind, unsigned base, int q) {
      const unsigned of = (unsigned)(q < ngk ? q : ngk - 1) << 10;
      const unsigned ob = tab(q & 31);
      return base + (kind == 2 ? ob : of);
    };
    // The stream of positions runs through all phases of the panel: a step at position p requests position p + kRing, past
    // the end of its phase a position of the wave's NEXT phase -- across the barrier too, the packed basis does not depend
    // on the element waves -- so a phase starts with its first operands in LDS instead of an exposed L2 round trip.
    // kind: 0 none, 1 forward of chunk kx, 2 adjoint of chunk kx.
    // (a phase shorter than the ring would need requests two phases ahead: panels with nvec <= 56 do not chain)
    const bool chain_ok = GRAD && ngk >= kRing && nitem >= kRing;
    auto phase_after_F = [&](int k, int& kind, int& kx) {  // what this wave runs after F(k)
      kind = 0; kx = 0;
      if (!chain_ok) return;  // also: the forward-only pass is bound by its element waves, chaining only adds scalar work there
      if (k >= 2) { kind = 2; kx = k - 2; }
      else if (k + 1 < nchunks) { kind = 1; kx = k + 1; }
      else if (k + 1 < nticks && k + 1 >= 2) { kind = 2; kx = k - 1; }
    };
    auto phase_after_B = [&](int k, int& kind, int& kx) {  // what this wave runs after B(k - 2), i.e. in tick k + 1
      kind = 0; kx = 0;
      if (!chain_ok || k + 1 >= nticks) return;
      if (k + 1 < nchunks) { kind = 1; kx = k + 1; }
      else { kind = 2; kx = k - 1; }
    };
    int cons = 0;         // stream index of the position the next step consumes; its slot is cons % kRing
    bool primed = false;  // the ring already holds (landed or in flight) the first kRing positions of the phase about to start
    f32x4 r_cur = f32x4{0.f, 0.f, 0.f, 0.f}, r_nxt;  // operand quad of the current / the next position
    // start a phase the previous one did not chain to: retire whatever is in flight, request kRing positions, read the first
#define RING_PRIME(KIND, BASE)                                                                   \
  {                                                                                              \
    RING_WAIT(0);                                                                                \
    for (int j = 0; j < kRing; ++j) ring_issue(ring_lds + (unsigned)((cons + j) & (kRing - 1)) * 1024u, ops, voff, pos_off(KIND, BASE, j)); \
    RING_WAIT(kRing - 1);                                                                        \
    r_cur = ring_rd[(cons & (kRing - 1)) * 64];                                                  \
  }
    // One position = 4 MFMAs (256 cycles of the SIMD's matrix pipe) + everything that feeds the next ones.  A wave issues
    // in order and an MFMA waits for the pipe, so whatever stands BEHIND the four MFMAs runs in the shadow of the last one
    // only (64 cycles; measured: 400-560 cycles per position instead of 256 with ~35 instructions and a DMA there).  The
    // feed work is therefore placed BETWEEN the MFMAs, a piece per 64-cycle gap, and pinned there with scheduling barriers:
    //   MFMA 0 | A operand of the next position (LDS)            (STEP_A)
    //   MFMA 1 | next position's operand has landed -> read back (RING_NEXT: all but the kRing - 2 youngest requests)
    //   MFMA 2 | this position's slot is free (read one step ago) -> request position + kRing into it (RING_REQ)
    //   MFMA 3 | loop bookkeeping
#define RING_NEXT()                                        \
  __builtin_amdgcn_sched_barrier(0);                       \
  RING_WAIT(kRing - 2);                                    \
  r_nxt = ring_rd[((cons + 1) & (kRing - 1)) * 64];        \
  __builtin_amdgcn_sched_barrier(0);
#define RING_REQ(REQOFF)                                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  ring_issue(ring_lds + (unsigned)(cons & (kRing - 1)) * 1024u, ops, voff, REQOFF);              \
  __builtin_amdgcn_sched_barrier(0);
#define RING_ADVANCE() \
  r_cur = r_nxt;       \
  ++cons;
    for (int k = 0; k < nticks; ++k) {
      WS_STAMP(sblk, k, 0);
      if (k < nchunks) {
        // ---- F(k): rows = (slot, re|im), columns = this wave's 32 channels, K = vectors: position = k-group g (4 MFMAs)
        f32x16 acc, acc2;  // two accumulator chains
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = acc2[j] = 0.f;
        int nx_kind, nx_k;
        phase_after_F(k, nx_kind, nx_k);
        const unsigned cur_base = f_base(k);
        const unsigned nx_base = nx_kind == 1 ? f_base(nx_k) : nx_kind == 2 ? b_base(nx_k) : cur_base;
        const int nx_kd = nx_kind == 0 ? 1 : nx_kind;  // nothing to chain to: re-request this phase's (clamped) positions
        if (!primed) RING_PRIME(1, cur_base)
        primed = nx_kind != 0;
        f32x4 a_cur = (sc4 + 0)[lane], a_nxt;
        for (int g = 0; g < ngk; ++g) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], r_cur[0], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          a_nxt = (sc4 + (g + 1 < ngk ? g + 1 : ngk - 1) * 64)[lane];
          __builtin_amdgcn_sched_barrier(0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], r_cur[1], acc2, 0, 0, 0);
          RING_NEXT()
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], r_cur[2], acc, 0, 0, 0);
          const int q = g + kRing;  // position q of THIS phase, or position q - ngk of the next one
          RING_REQ(q < ngk ? pos_off(1, cur_base, q) : pos_off(nx_kd, nx_base, q - ngk))
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], r_cur[3], acc2, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          RING_ADVANCE()
          a_cur = a_nxt;
        }
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
      if (GRAD && k >= 2) {
        // ---- B(k-2): rows = (slot, re|im), columns = vectors of a tile, K = 8 channels of the chunk per position (4 MFMAs)
        const int ci = k - 2;
        const float* gp = s_g + (size_t)(ci & 1) * kChunk * kSW + half * kSW + col;       // + (8 cg + 2 v) kSW
        int nx_kind, nx_k;
        phase_after_B(k, nx_kind, nx_k);
        const unsigned cur_base = b_base(ci);
        const unsigned nx_base = nx_kind == 1 ? f_base(nx_k) : nx_kind == 2 ? b_base(nx_k) : cur_base;
        const int nx_kd = nx_kind == 0 ? 2 : nx_kind;
        auto lds_a = [&](int p, float (&a)[4]) {  // gbar rows of position p (table entries past the last position repeat it)
          const float* g8 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(gp) + tab(32 + (p & 31)));
#pragma unroll
          for (int v = 0; v < 4; ++v) a[v] = g8[2 * v * kSW];
        };
        if (!primed) RING_PRIME(2, cur_base)
        primed = nx_kind != 0;
        float a_cur[4], a_nxt[4];
        lds_a(0, a_cur);
        // one loop per segment: the accumulator of a step is static
#define B_SEG(GACC, PBEG, PEND)                                                                  \
  for (int pp = (PBEG); pp < (PEND); ++pp) {                                                     \
    GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], r_cur[0], GACC, 0, 0, 0);              \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    lds_a(pp + 1, a_nxt);                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], r_cur[1], GACC, 0, 0, 0);              \
    RING_NEXT()                                                                                  \
    GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], r_cur[2], GACC, 0, 0, 0);              \
    const int q = pp + kRing;                                                                    \
    RING_REQ(q < nitem ? pos_off(2, cur_base, q) : pos_off(nx_kd, nx_base, q - nitem))           \
    GACC = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], r_cur[3], GACC, 0, 0, 0);              \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    RING_ADVANCE()                                                                               \
    a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; a_cur[2] = a_nxt[2]; a_cur[3] = a_nxt[3];          \
  }
        B_SEG(gacc0, 0, e0)
        B_SEG(gacc1, e0, e1)
        B_SEG(gacc2, e1, e2)
        B_SEG(gacc3, e2, nitem)
#undef B_SEG
      }
      WS_STAMP(sblk, k, 2);
      __syncthreads();
      WS_STAMP(sblk, k, 3);
#ifdef CAL_WS_STAMP
      if (sblk >= 0 && lane == 0 && (k == 0 || k == nticks - 1)) { g_ws_stamps[sblk][wave][k == 0 ? 14 : 15][0] = (long long)__builtin_amdgcn_s_memrealtime(); g_ws_stamps[sblk][wave][14][1] = nvec; g_ws_stamps[sblk][wave][14][2] = NT; g_ws_stamps[sblk][wave][14][3] = nfull4 * 10 + nrem; }
#endif
    }
#undef RING_PRIME
#undef RING_NEXT
#undef RING_REQ
#undef RING_ADVANCE
    RING_WAIT(0);  // retire the last phase's trailing requests: the epilogue reuses LDS
    if (!GRAD) return;
    // ---- coefficient gradients.  Whole tiles are final in their accumulators.  The partial sums of the K-split tiles
    // (segments 2 and 3 of every wave) meet in LDS -- the V / gbar chunk buffers are idle now -- and wave jr adds, in
    // fixed order, the parts that belong to tile nfullt + jr and stores it.
    auto store_tile = [&](const f32x16& g, int t) {
      const int n = t * 32 + col;
      if (n < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int slot = (j & 3) + 8 * (j >> 2) + 4 * half;
          const int b = P.bl[slot];
          if (b >= 0) {
            const int coff = A.bl_coff[b];
            A.gc_r[coff + n] = g[j];
            A.gc_i[coff + n] = g[j + 8];
          }
        }
      }
    };
    float* s_x = s_v;  // [wave][segment 2 | 3][16 regs][64 lanes] = 32 KB inside the s_v / s_g buffers (72 KB)
    if (n2 > 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s_x[((w4 * 2 + 0) * 16 + j) * 64 + lane] = gacc2[j];
    }
    if (n3 > 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s_x[((w4 * 2 + 1) * 16 + j) * 64 + lane] = gacc3[j];
    }
    __syncthreads();  // pairs with the element waves' barrier after their tick loop
    if (n0 > 0) store_tile(gacc0, w4);
    if (n1 > 0) store_tile(gacc1, w4 + 4);
    if (w4 < nrem) {
      f32x16 g;
#pragma unroll
      for (int j = 0; j < 16; ++j) g[j] = 0.f;
      for (int w = 0; w < 4; ++w) {  // the runs of the four waves, in wave order (fixed summation order)
        const int ws = 4 * nrem * w, wt = ws >> 4, wc = ws & 15;
        const int wn2 = 16 - wc < 4 * nrem ? 16 - wc : 4 * nrem, wn3 = 4 * nrem - wn2;
        if (wt == w4) {
#pragma unroll
          for (int j = 0; j < 16; ++j) g[j] += s_x[((w * 2 + 0) * 16 + j) * 64 + lane];
        }
        if (wn3 > 0 && wt + 1 == w4) {
#pragma unroll
          for (int j = 0; j < 16; ++j) g[j] += s_x[((w * 2 + 1) * 16 + j) * 64 + lane];
        }
      }
      store_tile(g, nfullt + w4);
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
      if (k <= nchunks) {
        process(k - 1, X);
        WS_STAMP(sblk, k, 1);
        if (k + 1 < nchunks) request(k + 1, X);
      }
      WS_STAMP(sblk, k, 2);
      __syncthreads();
      WS_STAMP(sblk, k, 3);
      if (k + 1 < nticks) {
        WS_STAMP(sblk, k + 1, 0);
        if (k + 1 <= nchunks) {
          process(k, Y);
          WS_STAMP(sblk, k + 1, 1);
          if (k + 2 < nchunks) request(k + 2, Y);
        }
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
  return 4 * (size_t)kRing * 1024 + ((size_t)((nvec_max + 7) / 8) * 256 + 4 * (size_t)kChunk * kSW) * sizeof(float) + 48 * sizeof(int) +
         12 * sizeof(double) + 64;
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
