// dense_kernels.hpp -- dense (matrix-core) formulation of the fit step for the SHARED layout, fp32.
//
// When every baseline of a distinct delay shares ONE basis block A_u (the operator cache of
// /root/reference/calamity/modeling.py:291-301; 120 blocks / 68 MB at HERA-350), fg_model (calibration.py:1587-1590) over all
// those baselines is a real dense contraction and so is its adjoint (BASELINE config 5).  Both
// run here on v_mfma_f32_32x32x2_f32 -- exact fp32, bit-for-bit a k-ordered fmaf chain -- with everything between them
// (gains, residual, chi^2, gbar_v, gbar_G; calibration.py:1593-1609 and their adjoints) done in the accumulator registers.
//
// Work item = a PANEL of 16 baselines with the same basis block, one 256-thread workgroup (4 waves, two workgroups per
// CU, i.e. two independent waves per SIMD: one wave's loads, waits and element-wise arithmetic run under the other's
// MFMAs).  Wave w owns the channel blocks cb = w, w + 4, ... (32 channels each) of the panel and does, per block,
//   F  V^T (32 ch x 32 cols)   = A_u[ch, :] (32 x nvec) . C^T (nvec x 32 cols)      cols = (slot, re | im) of the panel
//   E  element-wise on the accumulator: lane (col, half) holds 16 channels of its column; re and im of a slot sit 16 lanes
//      apart and are paired with v_permlane16_swap; G = g_i conj(g_j), m = G v, r = d - m, chi^2, e = -2 w r,
//      gbar_v = conj(G) e back into the SAME registers, gbar_G = conj(v) e to HBM
//   B  dC^T[t] (32 vec x 32 cols) += A_u[ch, 32 t ...]^T (32 x 32 ch) . gbar_v^T (32 ch x 32 cols)  for every vector tile t
// The accumulator of F is, register by register, the B operand of the adjoint MFMA (lane = column, the two lane halves =
// the two k of a 32x32x2 step: register r holds channels c(r) and c(r) + 4, c(r) = (r & 3) + 8 (r >> 2); the packed
// adjoint operand is stored in that k order), so V and gbar_v never leave the registers: no LDS round trip, no transposition,
// no barrier inside a panel.  The coefficient gradients of a panel stay in registers for the whole sweep (16 NT of them);
// the four waves' partial sums (each over its channel blocks) meet in LDS once, at the end.
#pragma once
#include "fit_kernels.hpp"

namespace calk {

constexpr int kPanel = 16;        // baselines per panel
constexpr int kChunk = 128;       // the row padding the dense path needs: 4 waves x one 32-channel block
constexpr int kCB = 32;           // channels per block = one MFMA tile edge
constexpr int kMaxNT = 8;         // vector tiles of 32 (nvec <= 256)
constexpr int kDenseThreads = 256;
constexpr int kSmpBytes = 6 * 1024;  // one channel block's samples of a wave's 16 baselines: 3 arrays x 32 channels x 16 x 4 B

typedef float f32x16 __attribute__((ext_vector_type(16)));

// which solver dtypes have a dense kernel, and the widest basis block it takes
template <typename T> struct DenseCfg { static constexpr int max_nvec = 0; };
template <> struct DenseCfg<float> { static constexpr int max_nvec = 32 * kMaxNT; };

struct PanelItem {
  int bl[kPanel];     // baseline ids (-1: padding slot)
  int nvec, nvp2, nvp32;
  int slice;          // the time slice all its baselines belong to (panels never mix slices): which DevState governs it
  int coff[kPanel];   // coefficient offset of each slot's group (0 for padding slots)
  int2 ant[kPanel];   // antenna pair of each slot (0, 0 for padding slots)
  // element offsets into MfmaArgs::ops (the whole buffer stays below 4 GB: the kernel adds 32-bit byte offsets to ONE base)
  long long a_kf4;    // the basis block's packed operands (mfma_pack_kernel): per channel block, forward positions
                      //   [ceil(nvec/8)][64 lanes][4]: lane (c, half), u -> A[32 cb + c][8 g + 2 u + half], then adjoint positions
                      //   [nvp32/32][4][64 lanes][4]: lane (i, half), u -> A[32 cb + 8 q + 4 half + u][32 t + i]
  long long a_fk4;    // (double precision: the adjoint operands are a separate run)
  int tile0;          // split-bf16 kernel (split_kernels.hpp): the first of the block's vector tiles this item carries (nvp32 / 32 of them)
  int pad_;
};

struct MfmaArgs {
  const float* ops;            // packed operands of every basis block
  const PanelItem* panels;
  const float* data_r;         // [nbls + 1][fpad]
  const float* data_i;
  const float* wgts;
  const float2* gains;         // [nants][fpad]
  const float* c_r;
  const float* c_i;
  float2* q0;                  // [nbls + 1][fpad]
  float* gc_r;                 // [ncoef] final coefficient gradient (every baseline owns its coefficients)
  float* gc_i;
  double* part;                // [npanels][4]
  const DevState* state;       // [nslices]
  int nslices;
  int fpad;
  int use_alpha;               // "sum" regulariser, second pass: e = -2 w r + alpha w with alpha = 2 (S - P) read from state
  int nbls;                    // row nbls of data_r / data_i / wgts / q0 is an all-zero spare row for padding slots
  const int* slot_map;         // [grid] workgroup -> panel, -1 for an empty slot: XCD-affine dispatch, see fused_dense_kernel
};

// ---- operand stream: an LDS ring filled by direct-to-LDS loads.
// hipcc places the s_waitcnt for a load in front of its first use, and around loop back-edges it falls back to draining the
// queue (vmcnt(0)); the packed basis operands are therefore requested a fixed number of positions ahead (kRing) with loads
// the compiler does not track, and consumed behind explicit counted waits.  The loads are LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 B = one 1-KB ring slot per instruction, no register destination): an inline-asm
// load INTO REGISTERS is only safe while hipcc never copies or re-assigns the destination before the data has landed, and
// nothing guarantees that (a register ring worked for one shape of the loops and silently broke -- renamed slots, moves of
// registers with a load in flight -- when they were restructured).  With the data in LDS every register the compiler sees
// is written by an instruction it tracks (ds_read_b128 of the slot, behind the counted wait).
constexpr int dense_ring_slots(bool grad, int ntmax) { return grad ? (ntmax > 4 ? 6 : 8) : 4; }  // operand-ring slots (1 KB each) per wave
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

#ifdef CAL_STAMP
__device__ long long g_dense_stamps[4096][4][8];  // diagnostic build only: [panel][wave][F, E, B, loop, prologue, epilogue cycles, entry time, exit time]
#define STAMP_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#else
#define STAMP_T(var)
#endif
template <bool GRAD, int NTMAX>
__device__ __forceinline__ void dense_panel(const MfmaArgs& A, unsigned char* smem_raw) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // ring depth: as deep as two workgroups per CU allow (measured at HERA-350, gradient pass: 8 slots 0.955 ms, 4 slots
  // 1.00 ms); the class with more than four vector tiles has the larger coefficient panel and gets 6
  constexpr int kRing = dense_ring_slots(GRAD, NTMAX);
  STAMP_T(t_entry);
  const int panel_idx = A.slot_map[blockIdx.x];  // never negative here: the kernel returns for empty slots
  const PanelItem& P = A.panels[panel_idx];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave-uniform quantities live in SGPRs: every operand address below is (scalar base) + (lane offset)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;       // MFMA column: (slot, re | im) of the panel
  const int half = lane >> 5;      // the k of a 32x32x2 step this lane feeds / the channel sub-group (+4) it holds
  const int slot = col & 15;
  const bool im_lane = (col & 16) != 0;
  // this lane's slot: sample row (padding slots use the all-zero spare row: weight 0), antenna pair, coefficient offset.
  // The panel's whole prologue is three dependent round trips (panel record -> samples + operands + coefficients -> LDS);
  // nothing in it sits behind a branch (a load behind a branch costs a round trip of its own)
  const int my_bl = P.bl[slot];
  const int2 my_ant = P.ant[slot];
  const int my_coff = P.coff[slot];
  const int fill_bl = P.bl[(tid >> 2) & 15];     // the column this thread fills of the coefficient panel (below)
  const int fill_coff = P.coff[(tid >> 2) & 15];
  const int nvec = P.nvec, NT = P.nvp32 / 32;
  const int ngk = (nvec + 7) / 8;  // forward k-groups of 8 vectors (4 k-steps)
  const int ncb = A.fpad / kCB;
  const DevState* sst = A.state;
  if (A.nslices > 1) sst += P.slice;  // (several time slices: the panel's own, one dependent load beside the slot table's)
  const int stopped = sst->done | sst->done_after;

  unsigned char* s_ring = smem_raw;                                                // [4 waves][kRing][1 KB] operand rings
  unsigned char* s_smp = smem_raw + 4 * kRing * 1024;                              // [4 waves][6 KB]: one channel block's samples
  double* s_red = reinterpret_cast<double*>(s_smp + 4 * kSmpBytes);                // [4 waves][3]: loss, S_r, S_i partials
  float* s_c = reinterpret_cast<float*>(s_smp + 4 * kSmpBytes + 128);              // [ngk][64 lanes][4] packed coefficient operand

  const f32x4* ops = reinterpret_cast<const f32x4*>(A.ops);  // ONE scalar base for every operand request
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(s_c) + lane;
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned ring_lds = (unsigned)reinterpret_cast<unsigned long long>(s_ring + wave * kRing * 1024);
  const f32x4* ring_rd = reinterpret_cast<const f32x4*>(s_ring + wave * kRing * 1024) + lane;

  // The wave's operand stream, position by position (1 KB = 4 MFMAs each): for each of its channel blocks the ngk forward
  // positions, then the 4 NT adjoint positions.  The packed block stores them in exactly that order, wave after wave
  // (mfma_pack_kernel), so the gradient pass requests ONE contiguous run; the loss-only pass skips the adjoint positions.
  // A request cursor runs kRing positions ahead of the consumer through phases and channel blocks alike; past the end it
  // re-requests the last position.
  const int nb_pos = 4 * NT;
  // the wave's channel blocks: cb_of(n) = wave + 4 n, n = 0 .. nper - 1.  (Starting every panel at a different block, so
  // that panels of one basis block do not request the same operand lines at the same moment, measured no difference.)
  const int nper = ncb / 4;  // the row padding makes ncb a multiple of 4
  auto cb_of = [&](int n) { return wave + 4 * n; };
  const unsigned wblk = (unsigned)P.a_kf4 * 4u + (unsigned)(wave * nper * (ngk + nb_pos)) * 1024u;  // byte offset of this wave's run
  unsigned rq_off = wblk;
  const unsigned rq_last = wblk + (unsigned)(nper * (ngk + nb_pos) - (GRAD ? 1 : nb_pos + 1)) * 1024u;
  int rq_o = 0;
  auto req_advance = [&]() {
    if (GRAD) {
      rq_off = rq_off < rq_last ? rq_off + 1024u : rq_last;
    } else {
      ++rq_o;
      unsigned nxt = rq_off + 1024u;
      if (rq_o == ngk) { rq_o = 0; nxt += (unsigned)nb_pos * 1024u; }
      if (rq_off < rq_last) rq_off = nxt; else rq_o = 0;
    }
  };
  // ---- samples (d_r, d_i, w) of one channel block: they stream from HBM, so they are requested a whole block ahead -- by
  // LDS-DMA into a 6-KB area of the wave, not into registers: a register double buffer carried around the block loop makes
  // hipcc copy it at the loop edge and wait for the loads there.  Request k (0..5) fetches array k / 2, register groups
  // 2 (k & 1) + (lane >> 5): lane L brings the 16 bytes (4 channels) of slot L & 15, channel sub-group (L >> 4) & 1; the re and
  // im lane of a slot each read back their 8 bytes of it.
  const unsigned row = (unsigned)(my_bl >= 0 ? my_bl : A.nbls);
  const unsigned smp_lds = (unsigned)reinterpret_cast<unsigned long long>(s_smp + wave * kSmpBytes);
  const unsigned smp_voff = (row * (unsigned)A.fpad + 4u * ((unsigned)(lane >> 4) & 1u)) * 4u + (unsigned)(lane >> 5) * 32u;
  const unsigned char* smp_rd = s_smp + wave * kSmpBytes + (slot + 16 * half) * 16 + (im_lane ? 8 : 0);
  auto smp_issue = [&](int cbn) {
    const unsigned o = (unsigned)cbn * (kCB * 4u);
    ring_issue(smp_lds + 0u * 1024u, reinterpret_cast<const f32x4*>(A.data_r), smp_voff, o);
    ring_issue(smp_lds + 1u * 1024u, reinterpret_cast<const f32x4*>(A.data_r), smp_voff, o + 64u);
    ring_issue(smp_lds + 2u * 1024u, reinterpret_cast<const f32x4*>(A.data_i), smp_voff, o);
    ring_issue(smp_lds + 3u * 1024u, reinterpret_cast<const f32x4*>(A.data_i), smp_voff, o + 64u);
    ring_issue(smp_lds + 4u * 1024u, reinterpret_cast<const f32x4*>(A.wgts), smp_voff, o);
    ring_issue(smp_lds + 5u * 1024u, reinterpret_cast<const f32x4*>(A.wgts), smp_voff, o + 64u);
  };
  constexpr int kSmpReq = 6;
  const bool few_positions = (GRAD ? ngk + nb_pos : ngk) < kRing - 1;
  if (stopped) return;
  smp_issue(cb_of(0));  // older than every operand request: RING_WAIT(kRing - 1) in front of the first element stage covers them
  int cslot = 0;  // ring slot of the position the next step consumes
  f32x4 r_cur, r_nxt;
  for (int j = 0; j < kRing; ++j) {
    ring_issue(ring_lds + (unsigned)j * 1024u, ops, voff, rq_off);
    req_advance();
  }
  const unsigned ob = (row * (unsigned)A.fpad + 4u * half) * 4u;                  // float arrays
  const unsigned og0 = ((unsigned)my_ant.x * (unsigned)A.fpad + 4u * half) * 8u;  // float2 arrays
  const unsigned og1 = ((unsigned)my_ant.y * (unsigned)A.fpad + 4u * half) * 8u;
  {
    // coefficient panel in the packed operand layout: s_c[(g * 64 + l) * 4 + u] = c_part[col = l & 31][k = 8 g + 2 u + (l >> 5)],
    // cols 0-15 re, 16-31 im of the panel slots; zero beyond nvec and for padding slots.  Thread tid fills (l, u) =
    // ((tid >> 2) & 63, tid & 3) of every k-group: its column is fixed, and the loads of eight k-groups go out together,
    // unconditionally (clamped index)
    const int u = tid & 3, l = (tid >> 2) & 63;
    const float* src = ((l & 16) ? A.c_i : A.c_r) + fill_coff;
    const int k0 = 2 * u + (l >> 5);
    const int klast = nvec - 1;
    for (int g0 = 0; g0 < ngk; g0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * (g0 + j) + k0;
        v[j] = src[k < klast ? k : klast];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * (g0 + j) + k0;
        if (g0 + j < ngk) s_c[(g0 + j) * 256 + tid] = (fill_bl >= 0 && k < nvec) ? v[j] : 0.f;
      }
    }
  }
  __syncthreads();
  RING_WAIT(kRing - 1);
  r_cur = ring_rd[0];
  // One position = 4 MFMAs (256 cycles of the SIMD's matrix pipe) + what feeds the next ones.  A wave issues in order and
  // an MFMA waits for the pipe, so work placed BEHIND the four MFMAs runs in the shadow of the last one only; the feed work
  // therefore sits BETWEEN the MFMAs, a piece per 64-cycle gap, pinned with scheduling barriers:
  //   MFMA 1 | next position's operand has landed (all but the kRing - 2 youngest requests) -> read it back  (STREAM_NEXT)
  //   MFMA 2 | this position's slot is free (read one step ago) -> request position + kRing into it          (STREAM_REQ)
  // A wave's vector-memory operations retire in issue order and s_waitcnt counts them together.  The kSmpReq sample
  // requests of the element stage are YOUNGER than the kRing - 1 operand requests in flight at that moment, so the next
  // kRing - 1 STREAM_NEXT waits leave them out of the count (`relax`); after that they are older than anything awaited.
  // In the gradient pass those kRing - 1 waits are the first adjoint positions (compile-time); in the loss-only pass the
  // next block's forward positions (a counter).
  int relax = 0;
#define RING_WAIT_NEXT(RELAXED)                            \
  if (RELAXED) {                                           \
    RING_WAIT(kRing - 2 + kSmpReq);                        \
  } else {                                                 \
    RING_WAIT(kRing - 2);                                  \
  }
#define STREAM_NEXT_INTO(R, RELAXED)                       \
  __builtin_amdgcn_sched_barrier(0);                       \
  RING_WAIT_NEXT(RELAXED)                                  \
  R = ring_rd[(cslot + 1 == kRing ? 0 : cslot + 1) * 64];  \
  __builtin_amdgcn_sched_barrier(0);
#define STREAM_REQ()                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  ring_issue(ring_lds + (unsigned)cslot * 1024u, ops, voff, rq_off);                             \
  req_advance();                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
#define STREAM_STEP() cslot = cslot + 1 == kRing ? 0 : cslot + 1;

  f32x16 dC[NTMAX];  // coefficient-gradient tiles: lane (col, half), reg r -> vector 32 t + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
  for (int t = 0; t < NTMAX; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) dC[t][j] = 0.f;
  double loss_acc = 0.0, sr_acc = 0.0, si_acc = 0.0;
  const float al_r = A.use_alpha ? (float)sst->alpha_r : 0.f, al_i = A.use_alpha ? (float)sst->alpha_i : 0.f;
  const char* p_g = reinterpret_cast<const char*>(A.gains);
  char* p_q = reinterpret_cast<char*>(A.q0);
  // this lane's two channels of every register group: + 0, 1 on the re lane, + 2, 3 on the im lane
  const unsigned pl = im_lane ? 2u : 0u;
  const unsigned obp = ob + 4u * pl, og0p = og0 + 8u * pl, og1p = og1 + 8u * pl;
  typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef CAL_STAMP
  long long cyc_f = 0, cyc_e = 0, cyc_b = 0;
  const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int nb = 0; nb < nper; ++nb) {
    const int cb = cb_of(nb);
    STAMP_T(t0);
    // ---- F: rows = this block's 32 channels, cols = (slot, re | im), K = vectors; position = k-group (4 MFMAs)
    f32x16 acc;  // one chain: a dependent 32x32x2 may issue as soon as the pipe is free again (64 cycles either way)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    // two positions per trip, the operand registers alternating roles (a copy per position otherwise); the coefficient
    // panel has a spare position behind its last, so the read-ahead needs no clamp
    f32x4 c_cur = sc4[0], c_nxt;
    int gq = 0;
#define F_POS(RC, RN, CC, CN)                                                     \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(RC[0], CC[0], acc, 0, 0, 0);         \
  __builtin_amdgcn_sched_barrier(0);                                              \
  ++gq;                                                                           \
  CN = sc4[gq * 64];                                                              \
  __builtin_amdgcn_sched_barrier(0);                                              \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(RC[1], CC[1], acc, 0, 0, 0);         \
  {                                                                               \
    const bool rlx = !GRAD && relax > 0;                                          \
    STREAM_NEXT_INTO(RN, rlx)                                                     \
    if (!GRAD && rlx) --relax;                                                    \
  }                                                                               \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(RC[2], CC[2], acc, 0, 0, 0);         \
  STREAM_REQ()                                                                    \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(RC[3], CC[3], acc, 0, 0, 0);         \
  __builtin_amdgcn_sched_barrier(0);                                              \
  STREAM_STEP()
    for (int g = 0; g + 2 <= ngk; g += 2) {
      F_POS(r_cur, r_nxt, c_cur, c_nxt)
      F_POS(r_nxt, r_cur, c_nxt, c_cur)
    }
    if (ngk & 1) {
      F_POS(r_cur, r_nxt, c_cur, c_nxt)
      r_cur = r_nxt;
    }
#undef F_POS
    // acc[r] of lane (col, half) = v(channel 32 cb + (r & 3) + 8 (r >> 2) + 4 half) of column col

    STAMP_T(t1);
    // ---- E: element-wise.  Re and im of a slot sit 16 lanes apart; the pair shares the work: of the four channels of a
    // register group the re lane takes the first two, the im lane the last two.  One v_permlane16_swap of (acc[4g + i],
    // acc[4g + i + 2]) hands the re lane (re, im) of channel i and the im lane (re, im) of channel i + 2; a second swap
    // returns the halves of gbar_v the other lane needs, so that acc[] ends up as gbar_v in the layout v had.
    const unsigned cb8 = (unsigned)cb * (kCB * 8u);
    float lt = 0.f, st_r = 0.f, st_i = 0.f;
    // this block's samples were requested one block ago: they are older than the kRing - 1 youngest operand requests when
    // at least that many positions lie between (a block of fewer -- 16 vectors or less -- drains the queue instead)
    if (few_positions) {
      RING_WAIT(0);
    } else {
      RING_WAIT(kRing - 1);
    }
    f32x2 s_dr[4], s_di[4], s_w[4];  // [register group][this lane's two channels]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int o = ((g >> 1) * 64 + 32 * (g & 1)) * 16;
      s_dr[g] = *reinterpret_cast<const f32x2*>(smp_rd + o);
      s_di[g] = *reinterpret_cast<const f32x2*>(smp_rd + 2048 + o);
      s_w[g] = *reinterpret_cast<const f32x2*>(smp_rd + 4096 + o);
    }
    // the two antennas' gains (they come from L2): all four register groups at once where the registers allow it (one
    // round trip instead of four), else one group ahead
    constexpr int GA = NTMAX <= 4 ? 4 : 2;  // gain quads per antenna in flight
    f32x4 ga4[GA], gb4[GA];
#pragma unroll
    for (int g = 0; g < (GA == 4 ? 4 : 1); ++g) {
      ga4[g] = *reinterpret_cast<const f32x4*>(p_g + (og0p + cb8 + 64u * g));
      gb4[g] = *reinterpret_cast<const f32x4*>(p_g + (og1p + cb8 + 64u * g));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (GA == 2 && g < 3) {
        ga4[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(p_g + (og0p + cb8 + 64u * (g + 1)));
        gb4[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(p_g + (og1p + cb8 + 64u * (g + 1)));
      }
      const f32x4 ga = ga4[GA == 4 ? g : (g & 1)], gb = gb4[GA == 4 ? g : (g & 1)];
      f32x4 qs;  // gbar_G of this lane's two channels, (re, im) interleaved
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        // (bit_cast applied to a vector ELEMENT expression reads element 0 with this hipcc: go through scalars)
        const float xa = acc[4 * g + i], xb = acc[4 * g + i + 2];
        const u2 pr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, xa), __builtin_bit_cast(unsigned, xb), false, false);
        const unsigned p0 = pr[0], p1 = pr[1];
        const float vr = __builtin_bit_cast(float, p0), vi = __builtin_bit_cast(float, p1);  // channel i (re lane) / i + 2 (im lane)
        const float d_r = s_dr[g][i], d_i = s_di[g][i], w = s_w[g][i];
        const float g0x = ga[2 * i], g0y = ga[2 * i + 1], g1x = gb[2 * i], g1y = gb[2 * i + 1];
        // G = g0 conj(g1)   (calibration.py:1598-1601: grgr + gigi, gigr - grgi)
        const float G_r = g0x * g1x + g0y * g1y;
        const float G_i = g0y * g1x - g0x * g1y;
        const float m_r = G_r * vr - G_i * vi;
        const float m_i = G_i * vr + G_r * vi;
        const float r_r = d_r - m_r, r_i = d_i - m_i;
        lt += w * (r_r * r_r + r_i * r_i);
        st_r += w * m_r;  // S = sum w m of the "sum" regulariser (calibration.py:1648-1649)
        st_i += w * m_i;
        if (GRAD) {
          const float e_r = -2.f * w * r_r + al_r * w, e_i = -2.f * w * r_i + al_i * w;
          // gbar_v = conj(G) e.  The re lane keeps the real part of its channel and needs the real part of the im lane's
          // channel; the im lane keeps the imaginary part of its channel and needs the imaginary part of the re lane's.
          const float gv_r = G_r * e_r + G_i * e_i, gv_i = G_r * e_i - G_i * e_r;
          const float give = im_lane ? gv_r : gv_i;
          const u2 qr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, give), __builtin_bit_cast(unsigned, give), false, false);
          const unsigned q0b = qr[0], q1b = qr[1];
          // result 0 = the re lane's value, result 1 = the im lane's value, on both lanes
          const float from_re = __builtin_bit_cast(float, q0b), from_im = __builtin_bit_cast(float, q1b);
          acc[4 * g + i] = im_lane ? from_re : gv_r;      // channel i:     re lane: own real part;        im lane: re lane's imaginary part
          acc[4 * g + i + 2] = im_lane ? gv_i : from_im;  // channel i + 2: re lane: im lane's real part;  im lane: own imaginary part
          // gbar_G = conj(v) e
          qs[2 * i] = vr * e_r + vi * e_i;
          qs[2 * i + 1] = vr * e_i - vi * e_r;
        }
      }
      if (GRAD) *reinterpret_cast<f32x4*>(p_q + (2u * obp + cb8 + 64u * g)) = qs;
    }
    loss_acc += (double)lt;
    sr_acc += (double)st_r;
    si_acc += (double)st_i;
    // the next block's samples (the last block requests itself again: the count of requests in flight stays what the
    // waits assume)
    __builtin_amdgcn_sched_barrier(0);
    smp_issue(cb_of(nb + 1 < nper ? nb + 1 : nb));
    if (!GRAD) relax = kRing - 1;
    __builtin_amdgcn_sched_barrier(0);

    STAMP_T(t2);
    // ---- B: rows = vectors of tile t, cols = (slot, re | im), K = this block's channels; position = (tile, register group q)
    if (GRAD) {
#define B_POS(T, Q)                                                                              \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[0], acc[4 * (Q) + 0], dC[T], 0, 0, 0);      \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[1], acc[4 * (Q) + 1], dC[T], 0, 0, 0);      \
  STREAM_NEXT_INTO(r_nxt, 4 * (T) + (Q) < kRing - 1)                                             \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[2], acc[4 * (Q) + 2], dC[T], 0, 0, 0);      \
  STREAM_REQ()                                                                                   \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[3], acc[4 * (Q) + 3], dC[T], 0, 0, 0);      \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  r_cur = r_nxt;                                                                                 \
  STREAM_STEP()
#pragma unroll
      for (int t = 0; t < NTMAX; ++t) {
        if (t < NT) {  // wave-uniform
          B_POS(t, 0)
          B_POS(t, 1)
          B_POS(t, 2)
          B_POS(t, 3)
        }
      }
#undef B_POS
    }
#ifdef CAL_STAMP
    {
      const long long t3 = (long long)__builtin_amdgcn_s_memtime();
      cyc_f += t1 - t0; cyc_e += t2 - t1; cyc_b += t3 - t2;
    }
#endif
  }
#ifdef CAL_STAMP
  if (lane == 0 && panel_idx < 4096) {
    long long* o = g_dense_stamps[panel_idx][wave];
    o[0] = cyc_f; o[1] = cyc_e; o[2] = cyc_b; o[3] = (long long)__builtin_amdgcn_s_memtime() - t_begin;
    o[4] = t_begin - t_entry; o[6] = t_entry;
  }
  const long long t_loop_end = (long long)__builtin_amdgcn_s_memtime();
#define STAMP_EXIT()                                                              \
  if (lane == 0 && panel_idx < 4096) {                                            \
    long long* o = g_dense_stamps[panel_idx][wave];                               \
    o[7] = (long long)__builtin_amdgcn_s_memtime(); o[5] = o[7] - t_loop_end;     \
  }
#else
#define STAMP_EXIT()
#endif
#undef STREAM_NEXT_INTO
#undef RING_WAIT_NEXT
#undef STREAM_REQ
#undef STREAM_STEP
  RING_WAIT(0);  // retire the trailing requests: the epilogue reuses the ring area

  // ---- panel epilogue: loss partials (double, fixed order), then the coefficient gradients
  {
    const double l = ldsum(loss_acc), sr = ldsum(sr_acc), si = ldsum(si_acc);
    if (lane == 0) {
      s_red[wave * 3 + 0] = l;
      s_red[wave * 3 + 1] = sr;
      s_red[wave * 3 + 2] = si;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const size_t pi = (size_t)panel_idx * 4;
    A.part[pi + 0] = s_red[0] + s_red[3] + s_red[6] + s_red[9];
    A.part[pi + 1] = s_red[1] + s_red[4] + s_red[7] + s_red[10];
    A.part[pi + 2] = s_red[2] + s_red[5] + s_red[8] + s_red[11];
  }
  if (!GRAD) { STAMP_EXIT() return; }
  // each wave holds the sums over ITS channel blocks; two tiles at a time the four parts meet in the (now idle) ring and
  // sample areas ([2 tiles][4 waves][16 regs][64 lanes] = 32 KB of their 40 KB or more) and ALL 256 threads add them, in wave order: thread (tile tid >> 7,
  // vector octet (tid >> 5) & 3, column tid & 31) owns 8 consecutive vectors of its column -- registers 4 o .. 4 o + 3 of the
  // two lane halves -- and stores them as one contiguous run
  float* s_x = reinterpret_cast<float*>(s_ring);
  const int e_tile = tid >> 7, e_oct = (tid >> 5) & 3;
  float* gc = (im_lane ? A.gc_i : A.gc_r) + my_coff;  // the thread's column is its MFMA column: tid & 31 == lane & 31
#pragma unroll
  for (int t0 = 0; t0 < NTMAX; t0 += 2) {
    if (t0 < NT) {  // wave-uniform
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        if (t0 + tt < NTMAX && t0 + tt < NT) {
#pragma unroll
          for (int j = 0; j < 16; ++j) s_x[((tt * 4 + wave) * 16 + j) * 64 + lane] = dC[t0 + tt][j];
        }
      __syncthreads();
      const int t = t0 + e_tile;
      if (t < NT && my_bl >= 0) {
        const int nbase = 32 * t + 8 * e_oct;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float* px = s_x + ((e_tile * 4) * 16 + 4 * e_oct + jj) * 64 + col + 32 * hf;
            const float v = ((px[0] + px[16 * 64]) + px[2 * 16 * 64]) + px[3 * 16 * 64];
            const int n = nbase + 4 * hf + jj;
            if (n < nvec) gc[n] = v;
          }
      }
      __syncthreads();
    }
  }
  STAMP_EXIT()
}

// One launch for all panels: the body is instantiated for panels of up to 4 and of up to 8 vector tiles (registers for 64 / 128
// gradient accumulators; two launches, one per class, leave the CUs idle while the first drains).
// XCD-affine dispatch: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), and each XCD has its
// own 4-MiB L2.  The host gives every basis block -- all its panels -- to ONE of 8 lists of equal cost (SolverT::order_panels)
// and workgroup b takes entry b / 8 of list b % 8 (slot_map; lists shorter than the longest leave empty slots), heaviest
// panels first in every list.  A block's packed operands then stream through one XCD's L2 instead of all eight: with panels
// dealt in one global order every XCD fetched every block.  (Placement is a speed matter only: any workgroup computes any panel.)
template <bool GRAD>
__global__ __launch_bounds__(kDenseThreads, 2) void fused_dense_kernel(const MfmaArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int slot = A.slot_map[blockIdx.x];
  if (slot < 0) return;
  if (A.panels[slot].nvp32 > 128)
    dense_panel<GRAD, 8>(A, smem_raw);
  else
    dense_panel<GRAD, 4>(A, smem_raw);
}

inline size_t dense_lds_bytes(int nvec_max, bool grad, int ntmax) {
  return 4 * (size_t)dense_ring_slots(grad, ntmax) * 1024 + 128 + 4 * (size_t)kSmpBytes + (size_t)((nvec_max + 7) / 8 + 1) * 1024;  // + the spare coefficient position
}

// packed MFMA-native operand layout of the dense kernel (see PanelItem): per basis block, wave after wave (w = 0..3), the
// wave's channel blocks cb = w + 4 n in order, each as [ngk forward KB][4 NT adjoint KB] -- the order the kernel consumes
__global__ void mfma_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int nfreqs, int fpad, int nvec, int nvp32) {
  const int ngk = (nvec + 7) / 8, NT = nvp32 / 32;
  const int nper = fpad / 128;
  const long long per_cb = (long long)(ngk + 4 * NT) * 256, total = per_cb * (fpad / 32);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cbp = (int)(i / per_cb);
    const int r = (int)(i % per_cb);
    const int cb = cbp / nper + 4 * (cbp % nper);
    const int u = r & 3, l = (r >> 2) & 63;
    float v = 0.f;
    if (r < ngk * 256) {
      const int g = r >> 8;
      const int f = cb * 32 + (l & 31), k = 8 * g + 2 * u + (l >> 5);
      if (f < nfreqs && k < nvec) v = src[(long long)f * nvec + k];
    } else {
      const int pos = (r >> 8) - ngk;
      const int qd = pos & 3, t = pos >> 2;
      const int f = 32 * cb + 8 * qd + 4 * (l >> 5) + u, n = 32 * t + (l & 31);
      if (f < nfreqs && n < nvec) v = src[(long long)f * nvec + n];
    }
    dst[i] = v;
  }
}

}  // namespace calk
