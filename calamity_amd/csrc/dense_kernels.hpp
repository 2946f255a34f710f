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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// which solver dtypes have a dense kernel, and the widest basis block it takes
template <typename T> struct DenseCfg { static constexpr int max_nvec = 0; };
template <> struct DenseCfg<float> { static constexpr int max_nvec = 32 * kMaxNT; };

struct PanelItem {
  int bl[kPanel];     // baseline ids (-1: padding slot)
  int nvec, nvp2, nvp32;
  int pad;
  // element offsets into MfmaArgs::ops (the whole buffer stays below 4 GB: the kernel adds 32-bit byte offsets to ONE base)
  long long a_kf4;    // packed forward operand  [F/32][ceil(nvec/8)][64 lanes][4]: lane (c, half), u -> A[32 cb + c][8 g + 2 u + half]
  long long a_fk4;    // packed adjoint operand  [F/32][nvp32/32][4][64 lanes][4]: lane (i, half), u -> A[32 cb + 8 q + 4 half + u][32 t + i]
};

struct MfmaArgs {
  const float* ops;            // packed operands of every basis block
  const PanelItem* panels;
  const int2* bl_ant;
  const int* bl_coff;          // coefficient offset of each baseline's group
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
  const DevState* state;
  int fpad;
  int use_alpha;               // "sum" regulariser, second pass: e = -2 w r + alpha w with alpha = 2 (S - P) read from state
  int nbls;                    // row nbls of data_r / data_i / wgts / q0 is an all-zero spare row for padding slots
  int panel_base;              // first panel of this launch (panels are launched in two classes, by vector-tile count)
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
constexpr int kRingMax = 8;                 // operand-ring slots (1 KB each) per wave: 8 in the gradient pass, 4 in the loss-only pass
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
__device__ long long g_dense_stamps[4096][4][4];  // diagnostic build only: [panel][wave][F, E, B, total] cycles
#define STAMP_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#else
#define STAMP_T(var)
#endif
template <bool GRAD, int NTMAX>
__global__ __launch_bounds__(kDenseThreads, 2) void fused_dense_kernel(const MfmaArgs A) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // measured at HERA-350: 8 slots 0.955 ms / 4 slots 1.00 ms for the gradient pass (a third to a half of the operand
  // requests miss L2: more of them in flight), 0.53 / 0.50 ms for the loss-only pass (whose smaller LDS footprint buys a
  // third workgroup per CU)
  constexpr int kRing = GRAD ? 8 : 4;
  if (A.state->done | A.state->done_after) return;
  const int panel_idx = A.panel_base + (int)blockIdx.x;
  const PanelItem& P = A.panels[panel_idx];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave-uniform quantities live in SGPRs: every operand address below is (scalar base) + (lane offset)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31;       // MFMA column: (slot, re | im) of the panel
  const int half = lane >> 5;      // the k of a 32x32x2 step this lane feeds / the channel sub-group (+4) it holds
  const int slot = col & 15;
  const bool im_lane = (col & 16) != 0;
  const int nvec = P.nvec, NT = P.nvp32 / 32;
  const int ngk = (nvec + 7) / 8;  // forward k-groups of 8 vectors (4 k-steps)
  const int ncb = A.fpad / kCB;

  unsigned char* s_ring = smem_raw;                                                // [4 waves][kRing][1 KB] operand rings, lowest LDS addresses
  double* s_red = reinterpret_cast<double*>(smem_raw + 4 * kRing * 1024);          // [4 waves][3]: loss, S_r, S_i partials
  float* s_c = reinterpret_cast<float*>(smem_raw + 4 * kRing * 1024 + 128);        // [ngk][64 lanes][4] packed coefficient operand
  {
    // coefficient panel in the packed operand layout: s_c[(g * 64 + lane) * 4 + u] = c_part[col = lane & 31][k = 8 g + 2 u + (lane >> 5)],
    // cols 0-15 re, 16-31 im of the panel slots; zero beyond nvec and for padding slots
    const int n = ngk * 256;
    for (int i = tid; i < n; i += kDenseThreads) {
      const int u = i & 3, l = (i >> 2) & 63, g = i >> 8;
      const int c = l & 31, k = 8 * g + 2 * u + (l >> 5);
      const int b = P.bl[c & 15];
      float v = 0.f;
      if (b >= 0 && k < nvec) v = (c < 16 ? A.c_r : A.c_i)[A.bl_coff[b] + k];
      s_c[i] = v;
    }
  }
  // this lane's slot: sample row (padding slots use the all-zero spare row: weight 0) and antenna pair, as 32-bit BYTE offsets
  // from the kernel-argument bases (scalar base + unsigned 32-bit offset is the form hipcc turns into `global_load v, v_off,
  // s[base]`; with element offsets it builds a 64-bit address pair per load)
  const int my_bl = P.bl[slot];
  const int2 my_ant = my_bl >= 0 ? A.bl_ant[my_bl] : make_int2(0, 0);
  const unsigned row = (unsigned)(my_bl >= 0 ? my_bl : A.nbls);
  const unsigned ob = (row * (unsigned)A.fpad + 4u * half) * 4u;            // float arrays
  const unsigned og0 = ((unsigned)my_ant.x * (unsigned)A.fpad + 4u * half) * 8u;  // float2 arrays
  const unsigned og1 = ((unsigned)my_ant.y * (unsigned)A.fpad + 4u * half) * 8u;
  __syncthreads();

  const f32x4* ops = reinterpret_cast<const f32x4*>(A.ops);  // ONE scalar base for every operand request
  const unsigned fblk = (unsigned)P.a_kf4 * 4u, bblk = (unsigned)P.a_fk4 * 4u;  // byte offsets of this panel's two packed blocks
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(s_c) + lane;
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned ring_lds = (unsigned)reinterpret_cast<unsigned long long>(s_ring + wave * kRing * 1024);
  const f32x4* ring_rd = reinterpret_cast<const f32x4*>(s_ring + wave * kRing * 1024) + lane;

  // The wave's operand stream, position by position (1 KB = 4 MFMAs each): for each of its channel blocks the ngk forward
  // positions, then (GRAD) the 4 NT adjoint positions -- both contiguous in the packed blocks.  A request cursor runs kRing
  // positions ahead of the consumer through phases and channel blocks alike; past the last block it re-requests the last.
  const int nb_pos = GRAD ? 4 * NT : 0;
  // the wave's channel blocks: cb_of(n) = wave + 4 n, n = 0 .. nper - 1.  (Starting every panel at a different block, so
  // that panels of one basis block do not request the same operand lines at the same moment, measured no difference.)
  const int nper = ncb / 4;  // the row padding makes ncb a multiple of 4
  auto cb_of = [&](int n) { return wave + 4 * n; };
  int rq_n = 0, rq_o = 0;
  auto req_off = [&]() {
    const int cbv = cb_of(rq_n < nper ? rq_n : nper - 1);
    return rq_o < ngk ? fblk + (unsigned)(cbv * ngk + rq_o) * 1024u : bblk + (unsigned)(cbv * nb_pos + (rq_o - ngk)) * 1024u;
  };
  auto req_advance = [&]() {
    ++rq_o;
    if (rq_o == ngk + nb_pos) { rq_o = 0; ++rq_n; }
  };
  int cons = 0;  // stream index of the position the next step consumes; its slot is cons % kRing
  f32x4 r_cur, r_nxt;
  for (int j = 0; j < kRing; ++j) {
    ring_issue(ring_lds + (unsigned)j * 1024u, ops, voff, req_off());
    req_advance();
  }
  RING_WAIT(kRing - 1);
  r_cur = ring_rd[0];
  // One position = 4 MFMAs (256 cycles of the SIMD's matrix pipe) + what feeds the next ones.  A wave issues in order and
  // an MFMA waits for the pipe, so work placed BEHIND the four MFMAs runs in the shadow of the last one only; the feed work
  // therefore sits BETWEEN the MFMAs, a piece per 64-cycle gap, pinned with scheduling barriers:
  //   MFMA 1 | next position's operand has landed (all but the kRing - 2 youngest requests) -> read it back  (STREAM_NEXT)
  //   MFMA 2 | this position's slot is free (read one step ago) -> request position + kRing into it          (STREAM_REQ)
#define STREAM_NEXT()                                      \
  __builtin_amdgcn_sched_barrier(0);                       \
  RING_WAIT(kRing - 2);                                    \
  r_nxt = ring_rd[((cons + 1) & (kRing - 1)) * 64];        \
  __builtin_amdgcn_sched_barrier(0);
#define STREAM_REQ()                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  ring_issue(ring_lds + (unsigned)(cons & (kRing - 1)) * 1024u, ops, voff, req_off());           \
  req_advance();                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
#define STREAM_ADVANCE() \
  r_cur = r_nxt;         \
  ++cons;

  f32x16 dC[NTMAX];  // coefficient-gradient tiles: lane (col, half), reg r -> vector 32 t + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
  for (int t = 0; t < NTMAX; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) dC[t][j] = 0.f;
  double loss_acc = 0.0, sr_acc = 0.0, si_acc = 0.0;
  const float al_r = A.use_alpha ? (float)A.state->alpha_r : 0.f, al_i = A.use_alpha ? (float)A.state->alpha_i : 0.f;
  const char* p_dr = reinterpret_cast<const char*>(A.data_r);
  const char* p_di = reinterpret_cast<const char*>(A.data_i);
  const char* p_w = reinterpret_cast<const char*>(A.wgts);
  const char* p_g = reinterpret_cast<const char*>(A.gains);
  char* p_q = reinterpret_cast<char*>(A.q0);
  // this lane's two channels of every register group: + 0, 1 on the re lane, + 2, 3 on the im lane
  const unsigned pl = im_lane ? 2u : 0u;
  const unsigned obp = ob + 4u * pl, og0p = og0 + 8u * pl, og1p = og1 + 8u * pl;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  struct Samples { f32x2 dr[4], di[4], w[4]; };  // one channel block: [register group][this lane's two channels]
  auto load_samples = [&](int cbn, Samples& S) {
    const unsigned o = obp + (unsigned)cbn * (kCB * 4u);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      S.dr[g] = *reinterpret_cast<const f32x2*>(p_dr + (o + 32u * g));
      S.di[g] = *reinterpret_cast<const f32x2*>(p_di + (o + 32u * g));
      S.w[g] = *reinterpret_cast<const f32x2*>(p_w + (o + 32u * g));
    }
  };
  Samples S_cur, S_nxt;
  load_samples(cb_of(0), S_nxt);

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
    f32x4 c_cur = sc4[0], c_nxt;
    for (int g = 0; g < ngk; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[0], c_cur[0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      c_nxt = sc4[(g + 1 < ngk ? g + 1 : ngk - 1) * 64];
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[1], c_cur[1], acc, 0, 0, 0);
      STREAM_NEXT()
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[2], c_cur[2], acc, 0, 0, 0);
      STREAM_REQ()
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[3], c_cur[3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      STREAM_ADVANCE()
      c_cur = c_nxt;
    }
    // acc[r] of lane (col, half) = v(channel 32 cb + (r & 3) + 8 (r >> 2) + 4 half) of column col

    STAMP_T(t1);
    // ---- E: element-wise.  Re and im of a slot sit 16 lanes apart; the pair shares the work: of the four channels of a
    // register group the re lane takes the first two, the im lane the last two.  One v_permlane16_swap of (acc[4g + i],
    // acc[4g + i + 2]) hands the re lane (re, im) of channel i and the im lane (re, im) of channel i + 2; a second swap
    // returns the halves of gbar_v the other lane needs, so that acc[] ends up as gbar_v in the layout v had.
    const unsigned cb8 = (unsigned)cb * (kCB * 8u);
    float lt = 0.f, st_r = 0.f, st_i = 0.f;
    // the two antennas' gains (they come from L2): all four register groups at once where the registers allow it (one
    // round trip instead of four), else one group ahead
    constexpr int GA = NTMAX <= 4 ? 4 : 2;  // gain quads per antenna in flight
    f32x4 ga4[GA], gb4[GA];
#pragma unroll
    for (int g = 0; g < (GA == 4 ? 4 : 1); ++g) {
      ga4[g] = *reinterpret_cast<const f32x4*>(p_g + (og0p + cb8 + 64u * g));
      gb4[g] = *reinterpret_cast<const f32x4*>(p_g + (og1p + cb8 + 64u * g));
    }
    // this block's samples were requested one block ago (the copy waits for them HERE, where they are needed, not where
    // they were requested); the NEXT block's are requested now: they stream from HBM, and a wave's memory operations
    // retire in order, so the only place such a request does not park the operand stream behind it is in front of this
    // arithmetic
    S_cur = S_nxt;
    if (nb + 1 < nper) load_samples(cb_of(nb + 1), S_nxt);
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
        const float d_r = S_cur.dr[g][i], d_i = S_cur.di[g][i], w = S_cur.w[g][i];
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

    STAMP_T(t2);
    // ---- B: rows = vectors of tile t, cols = (slot, re | im), K = this block's channels; position = (tile, register group q)
    if (GRAD) {
#define B_POS(T, Q)                                                                              \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[0], acc[4 * (Q) + 0], dC[T], 0, 0, 0);      \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[1], acc[4 * (Q) + 1], dC[T], 0, 0, 0);      \
  STREAM_NEXT()                                                                                  \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[2], acc[4 * (Q) + 2], dC[T], 0, 0, 0);      \
  STREAM_REQ()                                                                                   \
  dC[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(r_cur[3], acc[4 * (Q) + 3], dC[T], 0, 0, 0);      \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  STREAM_ADVANCE()
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
  }
#endif
#undef STREAM_NEXT
#undef STREAM_REQ
#undef STREAM_ADVANCE
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
  if (!GRAD) return;
  // each wave holds the sums over ITS channel blocks; tile by tile the four parts meet in the (now idle) ring area and
  // wave t % 4 adds them in wave order and stores the tile
  float* s_x = reinterpret_cast<float*>(s_ring);  // [4 waves][16 regs][64 lanes] = 16 KB
  const int coff = my_bl >= 0 ? A.bl_coff[my_bl] : 0;
  float* gc = im_lane ? A.gc_i : A.gc_r;
#pragma unroll
  for (int t = 0; t < NTMAX; ++t) {
    if (t < NT) {
#pragma unroll
      for (int j = 0; j < 16; ++j) s_x[(wave * 16 + j) * 64 + lane] = dC[t][j];
      __syncthreads();
      if (wave == (t & 3) && my_bl >= 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int n = 32 * t + (j & 3) + 8 * (j >> 2) + 4 * half;
          const float v = ((s_x[(0 * 16 + j) * 64 + lane] + s_x[(1 * 16 + j) * 64 + lane]) + s_x[(2 * 16 + j) * 64 + lane]) + s_x[(3 * 16 + j) * 64 + lane];
          if (n < nvec) gc[coff + n] = v;
        }
      }
      __syncthreads();
    }
  }
}

inline size_t dense_lds_bytes(int nvec_max, bool grad) { return 4 * (size_t)(grad ? 8 : 4) * 1024 + 128 + (size_t)((nvec_max + 7) / 8) * 1024; }

// packed MFMA-native operand layouts of the dense kernel (see PanelItem)
__global__ void mfma_pack_kernel(const float* __restrict__ src, float* __restrict__ a_kf4, float* __restrict__ a_fk4, int nfreqs, int fpad,
                                 int nvec, int nvp32) {
  const int ngk = (nvec + 7) / 8, NT = nvp32 / 32;
  const long long n1 = (long long)(fpad / 32) * ngk * 256, n2 = (long long)(fpad / 32) * NT * 4 * 256;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int u = (int)(i & 3), l = (int)((i >> 2) & 63);
      const long long r = i >> 8;
      const int g = (int)(r % ngk), cb = (int)(r / ngk);
      const int f = cb * 32 + (l & 31), k = 8 * g + 2 * u + (l >> 5);
      a_kf4[i] = (f < nfreqs && k < nvec) ? src[(long long)f * nvec + k] : 0.f;
    } else {
      const long long q = i - n1;
      const int u = (int)(q & 3), l = (int)((q >> 2) & 63);
      long long r = q >> 8;
      const int qd = (int)(r & 3);
      r >>= 2;
      const int t = (int)(r % NT), cb = (int)(r / NT);
      const int f = 32 * cb + 8 * qd + 4 * (l >> 5) + u, n = 32 * t + (l & 31);
      a_fk4[q] = (f < nfreqs && n < nvec) ? src[(long long)f * nvec + n] : 0.f;
    }
  }
}

}  // namespace calk
