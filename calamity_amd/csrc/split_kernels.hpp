// split_kernels.hpp -- the dense formulation of the fit step for the SHARED layout (dense_kernels.hpp) on
// v_mfma_f32_32x32x16_bf16 with split-bf16 operands: the fp32 product a c is formed as
//     a c ~= a1 c1 + a1 c2 + a2 c1 + a1 c3 + a2 c2 + a3 c1,      x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (x1 + x2 + x3 == x for a normal fp32 x; every bf16 x bf16 product is exact in the fp32 accumulator; the three dropped terms are
// <= 2^-24 |a c| each: tools/micro/bf16x3.hip measures the six products MORE accurate than the fp32 MFMA chain they replace, at 2.1-2.2 x its
// rate).  Same arithmetic as dense_kernels.hpp: fg_model (/root/reference/calamity/calibration.py:1587-1590), data_model (:1593-1604),
// mse (:1607-1609) and their adjoints, per step of fit_gains_and_foregrounds (:663-668).
//
// Partition.  The bf16 instruction is 2.7 x faster per flop and takes three operand planes, so a wave consumes operand bytes four
// times faster than in the f32 kernel: the basis operands must be SHARED.  A workgroup = 4 waves = a SUPER-PANEL of four panels of 16 baselines
// with the same basis block; wave w owns panel w (32 columns = (slot, re | im)) for the whole sweep: its coefficient gradients stay in its
// registers, there is no cross-wave sum.  All four waves walk the same sequence of operand GROUPS (4 positions of 3 KB: three bf16 planes of a
// 32 x 16 piece of the block each), which stream through ONE LDS ring filled by LDS-DMA: wave w requests position w of every group; per group
// one counted vmcnt wait for the own request and one s_barrier (which also frees the previous group's slots).  A group is one step of the code:
// per pair of channel blocks (cb0, cb1 = 2 cp, 2 cp + 1; 32 channels each)
//   F  per group two K-steps of 16 vectors x two channel blocks: the wave's coefficient operand of the K-step (32 columns x 16 vectors, fp32,
//      LDS-DMA into a per-wave buffer, split into three planes in registers -- once for BOTH channel blocks), 6 MFMAs each on V(cb0), V(cb1)
//   E  element stage of cb0 in the accumulator registers, gbar_v split into three planes: the accumulator is, register by register, the B
//      operand of the adjoint (register 8 q + j of lane (col, half) = channel 16 q + (j & 3) + 8 (j >> 2) + 4 half: the packed adjoint operand is
//      stored in that k order)
//   B  per group two vector tiles x two K-steps of 16 channels: 12 MFMAs on dC[t], 12 on dC[t + 1]
//   E, B of cb1.
// An item carries at most 4 vector tiles (64 gradient accumulators: two workgroups per CU); a basis block of more than 128 vectors is TWO items
// per super-panel, each with the whole forward and half of the tiles (the second one writes no loss and no gbar_G).
// Vector-memory accounting: every asm request is counted (`issued`), every awaited request remembers the count at its issue, and a wait is
// s_waitcnt vmcnt(issued - mark) -- a jump into a table of s_waitcnt instructions (the instruction takes an immediate).  Operations the compiler
// issues itself (gain loads, gbar_G stores) are not in the count: they only make a wait stricter -- towards OLDER operations -- never weaker.
#pragma once
#include <type_traits>

#include "dense_kernels.hpp"

namespace calk {

constexpr int kSpWaves = 4;            // panels (waves) per super-panel
constexpr int kPosBytes = 3072;        // one operand position: [3 planes][64 lanes][16 B]
constexpr int kGroupBytes = 4 * kPosBytes;
constexpr int kSpRingMax = 2;             // groups in the operand ring (measured at HERA-350, one workgroup per CU: 2 groups 0.728 ms, 3 groups 0.752, up to 6 as the LDS allows 0.756)
constexpr int kSplitLds = 160 * 1024;     // the workgroup's LDS: all of the CU's
constexpr int kSplitNT = 8;              // vector tiles per item at most (128 gradient accumulators)
constexpr int kSplitMaxNvec = 224;       // 7 forward groups: the coefficient panels of the four waves stay in LDS (4 x 28 KB) beside ring and samples

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS of a workgroup: the operand ring, per wave its coefficient panel (fp32, c_wave_bytes = 4 KB per forward group of the widest block) and the
// samples of one channel block.  ONE workgroup per CU (one wave per SIMD, 512 registers): halving the occupancy costs this kernel 13 % -- it
// is bound by the traffic it draws from the fabric, and the LDS a single workgroup has keeps the coefficients resident (they were 1.3 GB of
// re-reads per pass) and the registers carry all eight vector tiles of the widest blocks (they were two items, each with the whole forward).
inline int split_c_wave_bytes(int nvec) { return (((nvec + 15) / 16 + 1) / 2) * 4096; }
inline size_t split_lds_bytes() { return kSplitLds; }
// groups of one item's packed stream per channel-block pair: forward (two K-steps each), then the adjoint of cb0 and of cb1 (two tiles each)
inline int split_groups_per_pair(int nvec, int ntiles) { return ((nvec + 15) / 16 + 1) / 2 + 2 * ((ntiles + 1) / 2); }
inline long long split_stream_bytes(int fpad, int nvec, int ntiles) { return (long long)(fpad / 64) * split_groups_per_pair(nvec, ntiles) * kGroupBytes; }

// s_waitcnt vmcnt(n), n wave-uniform at run time: a jump into a table of (s_waitcnt vmcnt(k); s_branch end) pairs.  ONE asm statement: the
// compiler sees no control flow (a switch at every wait made the kernel's flow graph -- and its register allocation -- unmanageable).
__device__ __forceinline__ void wait_vm_dyn(int n) {
  // (n comes from scalar arithmetic on wave-uniform counters; the clamp is done inside the asm: written in C++ hipcc evaluated it in the
  // vector ALU and handed the asm a VGPR.)  More outstanding requests than the table holds: the wait is stricter, never weaker.
  asm volatile(
      "s_max_i32 s97, %0, 0\n\t"
      "s_min_i32 s97, s97, 31\n\t"
      "s_getpc_b64 s[98:99]\n\t"      // = address of the next instruction
      "s_lshl_b32 s97, s97, 3\n\t"     // 8 bytes per table entry
      "s_add_u32 s97, s97, 20\n\t"     // the five 4-byte instructions from there to the table
      "s_add_u32 s98, s98, s97\n\t"
      "s_addc_u32 s99, s99, 0\n\t"
      "s_setpc_b64 s[98:99]\n\t"
#define CAL_WV(N) "s_waitcnt vmcnt(" #N ")\n\ts_branch .Lcal_wv_end%=\n\t"
      CAL_WV(0) CAL_WV(1) CAL_WV(2) CAL_WV(3) CAL_WV(4) CAL_WV(5) CAL_WV(6) CAL_WV(7) CAL_WV(8) CAL_WV(9) CAL_WV(10) CAL_WV(11)
      CAL_WV(12) CAL_WV(13) CAL_WV(14) CAL_WV(15) CAL_WV(16) CAL_WV(17) CAL_WV(18) CAL_WV(19) CAL_WV(20) CAL_WV(21) CAL_WV(22)
      CAL_WV(23) CAL_WV(24) CAL_WV(25) CAL_WV(26) CAL_WV(27) CAL_WV(28) CAL_WV(29) CAL_WV(30) CAL_WV(31)
#undef CAL_WV
      ".Lcal_wv_end%=:"
      :
      : "s"(n)
      : "memory", "s97", "s98", "s99", "scc");
  __builtin_amdgcn_sched_barrier(0);
}

// LDS-DMA requests (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KB, no register destination).  M0 carries the LDS address; it is a register
// the compiler reserves and re-initialises in front of every use of its own, so it is set and left.  `base` must be wave-uniform (dma_base).
__device__ __forceinline__ const void* dma_base(const void* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void dma1(unsigned lds, const void* base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}
// three consecutive kilobytes: the instruction offset moves the global and the LDS address alike
__device__ __forceinline__ void dma3(unsigned lds, const void* base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048" ::"s"(lds), "v"(voff), "s"(base) : "memory");
}

// x (8 fp32) -> three bf16 planes, round to nearest even at every level (v_cvt_pk_bf16_f32)
template <bool CHEAP = false>
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  if (CHEAP) {  // (ablation builds: one conversion, the other planes copies)
#pragma unroll
    for (int j = 0; j < 8; ++j) p1[j] = (__bf16)x[j];
    p2 = p1;
    p3 = p1;
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h1 = (__bf16)x[j];
    const float r1 = x[j] - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    p1[j] = h1;
    p2[j] = h2;
    p3[j] = (__bf16)r2;
  }
}

#ifdef CAL_X_NOMFMA  // (ablation: everything but the matrix instructions)
#define CAL_MFMA_BF16(A_, B_, C_) asm volatile("" : "+v"(C_) : "v"(A_), "v"(B_))
#else
#define CAL_MFMA_BF16(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
#endif

typedef float cf2 __attribute__((ext_vector_type(2)));  // a pair of fp32: the element stage computes on the lane's two channels at once (v_pk_*)

#ifdef CAL_STAMP
// diagnostic build only: [item][wave][F, E, B, group waits + barriers, coefficient waits, sample waits, entry time, exit time, HW_ID, XCC_ID] in s_memtime ticks
__device__ long long g_split_stamps[4096][4][14];
#define SPL_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#define SPL_ADD(acc, t0)                                             \
  do {                                                               \
    acc += (long long)__builtin_amdgcn_s_memtime() - (t0);           \
  } while (0)
#else
#define SPL_T(var)
#define SPL_ADD(acc, t0)
#endif

template <bool GRAD>
__device__ __forceinline__ void split_panel(const MfmaArgs& A, unsigned char* smem_raw) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int NTMAX = kSplitNT;
  const int sp = A.slot_map[blockIdx.x];  // never negative here
  SPL_T(t_entry);
#ifdef CAL_STAMP
  long long cyc_f = 0, cyc_e = 0, cyc_b = 0, cyc_sync = 0, cyc_cw = 0, cyc_sw = 0, cyc_eg = 0, cyc_el = 0, cyc_es = 0, cyc_bs = 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const PanelItem& P = A.panels[sp * kSpWaves + wave];  // this wave's panel (padding panels of a super-panel: every slot -1, the block's nvec / operands)
  const int col = lane & 31, half = lane >> 5, slot = col & 15;
  const bool im_lane = (col & 16) != 0;
  const int my_bl = P.bl[slot];
  const int2 my_ant = P.ant[slot];
  const int my_coff = P.coff[slot];
  const int nvec = __builtin_amdgcn_readfirstlane(P.nvec);
  const int NT = __builtin_amdgcn_readfirstlane(P.nvp32 / 32);  // vector tiles of THIS item (<= kSplitNT), the first one being tile0 of the block
  const int tile0 = __builtin_amdgcn_readfirstlane(P.tile0);
  const int ngk = (nvec + 15) / 16;  // forward K-steps of 16 vectors
  const int ngd = (ngk + 1) / 2;     // forward groups
  const int ntd = (NT + 1) / 2;      // adjoint groups per channel block
  const int ncp = A.fpad / 64;       // channel-block pairs
  const DevState* sst = A.state;
  if (A.nslices > 1) sst += P.slice;
  const int stopped = __builtin_amdgcn_readfirstlane(sst->done | sst->done_after);
  if (stopped) return;  // (the four panels of a super-panel belong to one slice: workgroup-uniform)

  // LDS of the item: [4 waves][6 KB] samples | [4 waves][4 KB x ngd] coefficient panels | the operand ring: the rest, nring groups of 12 KB
  const int c_wave_bytes = ngd * 4096;
  unsigned char* s_smp = smem_raw;
  unsigned char* s_cr = smem_raw + kSpWaves * kSmpBytes;
  unsigned char* s_ring = s_cr + kSpWaves * c_wave_bytes;
  int nring = (kSplitLds - kSpWaves * kSmpBytes - kSpWaves * c_wave_bytes) / kGroupBytes;
  nring = nring > kSpRingMax ? kSpRingMax : nring;  // >= 2 for every block the host admits (kSplitMaxNvec)
#ifdef CAL_X_RINGCAP
  nring = nring > CAL_X_RINGCAP ? CAL_X_RINGCAP : nring;
#endif
  // wave-uniform bases of everything the LDS-DMA requests read ("s" operands of the asm)
  const void* ops_u = dma_base(A.ops);
  const void* c_u = dma_base(A.c_r);
  const void* dr_u = dma_base(A.data_r);
  const void* di_u = dma_base(A.data_i);
  const void* w_u = dma_base(A.wgts);
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned ring_lds = (unsigned)reinterpret_cast<unsigned long long>(s_ring);
  const f32x4* ring_rd = reinterpret_cast<const f32x4*>(s_ring) + lane;
  const unsigned cr_lds = (unsigned)reinterpret_cast<unsigned long long>(s_cr + wave * c_wave_bytes);
  const f32x4* cr_rd = reinterpret_cast<const f32x4*>(s_cr + wave * c_wave_bytes) + lane;

  int issued = 0;  // vector-memory requests this wave has issued through asm

  // ---- the operand stream: per channel-block pair GP groups, [ngd forward][ntd adjoint of cb0][ntd adjoint of cb1]; the gradient pass
  // consumes all of them in order, the loss-only pass the forward groups.  Wave w requests position w of a group; past the end the last
  // group again (the number of requests in flight stays what the marks assume).
  const int GP = ngd + 2 * ntd;
  const int g_used = GRAD ? GP : ngd;
  const unsigned base = (unsigned)P.a_kf4 * 4u + (unsigned)wave * (unsigned)kPosBytes;
  int rq_cp = 0, rq_d = 0;  // the group this wave requests next
  int rq_slot = 0;          // ring slot it goes to
  auto request_group = [&]() {
#ifdef CAL_X_A0  // (ablation: every pair re-reads the first pair's groups -- an operand stream that always hits L2)
    const unsigned off = base + (unsigned)(rq_d) * (unsigned)kGroupBytes;
#else
    const unsigned off = base + (unsigned)(rq_cp * GP + rq_d) * (unsigned)kGroupBytes;
#endif
    const unsigned lds = ring_lds + (unsigned)(rq_slot * 4 + wave) * (unsigned)kPosBytes;
#ifndef CAL_X_NOAREQ  // (ablation: no operand requests)
    dma3(lds, ops_u, voff + off);
#endif
    issued += 3;
    if (rq_cp < ncp - 1 || rq_d < g_used - 1) {
      ++rq_d;
      if (rq_d == g_used) { rq_d = 0; ++rq_cp; }
    }
    rq_slot = rq_slot + 1 == nring ? 0 : rq_slot + 1;
  };
  int markA[kSpRingMax - 1];  // issue counts of the group requests in flight (nring - 1 of them), oldest first
  // ---- the coefficient operand of a forward group (two K-steps): 4 KB, request i = 0..3 brings K-step i >> 1, vectors 4 (i & 1) .. + 3 of
  // lane L's eight: c[col L & 31][16 kk + 8 (L >> 5) + 4 (i & 1) ..]
#ifdef CAL_X_C0  // (ablation: every panel reads the same coefficients)
  const unsigned c_voff = (unsigned)((im_lane ? (int)(A.c_i - A.c_r) : 0) + 8 * half) * 4u;
#else
  const unsigned c_voff = (unsigned)((im_lane ? (int)(A.c_i - A.c_r) : 0) + my_coff + 8 * half) * 4u;
#endif
  // (loaded once per item, in front of the sweep: request i = 0..3 of forward group d at d x 4 KB + i KB)
  int markC = 0;
  auto request_c_panel = [&]() {
    for (int d = 0; d < ngd; ++d) {
      const unsigned vo = c_voff + (unsigned)d * 128u;  // 32 vectors x 4 B
      const unsigned lds = cr_lds + (unsigned)d * 4096u;
      dma1(lds, c_u, vo);
      dma1(lds + 1024u, c_u, vo + 16u);
      dma1(lds + 2048u, c_u, vo + 64u);
      dma1(lds + 3072u, c_u, vo + 80u);
      issued += 4;
    }
    markC = issued;
  };
  // ---- samples of one channel block (as dense_kernels.hpp): request k fetches array k / 2, register groups 2 (k & 1) + (lane >> 5)
  const unsigned row = (unsigned)(my_bl >= 0 ? my_bl : A.nbls);
  const unsigned smp_lds = (unsigned)reinterpret_cast<unsigned long long>(s_smp + wave * kSmpBytes);
#ifdef CAL_X_S0  // (ablation: every panel reads the samples of baselines 0..15)
  const unsigned srow = (unsigned)slot;
#else
  const unsigned srow = row;
#endif
  const unsigned smp_voff = (srow * (unsigned)A.fpad + 4u * ((unsigned)(lane >> 4) & 1u)) * 4u + (unsigned)(lane >> 5) * 32u;
  const unsigned char* smp_rd = s_smp + wave * kSmpBytes + (slot + 16 * half) * 16 + (im_lane ? 8 : 0);
  int markS = 0;
  auto smp_issue = [&](int cb) {
    const unsigned o = (unsigned)cb * (kCB * 4u);
    const unsigned vo = smp_voff + o;
    dma1(smp_lds + 0u * 1024u, dr_u, vo);
    dma1(smp_lds + 1u * 1024u, dr_u, vo + 64u);
    dma1(smp_lds + 2u * 1024u, di_u, vo);
    dma1(smp_lds + 3u * 1024u, di_u, vo + 64u);
    dma1(smp_lds + 4u * 1024u, w_u, vo);
    dma1(smp_lds + 5u * 1024u, w_u, vo + 64u);
    issued += 6;
    markS = issued;
  };

  RING_WAIT(0);  // the record loads above are the compiler's: nothing of them is in flight when the counting starts
  smp_issue(0);
#pragma unroll
  for (int j = 0; j < kSpRingMax - 1; ++j) {
    markA[j] = 0;
    if (j < nring - 1) {
      request_group();
      markA[j] = issued;
    }
  }
  request_c_panel();

  // one step of the code = one group: wait for the wave's own request of it, meet the other waves (their requests have landed too, and they
  // have left the previous group), request the group nring - 1 ahead into the slots just freed
  int cur_slot = 0;
  auto group_begin = [&]() -> int {  // -> f32x4 index (without the lane) of the group's first plane
    SPL_T(ts0);
    wait_vm_dyn(issued - markA[0]);
#ifndef CAL_X_NOBAR  // (ablation: the waves do not meet)
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
    SPL_ADD(cyc_sync, ts0);
#pragma unroll
    for (int j = 0; j + 1 < kSpRingMax - 1; ++j) markA[j] = markA[j + 1];
    request_group();
#pragma unroll
    for (int j = 0; j < kSpRingMax - 1; ++j)
      if (j == nring - 2) markA[j] = issued;
    __builtin_amdgcn_sched_barrier(0);
    const int at = cur_slot * (kGroupBytes / 16);
    cur_slot = cur_slot + 1 == nring ? 0 : cur_slot + 1;
    return at;
  };

  f32x16 dC[NTMAX];
#pragma unroll
  for (int t = 0; t < NTMAX; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) dC[t][j] = 0.f;
  double loss_acc = 0.0, sr_acc = 0.0, si_acc = 0.0;
  const cf2 alpha = A.use_alpha ? cf2{(float)sst->alpha_r, (float)sst->alpha_i} : cf2{0.f, 0.f};
  const char* p_g = reinterpret_cast<const char*>(A.gains);
  char* p_q = reinterpret_cast<char*>(A.q0);
  // this lane's two channels of every register group: + 0, 1 on the re lane, + 2, 3 on the im lane (see the element stage)
  const unsigned chan0 = 4u * half + (im_lane ? 2u : 0u);
  const unsigned obq = (row * (unsigned)A.fpad + chan0) * 8u;                 // gbar_G, (re, im) pairs
  const unsigned og0 = ((unsigned)my_ant.x * (unsigned)A.fpad + chan0) * 8u;  // gains
  const unsigned og1 = ((unsigned)my_ant.y * (unsigned)A.fpad + chan0) * 8u;

  // ---- the two antennas' gains of this lane's eight channels of a channel block (they come from L2).  They are requested one GROUP before the
  // element stage that uses them -- in the last forward group for cb0, in the last adjoint group of cb0 for cb1 -- as ordinary loads the compiler
  // tracks: its wait in front of their first use counts only its own loads, i.e. it also waits for every LDS-DMA request issued after them;
  // a group later those have landed anyway.  (Requested inside the element stage the round trip was exposed: 2 500 of its 5 400 ticks.)
  struct GainRegs { f32x4 a[4], b[4]; };
  auto gains_request = [&](GainRegs& R, int cb) {
#ifdef CAL_X_NOGAIN  // (ablation: the same gains for every block)
    const unsigned cb8 = 0;
    if (cb > 1) return;
#else
    const unsigned cb8 = (unsigned)cb * (kCB * 8u);
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      R.a[g] = *reinterpret_cast<const f32x4*>(p_g + (og0 + cb8 + 64u * g));
      R.b[g] = *reinterpret_cast<const f32x4*>(p_g + (og1 + cb8 + 64u * g));
    }
  };

  // ---- E: element stage of one channel block on its forward accumulator (calibration.py:1593-1609 and their adjoints).  Lane (col, half) holds
  // v of column col = (slot, re | im) at 16 channels: register 4 g + r -> channel 8 g + 4 half + r.  The re and im lane of a slot sit 16 lanes
  // apart and share the work: v_permlane16_swap(v[4 g + i], v[4 g + i + 2]) hands the re lane (re, im) of channel i and the im lane (re, im)
  // of channel i + 2; after the arithmetic ONE more swap of (gbar_v.re, gbar_v.im) puts gbar_v into the layout v had (the re lane keeps the
  // real part of its channel and receives the real part of the im lane's channel; the im lane the imaginary parts), no selects.
  // In three parts, so that the stage of a pair's SECOND block can run between the MFMAs of the first block's adjoint (a wave alone on its
  // SIMD has nobody else to fill the issue slots a chain of dependent MFMAs leaves free):
  //   e_begin  waits for the block's samples, reads them into registers, requests the next block's into the staging area
  //   e_chunk  one register group: arithmetic only (and one gbar_G store) -- nothing the compiler may not move between MFMAs
  //   e_end    the loss partials
  // G = g_i conj(g_j) (calibration.py:1598-1601) of the lane's eight channels, as pairs per register group.  Formed from BOTH blocks' gains at the
  // start of the first block's stage, in front of its sample request: the compiler waits for a load it tracks with s_waitcnt vmcnt(its own
  // younger loads) -- a count that knows nothing of the LDS-DMA requests, so a gain consumed AFTER a request went out waits for that request
  // too (the samples come from HBM: two exposed round trips per pair of channel blocks, 40 % of the wave's time, before this was moved).
  struct GProd { cf2 r[4], i[4]; };
  auto gains_product = [&](const GainRegs& GR, GProd& G) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const cf2 g0r = {GR.a[g][0], GR.a[g][2]}, g0i = {GR.a[g][1], GR.a[g][3]}, g1r = {GR.b[g][0], GR.b[g][2]}, g1i = {GR.b[g][1], GR.b[g][3]};
      G.r[g] = g0r * g1r + g0i * g1i;
      G.i[g] = g0i * g1r - g0r * g1i;
    }
  };
  struct EState { cf2 s_dr[4], s_di[4], s_w[4], lt, st_r, st_i; };
  auto e_begin = [&](EState& S, int next_cb) {
    S.lt = cf2{0.f, 0.f}; S.st_r = cf2{0.f, 0.f}; S.st_i = cf2{0.f, 0.f};
    SPL_T(tw0);
    wait_vm_dyn(issued - markS);
    SPL_ADD(cyc_sw, tw0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int so = ((g >> 1) * 64 + 32 * (g & 1)) * 16;
      S.s_dr[g] = *reinterpret_cast<const cf2*>(smp_rd + so);
      S.s_di[g] = *reinterpret_cast<const cf2*>(smp_rd + 2048 + so);
      S.s_w[g] = *reinterpret_cast<const cf2*>(smp_rd + 4096 + so);
    }
    if (next_cb >= 0) {
      // the staging area is free once the reads above have returned: the next block's samples go into it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      smp_issue(next_cb);
      __builtin_amdgcn_sched_barrier(0);
    }
    SPL_ADD(cyc_eg, tw0);
  };
  // Per register group the lane's two channels a, b are computed side by side: every quantity is a pair (x_a, x_b) in two adjacent registers
  // and every operation one packed instruction, no shuffles: after the two swaps (v[4 g], v[4 g + 1]) = (v_re a, v_re b) and
  // (v[4 g + 2], v[4 g + 3]) = (v_im a, v_im b) ARE such pairs, the samples arrive as (d a, d b), and gbar_v leaves the same way.
  auto e_chunk = [&](EState& S, const f32x16& v, f32x16& gout, const GProd& GP_, int cb, int g) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
#ifdef CAL_X_NOE  // (ablation: no element arithmetic, no gbar_G stores)
    gout[4 * g] = v[4 * g] + S.s_dr[g].x; gout[4 * g + 1] = v[4 * g + 1] + GP_.r[g].x; gout[4 * g + 2] = v[4 * g + 2]; gout[4 * g + 3] = v[4 * g + 3];
    return;
#endif
    const unsigned cb8 = (unsigned)cb * (kCB * 8u);
    cf2 vr, vi;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // (bit_cast applied to a vector ELEMENT expression reads element 0 with this hipcc: go through scalars)
      const float xa = v[4 * g + i], xb = v[4 * g + i + 2];
      const u2 pr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, xa), __builtin_bit_cast(unsigned, xb), false, false);
      const unsigned p0 = pr[0], p1 = pr[1];
      vr[i] = __builtin_bit_cast(float, p0);
      vi[i] = __builtin_bit_cast(float, p1);
    }
    const cf2 Gr = GP_.r[g], Gi = GP_.i[g];
    const cf2 mr = Gr * vr - Gi * vi, mi = Gi * vr + Gr * vi;  // model = G v  (:1602-1604)
    const cf2 rr = S.s_dr[g] - mr, ri = S.s_di[g] - mi;
    const cf2 w = S.s_w[g];
    S.lt += (rr * rr + ri * ri) * w;  // (:1609)
    S.st_r += mr * w;                 // S = sum w m of the "sum" regulariser (calibration.py:1648-1649)
    S.st_i += mi * w;
    if (GRAD) {
      const cf2 er = (rr * -2.f + alpha.x) * w, ei = (ri * -2.f + alpha.y) * w;  // e = -2 w r + alpha w
      const cf2 gvr = Gr * er + Gi * ei, gvi = Gr * ei - Gi * er;                // gbar_v = conj(G) e
      const cf2 gqr = vr * er + vi * ei, gqi = vr * ei - vi * er;                // gbar_G = conj(v) e
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float fx = gvr[i], fy = gvi[i];
        const u2 qr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, fx), __builtin_bit_cast(unsigned, fy), false, false);
        const unsigned q0b = qr[0], q1b = qr[1];
        gout[4 * g + i] = __builtin_bit_cast(float, q0b);
        gout[4 * g + i + 2] = __builtin_bit_cast(float, q1b);
      }
#if defined(CAL_X_QSMALL)  // (ablation: the stores go to 256 KB that stay in L2)
      *reinterpret_cast<f32x4*>(p_q + ((obq + cb8 + 64u * g) & 0x3FFF0u)) = f32x4{gqr[0], gqi[0], gqr[1], gqi[1]};
#elif !defined(CAL_X_NOQ)
      *reinterpret_cast<f32x4*>(p_q + (obq + cb8 + 64u * g)) = f32x4{gqr[0], gqi[0], gqr[1], gqi[1]};
#endif
    }
  };
  auto e_end = [&](const EState& S) {
    loss_acc += (double)(S.lt.x + S.lt.y);
    sr_acc += (double)(S.st_r.x + S.st_r.y);
    si_acc += (double)(S.st_i.x + S.st_i.y);
  };
  // gbar_v (16 fp32 per lane: two K-steps of 16 channels) -> three bf16 planes: the B operand of the adjoint
  struct Planes { bf16x8 p1[2], p2[2], p3[2]; };
  auto split_gbar = [&](const f32x16& gv, Planes& P) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = gv[8 * q + j];
#ifdef CAL_X_NOGSPLIT
      split3<true>(x, P.p1[q], P.p2[q], P.p3[q]);
#else
      split3(x, P.p1[q], P.p2[q], P.p3[q]);
#endif
    }
  };
  // ---- B: one adjoint group: dC[2 e], dC[2 e + 1] += A[ch, tile]^T gbar_v over two K-steps of 16 channels, the two tiles' chains interleaved
  // (a dependent MFMA waits for its predecessor).  `between(k)`, k = 0..3: called between the four batches of MFMAs -- vector work placed there
  // issues in the slots the MFMA chains leave free
  auto adjoint_group = [&](const Planes& P, int e, auto two_tag, auto&& between) {
    constexpr bool two = decltype(two_tag)::value;
    const int at = group_begin();
    f32x16& D0 = dC[2 * e];
    f32x16& D1 = dC[two ? 2 * e + 1 : 2 * e];
    // every operand of the group is requested from LDS before the first MFMA: one read latency per group, not one per batch (the wave is
    // alone on its SIMD: nobody else covers an LDS round trip)
    bf16x8 a[2][3], b[2][3];
#ifdef CAL_X_NOLDS  // (ablation: no operand reads)
    const int at_ = 0;
    (void)at;
#else
    const int at_ = at;
#endif
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#ifdef CAL_X_NOLDS
        a[q][pl] = __builtin_bit_cast(bf16x8, f32x4{(float)at_, 1.f, 2.f, (float)pl});
        if (two) b[q][pl] = a[q][pl];
#else
        a[q][pl] = __builtin_bit_cast(bf16x8, ring_rd[at_ + (q * 3 + pl) * 64]);
        if (two) b[q][pl] = __builtin_bit_cast(bf16x8, ring_rd[at_ + ((2 + q) * 3 + pl) * 64]);
#endif
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (two) {
        CAL_MFMA_BF16(a[q][2], P.p1[q], D0);
        CAL_MFMA_BF16(b[q][2], P.p1[q], D1);
        CAL_MFMA_BF16(a[q][1], P.p2[q], D0);
        CAL_MFMA_BF16(b[q][1], P.p2[q], D1);
        CAL_MFMA_BF16(a[q][0], P.p3[q], D0);
        CAL_MFMA_BF16(b[q][0], P.p3[q], D1);
        between(2 * q);
        CAL_MFMA_BF16(a[q][1], P.p1[q], D0);
        CAL_MFMA_BF16(b[q][1], P.p1[q], D1);
        CAL_MFMA_BF16(a[q][0], P.p2[q], D0);
        CAL_MFMA_BF16(b[q][0], P.p2[q], D1);
        CAL_MFMA_BF16(a[q][0], P.p1[q], D0);
        CAL_MFMA_BF16(b[q][0], P.p1[q], D1);
        between(2 * q + 1);
      } else {
        CAL_MFMA_BF16(a[q][2], P.p1[q], D0);
        CAL_MFMA_BF16(a[q][1], P.p2[q], D0);
        CAL_MFMA_BF16(a[q][0], P.p3[q], D0);
        between(2 * q);
        CAL_MFMA_BF16(a[q][1], P.p1[q], D0);
        CAL_MFMA_BF16(a[q][0], P.p2[q], D0);
        CAL_MFMA_BF16(a[q][0], P.p1[q], D0);
        between(2 * q + 1);
      }
    }
  };
  // the adjoint of one channel block; `first()` runs in front of the MFMAs of the block's LAST group, `between(k)` between its batches
  auto adjoint = [&](const Planes& P, auto&& first, auto&& between) {
    SPL_T(tb0);
    auto none = [](int) {};
#pragma unroll
    for (int e = 0; e < NTMAX / 2; ++e) {
      if (e < ntd) {  // wave-uniform
        const bool last = e == ntd - 1;
        const bool two = 2 * e + 1 < NT;  // wave-uniform: the group's second tile exists
        if (!last) {  // (every group but a block's last carries two tiles)
          adjoint_group(P, e, std::true_type{}, none);
        } else {
          first();
          if (two) adjoint_group(P, e, std::true_type{}, between);
          else adjoint_group(P, e, std::false_type{}, between);
        }
      }
    }
    SPL_ADD(cyc_bs, tb0);
  };
  // ---- F: one forward group: two K-steps x (cb0, cb1)
  auto forward_group = [&](f32x16& acc0, f32x16& acc1, int d, auto&& pre) {
    const int at = group_begin();
    // the two K-steps' coefficients (from the wave's resident panel) and every operand of the group, requested from LDS before the first MFMA
    f32x4 cq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cq[j] = cr_rd[d * 256 + j * 64];
    bf16x8 a0[2][3], a1[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#ifdef CAL_X_NOLDS
        a0[ks][p] = __builtin_bit_cast(bf16x8, f32x4{(float)at, 1.f, 2.f, (float)p});
        a1[ks][p] = a0[ks][p];
#else
        a0[ks][p] = __builtin_bit_cast(bf16x8, ring_rd[at + ((2 * ks) * 3 + p) * 64]);
        a1[ks][p] = __builtin_bit_cast(bf16x8, ring_rd[at + ((2 * ks + 1) * 3 + p) * 64]);
#endif
      }
    pre();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0 || 2 * d + 1 < ngk) {  // wave-uniform: the second K-step of the last group may lie past the block's vectors
        bf16x8 c1, c2, c3;
        {
          const float x[8] = {cq[2 * ks][0], cq[2 * ks][1], cq[2 * ks][2], cq[2 * ks][3], cq[2 * ks + 1][0], cq[2 * ks + 1][1], cq[2 * ks + 1][2], cq[2 * ks + 1][3]};
#ifdef CAL_X_NOCSPLIT
          split3<true>(x, c1, c2, c3);
#else
          split3(x, c1, c2, c3);
#endif
        }
        CAL_MFMA_BF16(a0[ks][2], c1, acc0);
        CAL_MFMA_BF16(a1[ks][2], c1, acc1);
        CAL_MFMA_BF16(a0[ks][1], c2, acc0);
        CAL_MFMA_BF16(a1[ks][1], c2, acc1);
        CAL_MFMA_BF16(a0[ks][0], c3, acc0);
        CAL_MFMA_BF16(a1[ks][0], c3, acc1);
        CAL_MFMA_BF16(a0[ks][1], c1, acc0);
        CAL_MFMA_BF16(a1[ks][1], c1, acc1);
        CAL_MFMA_BF16(a0[ks][0], c2, acc0);
        CAL_MFMA_BF16(a1[ks][0], c2, acc1);
        CAL_MFMA_BF16(a0[ks][0], c1, acc0);
        CAL_MFMA_BF16(a1[ks][0], c1, acc1);
      }
    }
  };

  {
    SPL_T(tc0);
    wait_vm_dyn(issued - markC);  // the coefficient panel has landed
    SPL_ADD(cyc_cw, tc0);
  }
  for (int cp = 0; cp < ncp; ++cp) {
    SPL_T(t_f0);
    f32x16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
    GainRegs GR0, GR1;
    for (int d = 0; d + 1 < ngd; ++d) forward_group(acc0, acc1, d, [] {});
    forward_group(acc0, acc1, ngd - 1, [&] {  // (the gains of both blocks: 64 of the wave's 512 registers)
      gains_request(GR0, 2 * cp);
      gains_request(GR1, 2 * cp + 1);
    });
    SPL_T(t1);
    SPL_ADD(cyc_f, t_f0);
    const int next_pair_cb = cp + 1 < ncp ? 2 * cp + 2 : -1;
    f32x16 gv0, gv1;
    EState S0, S1;
    GProd G0, G1;
    gains_product(GR0, G0);
    gains_product(GR1, G1);
    // (pin the products here: everything the compiler waits for must have been waited for before the next request is issued)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(G0.r[g]), "+v"(G0.i[g]), "+v"(G1.r[g]), "+v"(G1.i[g]));
    __builtin_amdgcn_sched_barrier(0);
    e_begin(S0, 2 * cp + 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) e_chunk(S0, acc0, gv0, G0, 2 * cp, g);
    e_end(S0);
    SPL_T(t2);
    SPL_ADD(cyc_e, t1);
    if (GRAD) {
      Planes P0, P1;
      split_gbar(gv0, P0);
      // the adjoint of cb0, with the element stage of cb1 between the MFMAs of its last group
      adjoint(P0, [&] { e_begin(S1, next_pair_cb); }, [&](int k) { e_chunk(S1, acc1, gv1, G1, 2 * cp + 1, k); });
      e_end(S1);
      SPL_T(t3);
      SPL_ADD(cyc_b, t2);
      split_gbar(gv1, P1);
      adjoint(P1, [] {}, [](int) {});
      SPL_ADD(cyc_b, t3);
    } else {
      e_begin(S1, next_pair_cb);
#pragma unroll
      for (int g = 0; g < 4; ++g) e_chunk(S1, acc1, gv1, G1, 2 * cp + 1, g);
      e_end(S1);
    }
  }
  RING_WAIT(0);  // nothing may still be writing into this workgroup's LDS when it ends

  // ---- epilogue: the panel's loss partials (double), then its coefficient gradients -- the wave owns them completely
  {
    const double l = ldsum(loss_acc), sr = ldsum(sr_acc), si = ldsum(si_acc);
    if (lane == 0) {
      const size_t pi = (size_t)(sp * kSpWaves + wave) * 4;
      A.part[pi + 0] = l;
      A.part[pi + 1] = sr;
      A.part[pi + 2] = si;
    }
  }
#ifdef CAL_STAMP
  if (lane == 0 && sp < 4096) {
    long long* o = g_split_stamps[sp][wave];
    o[0] = cyc_f; o[1] = cyc_e; o[2] = cyc_b; o[3] = cyc_sync; o[4] = cyc_cw; o[5] = cyc_sw; o[6] = t_entry;
    o[7] = (long long)__builtin_amdgcn_s_memtime();
    o[8] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    o[9] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    o[10] = cyc_eg; o[11] = cyc_el; o[12] = cyc_es; o[13] = cyc_bs;  // element stage: until the gains are there, the loop, the tail; adjoint: the split
  }
#endif
  if (!GRAD) return;
  if (my_bl >= 0) {
    float* gc = (im_lane ? A.gc_i : A.gc_r) + my_coff + 32 * tile0;
    const int nleft = nvec - 32 * tile0;
#pragma unroll
    for (int t = 0; t < NTMAX; ++t) {
      if (t < NT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n0 = 32 * t + 8 * i + 4 * half;  // registers 4 i .. 4 i + 3 = vectors n0 .. n0 + 3 of the item
          if (n0 + 3 < nleft) {
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // (a group's coefficients start at any multiple of 4 bytes)
            *reinterpret_cast<f32x4u*>(gc + n0) = f32x4u{dC[t][4 * i], dC[t][4 * i + 1], dC[t][4 * i + 2], dC[t][4 * i + 3]};
          } else {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
              if (n0 + jj < nleft) gc[n0 + jj] = dC[t][4 * i + jj];
          }
        }
      }
    }
  }
}

// One launch for all items; XCD-affine dispatch through slot_map as fused_dense_kernel.
template <bool GRAD>
__global__ __launch_bounds__(kDenseThreads, 1) void fused_dense_split_kernel(const MfmaArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int sp = A.slot_map[blockIdx.x];
  if (sp < 0) return;
  split_panel<GRAD>(A, smem_raw);
}

// packed split-bf16 operands of one item class of a basis block (tiles tile0 .. tile0 + ntiles - 1 of its vectors): groups of 4 positions of
// 3 KB = [3 planes][64 lanes][8 bf16]; per channel-block pair cp, with lane = (row, kg) and j = 0..7
//   forward group d, position 2 ks + s      -> A[32 (2 cp + s) + row][16 (2 d + ks) + 8 kg + j]
//   adjoint group e of channel block s (after the ngd forward groups: ngd + s ntd + e), position 2 tt + q
//                                           -> A[32 (2 cp + s) + 16 q + (j & 3) + 8 (j >> 2) + 4 kg][32 (tile0 + 2 e + tt) + row]
// zero outside the block (channels >= nfreqs, vectors >= nvec, tiles >= tile0 + ntiles)
__global__ void split_pack_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int nfreqs, int fpad, int nvec, int tile0, int ntiles) {
  const int ngk = (nvec + 15) / 16, ngd = (ngk + 1) / 2, ntd = (ntiles + 1) / 2;
  const int GP = ngd + 2 * ntd;
  const long long total = (long long)(fpad / 64) * GP * 4 * 512;  // (position, lane, j)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
    const long long pos = i >> 9;
    const int p = (int)(pos & 3);
    const long long grp = pos >> 2;
    const int cp = (int)(grp / GP), gi = (int)(grp % GP);
    const int row = l & 31, kg = l >> 5;
    int f, k;
    if (gi < ngd) {
      f = 32 * (2 * cp + (p & 1)) + row;
      k = 16 * (2 * gi + (p >> 1)) + 8 * kg + j;
    } else {
      const int s = (gi - ngd) / ntd, e = (gi - ngd) % ntd;
      const int tt = p >> 1, q = p & 1;
      f = 32 * (2 * cp + s) + 16 * q + (j & 3) + 8 * (j >> 2) + 4 * kg;
      k = 2 * e + tt < ntiles ? 32 * (tile0 + 2 * e + tt) + row : nvec;
    }
    float x = 0.f;
    if (f < nfreqs && k < nvec) x = src[(long long)f * nvec + k];
    const __bf16 h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const __bf16 h3 = (__bf16)(r1 - (float)h2);
    unsigned short* o = dst + pos * 1536 + l * 8 + j;
    o[0] = __builtin_bit_cast(unsigned short, h1);
    o[512] = __builtin_bit_cast(unsigned short, h2);
    o[1024] = __builtin_bit_cast(unsigned short, h3);
  }
}

}  // namespace calk
