// split2_kernels.hpp -- the split-bf16 dense step of split_kernels.hpp with ONE operand image instead of two packed streams (DESIGN 3.1e,
// "what a successor would change"; the loop was measured in isolation first: tools/micro/tr_image.hip).  Same arithmetic, same partition
// (a workgroup = a super-panel of four 16-baseline panels, wave w owns panel w), same element stage, sample staging and epilogue -- those
// parts of the function are the text of split_kernels.hpp.  What differs:
//   * the basis operand of a UNIT of channels is one image [3 bf16 planes][RC channels][NVP vectors] of 48 KB (RC = 64, NVP = 128 for blocks
//     of at most 128 vectors; RC = 32, NVP = 256 above), rows of 256 bytes per 128-vector tile, 16-byte chunks XOR-swizzled
//     (cdna_hip_programming.md T10 (b)); the packed global copy IS the LDS byte image.  Two buffers; per unit ONE counted wait and ONE barrier
//     (then the next unit is requested into the buffer everybody has just left);
//   * F reads rows of it (ds_read_b128: 8 consecutive vectors of a channel), B reads it TRANSPOSED (ds_read_tr16_b64: lane 4 q + p of a
//     16-lane group supplies row R + q, columns C + 4 p ..; lane i receives vector C + i at channels R .. R + 3) -- the K order of two such
//     reads (channels 16 s + 4 half + 0..3, then + 8) is the order of the accumulator registers 8 s .. 8 s + 7, i.e. of gbar_v as the element
//     stage leaves it: no adjoint stream, no requests or barriers of its own;
//   * the coefficient operand of the wave's panel lives in REGISTERS (fp32, 8 per 16-vector step), split into planes on the fly: the body is
//     instantiated per number of 32-vector tiles (NTC = 1 .. 7), and the LDS holds only images (96 KB) and samples (24 KB).
#pragma once
#include "split_kernels.hpp"

namespace calk {

constexpr int kS2UnitBytes = 48 * 1024;   // one unit's image: 3 planes x RC x NVP x 2 B
constexpr int kS2PlaneBytes = 16 * 1024;
constexpr int kS2Lds = kSpWaves * kSmpBytes + 2 * kS2UnitBytes;
__host__ __device__ inline int split2_rc(int nvec) { return nvec <= 128 ? 64 : 32; }
inline long long split2_stream_bytes(int fpad, int nvec) { return (long long)(fpad / split2_rc(nvec)) * kS2UnitBytes; }
// byte offset of 16-byte chunk `chunk` (0..15) of row `row` inside a [RC][128 x bf16] tile
__host__ __device__ inline unsigned s2_off(int row, int chunk) { return 256u * (unsigned)row + 16u * (unsigned)(chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

typedef short s2_s16x4 __attribute__((ext_vector_type(4)));
typedef short s2_s16x8 __attribute__((ext_vector_type(8)));

template <bool GRAD, int NTC>
__device__ __forceinline__ void split2_panel(const MfmaArgs& A, unsigned char* smem_raw) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int NTMAX = NTC;
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
  const DevState* sst = A.state;
  if (A.nslices > 1) sst += P.slice;
  const int stopped = __builtin_amdgcn_readfirstlane(sst->done | sst->done_after);
  if (stopped) return;  // (the four panels of a super-panel belong to one slice: workgroup-uniform)


  constexpr int RC = NTC <= 4 ? 64 : 32, NCB = RC / 32, NSTEP = 2 * NTC;
  const int nunits = A.fpad / RC;
  // LDS of the item: [4 waves][6 KB] samples | two image buffers of 48 KB
  unsigned char* s_smp = smem_raw;
  unsigned char* s_img = smem_raw + kSpWaves * kSmpBytes;
  const void* ops_u = dma_base(A.ops);
  const void* dr_u = dma_base(A.data_r);
  const void* di_u = dma_base(A.data_i);
  const void* w_u = dma_base(A.wgts);
  const unsigned voff = (unsigned)lane * 16u;
  int issued = 0;  // vector-memory requests this wave has issued through asm

  // ---- the coefficient operand of the panel: lane (col, half), K-slot j of step s <-> vector 16 s + 8 half + j (zero past nvec)
  float creg[NSTEP][8];
  {
    const float* cp_ = (im_lane ? A.c_i : A.c_r) + my_coff;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * half + j;
        creg[s][j] = k < nvec ? cp_[k] : 0.f;
      }
  }
  // ---- the image of a unit: this wave requests its quarter (12 KB = four times three kilobytes)
  const unsigned img_lds = (unsigned)reinterpret_cast<unsigned long long>(s_img);
  const unsigned ibase = (unsigned)P.a_kf4 * 4u + (unsigned)wave * 12288u;
  int markI = 0;
  auto request_image = [&](int u, int buf) {
    const unsigned off = ibase + (unsigned)u * (unsigned)kS2UnitBytes;
    const unsigned lds = img_lds + (unsigned)buf * (unsigned)kS2UnitBytes + (unsigned)wave * 12288u;
#pragma unroll
    for (int r = 0; r < 4; ++r) dma3(lds + r * 3072u, ops_u, voff + off + r * 3072u);
    issued += 12;
    markI = issued;
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

  // lane parts of the image addresses.  Row read (F): row cb * 32 + col, chunk (2 s + half) & 15 of tile (16 s) >> 7.
  // Transposed read (B): lane 4 q + p of a 16-lane group supplies row R + q, columns C + 4 p .. + 3.
  const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  auto row_read = [&](const unsigned char* ub, int s, int cb, int pl) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(ub + pl * kS2PlaneBytes + ((16 * s) >> 7) * (RC * 256) + s2_off(cb * 32 + col, (2 * s + half) & 15)));
  };
  auto tr_read = [&](const unsigned char* ub, int cb, int cs, int t, int pl) -> bf16x8 {
    const int R = cb * 32 + 16 * cs + 4 * half + q4;
    const int ch = ((4 * t + 2 * g16) & 15) + (p4 >> 1);
    const unsigned char* b = ub + pl * kS2PlaneBytes + ((32 * t) >> 7) * (RC * 256) + 8 * (p4 & 1);
    typedef __attribute__((address_space(3))) s2_s16x4* lds_p;
    const s2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + s2_off(R, ch)));
    const s2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + s2_off(R + 8, ch)));
    return __builtin_bit_cast(bf16x8, s2_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };

  RING_WAIT(0);  // the record and coefficient loads above are the compiler's: nothing of them is in flight when the counting starts
  smp_issue(0);
  request_image(0, 0);
  GainRegs GR[NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c) gains_request(GR[c], c);
  const int ncb_all = A.fpad / kCB;
  for (int u = 0; u < nunits; ++u) {
    const int buf = u & 1;
    wait_vm_dyn(issued - markI);  // this wave's quarter of unit u has landed
    __builtin_amdgcn_s_barrier();  // ... everybody's has, and everybody has left unit u - 1
    __builtin_amdgcn_sched_barrier(0);
    // the gain products of this unit's blocks, in front of the next requests (a tracked load consumed after an LDS-DMA request went out
    // waits for that request too: split_kernels.hpp)
    GProd G[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      gains_product(GR[c], G[c]);
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(G[c].r[g]), "+v"(G[c].i[g]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (u + 1 < nunits) request_image(u + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* ub = s_img + buf * kS2UnitBytes;
    // ---- F
    f32x16 acc[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      bf16x8 c1, c2, c3;
      split3(creg[s], c1, c2, c3);
      bf16x8 a[NCB][3];
#pragma unroll
      for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[c][pl] = row_read(ub, s, c, pl);
#pragma unroll
      for (int c = 0; c < NCB; ++c) CAL_MFMA_BF16(a[c][2], c1, acc[c]);
#pragma unroll
      for (int c = 0; c < NCB; ++c) CAL_MFMA_BF16(a[c][1], c2, acc[c]);
#pragma unroll
      for (int c = 0; c < NCB; ++c) CAL_MFMA_BF16(a[c][0], c3, acc[c]);
#pragma unroll
      for (int c = 0; c < NCB; ++c) CAL_MFMA_BF16(a[c][1], c1, acc[c]);
#pragma unroll
      for (int c = 0; c < NCB; ++c) CAL_MFMA_BF16(a[c][0], c2, acc[c]);
#pragma unroll
      for (int c = 0; c < NCB; ++c) CAL_MFMA_BF16(a[c][0], c1, acc[c]);
    }
    // the next unit's gains (ordinary loads, consumed at the next unit's start)
    if (u + 1 < nunits) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) gains_request(GR[c], (u + 1) * NCB + c);
    }
    // ---- per channel block: E, then B on all gradient tiles
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int cb = u * NCB + c;
      EState S;
      f32x16 gv;
      e_begin(S, cb + 1 < ncb_all ? cb + 1 : -1);
#pragma unroll
      for (int g = 0; g < 4; ++g) e_chunk(S, acc[c], gv, G[c], cb, g);
      e_end(S);
      if (GRAD) {
        Planes Pl;
        split_gbar(gv, Pl);
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
#pragma unroll
          for (int t = 0; t < NTC; t += 2) {
            constexpr int kLast = NTC - 1;
            const int t1 = t + 1 <= kLast ? t + 1 : t;  // (an odd tile count: the last pair is a single tile)
            const bool two = t + 1 <= kLast;
            bf16x8 x[3], y[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              x[pl] = tr_read(ub, c, cs, t, pl);
              if (two) y[pl] = tr_read(ub, c, cs, t1, pl);
            }
            CAL_MFMA_BF16(x[2], Pl.p1[cs], dC[t]);
            if (two) CAL_MFMA_BF16(y[2], Pl.p1[cs], dC[t1]);
            CAL_MFMA_BF16(x[1], Pl.p2[cs], dC[t]);
            if (two) CAL_MFMA_BF16(y[1], Pl.p2[cs], dC[t1]);
            CAL_MFMA_BF16(x[0], Pl.p3[cs], dC[t]);
            if (two) CAL_MFMA_BF16(y[0], Pl.p3[cs], dC[t1]);
            CAL_MFMA_BF16(x[1], Pl.p1[cs], dC[t]);
            if (two) CAL_MFMA_BF16(y[1], Pl.p1[cs], dC[t1]);
            CAL_MFMA_BF16(x[0], Pl.p2[cs], dC[t]);
            if (two) CAL_MFMA_BF16(y[0], Pl.p2[cs], dC[t1]);
            CAL_MFMA_BF16(x[0], Pl.p1[cs], dC[t]);
            if (two) CAL_MFMA_BF16(y[0], Pl.p1[cs], dC[t1]);
          }
        }
      }
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

// One launch for all items; XCD-affine dispatch through slot_map as fused_dense_kernel; the body per number of vector tiles of the block.
template <bool GRAD>
__global__ __launch_bounds__(kDenseThreads, 1) void fused_dense_split2_kernel(const MfmaArgs A) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
  const int sp = A.slot_map[blockIdx.x];
  if (sp < 0) return;
  switch (A.panels[sp * kSpWaves].nvp32 >> 5) {  // workgroup-uniform
    case 1: split2_panel<GRAD, 1>(A, smem_raw); break;
    case 2: split2_panel<GRAD, 2>(A, smem_raw); break;
    case 3: split2_panel<GRAD, 3>(A, smem_raw); break;
    case 4: split2_panel<GRAD, 4>(A, smem_raw); break;
    case 5: split2_panel<GRAD, 5>(A, smem_raw); break;
    case 6: split2_panel<GRAD, 6>(A, smem_raw); break;
    case 7: split2_panel<GRAD, 7>(A, smem_raw); break;
    default: break;  // (the host admits blocks of at most kSplitMaxNvec vectors)
  }
}

// the images of one basis block: unit n = channels n RC .. n RC + RC - 1, [3 planes][RC][NVP] in the byte layout the kernel reads; zero outside the block
__global__ void split2_pack_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int nfreqs, int fpad, int nvec) {
  const int rc = split2_rc(nvec), nvp = nvec <= 128 ? 128 : 256;
  const long long total = (long long)fpad * nvp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i / nvp), k = (int)(i % nvp);
    float x = 0.f;
    if (f < nfreqs && k < nvec) x = src[(long long)f * nvec + k];
    const __bf16 h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const __bf16 h3 = (__bf16)(r1 - (float)h2);
    unsigned char* o = dst + (long long)(f / rc) * kS2UnitBytes + (k >> 7) * (rc * 256) + s2_off(f % rc, (k & 127) >> 3) + 2 * (k & 7);
    *reinterpret_cast<unsigned short*>(o) = __builtin_bit_cast(unsigned short, h1);
    *reinterpret_cast<unsigned short*>(o + kS2PlaneBytes) = __builtin_bit_cast(unsigned short, h2);
    *reinterpret_cast<unsigned short*>(o + 2 * kS2PlaneBytes) = __builtin_bit_cast(unsigned short, h3);
  }
}

}  // namespace calk
