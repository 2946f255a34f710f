// dense64_kernels.hpp -- the dense (matrix-core) formulation of dense_kernels.hpp in double precision, on
// v_mfma_f64_16x16x4_f64 (--precision 64 of the reference, /root/reference/calamity/calibration.py:1857, :1795).
//
// Same structure as the fp32 kernel -- one 4-wave workgroup per panel, two workgroups per CU, each wave owns channel blocks
// and runs forward MFMA, element-wise stage in the accumulator registers and the adjoint MFMA that takes those registers as
// its B operand, coefficient gradients in registers for the whole panel, operand stream through a per-wave LDS-DMA ring --
// with the tile shapes of the f64 instruction:
//   * one MFMA = 16 x 16 x 4; a channel block = 16 channels; a column tile = 16 columns = 8 baselines x (re | im);
//   * accumulator of lane l: column j = l & 15, rows (l >> 4) + 4 r, r = 0..3: as the B operand of the adjoint MFMA register
//     r carries k = l >> 4 <-> channel (l >> 4) + 4 r, which is how the packed adjoint operand is ordered;
//   * re and im of a slot sit 8 lanes apart inside a row of 16 lanes and are exchanged with a DPP row rotation by 8; of the
//     four channels a lane holds per block the re lane takes the first two, the im lane the last two.
// One kilobyte of packed basis feeds only TWO f64 MFMAs per column tile (eight bytes per operand), so a panel of NC = 2
// column tiles (16 baselines) is used wherever the registers allow it (at most 128 vectors: 2 x 8 gradient tiles); wider
// blocks run with NC = 1.
#pragma once
#include "dense_kernels.hpp"

namespace calk {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <> struct DenseCfg<double> { static constexpr int max_nvec = 256; };
constexpr int kCB64 = 16;    // channels per block
constexpr int kVT64 = 16;    // vectors per gradient tile

struct Dense64Args {
  const double* ops;           // packed operands of every basis block
  const PanelItem* panels;     // bl[0 .. 8 NC)
  const double* data_r;        // [nbls + 1][fpad]
  const double* data_i;
  const double* wgts;
  const double2* gains;        // [nants][fpad]
  const double* c_r;
  const double* c_i;
  double2* q0;                 // [nbls + 1][fpad]
  double* gc_r;
  double* gc_i;
  double* part;                // [npanels][4]
  const DevState* state;       // [nslices]
  int nslices;
  int fpad;
  int use_alpha;
  int nbls;
  const int* slot_map;         // [grid] workgroup -> panel, -1 for an empty slot (XCD-affine dispatch: dense_kernels.hpp)
};

__device__ __forceinline__ double swap8(double x) {  // the value of lane l ^ 8 (rotation by 8 inside each row of 16 lanes)
  const long long b = __builtin_bit_cast(long long, x);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

template <bool GRAD, int NC, int NTMAX>
__device__ __forceinline__ void dense64_panel(const Dense64Args& A, unsigned char* smem_raw) {
  constexpr int kRing = GRAD ? 8 : 4;  // powers of two (slot index by mask)
  const int panel_idx = A.slot_map[blockIdx.x];  // never negative here: the kernel returns for empty slots
  const PanelItem& P = A.panels[panel_idx];
  const DevState* sst = A.state;
  if (A.nslices > 1) sst += P.slice;
  const int stopped = sst->done | sst->done_after;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15;       // MFMA column of a tile: (slot, re | im)
  const int kq = lane >> 4;        // the k of a 16x16x4 step this lane feeds / the channel sub-offset it holds
  const bool im_lane = (col & 8) != 0;
  const int nvec = P.nvec;
  const int NT = (nvec + kVT64 - 1) / kVT64;
  const int ngk = (nvec + 7) / 8;  // forward positions: 8 vectors (two k-steps of 4) per kilobyte
  const int ncb = A.fpad / kCB64;

  unsigned char* s_ring = smem_raw;                                                // [4 waves][kRing][1 KB]
  double* s_red = reinterpret_cast<double*>(smem_raw + 4 * kRing * 1024);          // [4 waves][3]
  double* s_c = reinterpret_cast<double*>(smem_raw + 4 * kRing * 1024 + 128);      // [NC][ngk + 1][64 lanes][2] (one spare position per tile)
  if (stopped) return;
  {
    // coefficient operand of column tile c: s_c[((c * ngk + p) * 64 + lane) * 2 + u] = C[col = lane & 15][vector 8 p + 4 u + (lane >> 4)].
    // Thread tid fills (u, lane) = (tid & 1, (tid >> 1) & 63) of the positions p = (tid >> 7) + 2 j of every tile: its column
    // is fixed per tile, the slot table is read once, and the loads of a tile's positions go out in batches of eight,
    // unconditionally (clamped index; a load behind a branch costs a round trip each)
    const int u = tid & 1, l = (tid >> 1) & 63, cj = l & 15;
    const int k0 = 4 * u + (l >> 4);
    const int klast = nvec - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int fb = P.bl[c * 8 + (cj & 7)];
      const double* src = (cj < 8 ? A.c_r : A.c_i) + P.coff[c * 8 + (cj & 7)];
      for (int p0 = tid >> 7; p0 < ngk; p0 += 16) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * (p0 + 2 * j) + k0;
          v[j] = src[k < klast ? k : klast];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int p = p0 + 2 * j, k = 8 * p + k0;
          if (p < ngk) s_c[((c * (ngk + 1) + p) * 64 + l) * 2 + u] = (fb >= 0 && k < nvec) ? v[j] : 0.0;
        }
      }
    }
  }
  // this lane's slots (one per column tile): sample row and antenna pair as 32-bit BYTE offsets from the kernel-argument bases
  // (+ kq channels: the lane holds channels kq + 4 r of a block)
  const unsigned pl = im_lane ? 2u : 0u;  // the re lane works on r = 0, 1, the im lane on r = 2, 3
  int my_bl[NC];
  unsigned ob[NC], og0[NC], og1[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    my_bl[c] = P.bl[c * 8 + (col & 7)];
    const int2 ant = P.ant[c * 8 + (col & 7)];
    const unsigned row = (unsigned)(my_bl[c] >= 0 ? my_bl[c] : A.nbls);
    ob[c] = (row * (unsigned)A.fpad + (unsigned)kq + 4u * pl) * 8u;              // double arrays
    og0[c] = ((unsigned)ant.x * (unsigned)A.fpad + (unsigned)kq + 4u * pl) * 16u;  // double2 arrays
    og1[c] = ((unsigned)ant.y * (unsigned)A.fpad + (unsigned)kq + 4u * pl) * 16u;
  }
  __syncthreads();

  const f32x4_t* ops = reinterpret_cast<const f32x4_t*>(A.ops);
  const f64x2* sc2 = reinterpret_cast<const f64x2*>(s_c) + lane;
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned ring_lds = (unsigned)reinterpret_cast<unsigned long long>(s_ring + wave * kRing * 1024);
  const f64x2* ring_rd = reinterpret_cast<const f64x2*>(s_ring + wave * kRing * 1024) + lane;

  // the wave's operand stream (see dense_kernels.hpp): per channel block ngk forward positions, then 2 NT adjoint positions,
  // stored in exactly that order, wave after wave (mfma_pack64_kernel): the gradient pass requests ONE contiguous run, the
  // loss-only pass skips the adjoint positions; past the end the last position is requested again
  const int nb_pos = 2 * NT;
  const int nper = ncb / 4;  // the row padding (a multiple of 128 channels) makes ncb a multiple of 4
  auto cb_of = [&](int n) { return wave + 4 * n; };
  const unsigned wblk = (unsigned)P.a_kf4 * 8u + (unsigned)(wave * nper * (ngk + nb_pos)) * 1024u;  // byte offset of this wave's run
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
  int cslot = 0;  // ring slot of the position the next step consumes
  f64x2 r_cur, r_nxt;
  for (int j = 0; j < kRing; ++j) {
    ring_issue(ring_lds + (unsigned)j * 1024u, ops, voff, rq_off);
    req_advance();
  }
  RING_WAIT(kRing - 1);
  r_cur = ring_rd[0];
  // (Unlike the fp32 kernel this one never leaves the sample refill out of the count: its samples are ordinary loads, and a
  // relaxed wait would rest on hipcc emitting exactly 6 NC of them per block -- ADVICE round 2.  Measured cost of the plain
  // wait: 1.2 % of the gradient pass.)
#define STREAM_NEXT_INTO(R, UNUSED)                       \
  __builtin_amdgcn_sched_barrier(0);                       \
  RING_WAIT(kRing - 2);                                    \
  R = ring_rd[((cslot + 1) & (kRing - 1)) * 64];           \
  __builtin_amdgcn_sched_barrier(0);
#define STREAM_REQ()                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  ring_issue(ring_lds + (unsigned)cslot * 1024u, ops, voff, rq_off);                             \
  req_advance();                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
#define STREAM_STEP() cslot = (cslot + 1) & (kRing - 1);

  f64x4 dC[NC][NTMAX];  // gradient tiles: lane (col, kq), element r -> vector 16 t + kq + 4 r
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < NTMAX; ++t) dC[c][t] = f64x4{0.0, 0.0, 0.0, 0.0};
  double loss_acc = 0.0, sr_acc = 0.0, si_acc = 0.0;
  const double al_r = A.use_alpha ? sst->alpha_r : 0.0, al_i = A.use_alpha ? sst->alpha_i : 0.0;
  const char* p_dr = reinterpret_cast<const char*>(A.data_r);
  const char* p_di = reinterpret_cast<const char*>(A.data_i);
  const char* p_w = reinterpret_cast<const char*>(A.wgts);
  const char* p_g = reinterpret_cast<const char*>(A.gains);
  char* p_q = reinterpret_cast<char*>(A.q0);
  struct Samples { double dr[NC][2], di[NC][2], w[NC][2]; };  // one channel block: [column tile][this lane's two channels]
  auto load_samples = [&](int cbn, Samples& S) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned o = ob[c] + (unsigned)cbn * (kCB64 * 8u) + 32u * i;  // channel 16 cb + kq + 4 (pl + i)
        S.dr[c][i] = *reinterpret_cast<const double*>(p_dr + o);
        S.di[c][i] = *reinterpret_cast<const double*>(p_di + o);
        S.w[c][i] = *reinterpret_cast<const double*>(p_w + o);
      }
  };
  // ONE sample buffer, refilled behind the element stage that has just used it: a (current, next) pair carried around the
  // block loop costs 24 more registers and is copied at the loop edge, behind a wait for the loads.
  Samples S_cur;
  load_samples(cb_of(0), S_cur);

  for (int nb = 0; nb < nper; ++nb) {
    const int cb = cb_of(nb);
    // ---- F: rows = this block's 16 channels, cols = (slot, re | im), K = vectors; position = 8 vectors (2 MFMAs per column tile)
    f64x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
    // the coefficient operand has a spare position behind its last, so the read-ahead needs no clamp
    f64x2 c_cur[NC], c_nxt[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) c_cur[c] = sc2[(c * (ngk + 1)) * 64];
    int gq = 0;
#define F_POS64(RC, RN, CC, CN)                                                                                       \
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(RC[0], CC[0][0], acc[0], 0, 0, 0);                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  ++gq;                                                                                                               \
  _Pragma("unroll") for (int c = 0; c < NC; ++c) CN[c] = sc2[(c * (ngk + 1) + gq) * 64];                              \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(RC[1], CC[0][1], acc[0], 0, 0, 0);                                    \
  STREAM_NEXT_INTO(RN, false)                                                                                         \
  if (NC > 1) acc[NC - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(RC[0], CC[NC - 1][0], acc[NC - 1], 0, 0, 0);         \
  STREAM_REQ()                                                                                                        \
  if (NC > 1) acc[NC - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(RC[1], CC[NC - 1][1], acc[NC - 1], 0, 0, 0);         \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  STREAM_STEP()
    // (one position per trip: with the operand registers of two positions live the gradient pass spills, and scratch
    // traffic is vector-memory traffic -- it would break the counted waits of the ring)
    for (int g = 0; g < ngk; ++g) {
      F_POS64(r_cur, r_nxt, c_cur, c_nxt)
      r_cur = r_nxt;
#pragma unroll
      for (int c = 0; c < NC; ++c) c_cur[c] = c_nxt[c];
    }
#undef F_POS64
    // acc[c][r] of lane (col, kq) = v(channel 16 cb + kq + 4 r) of column col of tile c

    // ---- E: element-wise; per column tile the re lane evaluates channels r = 0, 1 and the im lane r = 2, 3 (see the header)
    const unsigned cb16 = (unsigned)cb * (kCB64 * 16u);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // the tile's four gain loads go out together (they come from L2: one round trip per tile, not one per channel pair)
      double2 g0s[2], g1s[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        g0s[i] = *reinterpret_cast<const double2*>(p_g + (og0[c] + cb16 + 64u * i));
        g1s[i] = *reinterpret_cast<const double2*>(p_g + (og1[c] + cb16 + 64u * i));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double2 g0 = g0s[i], g1 = g1s[i];
        // the re lane gets the imaginary part of channel i from its partner, the im lane the real part of channel i + 2
        const double mine_re = acc[c][i], mine_im = acc[c][i + 2];
        const double got = swap8(im_lane ? mine_re : mine_im);
        const double vr = im_lane ? got : mine_re, vi = im_lane ? mine_im : got;
        const double d_r = S_cur.dr[c][i], d_i = S_cur.di[c][i], w = S_cur.w[c][i];
        // G = g0 conj(g1)   (calibration.py:1598-1601: grgr + gigi, gigr - grgi)
        const double G_r = g0.x * g1.x + g0.y * g1.y;
        const double G_i = g0.y * g1.x - g0.x * g1.y;
        const double m_r = G_r * vr - G_i * vi;
        const double m_i = G_i * vr + G_r * vi;
        const double r_r = d_r - m_r, r_i = d_i - m_i;
        loss_acc += w * (r_r * r_r + r_i * r_i);
        sr_acc += w * m_r;  // S = sum w m of the "sum" regulariser (calibration.py:1648-1649)
        si_acc += w * m_i;
        if (GRAD) {
          const double e_r = -2.0 * w * r_r + al_r * w, e_i = -2.0 * w * r_i + al_i * w;
          // gbar_v = conj(G) e: the re lane keeps the real part of its channel and needs the real part of the im lane's
          // channel; the im lane keeps the imaginary part of its channel and needs the imaginary part of the re lane's
          const double gv_r = G_r * e_r + G_i * e_i, gv_i = G_r * e_i - G_i * e_r;
          const double back = swap8(im_lane ? gv_r : gv_i);
          acc[c][i] = im_lane ? back : gv_r;      // channel i:     re lane: own real part;       im lane: re lane's imaginary part
          acc[c][i + 2] = im_lane ? gv_i : back;  // channel i + 2: re lane: im lane's real part; im lane: own imaginary part
          // gbar_G = conj(v) e
          double2 q;
          q.x = vr * e_r + vi * e_i;
          q.y = vr * e_i - vi * e_r;
          *reinterpret_cast<double2*>(p_q + (2u * ob[c] + cb16 + 64u * i)) = q;
        }
      }
    }

    // the next block's samples (the last block loads itself again: the count of loads in flight stays what the waits assume)
    __builtin_amdgcn_sched_barrier(0);
    load_samples(cb_of(nb + 1 < nper ? nb + 1 : nb), S_cur);
    __builtin_amdgcn_sched_barrier(0);

    // ---- B: rows = vectors of tile t, cols = (slot, re | im), K = this block's channels; position = (tile, half v): r = 2 v, 2 v + 1
    if (GRAD) {
#define B_POS64(T, V)                                                                                                 \
  dC[0][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_cur[0], acc[0][2 * (V) + 0], dC[0][T], 0, 0, 0);                    \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  dC[0][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_cur[1], acc[0][2 * (V) + 1], dC[0][T], 0, 0, 0);                    \
  STREAM_NEXT_INTO(r_nxt, 2 * (T) + (V) < kRing - 1)                                                                  \
  if (NC > 1) dC[NC - 1][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_cur[0], acc[NC - 1][2 * (V) + 0], dC[NC - 1][T], 0, 0, 0); \
  STREAM_REQ()                                                                                                        \
  if (NC > 1) dC[NC - 1][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_cur[1], acc[NC - 1][2 * (V) + 1], dC[NC - 1][T], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  r_cur = r_nxt;                                                                                                      \
  STREAM_STEP()
#pragma unroll
      for (int t = 0; t < NTMAX; ++t) {
        if (t < NT) {  // wave-uniform
          B_POS64(t, 0)
          B_POS64(t, 1)
        }
      }
#undef B_POS64
    }
  }
#undef STREAM_NEXT_INTO
#undef STREAM_REQ
#undef STREAM_STEP
  RING_WAIT(0);

  // ---- panel epilogue: loss partials (fixed order), then the coefficient gradients
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
  // each wave holds the sums over ITS channel blocks; eight tiles at a time the four parts meet in the (now idle) ring area
  // ([8 tiles][4 waves][4 elements][64 lanes] doubles = 64 KB would not fit: [4 tiles] = 32 KB does) and ALL 256 threads add
  // them, in wave order: thread (tile tid >> 6, column tid & 15, quarter (tid >> 4) & 3) owns the 4 consecutive vectors
  // 4 quarter .. + 3 of its column -- element `quarter` of the four lanes kq = 0..3 -- and stores them as one run
  double* s_x = reinterpret_cast<double*>(s_ring);
  const int e_tile = tid >> 6, e_q = (tid >> 4) & 3;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    double* gc = (im_lane ? A.gc_i : A.gc_r) + P.coff[c * 8 + (col & 7)];  // the thread's column is its MFMA column: tid & 15 == lane & 15
#pragma unroll
    for (int t0 = 0; t0 < NTMAX; t0 += 4) {
      if (t0 < NT) {  // wave-uniform
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (t0 + tt < NTMAX && t0 + tt < NT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s_x[((tt * 4 + wave) * 4 + j) * 64 + lane] = dC[c][t0 + tt][j];
          }
        __syncthreads();
        const int t = t0 + e_tile;
        if (t < NT && my_bl[c] >= 0) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const double* px = s_x + ((e_tile * 4) * 4 + e_q) * 64 + col + 16 * kk;
            const double v = ((px[0] + px[4 * 64]) + px[2 * 4 * 64]) + px[3 * 4 * 64];
            const int n = kVT64 * t + 4 * e_q + kk;
            if (n < nvec) gc[n] = v;
          }
        }
        __syncthreads();
      }
    }
  }
}

// One launch for all panels, heaviest first: blocks of more than 128 vectors with one column tile (8 baselines, 16
// gradient tiles), the others with two (16 baselines, 8 tiles) -- see dense_kernels.hpp
template <bool GRAD>
__global__ __launch_bounds__(kDenseThreads, 2) void fused_dense64_kernel(const Dense64Args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int slot = A.slot_map[blockIdx.x];
  if (slot < 0) return;
  if (A.panels[slot].nvec > 128)
    dense64_panel<GRAD, 1, 16>(A, smem_raw);
  else
    dense64_panel<GRAD, 2, 8>(A, smem_raw);
}

inline size_t dense64_lds_bytes(int nvec_max, int nc, bool grad) {
  return 4 * (size_t)(grad ? 8 : 4) * 1024 + 128 + (size_t)nc * ((nvec_max + 7) / 8 + 1) * 1024;  // + the spare coefficient position per tile
}

// packed operands of the f64 kernel, per basis block: wave after wave (w = 0..3), the wave's channel blocks cb = w + 4 n in
// order, each as [ngk forward KB][2 NT adjoint KB] -- the order the kernel consumes:
//   forward position p  [64 lanes][2]: lane (i, kq), u -> A[16 cb + i][8 p + 4 u + kq]
//   adjoint position (t, v) [64 lanes][2]: lane (i, kq), u -> A[16 cb + kq + 4 (2 v + u)][16 t + i]
__global__ void mfma_pack64_kernel(const double* __restrict__ src, double* __restrict__ dst, int nfreqs, int fpad, int nvec) {
  const int ngk = (nvec + 7) / 8, NT = (nvec + kVT64 - 1) / kVT64;
  const int nper = fpad / (4 * kCB64);
  const long long per_cb = (long long)(ngk + 2 * NT) * 128, total = per_cb * (fpad / kCB64);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cbp = (int)(i / per_cb);
    const int r = (int)(i % per_cb);
    const int cb = cbp / nper + 4 * (cbp % nper);
    const int u = r & 1, l = (r >> 1) & 63;
    double val = 0.0;
    if (r < ngk * 128) {
      const int p = r >> 7;
      const int f = cb * kCB64 + (l & 15), k = 8 * p + 4 * u + (l >> 4);
      if (f < nfreqs && k < nvec) val = src[(long long)f * nvec + k];
    } else {
      const int pos = (r >> 7) - ngk;
      const int v = pos & 1, t = pos >> 1;
      const int f = kCB64 * cb + (l >> 4) + 4 * (2 * v + u), n = kVT64 * t + (l & 15);
      if (f < nfreqs && n < nvec) val = src[(long long)f * nvec + n];
    }
    dst[i] = val;
  }
}

}  // namespace calk
