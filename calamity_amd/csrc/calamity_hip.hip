// calamity_hip.hip -- C-ABI (include/calamity_hip.h) and host orchestration of the gfx950 fit kernels.
// No PyTorch, no TensorFlow, no CPU fallback: every compute entry point launches the HIP kernels of fit_kernels.hpp.
#include "../../include/calamity_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "fit_kernels.hpp"
#include "dense_kernels.hpp"
#include "split_kernels.hpp"
#include "split2_kernels.hpp"
#include "dense64_kernels.hpp"
#include "multi_mfma_kernels.hpp"
#include <cstdlib>
#include <dlfcn.h>
#include <type_traits>

using namespace calk;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                          \
  do {                                                                                                         \
    hipError_t _e = (expr);                                                                                    \
    if (_e != hipSuccess) return fail(CAL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define NCCL_TRY(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t _e = (expr);                                                                                  \
    if (_e != ncclSuccess) return fail(CAL_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CAL_TRY(expr)      \
  do {                     \
    int _r = (expr);       \
    if (_r != CAL_OK) return _r; \
  } while (0)

// Zero fills of new allocations run on a non-blocking utility stream of the current device, never on the legacy stream: a
// hipMemset there synchronises with every other stream, and HIP refuses it ("operation would make the legacy stream depend on
// a capturing ... stream") while ANOTHER thread's solver captures its step graph -- which is what parallel_fits does.
static hipError_t zero_fill(void* p, size_t n) {
  static std::mutex mu;
  static std::map<int, hipStream_t> streams;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipStream_t st;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(dev);
    if (it == streams.end()) {
      e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      if (e != hipSuccess) return e;
      streams[dev] = st;
    } else {
      st = it->second;
    }
  }
  e = hipMemsetAsync(p, 0, n, st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(st);
}

struct DevBuf {  // owning device allocation
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  int alloc(size_t n, bool zero = true) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(CAL_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
    }
    bytes = n;
    if (zero) {
      // complete before this returns: the solver works on its own non-blocking stream, and an unfinished fill could land on top
      // of a later upload
      e = zero_fill(p, n);
      if (e != hipSuccess) return fail(CAL_ERR_HIP, "zero fill of a new allocation failed: %s", hipGetErrorString(e));
    }
    return CAL_OK;
  }
  template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
};

// roctx ranges around the chunks of a PROFILED run (n_profile_steps of the reference wraps its profiled steps in
// tf.profiler.experimental.Trace, calibration.py:681-687): rocprofv3 --marker-trace then shows "calamity: train steps [a, b)"
// over the kernels of each chunk.  The marker library is looked up at run time; without it the ranges are no-ops.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
      }
    }
  }
};
inline Roctx& roctx() {
  static Roctx r;
  return r;
}

inline int grid_for(long long n, int block = 256, int cap = 16384) {
  long long g = (n + block - 1) / block;
  return (int)std::max<long long>(1, std::min<long long>(g, cap));
}

}  // namespace

// ================================================================================================================
struct cal_solver {
  virtual ~cal_solver() {}
  int device = 0;
  int dtype = CAL_F32;
  virtual int set_problem(const cal_problem_desc* d) = 0;
  virtual int set_data(const void* dr, const void* di, const void* w) = 0;
  virtual int set_regularization(int mode, const double* pr, const double* pi, bool per_slice) = 0;
  virtual int slice_count() = 0;
  virtual int get_slice_losses(double* out) = 0;
  virtual int set_optimizer(const cal_optimizer_desc* d) = 0;
  virtual int set_params(const void* g_r, const void* g_i, const void* c_r, const void* c_i) = 0;
  virtual int get_params(int which, void* g_r, void* g_i, void* c_r, void* c_i) = 0;
  virtual int get_moments(void* gm_r, void* gm_i, void* gv_r, void* gv_i, void* cm_r, void* cm_i, void* cv_r, void* cv_i,
                          int64_t* t) = 0;
  virtual int set_moments(const void* gm_r, const void* gm_i, const void* gv_r, const void* gv_i, const void* cm_r,
                          const void* cm_i, const void* cv_r, const void* cv_i, int64_t t) = 0;
  virtual int eval(bool grads, double* loss, void* gg_r, void* gg_i, void* gc_r, void* gc_i) = 0;
  virtual int run(const cal_run_desc* r, double* losses_out, cal_run_result* res, bool per_slice) = 0;
  virtual int model(void* mr, void* mi, bool with_gains) = 0;
  virtual int init_coeffs(const void* sr, const void* si) = 0;
  virtual int synchronize() = 0;
  virtual int timing_enable(int e) = 0;
  virtual int timing_get(cal_kernel_timing* out) = 0;
  virtual int memory_bytes(int64_t* b) = 0;
  virtual int comm_init(const void* id, int rank, int nranks) = 0;
  virtual int set_launch_mode(int mode) = 0;
  virtual int set_exchange_hook(cal_exchange_fn fn, void* ctx, int rank, int nranks) = 0;
  virtual int comm_size(int* nranks_seen) = 0;
};

template <typename T>
struct SolverT final : cal_solver {
  using T2 = vec2_t<T>;
  hipStream_t stream = nullptr;
  // Synchronous copies go through the solver's OWN (non-blocking) stream, never the legacy stream: a copy there synchronises with
  // every other stream, and HIP refuses it ("operation would make the legacy stream depend on a capturing ... stream") while ANOTHER
  // thread's solver captures its step graph -- parallel_fits, the workers of a SliceBatchFitter.
  hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(stream);
  }
  bool has_problem = false, has_data = false, has_gains = false, has_coef = false, has_opt = false;
  // problem
  int nants = 0, nfreqs = 0, fpad = 0, ngrps = 0, nbls = 0, ncoef = 0, nitems = 0, layout = 0;
  // time slices (cal_problem_desc::nslices): independent fits held together; slice t owns antennas [t na_slice, (t + 1) na_slice),
  // a contiguous run of the coefficient planes, its own loop state (state[par][t]), reduced sums (scal[4 t ..]) and loss history
  int nslices = 1, na_slice = 0;
  DevBuf slice_coff, slice_ipart_ptr, slice_ipart_idx, slice_ppart_ptr, slice_ppart_idx, slice_cblk;
  std::vector<int> h_slice_coff, h_slice_cblk;
  bool small_loads = false; // every single-baseline item's tile fits kSmallLoads loads per thread: the narrow instance of fused_basis_kernel serves the loss / gradient passes
  int nitems_simple = 0;   // items [0, nitems_simple) are single-baseline groups (fused_basis_kernel), the rest multi-baseline (fused_group_kernel)
  int nitems_plain = 0;    // items [0, nitems_plain) of those are not covered by a head item of the multi-slice kernels
  size_t lds_group_bytes = 0;
  bool gc_direct = true;
  size_t lds_bytes = 0;
  long long gcp_len = 0;
  std::vector<int> h_grp_coff;
  double basis_bytes = 0;
  // device buffers
  DevBuf tiles, bl_tile, bl_ant, runs, items, ant_ptr, ant_ent, coef_grp, grp_coff, grp_item_ptr, item_goff;
  DevBuf data_r, data_i, wgts;
  DevBuf gains, gains_m, gains_v, gains_snap;  // [nants][fpad] T2
  DevBuf gains_alt;                            // one-launch tail (step_tail_kernel): gains are read from one buffer and written to the other; `gains` is always the current one
  DevBuf coef, coef_m, coef_v, coef_snap;      // [2][ncoef] T (r plane then i plane)
  DevBuf q0, q1, comm;                         // comm: r0 | r1 | r2 (gain gradient parts), contiguous for the all-reduce
  DevBuf scal;                                 // [nslices] x 4 doubles: loss, s_r, s_i, spare
  DevBuf gcp0, gcp1, gc0, gc1;                 // coefficient-gradient partials and (multi-item groups) their sums
  DevBuf part, state, losses, scratch, model_buf;
  DevBuf members, heads;                       // baselines that share tiles (bl_alias): member lists of the head items, head item indices
  int nheads = 0;                              // heads[0 .. nheads_mfma): fused_multi_mfma_kernel (at most kMmMaxVec vectors); the rest: fused_multi_kernel
  int nheads_mfma = 0;
  bool heads_one_pass = false;                 // the regularised step of the heads in ONE pass (all of them on fused_multi_mfma_kernel<.., REG = 2>)
  int mm_grid = 0;                             // workgroups of the matrix-core multi-slice launch: its head list is dealt over the 8 XCDs (-1: empty slot)
  size_t lds_multi_bytes = 0, lds_multi_mfma_bytes = 0;
  // dense (MFMA) path of the SHARED layout, fp32, one baseline per fitting group
  DevBuf mf_ops, mf_panels;                    // mf_ops: every basis block's packed MFMA operands (see mfma_pack_kernel / mfma_pack64_kernel)
  int mf_npanels = 0;
  bool mf_split2 = false;                      // ... in its one-image form (split2_kernels.hpp)
  bool mf_split = false;                       // fp32: the split-bf16 kernel (split_kernels.hpp: super-panels of 4 panels) instead of fused_dense_kernel
  DevBuf mf_map;                               // [mf_grid] workgroup -> panel (-1: empty slot): XCD-affine dispatch of the dense launch
  int mf_grid = 0;
  size_t mf_lds_grad[2] = {0, 0}, mf_lds_loss[2] = {0, 0};  // per launch class
  bool mf_ok = false;
  int steps_per_sync = 1;                      // train steps enqueued between two host synchronisations of run()
  // small problems: a train step is two launches (fused pass, step_tail_kernel), replayed kGraphSteps at a time from a hipGraph
  int launch_mode = CAL_LAUNCH_AUTO;
  static constexpr int kGraphSteps = 16;       // even: the double-buffered loop state and gains end a replay where they began
  hipGraphExec_t graph_exec = nullptr;
  struct GraphKey {
    int optimizer, freeze, reg, losses_cap, st_par, tail, nsteps;
    const void *gains, *snap, *losses;
    bool operator==(const GraphKey& o) const {
      return optimizer == o.optimizer && freeze == o.freeze && reg == o.reg && losses_cap == o.losses_cap && st_par == o.st_par && tail == o.tail &&
             nsteps == o.nsteps && gains == o.gains && snap == o.snap && losses == o.losses;
    }
  } graph_key{};
  DevBuf agree_buf;
  DevState* h_state = nullptr;                 // pinned mirror: [nslices]
  int h_state_cap = 0;
  int st_par = 0;                              // which half of `state` ([2][nslices]) is current
  DevState* st_cur() { return state.as<DevState>() + (size_t)st_par * nslices; }
  DevState* st_nxt() { return state.as<DevState>() + (size_t)(st_par ^ 1) * nslices; }
  std::vector<double> prior_r_t, prior_i_t;    // per slice
  // which slice every kernel's work belongs to (fit_kernels.hpp: SliceMap); the loss partials are per item (general kernels) or
  // per panel (dense kernels)
  SliceMap smap(bool panels) const {
    SliceMap m{};
    m.coff = slice_coff.as<int>();
    m.nslices = nslices;
    m.na_slice = na_slice;
    if (nslices > 1) {
      m.part_ptr = (panels ? slice_ppart_ptr : slice_ipart_ptr).template as<int>();
      m.part_idx = (panels ? slice_ppart_idx : slice_ipart_idx).template as<int>();
    }
    return m;
  }
  // settings
  cal_optimizer_desc opt{CAL_OPT_ADAMAX, 1e-3, 0.9, 0.999, 1e-7, 0.9, 0.0, 0.1, 0, 0, -0.5, 0.0, 0.0, 0.0, 0.0, 0.0};
  // LAMB: the optimizer's variables (g_r, g_i of every slice; the coefficient runs of every (slice, cal_problem_desc::grp_var) per plane)
  DevBuf lamb_vars, lamb_cvar_ptr, lamb_partial, lamb_ratio;
  int lamb_nvar = 0, lamb_ncvar = 0;
  bool lamb_ok = true;
  // ... over several ranks: a coefficient variable (slice t, grp_var w) is spread over the ranks that own its groups -> its two sums of
  // squares are all-reduced in slot t * lamb_nv + w of lamb_glob (the gain variables are replicated: every rank already holds their norms)
  DevBuf lamb_glob, lamb_slot;
  std::vector<int> lamb_cvar_slice, lamb_cvar_id;
  int lamb_nv = 0;
  int reg = CAL_REG_NONE;
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  long long t_launches = 0;
  double t_total_ms = 0;
  // comm
  ncclComm_t nccl = nullptr;
  int nranks = 1, rank = 0;
  // caller-supplied exchange (cal_solver_set_exchange_hook): the same buffers and counts RCCL would reduce, staged through
  // pinned host memory and reduced by the callback
  cal_exchange_fn hook = nullptr;
  void* hook_ctx = nullptr;
  void* hook_buf = nullptr;
  size_t hook_bytes = 0;
  bool comm_on() const { return nccl != nullptr || hook != nullptr; }

  ~SolverT() override {
    (void)hipSetDevice(device);
    if (nccl) (void)ncclCommDestroy(nccl);
    if (hook_buf) (void)hipHostFree(hook_buf);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    for (auto& e : ev_pool) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    if (h_state) (void)hipHostFree(h_state);
    if (stream) (void)hipStreamDestroy(stream);
  }

  int init() {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    CAL_TRY(size_state(1));
    return CAL_OK;
  }
  // loop state, reduced sums and their host mirror for n time slices (a new problem may change the count)
  int size_state(int n) {
    if (n > h_state_cap) {
      if (h_state) (void)hipHostFree(h_state);
      h_state = nullptr;
      h_state_cap = 0;
      HIP_TRY(hipHostMalloc((void**)&h_state, (size_t)n * sizeof(DevState), hipHostMallocDefault));
      h_state_cap = n;
    }
    memset(h_state, 0, (size_t)n * sizeof(DevState));
    CAL_TRY(state.alloc(2 * (size_t)n * sizeof(DevState)));  // double-buffered: see step_update_kernel
    CAL_TRY(scal.alloc(4 * (size_t)n * sizeof(double)));
    st_par = 0;
    prior_r_t.assign(n, 0.0);
    prior_i_t.assign(n, 0.0);
    for (int t = 0; t < n; ++t) reset_loop_state(h_state[t]);
    return CAL_OK;
  }
  static void reset_loop_state(DevState& h) {  // a new fit begins: loop state of calibration.py:573-574
    h.t = 0;
    h.b1t = h.b2t = 1.0;
    h.nadam_sched = 1.0;
    h.min_loss = 9e99;
    h.prev_loss = 0;
    h.n_recorded_total = 0;
    h.nonfinite = 0;
  }

  static int choose_fb(int nvec, int nfreqs) {
    // widest channel block whose tile (nvec x FB) still fits the staging budget; never wider than the band needs
    int cap = FbSet<T>::fb_min;
    while (cap < FbSet<T>::fb_max && cap < nfreqs) cap *= 2;
    for (int fb = std::min(cap, FbSet<T>::fb_max); fb >= FbSet<T>::fb_min; fb /= 2)
      if ((long long)nvec * fb * (long long)sizeof(T) <= kTileBytes) return fb;
    return -1;
  }
  static size_t group_lds_for(int fb) {
    constexpr int M = FbSet<T>::fb_max;
    if (fb == M) return group_lds_bytes<T, M>();
    if (fb == M / 2) return group_lds_bytes<T, M / 2>();
    if (fb == M / 4) return group_lds_bytes<T, M / 4>();
    if (fb == M / 8) return group_lds_bytes<T, M / 8>();
    return group_lds_bytes<T, M / 16>();
  }
  static size_t multi_lds_for(int fb) {
    constexpr int M = FbSet<T>::fb_max;
    if (fb == M) return multi_lds_bytes<T, M>();
    if (fb == M / 2) return multi_lds_bytes<T, M / 2>();
    if (fb == M / 4) return multi_lds_bytes<T, M / 4>();
    if (fb == M / 8) return multi_lds_bytes<T, M / 8>();
    return multi_lds_bytes<T, M / 16>();
  }
  static size_t lds_for(int fb) {
    constexpr int M = FbSet<T>::fb_max;
    if (fb == M) return TileCfg<T, M>::lds_bytes();
    if (fb == M / 2) return TileCfg<T, M / 2>::lds_bytes();
    if (fb == M / 4) return TileCfg<T, M / 4>::lds_bytes();
    if (fb == M / 8) return TileCfg<T, M / 8>::lds_bytes();
    return TileCfg<T, M / 16>::lds_bytes();
  }

  // ------------------------------------------------------------------------------------------------------------
  // With a communicator attached, every decision a rank takes from its OWN shard and that changes what it exchanges must
  // be taken by all ranks together -- and a rank whose set-up fails must not leave the others waiting in a collective.
  // So the rank-local part (validation, eligibility, every allocation and upload: set_problem_local) contains no collective
  // at all, and ONE agreement follows it on every path, failure included: the minimum over ranks of
  // {set-up ok ? dense kernels built : -1, steps per host synchronisation}.  A negative minimum fails the call on every rank;
  // a rank that built the dense path while another could not drops back to the general kernels (same buffers).
  int set_problem(const cal_problem_desc* d) override {
    const int rc = set_problem_local(d);
    if (!comm_on()) return rc;
    return agree_problem(rc);
  }
  // the agreement: {set-up ok ? dense kernels built : -1, steps per host synchronisation, -(variables per slice), LAMB possible}
  int agree_problem(int rc) {
    const std::string msg = g_err;
    int local_nv = 0;
    for (int w : lamb_cvar_id) local_nv = std::max(local_nv, w + 1);
    int v[4] = {rc == CAL_OK ? (mf_ok ? 1 : 0) : -1, rc == CAL_OK ? steps_per_sync : (1 << 30), -local_nv, rc == CAL_OK && lamb_ok ? 1 : 0};
    const int arc = agree_min(v, 4);
    if (arc != CAL_OK) {
      has_problem = false;
      return arc;
    }
    if (rc != CAL_OK) {
      g_err = msg;
      return rc;
    }
    if (v[0] < 0) {
      has_problem = false;
      return fail(CAL_ERR_STATE, "set_problem failed on another rank of the communicator");
    }
    if (!v[0]) mf_ok = false;
    steps_per_sync = v[1];
    lamb_nv = -v[2];
    if (!v[3]) lamb_ok = false;
    if (lamb_ok) {
      std::vector<int> slot(std::max(lamb_ncvar, 1), 0);
      for (int k = 0; k < lamb_ncvar; ++k) slot[k] = lamb_cvar_slice[k] * lamb_nv + lamb_cvar_id[k];
      CAL_TRY(lamb_slot.alloc(slot.size() * sizeof(int), false));
      HIP_TRY(copy_sync(lamb_slot.p, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(lamb_glob.alloc((size_t)std::max(1, 2 * nslices * lamb_nv) * 2 * sizeof(double)));
    }
    return CAL_OK;
  }
  int set_problem_local(const cal_problem_desc* d) {
    HIP_TRY(hipSetDevice(device));
    has_problem = has_data = has_gains = has_coef = false;
    drop_graph();
    if (!d || d->nants <= 0 || d->nfreqs <= 0 || d->ngrps <= 0 || d->nbls <= 0 || d->nbasis <= 0)
      return fail(CAL_ERR_INVALID, "set_problem: non-positive dimension");
    if (!d->basis_offset || !d->basis_nvec || !d->basis_nrowblk || !d->basis_data || !d->grp_basis || !d->grp_bl_start ||
        !d->bl_ant0 || !d->bl_ant1)
      return fail(CAL_ERR_INVALID, "set_problem: null pointer in problem description");
    if (d->layout != CAL_LAYOUT_STREAM && d->layout != CAL_LAYOUT_SHARED) return fail(CAL_ERR_INVALID, "set_problem: bad layout");
    if (d->grp_bl_start[0] != 0 || d->grp_bl_start[d->ngrps] != d->nbls)
      return fail(CAL_ERR_INVALID, "set_problem: grp_bl_start must run from 0 to nbls");
    nants = d->nants; nfreqs = d->nfreqs; ngrps = d->ngrps; nbls = d->nbls; layout = d->layout;
    const int nbasis = d->nbasis;
    std::vector<int> fb_u(nbasis);
    int fb_used_max = 0;
    for (int u = 0; u < nbasis; ++u) {
      if (d->basis_nvec[u] <= 0 || d->basis_nrowblk[u] <= 0) return fail(CAL_ERR_INVALID, "set_problem: empty basis block %d", u);
      const long long want = (long long)d->basis_nvec[u] * d->basis_nrowblk[u] * nfreqs;
      if (d->basis_offset[u + 1] - d->basis_offset[u] != want)
        return fail(CAL_ERR_INVALID, "set_problem: basis block %d has %lld elements, expected %lld", u,
                    (long long)(d->basis_offset[u + 1] - d->basis_offset[u]), want);
      fb_u[u] = choose_fb(d->basis_nvec[u], nfreqs);
      if (fb_u[u] < 0)
        return fail(CAL_ERR_UNSUPPORTED, "set_problem: basis block %d has %d vectors; at most %d are supported for this dtype", u,
                    d->basis_nvec[u], (int)(kTileBytes / sizeof(T) / FbSet<T>::fb_min));
      fb_used_max = std::max(fb_used_max, fb_u[u]);
    }
    // Row padding is a function of nfreqs alone (never of this rank's basis blocks or kernel choice): every rank of a
    // sharded fit must lay the gains out identically and all-reduce the same number of reals.  pw = min(128,
    // next_pow2(nfreqs)) is a multiple of every tile width in use and, for nfreqs > 64, of the dense kernel's chunk.
    int pw = 8;
    while (pw < 128 && pw < nfreqs) pw *= 2;
    if (fb_used_max > pw) return fail(CAL_ERR_INVALID, "set_problem: internal error: tile width %d exceeds the row padding %d", fb_used_max, pw);
    fpad = (nfreqs + pw - 1) / pw * pw;
    if (d->kernel_path != CAL_PATH_AUTO && d->kernel_path != CAL_PATH_GENERAL && d->kernel_path != CAL_PATH_DENSE && d->kernel_path != CAL_PATH_DENSE_F32 &&
        d->kernel_path != CAL_PATH_DENSE_SPLIT1)
      return fail(CAL_ERR_INVALID, "set_problem: bad kernel_path %d", d->kernel_path);
    if (d->kernel_path == CAL_PATH_DENSE_F32 && !std::is_same<T, float>::value)
      return fail(CAL_ERR_UNSUPPORTED, "set_problem: CAL_PATH_DENSE_F32 is the fp32 kernel on v_mfma_f32_32x32x2_f32; this solver is fp64");
    if (d->kernel_path == CAL_PATH_DENSE_SPLIT1 && !std::is_same<T, float>::value)
      return fail(CAL_ERR_UNSUPPORTED, "set_problem: CAL_PATH_DENSE_SPLIT1 is an fp32 kernel (split-bf16 operands); this solver is fp64");
    const bool forced_dense = d->kernel_path == CAL_PATH_DENSE || d->kernel_path == CAL_PATH_DENSE_F32 || d->kernel_path == CAL_PATH_DENSE_SPLIT1;
    bool want_split = std::is_same<T, float>::value && d->kernel_path != CAL_PATH_DENSE_F32;
    for (int u = 0; u < nbasis && want_split; ++u) want_split = d->basis_nvec[u] <= kSplitMaxNvec;  // (wider blocks: the f32 kernel, up to 256 vectors)
    // dense (matrix-core) path: eligibility, then -- for CAL_PATH_AUTO -- whether the problem fills the chip
    bool dense_ok = layout == CAL_LAYOUT_SHARED && fpad % kChunk == 0;
    for (int g = 0; g < ngrps && dense_ok; ++g) dense_ok = (d->grp_bl_start[g + 1] - d->grp_bl_start[g]) == 1;
    for (int u = 0; u < nbasis && dense_ok; ++u) dense_ok = d->basis_nrowblk[u] == 1 && d->basis_nvec[u] <= DenseCfg<T>::max_nvec;
    // the dense kernels address the per-sample arrays with 32-bit BYTE offsets; the widest sample is one (re, im) pair
    if ((long long)(nbls + 2) * fpad * 2 * (long long)sizeof(T) >= (1LL << 32)) dense_ok = false;
    // ... and its packed operands (two MFMA-native copies of every unique block) with 32-bit byte offsets from one base
    long long dense_op_elems = 0;
    for (int u = 0; u < nbasis && dense_ok; ++u)  // kilobyte positions: forward + adjoint (the same count for both dtypes' layouts up to padding)
      dense_op_elems += want_split ? std::max(split_stream_bytes(fpad, d->basis_nvec[u], (d->basis_nvec[u] + 31) / 32), split2_stream_bytes(fpad, d->basis_nvec[u])) / 4
                        : std::is_same<T, float>::value
                            ? (long long)(fpad / 32) * ((d->basis_nvec[u] + 7) / 8) * 256 + (long long)(fpad / 32) * ((d->basis_nvec[u] + 31) / 32) * 4 * 256
                            : (long long)(fpad / 16) * ((d->basis_nvec[u] + 7) / 8) * 128 + (long long)(fpad / 16) * ((d->basis_nvec[u] + 15) / 16) * 2 * 128;
    if (dense_op_elems * (long long)(want_split ? 4 : sizeof(T)) >= (1LL << 32)) dense_ok = false;
    if (forced_dense && !dense_ok)
      return fail(CAL_ERR_UNSUPPORTED, "set_problem: CAL_PATH_DENSE needs the SHARED layout, one baseline per fitting group, "
                  "basis_nvec <= %d and nfreqs > 64", DenseCfg<T>::max_nvec);
    // a panel of 16 baselines occupies one CU for 60-70 us whatever the problem size; below ~2000 baselines the panels do
    // not fill the chip and the general kernel (one workgroup per baseline) is 2-3x faster (HERA-37 fp32: 25 vs 71 us)
    // (with a communicator the ranks then agree on ONE path -- the exchange payload of the "sum" regulariser differs between
    // the two -- in set_problem, behind all the rank-local work)
    const bool want_mfma = dense_ok && d->kernel_path != CAL_PATH_GENERAL && (forced_dense || nbls >= 2048);
    lds_bytes = 0;
    for (int u = 0; u < nbasis; ++u) lds_bytes = std::max(lds_bytes, lds_for(fb_u[u]));
    for (int b = 0; b < d->nbls; ++b) {
      if (d->bl_ant0[b] < 0 || d->bl_ant0[b] >= nants || d->bl_ant1[b] < 0 || d->bl_ant1[b] >= nants)
        return fail(CAL_ERR_INVALID, "set_problem: baseline %d has an antenna index outside [0, %d)", b, nants);
    }
    // ---- time slices: independent fits over disjoint antenna ranges, listed slice by slice
    const int NSL = d->nslices > 1 ? d->nslices : 1;
    if (NSL > CAL_MAX_SLICES) return fail(CAL_ERR_UNSUPPORTED, "set_problem: %d time slices; at most %d are supported", NSL, CAL_MAX_SLICES);
    if (nants % NSL != 0) return fail(CAL_ERR_INVALID, "set_problem: nants = %d is not a multiple of nslices = %d", nants, NSL);
    const int nas = nants / NSL;
    std::vector<int> grp_slice(ngrps, 0);
    if (NSL > 1) {
      std::vector<char> seen(NSL, 0);
      for (int g = 0; g < ngrps; ++g) {
        const int b0 = d->grp_bl_start[g], b1 = d->grp_bl_start[g + 1];
        if (b0 < 0 || b1 > nbls || b1 <= b0) return fail(CAL_ERR_INVALID, "set_problem: group %d has no baselines", g);
        const int t = d->bl_ant0[b0] / nas;
        for (int b = b0; b < b1; ++b)
          if (d->bl_ant0[b] / nas != t || d->bl_ant1[b] / nas != t)
            return fail(CAL_ERR_INVALID, "set_problem: baseline %d of group %d leaves time slice %d (antennas %d, %d; %d antennas per slice)", b, g, t,
                        d->bl_ant0[b], d->bl_ant1[b], nas);
        if (g > 0 && t < grp_slice[g - 1]) return fail(CAL_ERR_INVALID, "set_problem: fitting groups must be listed slice by slice (group %d)", g);
        grp_slice[g] = t;
        seen[t] = 1;
      }
      for (int t = 0; t < NSL; ++t)
        if (!seen[t]) return fail(CAL_ERR_INVALID, "set_problem: time slice %d has no fitting group", t);
    }
    nslices = NSL;
    na_slice = nas;
    CAL_TRY(size_state(NSL));
    // ---- groups, coefficient offsets
    h_grp_coff.assign(ngrps + 1, 0);
    std::vector<int> grp_of_bl(nbls);
    for (int g = 0; g < ngrps; ++g) {
      const int u = d->grp_basis[g];
      if (u < 0 || u >= nbasis) return fail(CAL_ERR_INVALID, "set_problem: group %d points at basis %d", g, u);
      if (d->grp_bl_start[g + 1] <= d->grp_bl_start[g]) return fail(CAL_ERR_INVALID, "set_problem: group %d has no baselines", g);
      h_grp_coff[g + 1] = h_grp_coff[g] + d->basis_nvec[u];
      for (int b = d->grp_bl_start[g]; b < d->grp_bl_start[g + 1]; ++b) {
        grp_of_bl[b] = g;
        const int rb = d->bl_rowblk ? d->bl_rowblk[b] : 0;
        if (rb < 0 || rb >= d->basis_nrowblk[u]) return fail(CAL_ERR_INVALID, "set_problem: baseline %d row block %d out of range", b, rb);
      }
    }
    ncoef = h_grp_coff[ngrps];
    h_slice_coff.assign(nslices + 1, ncoef);
    h_slice_coff[0] = 0;
    for (int g = ngrps - 1; g >= 0; --g) h_slice_coff[grp_slice[g]] = h_grp_coff[g];  // first group of every slice
    h_slice_coff[0] = 0;
    CAL_TRY(slice_coff.alloc((nslices + 1) * sizeof(int), false));
    HIP_TRY(copy_sync(slice_coff.p, h_slice_coff.data(), (nslices + 1) * sizeof(int), hipMemcpyHostToDevice));
    {
      // the optimizer's variables (LAMB): a new coefficient variable wherever the (slice, grp_var) of the groups changes
      lamb_ok = true;
      std::vector<int> cptr;
      lamb_cvar_slice.clear();
      lamb_cvar_id.clear();
      for (int g = 0; g < ngrps; ++g) {
        const int var = d->grp_var ? d->grp_var[g] : 0;
        if (var < 0) return fail(CAL_ERR_INVALID, "set_problem: grp_var[%d] = %d is negative", g, var);
        // (groups of one variable scattered over a slice: fine for every element-wise optimizer; LAMB is refused in set_optimizer)
        if (g > 0 && grp_slice[g] == grp_slice[g - 1] && d->grp_var && var < d->grp_var[g - 1]) lamb_ok = false;
        if (g == 0 || grp_slice[g] != grp_slice[g - 1] || (d->grp_var && var != d->grp_var[g - 1])) {
          cptr.push_back(h_grp_coff[g]);
          lamb_cvar_slice.push_back(grp_slice[g]);
          lamb_cvar_id.push_back(var);
        }
      }
      lamb_ncvar = (int)cptr.size();
      cptr.push_back(ncoef);
      lamb_nvar = 2 * nslices + 2 * lamb_ncvar;
      std::vector<LambVar> vars;
      for (int t = 0; t < nslices; ++t)
        for (int c = 0; c < 2; ++c) vars.push_back(LambVar{(long long)t * na_slice * fpad * 2 + c, (long long)na_slice * fpad, 2, 0});
      for (int plane = 0; plane < 2; ++plane)
        for (int k = 0; k < lamb_ncvar; ++k) vars.push_back(LambVar{(long long)plane * ncoef + cptr[k], (long long)(cptr[k + 1] - cptr[k]), 1, 1});
      CAL_TRY(lamb_vars.alloc(vars.size() * sizeof(LambVar), false));
      HIP_TRY(copy_sync(lamb_vars.p, vars.data(), vars.size() * sizeof(LambVar), hipMemcpyHostToDevice));
      CAL_TRY(lamb_cvar_ptr.alloc(cptr.size() * sizeof(int), false));
      HIP_TRY(copy_sync(lamb_cvar_ptr.p, cptr.data(), cptr.size() * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(lamb_partial.alloc((size_t)lamb_nvar * kLambSeg * 2 * sizeof(double)));
      CAL_TRY(lamb_ratio.alloc((size_t)lamb_nvar * sizeof(double)));
    }
    // ---- baselines that read another baseline's tiles (STREAM layout): the same physical baseline in several time slices
    std::vector<int> alias_root(nbls, -1);  // -1: owns its tiles
    if (d->bl_alias && layout == CAL_LAYOUT_STREAM) {
      for (int b = 0; b < nbls; ++b) {
        const int r = d->bl_alias[b];
        if (r < 0 || r == b) continue;
        if (r >= nbls || (d->bl_alias[r] >= 0 && d->bl_alias[r] != r))
          return fail(CAL_ERR_INVALID, "set_problem: bl_alias[%d] = %d must name a baseline that owns its tiles", b, r);
        const int g = grp_of_bl[b], gr = grp_of_bl[r];
        if (d->grp_basis[g] != d->grp_basis[gr] || (d->bl_rowblk ? d->bl_rowblk[b] != d->bl_rowblk[r] : false))
          return fail(CAL_ERR_INVALID, "set_problem: bl_alias[%d] = %d: the two baselines use different basis rows", b, r);
        if (d->grp_bl_start[g + 1] - d->grp_bl_start[g] != 1 || d->grp_bl_start[gr + 1] - d->grp_bl_start[gr] != 1)
          return fail(CAL_ERR_INVALID, "set_problem: bl_alias is for single-baseline fitting groups (baseline %d)", b);
        alias_root[b] = r;
      }
    }
    std::vector<char> in_alias_set(nbls, 0);
    const bool multi_ok = (long long)(nbls + 1) * fpad < (1LL << 31) && (long long)nants * fpad < (1LL << 31);  // the multi kernel's 32-bit sample offsets
    for (int b = 0; b < nbls; ++b)
      if (alias_root[b] >= 0 && multi_ok) in_alias_set[b] = in_alias_set[alias_root[b]] = 1;  // (slices_share_heads: see below)

    // ---- unique basis blocks -> tile-major device layout
    const long long raw_elems = d->basis_offset[nbasis];
    DevBuf raw, utiles;
    CAL_TRY(raw.alloc((size_t)raw_elems * sizeof(T), false));
    HIP_TRY(hipMemcpyAsync(raw.p, d->basis_data, (size_t)raw_elems * sizeof(T), hipMemcpyHostToDevice, stream));
    std::vector<long long> uoff(nbasis + 1, 0);
    for (int u = 0; u < nbasis; ++u) uoff[u + 1] = uoff[u] + (long long)d->basis_nrowblk[u] * fpad * d->basis_nvec[u];
    // (+ a zeroed pad: fused_multi_mfma_kernel reads on past the last rows of a tile, against zero coefficients)
    CAL_TRY(utiles.alloc(((size_t)uoff[nbasis] + kMmTilePadElems) * sizeof(T), false));
    HIP_TRY(hipMemsetAsync(utiles.as<T>() + uoff[nbasis], 0, kMmTilePadElems * sizeof(T), stream));
    for (int u = 0; u < nbasis; ++u) {
      const long long n = uoff[u + 1] - uoff[u];
      hipLaunchKernelGGL(retile_kernel<T>, dim3(grid_for(n)), dim3(256), 0, stream, raw.as<T>() + d->basis_offset[u],
                         utiles.as<T>() + uoff[u], nfreqs, fpad, d->basis_nvec[u], d->basis_nrowblk[u], fb_u[u]);
    }
    HIP_TRY(hipGetLastError());
    std::vector<long long> h_bl_tile(nbls);
    basis_bytes = 0;
    if (layout == CAL_LAYOUT_SHARED) {
      for (int b = 0; b < nbls; ++b) {
        const int u = d->grp_basis[grp_of_bl[b]];
        const int rb = d->bl_rowblk ? d->bl_rowblk[b] : 0;
        h_bl_tile[b] = uoff[u] + (long long)rb * fpad * d->basis_nvec[u];
      }
      HIP_TRY(hipStreamSynchronize(stream));
      tiles.release();
      tiles.p = utiles.p; tiles.bytes = utiles.bytes;
      utiles.p = nullptr; utiles.bytes = 0;
      basis_bytes = (double)uoff[nbasis] / fpad * nfreqs * sizeof(T);
    } else {
      // every baseline owns its tiles, except that consecutive baselines of one group with the same row block (a
      // redundant set: one forward product for all of them) share one copy
      std::vector<CopyJob> jobs;
      jobs.reserve(nbls);
      long long off = 0;
      for (int b = 0; b < nbls; ++b) {
        const int u = d->grp_basis[grp_of_bl[b]];
        const int rb = d->bl_rowblk ? d->bl_rowblk[b] : 0;
        const long long n = (long long)fpad * d->basis_nvec[u];
        const bool alias = b > 0 && grp_of_bl[b - 1] == grp_of_bl[b] && (d->bl_rowblk ? d->bl_rowblk[b - 1] : 0) == rb;
        if (alias) {
          h_bl_tile[b] = h_bl_tile[b - 1];
          continue;
        }
        if (alias_root[b] >= 0) continue;  // filled in below, once its owner's offset is known (the owner may come later)
        jobs.push_back(CopyJob{uoff[u] + (long long)rb * n, off, n});
        h_bl_tile[b] = off;
        off += n;
      }
      for (int b = 0; b < nbls; ++b)
        if (alias_root[b] >= 0) h_bl_tile[b] = h_bl_tile[alias_root[b]];
      CAL_TRY(tiles.alloc(((size_t)off + kMmTilePadElems) * sizeof(T), false));
      HIP_TRY(hipMemsetAsync(tiles.as<T>() + off, 0, kMmTilePadElems * sizeof(T), stream));
      DevBuf djobs;
      CAL_TRY(djobs.alloc(jobs.size() * sizeof(CopyJob), false));
      HIP_TRY(hipMemcpyAsync(djobs.p, jobs.data(), jobs.size() * sizeof(CopyJob), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(tile_copy_kernel<T>, dim3((unsigned)jobs.size()), dim3(256), 0, stream, utiles.as<T>(), tiles.as<T>(), djobs.as<CopyJob>());
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(stream));
      basis_bytes = (double)off / fpad * nfreqs * sizeof(T);
    }
    mf_ok = false;
    mf_split = false;
    mf_split2 = false;
    if (want_mfma && want_split) {
      if constexpr (std::is_same<T, float>::value) {
        // ---- fp32, split-bf16 operands (split_kernels.hpp): super-panels of kSpWaves panels (64 baselines) with the same basis block and slice;
        // an item carries at most kSplitNT vector tiles: a block of more than 128 vectors is two items per super-panel (each with its own packed
        // stream: the whole forward, half of the adjoint tiles)
        struct Half { int tile0, ntiles; long long obyte; };
        std::vector<std::vector<Half>> halves(nbasis);
        long long obytes = 0;
        const bool v2 = d->kernel_path != CAL_PATH_DENSE_SPLIT1;  // the one-image form (split2_kernels.hpp) unless the first one is asked for by name
        for (int u = 0; u < nbasis; ++u) {
          const int ntu = (d->basis_nvec[u] + 31) / 32;
          halves[u] = {Half{0, ntu, 0}};  // (kSplitNT = 8 tiles fit one item since the kernel runs one workgroup per CU)
          for (Half& h : halves[u]) {
            h.obyte = obytes;
            obytes += v2 ? split2_stream_bytes(fpad, d->basis_nvec[u]) : split_stream_bytes(fpad, d->basis_nvec[u], h.ntiles);
          }
        }
        CAL_TRY(mf_ops.alloc((size_t)obytes, false));
        for (int u = 0; u < nbasis; ++u)
          for (const Half& h : halves[u]) {
            if (v2)
              hipLaunchKernelGGL(split2_pack_kernel, dim3(grid_for((long long)fpad * 256)), dim3(256), 0, stream, raw.as<float>() + d->basis_offset[u],
                                 mf_ops.as<unsigned char>() + h.obyte, nfreqs, fpad, d->basis_nvec[u]);
            else
              hipLaunchKernelGGL(split_pack_kernel, dim3(grid_for(split_stream_bytes(fpad, d->basis_nvec[u], h.ntiles) / 6)), dim3(256), 0, stream,
                                 raw.as<float>() + d->basis_offset[u], reinterpret_cast<unsigned short*>(mf_ops.as<unsigned char>() + h.obyte), nfreqs, fpad,
                                 d->basis_nvec[u], h.tile0, h.ntiles);
          }
        HIP_TRY(hipGetLastError());
        std::vector<std::vector<int>> by_u((size_t)nbasis * nslices);
        for (int b = 0; b < nbls; ++b) by_u[(size_t)d->grp_basis[grp_of_bl[b]] * nslices + grp_slice[grp_of_bl[b]]].push_back(b);
        std::vector<int> uorder((size_t)nbasis * nslices);
        std::iota(uorder.begin(), uorder.end(), 0);
        std::stable_sort(uorder.begin(), uorder.end(), [&](int a, int b) { return d->basis_nvec[a / nslices] > d->basis_nvec[b / nslices]; });
        std::vector<PanelItem> h_panels;
        std::vector<double> h_cost;
        const int wide = kPanel * kSpWaves;
        for (int us : uorder) {
          const int u = us / nslices;
          const int nv = d->basis_nvec[u];
          for (size_t i = 0; i < by_u[us].size(); i += wide) {
            for (const Half& h : halves[u]) {
              for (int w = 0; w < kSpWaves; ++w) {
                PanelItem pi{};
                pi.slice = us % nslices;
                for (int k = 0; k < kPanel; ++k) {
                  const size_t at = i + (size_t)w * kPanel + k;
                  const int b = at < by_u[us].size() ? by_u[us][at] : -1;
                  pi.bl[k] = b;
                  pi.coff[k] = b >= 0 ? h_grp_coff[grp_of_bl[b]] : 0;
                  pi.ant[k] = b >= 0 ? make_int2(d->bl_ant0[b], d->bl_ant1[b]) : make_int2(0, 0);
                }
                pi.a_kf4 = h.obyte / 4;
                pi.a_fk4 = 0;
                pi.nvec = nv;
                pi.nvp2 = (nv + 15) / 16 * 16;
                pi.nvp32 = 32 * h.ntiles;
                pi.tile0 = h.tile0;
                h_panels.push_back(pi);
              }
              // an item's time on a CU: per channel-block pair two element stages + its groups of 24 MFMAs
              h_cost.push_back((fpad / 64) * (3000.0 + 900.0 * split_groups_per_pair(nv, h.ntiles)));
            }
          }
        }
        CAL_TRY(order_panels(h_panels, h_cost, kSpWaves));
        mf_lds_grad[0] = mf_lds_loss[0] = v2 ? (size_t)kS2Lds : split_lds_bytes();
        if (v2) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense_split2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_grad[0]));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense_split2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_loss[0]));
        } else {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_grad[0]));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_loss[0]));
        }
        mf_ok = true;
        mf_split = true;
        mf_split2 = v2;
      }
    } else if (want_mfma) {
      if constexpr (std::is_same<T, float>::value) {
        std::vector<long long> okf4(nbasis + 1, 0), ofk4(nbasis + 1, 0);
        std::vector<int> nvp2(nbasis), nvp32(nbasis);
        int nvec_max = 0;
        for (int u = 0; u < nbasis; ++u) {
          nvp2[u] = (d->basis_nvec[u] + 15) / 16 * 16;
          nvp32[u] = (d->basis_nvec[u] + 31) / 32 * 32;
          nvec_max = std::max(nvec_max, d->basis_nvec[u]);
          okf4[u + 1] = okf4[u] + (long long)(fpad / 32) * ((d->basis_nvec[u] + 7) / 8 + nvp32[u] / 32 * 4) * 256;
        }
        CAL_TRY(mf_ops.alloc((size_t)okf4[nbasis] * sizeof(float), false));
        for (int u = 0; u < nbasis; ++u)
          hipLaunchKernelGGL(mfma_pack_kernel, dim3(grid_for(okf4[u + 1] - okf4[u])), dim3(256), 0, stream,
                             raw.as<float>() + d->basis_offset[u], mf_ops.as<float>() + okf4[u], nfreqs, fpad, d->basis_nvec[u], nvp32[u]);
        HIP_TRY(hipGetLastError());
        // panels of kPanel baselines with the same basis, heaviest first
        // (a panel never mixes time slices: one loop state, one alpha per panel)
        std::vector<std::vector<int>> by_u((size_t)nbasis * nslices);
        for (int b = 0; b < nbls; ++b) by_u[(size_t)d->grp_basis[grp_of_bl[b]] * nslices + grp_slice[grp_of_bl[b]]].push_back(b);
        std::vector<int> uorder((size_t)nbasis * nslices);
        std::iota(uorder.begin(), uorder.end(), 0);
        std::stable_sort(uorder.begin(), uorder.end(), [&](int a, int b) { return d->basis_nvec[a / nslices] > d->basis_nvec[b / nslices]; });
        // Panels of more than four vector tiles first, then the rest (two bodies of ONE launch); inside a class the
        // heaviest panels come first (the hardware dispatches workgroups in index order, so the tail is made of the lightest).
        // Per-XCD panel lists (all panels of a basis block on one XCD, its packed operands L2-resident there: the L2 hit
        // rate of the operand requests is only 55-65 % without them) measured 3-5 % SLOWER with every generation of this
        // kernel: panels of one block then walk the same lines in step.
        std::vector<PanelItem> h_panels;
        std::vector<double> h_cost;
        for (int us : uorder) {
          const int u = us / nslices;
          for (size_t i = 0; i < by_u[us].size(); i += kPanel) {
            PanelItem pi{};
            pi.slice = us % nslices;
            for (int k = 0; k < kPanel; ++k) {
              const int b = i + k < by_u[us].size() ? by_u[us][i + k] : -1;
              pi.bl[k] = b;
              pi.coff[k] = b >= 0 ? h_grp_coff[grp_of_bl[b]] : 0;
              pi.ant[k] = b >= 0 ? make_int2(d->bl_ant0[b], d->bl_ant1[b]) : make_int2(0, 0);
            }
            pi.a_kf4 = okf4[u];
            pi.a_fk4 = 0;
            pi.nvec = d->basis_nvec[u];
            pi.nvp2 = nvp2[u];
            pi.nvp32 = nvp32[u];
            h_panels.push_back(pi);
            // a panel's time on a CU: a fixed part (prologue, element stage, epilogue) + its MFMA positions (stamps of the HERA-350 pass)
            h_cost.push_back(60e3 + 600.0 * (fpad / kChunk) * ((d->basis_nvec[u] + 7) / 8 + nvp32[u] / 32 * 4));
          }
        }
        CAL_TRY(order_panels(h_panels, h_cost));
        // one launch serves both panel classes (up to 4 / up to 8 vector tiles): the larger of their LDS footprints
        mf_lds_grad[0] = std::max(dense_lds_bytes(nvec_max, true, nvec_max > 128 ? 8 : 4), dense_lds_bytes(std::min(nvec_max, 128), true, 4));
        mf_lds_loss[0] = std::max(dense_lds_bytes(nvec_max, false, nvec_max > 128 ? 8 : 4), dense_lds_bytes(std::min(nvec_max, 128), false, 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_grad[0]));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_loss[0]));
        mf_ok = true;
      } else {
        // ---- double precision: v_mfma_f64_16x16x4_f64 (dense64_kernels.hpp).  Two panel classes (bodies of one launch): blocks
        // of more than 128 vectors with panels of 8 baselines (one column tile, 16 gradient tiles), the rest with panels of 16
        std::vector<long long> okf(nbasis + 1, 0);
        int nvec_a = 0, nvec_b = 0;
        for (int u = 0; u < nbasis; ++u) {
          const int nv = d->basis_nvec[u];
          (nv > 128 ? nvec_a : nvec_b) = std::max(nv > 128 ? nvec_a : nvec_b, nv);
          okf[u + 1] = okf[u] + (long long)(fpad / kCB64) * ((nv + 7) / 8 + (nv + kVT64 - 1) / kVT64 * 2) * 128;
        }
        CAL_TRY(mf_ops.alloc((size_t)okf[nbasis] * sizeof(double), false));
        for (int u = 0; u < nbasis; ++u)
          hipLaunchKernelGGL(mfma_pack64_kernel, dim3(grid_for(okf[u + 1] - okf[u])), dim3(256), 0, stream,
                             raw.as<double>() + d->basis_offset[u], mf_ops.as<double>() + okf[u], nfreqs, fpad, d->basis_nvec[u]);
        HIP_TRY(hipGetLastError());
        std::vector<std::vector<int>> by_u((size_t)nbasis * nslices);
        for (int b = 0; b < nbls; ++b) by_u[(size_t)d->grp_basis[grp_of_bl[b]] * nslices + grp_slice[grp_of_bl[b]]].push_back(b);
        std::vector<int> uorder((size_t)nbasis * nslices);
        std::iota(uorder.begin(), uorder.end(), 0);
        std::stable_sort(uorder.begin(), uorder.end(), [&](int a, int b) { return d->basis_nvec[a / nslices] > d->basis_nvec[b / nslices]; });
        std::vector<PanelItem> h_panels;
        std::vector<double> h_cost;
        for (int us : uorder) {
          const int u = us / nslices;
          const int width = d->basis_nvec[u] > 128 ? 8 : 16;
          for (size_t i = 0; i < by_u[us].size(); i += width) {
            PanelItem pi{};
            pi.slice = us % nslices;
            for (int k = 0; k < kPanel; ++k) {
              const int b = k < width && i + k < by_u[us].size() ? by_u[us][i + k] : -1;
              pi.bl[k] = b;
              pi.coff[k] = b >= 0 ? h_grp_coff[grp_of_bl[b]] : 0;
              pi.ant[k] = b >= 0 ? make_int2(d->bl_ant0[b], d->bl_ant1[b]) : make_int2(0, 0);
            }
            pi.a_kf4 = okf[u];
            pi.a_fk4 = 0;
            pi.nvec = d->basis_nvec[u];
            h_panels.push_back(pi);
            h_cost.push_back(60e3 + 300.0 * (width / 8) * (fpad / (4 * kCB64)) * ((d->basis_nvec[u] + 7) / 8 + (d->basis_nvec[u] + kVT64 - 1) / kVT64 * 2));
          }
        }
        CAL_TRY(order_panels(h_panels, h_cost));
        mf_lds_grad[0] = std::max(dense64_lds_bytes(nvec_a, 1, true), dense64_lds_bytes(nvec_b, 2, true));
        mf_lds_loss[0] = std::max(dense64_lds_bytes(nvec_a, 1, false), dense64_lds_bytes(nvec_b, 2, false));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_grad[0]));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_dense64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_lds_loss[0]));
        mf_ok = true;
      }
    }
    raw.release();
    CAL_TRY(bl_tile.alloc(nbls * sizeof(long long), false));
    HIP_TRY(copy_sync(bl_tile.p, h_bl_tile.data(), nbls * sizeof(long long), hipMemcpyHostToDevice));
    std::vector<int2> h_ant(nbls);
    for (int b = 0; b < nbls; ++b) h_ant[b] = make_int2(d->bl_ant0[b], d->bl_ant1[b]);
    CAL_TRY(bl_ant.alloc(nbls * sizeof(int2), false));
    HIP_TRY(copy_sync(bl_ant.p, h_ant.data(), nbls * sizeof(int2), hipMemcpyHostToDevice));

    // ---- runs of the multi-baseline groups: consecutive baselines with the same row block, cut to kRunMax
    std::vector<int2> h_runs;
    std::vector<int> h_grp_run0(ngrps + 1, 0);
    int nsimple_grps = 0;
    for (int g = 0; g < ngrps; ++g) {
      const int s0 = d->grp_bl_start[g], s1 = d->grp_bl_start[g + 1];
      h_grp_run0[g] = (int)h_runs.size();
      if (s1 - s0 == 1) {
        ++nsimple_grps;
        continue;
      }
      int lo = s0;
      for (int b = s0 + 1; b <= s1; ++b) {
        const bool cut = b == s1 || (d->bl_rowblk && d->bl_rowblk[b] != d->bl_rowblk[lo]) || b - lo == kRunMax;
        if (cut) {
          h_runs.push_back(make_int2(lo, b));
          lo = b;
        }
      }
    }
    h_grp_run0[ngrps] = (int)h_runs.size();
    CAL_TRY(runs.alloc(std::max<size_t>(1, h_runs.size()) * sizeof(int2), false));
    if (!h_runs.empty()) HIP_TRY(copy_sync(runs.p, h_runs.data(), h_runs.size() * sizeof(int2), hipMemcpyHostToDevice));

    // ---- work items: whole groups when that already fills the chip, otherwise split along tiles
    long long total_tiles = 0;
    for (int g = 0; g < ngrps; ++g)
      if (d->grp_bl_start[g + 1] - d->grp_bl_start[g] == 1) total_tiles += fpad / fb_u[d->grp_basis[g]];
    // one item per group when the groups alone fill the chip (256 CUs x ~4 resident workgroups, several waves of them);
    // otherwise split groups along their tiles (partial coefficient gradients are summed by coeff_partial_reduce_kernel)
    // With at least one group per CU an item is never smaller than 1024 channels of its group (8 tiles of the widest fp32
    // shape, 16 of the widest fp64 one): a group of up to 56 vectors is then ONE item -- no partial coefficient gradients,
    // no second launch to sum them -- and only wider groups (narrower tiles) are cut, which also evens the items out.
    // Measured at HERA-37 (666 groups): fp64 76 -> 61 us per step, fp32 45 -> 42; cutting every group into 4-tile items
    // (the earlier rule) bought parallelism the chip did not need and paid a prologue and a reduction for it.
    const long long target_items = 8192;
    const bool groups_fill_chip = nsimple_grps >= 2048;
    const long long min_tiles = nsimple_grps >= 256 ? 1024 / FbSet<T>::fb_max : 4;
    const long long tiles_per_item = groups_fill_chip ? std::max<long long>(64, 4 * total_tiles / std::max(1, nsimple_grps))
                                                      : std::max<long long>(min_tiles, total_tiles / target_items);
    std::vector<Item> h_items;
    std::vector<int> h_grp_item_ptr(ngrps + 1, 0);
    gc_direct = true;
    std::vector<long long> h_item_cost;
    std::vector<char> h_item_multi;
    lds_group_bytes = 0;
    for (int g = 0; g < ngrps; ++g) {
      const int u = d->grp_basis[g];
      const int ntpb = fpad / fb_u[u];
      int fl = 0;
      while ((1 << fl) < fb_u[u]) ++fl;
      Item it{};
      it.nvec = d->basis_nvec[u];
      it.coff = h_grp_coff[g];
      it.fb_log2 = fl;
      it.slice = grp_slice[g];
      if (d->grp_bl_start[g + 1] - d->grp_bl_start[g] == 1) {
        const long long nt = ntpb;
        // (a baseline that shares tiles stays ONE item: the multi kernel writes the whole coefficient gradient of its group)
        const int nparts = in_alias_set[d->grp_bl_start[g]] ? 1 : (int)std::max<long long>(1, (nt + tiles_per_item - 1) / tiles_per_item);
        if (nparts > 1) gc_direct = false;
        for (int p = 0; p < nparts; ++p) {
          it.bl0 = d->grp_bl_start[g];
          it.tile0 = (int)(nt * p / nparts);
          it.tile1 = (int)(nt * (p + 1) / nparts);
          it.tile_first = h_bl_tile[it.bl0];
          it.ant_first = make_int2(d->bl_ant0[it.bl0], d->bl_ant1[it.bl0]);
          h_items.push_back(it);
          h_item_cost.push_back((long long)(it.tile1 - it.tile0) * it.nvec * fb_u[u]);
          h_item_multi.push_back(0);
        }
      } else {
        // units = (run, channel block), run-major; a unit costs one tile (load, forward, adjoint: about 8 batches' worth)
        // plus one batch of the per-channel stage per kThreads / FB baselines.  Items are cut at ~96 batch equivalents
        // so that a big redundant set is spread over many workgroups (their partial coefficient gradients are summed).
        lds_group_bytes = std::max(lds_group_bytes, group_lds_for(fb_u[u]));
        const int bpt = kThreads / fb_u[u];
        const int r0 = h_grp_run0[g], r1 = h_grp_run0[g + 1];
        const long long budget = 96;
        long long cost = 0;
        int unit_lo = 0, nparts = 0;
        const int nunits = (r1 - r0) * ntpb;
        for (int uu = 0; uu < nunits; ++uu) {
          const int2 rn = h_runs[r0 + uu / ntpb];
          cost += 8 + (rn.y - rn.x + bpt - 1) / bpt;
          if (cost >= budget || uu + 1 == nunits) {
            it.bl0 = r0;
            it.tile0 = unit_lo;
            it.tile1 = uu + 1;
            h_items.push_back(it);
            h_item_cost.push_back(cost * 1024);  // same scale as nvec x FB of a full tile, roughly
            h_item_multi.push_back(1);
            unit_lo = uu + 1;
            cost = 0;
            ++nparts;
          }
        }
        if (nparts > 1) gc_direct = false;
      }
      h_grp_item_ptr[g + 1] = (int)h_items.size();
    }
    nitems = (int)h_items.size();
    small_loads = true;
    for (const Item& q : h_items) {
      const int lpr = (1 << q.fb_log2) / (16 / (int)sizeof(T));  // lanes per tile row; a load covers kThreads / lpr rows
      if (q.nvec > kSmallLoads * (kThreads / lpr)) small_loads = false;
    }
    // single-baseline items first, then the multi-baseline ones (two launches); inside each class heaviest first: the
    // hardware dispatches workgroups in index order, so the tail is made of the lightest items
    // (among the single-baseline items those that a head item of the multi-slice kernels covers come last: the loss and gradient
    // passes launch fused_basis_kernel over the plain ones only)
    std::vector<std::vector<int>> alias_sets(nbls);
    for (int b = 0; b < nbls; ++b)
      if (in_alias_set[b]) alias_sets[alias_root[b] >= 0 ? alias_root[b] : b].push_back(b);
    // members per head item: the matrix-core kernel takes 8 in both precisions (16 MFMA columns), fused_multi_kernel
    // MultiCfg<T>::nb_max (its gradient accumulators live in registers: 4 in fp64)
    auto mfma_shape = [&](const Item& q) { return q.nvec <= kMmMaxVec && (1 << q.fb_log2) >= kMmStrip && fpad % 128 == 0; };
    std::vector<int> set_cap(nbls, MultiCfg<T>::nb_max);
    for (int q = 0; q < nitems; ++q)
      if (!h_item_multi[q] && mfma_shape(h_items[q])) set_cap[h_items[q].bl0] = kMmMembers;
    std::vector<char> bl_covered(nbls, 0);
    for (int r = 0; r < nbls; ++r)
      for (size_t i = 0; i < alias_sets[r].size(); i += set_cap[r]) {
        const size_t n = std::min<size_t>(set_cap[r], alias_sets[r].size() - i);
        if (n >= 2)
          for (size_t k = 0; k < n; ++k) bl_covered[alias_sets[r][i + k]] = 1;
      }
    auto item_class = [&](int q) { return h_item_multi[q] ? 2 : (bl_covered[h_items[q].bl0] ? 1 : 0); };
    std::vector<int> order(nitems);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (item_class(a) != item_class(b)) return item_class(a) < item_class(b);
      return h_item_cost[a] > h_item_cost[b];
    });
    nitems_simple = 0;
    nitems_plain = 0;
    for (int q = 0; q < nitems; ++q) {
      nitems_simple += h_item_multi[q] ? 0 : 1;
      nitems_plain += item_class(q) == 0 ? 1 : 0;
    }
    std::vector<int> h_item_goff(nitems);
    gcp_len = 0;
    if (gc_direct) {
      for (int q = 0; q < nitems; ++q) h_item_goff[q] = h_items[q].coff;
      gcp_len = ncoef;
    } else {
      for (int q = 0; q < nitems; ++q) {
        h_item_goff[q] = (int)gcp_len;
        gcp_len += h_items[q].nvec;
      }
    }
    std::vector<Item> sorted(nitems);
    for (int q = 0; q < nitems; ++q) {
      sorted[q] = h_items[order[q]];
      sorted[q].goff = h_item_goff[order[q]];
      sorted[q].role_n = 0;
      sorted[q].member0 = 0;
    }
    // the loss partials (one per item) of every time slice, in item order
    if (nslices > 1) {
      std::vector<int> ptr(nslices + 1, 0), idx(nitems);
      for (int q = 0; q < nitems; ++q) ptr[sorted[q].slice + 1]++;
      for (int t = 0; t < nslices; ++t) ptr[t + 1] += ptr[t];
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (int q = 0; q < nitems; ++q) idx[fill[sorted[q].slice]++] = q;
      CAL_TRY(slice_ipart_ptr.alloc(ptr.size() * sizeof(int), false));
      HIP_TRY(copy_sync(slice_ipart_ptr.p, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(slice_ipart_idx.alloc(idx.size() * sizeof(int), false));
      HIP_TRY(copy_sync(slice_ipart_idx.p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // coefficient blocks of step_tail_kernel: every slice gets its own (a block works for ONE slice's decisions)
    h_slice_cblk.assign(nslices + 1, 0);
    for (int t = 0; t < nslices; ++t) {
      const long long nb = (2LL * (h_slice_coff[t + 1] - h_slice_coff[t]) + 255) / 256;
      h_slice_cblk[t + 1] = h_slice_cblk[t] + (int)std::max<long long>(1, std::min<long long>(nb, std::max(1, 4096 / nslices)));
    }
    if (nslices > 1) {
      CAL_TRY(slice_cblk.alloc((nslices + 1) * sizeof(int), false));
      HIP_TRY(copy_sync(slice_cblk.p, h_slice_cblk.data(), (nslices + 1) * sizeof(int), hipMemcpyHostToDevice));
    }
    // ---- sets of baselines that share tiles -> head items with member lists (at most MultiCfg<T>::nb_max baselines each)
    nheads = 0;
    nheads_mfma = 0;
    lds_multi_bytes = 0;
    lds_multi_mfma_bytes = 0;
    {
      std::vector<int> item_of_bl(nbls, -1);
      for (int q = 0; q < nitems; ++q)
        if (!h_item_multi[order[q]]) item_of_bl[sorted[q].bl0] = q;
      const std::vector<std::vector<int>>& sets = alias_sets;
      std::vector<Member> h_members;
      std::vector<int> h_heads;
      for (int r = 0; r < nbls; ++r) {
        const int NBM = set_cap[r];
        for (size_t i = 0; i < sets[r].size(); i += NBM) {
          const int n = (int)std::min<size_t>(NBM, sets[r].size() - i);
          if (n < 2) continue;  // a lone baseline runs as an ordinary item
          const int head = item_of_bl[sets[r][i]];
          sorted[head].role_n = (n << 2) | 1;
          sorted[head].member0 = (int)h_members.size();
          h_heads.push_back(head);
          for (int k = 0; k < n; ++k) {
            const int b = sets[r][i + k], q = item_of_bl[b];
            if (k > 0) sorted[q].role_n = 2;
            Member m{};
            m.bl = b;
            m.coff = sorted[q].coff;
            m.goff = sorted[q].goff;
            m.ant0 = d->bl_ant0[b];
            m.ant1 = d->bl_ant1[b];
            m.slice = sorted[q].slice;
            m.item = q;
            h_members.push_back(m);
          }
        }
      }
      // the matrix-core form first (blocks of at most kMmMaxVec vectors), each list heaviest first
      // (rows padded to a multiple of 128 channels -- any band of more than 64: its waves take an even number of 16-channel strips each)
      auto on_mfma = [&](int head) { return mfma_shape(sorted[head]); };
      std::stable_sort(h_heads.begin(), h_heads.end(), [&](int a, int b) {
        const bool ma = on_mfma(a), mb = on_mfma(b);
        if (ma != mb) return ma;
        return (long long)sorted[a].nvec * (sorted[a].role_n >> 2) > (long long)sorted[b].nvec * (sorted[b].role_n >> 2);
      });
      for (int head : h_heads) {
        if (on_mfma(head)) {
          ++nheads_mfma;
          lds_multi_mfma_bytes = std::max(lds_multi_mfma_bytes, multi_mfma_lds_bytes<T>(sorted[head].nvec));
        } else {
          lds_multi_bytes = std::max(lds_multi_bytes, multi_lds_for(1 << sorted[head].fb_log2));
        }
      }
      nheads = (int)h_heads.size();
      // the "sum" regulariser over heads: one pass with two adjoint sets when every head is on the matrix-core kernel and narrow enough
      // for it (multi_mfma_kernels.hpp, REG == 2); else a loss pass for the slices' sums in front of the gradient pass (enqueue_pass)
      heads_one_pass = nheads > 0 && nheads == nheads_mfma;
      for (int head : h_heads) heads_one_pass = heads_one_pass && sorted[head].nvec <= kMmMaxVecOnePass<T>;
      // XCD-affine, antenna-grouped dispatch of the matrix-core heads: a head reads 2 gain rows per member (8 slices x 2 x 8 KB of a
      // 1024-channel band) -- a fifth of its bytes, 1.0 GB per pass of an 8-GPU rank's share against 23 MB of distinct gains, because
      // with the heads in cost order nothing a workgroup brings into its XCD's L2 is wanted by its neighbours (hit rate 17 %).
      // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one): every first antenna -- all heads whose baseline
      // starts at it -- goes to ONE of 8 lists of equal cost (longest-processing-time-first), heaviest antenna groups first in a
      // list, heaviest head first in a group, and workgroup b takes entry b / 8 of list b % 8.  The ant0 rows of a group then stay
      // in that XCD's L2 for the whole group.  (-1 where a list is shorter.)
      mm_grid = nheads_mfma;
      if (nheads_mfma >= 64) {
        auto cost = [&](int h) { return (double)sorted[h].nvec * (sorted[h].role_n >> 2) + 64.0; };
        std::map<int, std::vector<int>> by_ant;
        for (int i = 0; i < nheads_mfma; ++i) by_ant[sorted[h_heads[i]].ant_first.x].push_back(h_heads[i]);  // (already heaviest first)
        std::vector<std::pair<double, int>> groups;
        for (auto& kv : by_ant) {
          double c = 0;
          for (int h : kv.second) c += cost(h);
          groups.push_back({c, kv.first});
        }
        std::stable_sort(groups.begin(), groups.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
        double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<std::vector<int>> lists(8);
        for (auto& g : groups) {
          const int x = (int)(std::min_element(load, load + 8) - load);
          load[x] += g.first;
          for (int h : by_ant[g.second]) lists[x].push_back(h);
        }
        size_t longest = 0;
        for (auto& l : lists) longest = std::max(longest, l.size());
        std::vector<int> dealt(8 * longest, -1);
        for (int x = 0; x < 8; ++x)
          for (size_t j = 0; j < lists[x].size(); ++j) dealt[j * 8 + x] = lists[x][j];
        dealt.insert(dealt.end(), h_heads.begin() + nheads_mfma, h_heads.end());
        mm_grid = (int)(8 * longest);
        h_heads.swap(dealt);
      }
      members.release();
      heads.release();
      if (nheads > 0) {
        CAL_TRY(members.alloc(h_members.size() * sizeof(Member), false));
        HIP_TRY(copy_sync(members.p, h_members.data(), h_members.size() * sizeof(Member), hipMemcpyHostToDevice));
        CAL_TRY(heads.alloc(h_heads.size() * sizeof(int), false));
        HIP_TRY(copy_sync(heads.p, h_heads.data(), h_heads.size() * sizeof(int), hipMemcpyHostToDevice));
        if (nheads > nheads_mfma) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_kernel<T, MODE_GRAD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_kernel<T, MODE_LOSS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_kernel<T, MODE_GRAD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_kernel<T, MODE_LOSS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_bytes));
        }
        if (nheads_mfma > 0) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_mfma_kernel<T, MODE_GRAD, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_mfma_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_mfma_kernel<T, MODE_LOSS, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_mfma_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_mfma_kernel<T, MODE_GRAD, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_mfma_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_mfma_kernel<T, MODE_LOSS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_mfma_bytes));
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_multi_mfma_kernel<T, MODE_GRAD, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi_mfma_bytes));
        }
      }
    }
    CAL_TRY(items.alloc(nitems * sizeof(Item), false));
    HIP_TRY(copy_sync(items.p, sorted.data(), nitems * sizeof(Item), hipMemcpyHostToDevice));
    if (!gc_direct) {
      std::vector<int> h_coef_grp(ncoef);
      for (int g = 0; g < ngrps; ++g)
        for (int n = h_grp_coff[g]; n < h_grp_coff[g + 1]; ++n) h_coef_grp[n] = g;
      CAL_TRY(coef_grp.alloc(ncoef * sizeof(int), false));
      HIP_TRY(copy_sync(coef_grp.p, h_coef_grp.data(), ncoef * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(grp_coff.alloc((ngrps + 1) * sizeof(int), false));
      HIP_TRY(copy_sync(grp_coff.p, h_grp_coff.data(), (ngrps + 1) * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(grp_item_ptr.alloc((ngrps + 1) * sizeof(int), false));
      HIP_TRY(copy_sync(grp_item_ptr.p, h_grp_item_ptr.data(), (ngrps + 1) * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(item_goff.alloc(nitems * sizeof(int), false));
      HIP_TRY(copy_sync(item_goff.p, h_item_goff.data(), nitems * sizeof(int), hipMemcpyHostToDevice));
      CAL_TRY(gc0.alloc(2 * (size_t)ncoef * sizeof(T)));
    }

    // ---- per-antenna CSR of (baseline, role, other antenna), in baseline order: a fixed summation order
    std::vector<int> h_ant_ptr(nants + 1, 0);
    for (int b = 0; b < nbls; ++b) {
      h_ant_ptr[d->bl_ant0[b] + 1]++;
      h_ant_ptr[d->bl_ant1[b] + 1]++;
    }
    for (int a = 0; a < nants; ++a) h_ant_ptr[a + 1] += h_ant_ptr[a];
    std::vector<int2> h_ent(2 * (size_t)nbls);
    std::vector<int> fill(h_ant_ptr.begin(), h_ant_ptr.end() - 1);
    for (int b = 0; b < nbls; ++b) {
      h_ent[fill[d->bl_ant0[b]]++] = make_int2(b * 2 + 0, d->bl_ant1[b]);
      h_ent[fill[d->bl_ant1[b]]++] = make_int2(b * 2 + 1, d->bl_ant0[b]);
    }
    CAL_TRY(ant_ptr.alloc((nants + 1) * sizeof(int), false));
    HIP_TRY(copy_sync(ant_ptr.p, h_ant_ptr.data(), (nants + 1) * sizeof(int), hipMemcpyHostToDevice));
    CAL_TRY(ant_ent.alloc(h_ent.size() * sizeof(int2), false));
    HIP_TRY(copy_sync(ant_ent.p, h_ent.data(), h_ent.size() * sizeof(int2), hipMemcpyHostToDevice));

    // ---- state arrays
    const size_t rowbytes = (size_t)(nbls + 1) * fpad * sizeof(T);  // + one all-zero spare row (padding slots of the dense path)
    CAL_TRY(data_r.alloc(rowbytes));
    CAL_TRY(data_i.alloc(rowbytes));
    CAL_TRY(wgts.alloc(rowbytes));
    const size_t gbytes = (size_t)nants * fpad * sizeof(T2);
    CAL_TRY(gains.alloc(gbytes));
    CAL_TRY(gains_m.alloc(gbytes));
    CAL_TRY(gains_v.alloc(gbytes));
    gains_snap.release();
    gains_alt.release();
    const size_t cbytes = 2 * (size_t)ncoef * sizeof(T);
    CAL_TRY(coef.alloc(cbytes + 256));  // (the split-bf16 kernel reads whole pairs of 16-vector steps: up to 31 reals past the last group's coefficients)
    CAL_TRY(coef_m.alloc(cbytes));
    CAL_TRY(coef_v.alloc(cbytes));
    coef_snap.release();
    CAL_TRY(q0.alloc((size_t)(nbls + 1) * fpad * sizeof(T2)));
    q1.release();
    CAL_TRY(comm.alloc(3 * gbytes));
    CAL_TRY(gcp0.alloc(2 * (size_t)gcp_len * sizeof(T)));
    gcp1.release();
    gc1.release();
    CAL_TRY(part.alloc((size_t)std::max(nitems, mf_npanels) * 4 * sizeof(double)));
    model_buf.release();
    scratch.release();
    has_problem = true;
    reg = CAL_REG_NONE;
    // ~ tens of milliseconds of GPU time between two host synchronisations of run(); the same on every rank, or ranks
    // would notice a tolerance stop after different step counts and issue different numbers of all-reduces
    steps_per_sync = (int)std::max(1.0, std::min(256.0, 2.0e11 / ((double)basis_bytes + 1.0)));
    return CAL_OK;
  }

  // XCD-affine dispatch of the dense launch (dense_kernels.hpp: fused_dense_kernel): every basis block -- all its panels -- goes
  // to ONE of 8 lists, blocks dealt longest-processing-time first so the lists carry equal cost; inside a list the heaviest
  // panels come first (the tail of the pass is made of the lightest).  Workgroup b takes entry b / 8 of list b % 8: the slot
  // map interleaves the lists, -1 where a list is shorter than the longest.  Uploads the records and the map.
  // `per` consecutive panel records form one work item (split-bf16 kernel: a super-panel); cost and the map count items.
  int order_panels(std::vector<PanelItem>& panels, const std::vector<double>& cost, int per = 1) {
    const int n = (int)panels.size() / per;
    std::vector<long long> keys;
    std::vector<double> kcost;
    std::vector<int> blk(n);
    for (int i = 0; i < n; ++i) {
      size_t k = std::find(keys.begin(), keys.end(), panels[(size_t)i * per].a_kf4) - keys.begin();
      if (k == keys.size()) { keys.push_back(panels[i].a_kf4); kcost.push_back(0.0); }
      kcost[k] += cost[i];
      blk[i] = (int)k;
    }
    std::vector<int> border(keys.size());
    std::iota(border.begin(), border.end(), 0);
    std::stable_sort(border.begin(), border.end(), [&](int a, int b) { return kcost[a] > kcost[b]; });
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<int> list_of(keys.size());
    for (int k : border) {
      const int x = (int)(std::min_element(load, load + 8) - load);
      list_of[k] = x;
      load[x] += kcost[k];
    }
    std::vector<std::vector<int>> lists(8);
    for (int i = 0; i < n; ++i) lists[list_of[blk[i]]].push_back(i);
    size_t longest = 0;
    for (auto& l : lists) {
      std::stable_sort(l.begin(), l.end(), [&](int a, int b) { return cost[a] > cost[b]; });
      longest = std::max(longest, l.size());
    }
    std::vector<int> h_map(8 * longest, -1);
    for (int x = 0; x < 8; ++x)
      for (size_t j = 0; j < lists[x].size(); ++j) h_map[j * 8 + x] = lists[x][j];
    const int np = (int)panels.size();
    mf_npanels = np;
    mf_grid = (int)h_map.size();
    if (nslices > 1) {  // the loss partials (one per panel) of every time slice, in panel order
      std::vector<int> ptr(nslices + 1, 0), idx(np);
      for (int i = 0; i < np; ++i) ptr[panels[i].slice + 1]++;
      for (int t = 0; t < nslices; ++t) ptr[t + 1] += ptr[t];
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (int i = 0; i < np; ++i) idx[fill[panels[i].slice]++] = i;
      CAL_TRY(slice_ppart_ptr.alloc(ptr.size() * sizeof(int), false));
      HIP_TRY(hipMemcpyAsync(slice_ppart_ptr.p, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice, stream));
      CAL_TRY(slice_ppart_idx.alloc(idx.size() * sizeof(int), false));
      HIP_TRY(hipMemcpyAsync(slice_ppart_idx.p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipStreamSynchronize(stream));  // (the vectors leave scope)
    }
    CAL_TRY(mf_panels.alloc((size_t)np * sizeof(PanelItem), false));
    HIP_TRY(hipMemcpyAsync(mf_panels.p, panels.data(), (size_t)np * sizeof(PanelItem), hipMemcpyHostToDevice, stream));
    CAL_TRY(mf_map.alloc(h_map.size() * sizeof(int), false));
    HIP_TRY(hipMemcpyAsync(mf_map.p, h_map.data(), h_map.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }

  // min over the ranks of the communicator (set-up decisions that every rank must take identically)
  // one in-place all-reduce over the ranks of the job, on the solver's stream: RCCL, or the caller's exchange hook (the
  // stream is drained, the buffer staged through pinned host memory, reduced by the callback and copied back)
  int all_reduce(void* dev, size_t count, int dtype_code, int op) {
    if (nccl) {
      const ncclDataType_t dt = dtype_code == CAL_XCHG_F32 ? ncclFloat : dtype_code == CAL_XCHG_F64 ? ncclDouble : ncclInt32;
      NCCL_TRY(ncclAllReduce(dev, dev, count, dt, op == CAL_XCHG_MIN ? ncclMin : ncclSum, nccl, stream));
      return CAL_OK;
    }
    if (!hook) return CAL_OK;
    const size_t bytes = count * (dtype_code == CAL_XCHG_F64 ? 8 : 4);
    if (hook_bytes < bytes) {
      if (hook_buf) (void)hipHostFree(hook_buf);
      hook_buf = nullptr;
      hook_bytes = 0;
      HIP_TRY(hipHostMalloc(&hook_buf, bytes, hipHostMallocDefault));
      hook_bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(hook_buf, dev, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const int rc = hook(hook_ctx, hook_buf, (int64_t)count, dtype_code, op);
    if (rc != 0) return fail(CAL_ERR_RCCL, "the exchange hook returned %d", rc);
    HIP_TRY(hipMemcpyAsync(dev, hook_buf, bytes, hipMemcpyHostToDevice, stream));
    return CAL_OK;
  }
  int agree_min(int* v, int n) {
    if (!comm_on()) return CAL_OK;
    if (!agree_buf.p) CAL_TRY(agree_buf.alloc(4 * sizeof(int)));
    HIP_TRY(hipMemcpyAsync(agree_buf.p, v, n * sizeof(int), hipMemcpyHostToDevice, stream));
    CAL_TRY(all_reduce(agree_buf.p, n, CAL_XCHG_I32, CAL_XCHG_MIN));
    HIP_TRY(hipMemcpyAsync(v, agree_buf.p, n * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }

  // ---- padded copies ------------------------------------------------------------------------------------------
  int ensure_scratch(size_t bytes) {
    if (scratch.bytes < bytes) CAL_TRY(scratch.alloc(bytes, false));
    return CAL_OK;
  }
  // host [rows][nfreqs] -> device [rows][fpad] * stride + off
  int upload_rows(const void* src, T* dst, long long rows, int stride, int off) {
    const size_t bytes = (size_t)rows * nfreqs * sizeof(T);
    if (stride == 1 && fpad == nfreqs) {
      HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      return CAL_OK;
    }
    CAL_TRY(ensure_scratch(bytes));
    HIP_TRY(hipMemcpyAsync(scratch.p, src, bytes, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(pad_rows_kernel<T>, dim3(grid_for(rows * fpad)), dim3(256), 0, stream, scratch.as<T>(), dst, rows, nfreqs, fpad,
                       stride, off);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }
  int download_rows(void* dst, const T* src, long long rows, int stride, int off) {
    const size_t bytes = (size_t)rows * nfreqs * sizeof(T);
    if (stride == 1 && fpad == nfreqs) {
      HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      return CAL_OK;
    }
    CAL_TRY(ensure_scratch(bytes));
    hipLaunchKernelGGL(unpad_rows_kernel<T>, dim3(grid_for(rows * nfreqs)), dim3(256), 0, stream, src, scratch.as<T>(), rows, nfreqs,
                       fpad, stride, off);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst, scratch.p, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }

  int set_data(const void* dr, const void* di, const void* w) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "set_data before set_problem");
    if (!dr || !di || !w) return fail(CAL_ERR_INVALID, "set_data: null pointer");
    CAL_TRY(upload_rows(dr, data_r.as<T>(), nbls, 1, 0));
    CAL_TRY(upload_rows(di, data_i.as<T>(), nbls, 1, 0));
    CAL_TRY(upload_rows(w, wgts.as<T>(), nbls, 1, 0));
    has_data = true;
    return CAL_OK;
  }

  int set_regularization(int mode, const double* pr, const double* pi, bool per_slice) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "set_regularization before set_problem");
    if (mode != CAL_REG_NONE && mode != CAL_REG_SUM) return fail(CAL_ERR_INVALID, "set_regularization: unknown mode %d", mode);
    if (!pr || !pi) return fail(CAL_ERR_INVALID, "set_regularization: null priors");
    if (mode != reg && nheads > 0) HIP_TRY(hipMemsetAsync(part.p, 0, part.bytes, stream));  // member items write their slot only in the regularised form
    if (mode != reg) drop_graph();
    reg = mode;
    for (int t = 0; t < nslices; ++t) {
      prior_r_t[t] = pr[per_slice ? t : 0];
      prior_i_t[t] = pi[per_slice ? t : 0];
    }
    if (reg == CAL_REG_SUM) {
      if (!q1.p) CAL_TRY(q1.alloc((size_t)(nbls + 1) * fpad * sizeof(T2)));
      if (!gcp1.p) CAL_TRY(gcp1.alloc(2 * (size_t)gcp_len * sizeof(T)));
      if (!gc_direct && !gc1.p) CAL_TRY(gc1.alloc(2 * (size_t)ncoef * sizeof(T)));
    }
    return CAL_OK;
  }
  int slice_count() override { return nslices; }
  int get_slice_losses(double* out) override {
    if (!has_problem) return fail(CAL_ERR_STATE, "get_slice_losses before set_problem");
    if (!out) return fail(CAL_ERR_INVALID, "get_slice_losses: null");
    for (int t = 0; t < nslices; ++t) out[t] = h_state[t].loss;
    return CAL_OK;
  }

  int set_optimizer(const cal_optimizer_desc* d) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "set_optimizer before set_problem");
    if (!d) return fail(CAL_ERR_INVALID, "set_optimizer: null");
    if (d->optimizer < CAL_OPT_ADAM || d->optimizer > CAL_OPT_LAMB)
      return fail(CAL_ERR_INVALID, "set_optimizer: unknown optimizer id %d", d->optimizer);
    if (d->optimizer == CAL_OPT_LAMB && !lamb_ok)
      return fail(CAL_ERR_UNSUPPORTED, "set_optimizer: LAMB takes one trust ratio per variable; the groups of a variable (cal_problem_desc::grp_var) must be "
                  "contiguous inside a time slice");
    opt = *d;
    HIP_TRY(hipMemsetAsync(gains_m.p, 0, gains_m.bytes, stream));
    HIP_TRY(hipMemsetAsync(coef_m.p, 0, coef_m.bytes, stream));
    if ((d->optimizer == CAL_OPT_ADAGRAD || d->optimizer == CAL_OPT_FTRL) && d->initial_accumulator_value != 0.0) {
      // Adagrad's / Ftrl's accumulator starts at initial_accumulator_value (Keras default 0.1)
      hipLaunchKernelGGL(fill_kernel<T>, dim3(grid_for((long long)(gains_v.bytes / sizeof(T)))), dim3(256), 0, stream, gains_v.as<T>(),
                         (long long)(gains_v.bytes / sizeof(T)), (T)d->initial_accumulator_value);
      hipLaunchKernelGGL(fill_kernel<T>, dim3(grid_for((long long)(coef_v.bytes / sizeof(T)))), dim3(256), 0, stream, coef_v.as<T>(),
                         (long long)(coef_v.bytes / sizeof(T)), (T)d->initial_accumulator_value);
      HIP_TRY(hipGetLastError());
    } else {
      HIP_TRY(hipMemsetAsync(gains_v.p, 0, gains_v.bytes, stream));
      HIP_TRY(hipMemsetAsync(coef_v.p, 0, coef_v.bytes, stream));
    }
    drop_graph();
    for (int t = 0; t < nslices; ++t) reset_loop_state(h_state[t]);  // a new fit begins
    has_opt = true;
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }

  int set_params(const void* g_r, const void* g_i, const void* c_r, const void* c_i) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "set_params before set_problem");
    if (g_r) CAL_TRY(upload_rows(g_r, gains.as<T>(), nants, 2, 0));
    if (g_i) CAL_TRY(upload_rows(g_i, gains.as<T>(), nants, 2, 1));
    if (c_r) HIP_TRY(copy_sync(coef.as<T>(), c_r, (size_t)ncoef * sizeof(T), hipMemcpyHostToDevice));
    if (c_i) HIP_TRY(copy_sync(coef.as<T>() + ncoef, c_i, (size_t)ncoef * sizeof(T), hipMemcpyHostToDevice));
    if (g_r && g_i) has_gains = true;
    if (c_r && c_i) has_coef = true;
    return CAL_OK;
  }

  int get_params(int which, void* g_r, void* g_i, void* c_r, void* c_i) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "get_params before set_problem");
    const T* g = gains.as<T>();
    const T* c = coef.as<T>();
    if (which == 1) {
      if (!gains_snap.p) return fail(CAL_ERR_STATE, "get_params(which=1): no use_min snapshot was taken");
      g = gains_snap.as<T>();
      c = coef_snap.as<T>();
    } else if (which != 0) {
      return fail(CAL_ERR_INVALID, "get_params: which must be 0 or 1");
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (g_r) CAL_TRY(download_rows(g_r, g, nants, 2, 0));
    if (g_i) CAL_TRY(download_rows(g_i, g, nants, 2, 1));
    if (c_r) HIP_TRY(copy_sync(c_r, c, (size_t)ncoef * sizeof(T), hipMemcpyDeviceToHost));
    if (c_i) HIP_TRY(copy_sync(c_i, c + ncoef, (size_t)ncoef * sizeof(T), hipMemcpyDeviceToHost));
    return CAL_OK;
  }

  int get_moments(void* gm_r, void* gm_i, void* gv_r, void* gv_i, void* cm_r, void* cm_i, void* cv_r, void* cv_i,
                  int64_t* t) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "get_moments before set_problem");
    HIP_TRY(hipStreamSynchronize(stream));
    if (t && nslices > 1) {
      // every slice counts its own updates (slices stop on their own and then freeze): ONE t only describes a solver whose slices agree
      for (int sl = 1; sl < nslices; ++sl)
        if (h_state[sl].t != h_state[0].t)
          return fail(CAL_ERR_STATE, "get_moments: the %d time slices of this solver have applied different numbers of updates (slice 0: %lld, slice %d: %lld); "
                      "one iteration count cannot resume them bit for bit", nslices, (long long)h_state[0].t, sl, (long long)h_state[sl].t);
    }
    if (gm_r) CAL_TRY(download_rows(gm_r, gains_m.as<T>(), nants, 2, 0));
    if (gm_i) CAL_TRY(download_rows(gm_i, gains_m.as<T>(), nants, 2, 1));
    if (gv_r) CAL_TRY(download_rows(gv_r, gains_v.as<T>(), nants, 2, 0));
    if (gv_i) CAL_TRY(download_rows(gv_i, gains_v.as<T>(), nants, 2, 1));
    const size_t cb = (size_t)ncoef * sizeof(T);
    if (cm_r) HIP_TRY(copy_sync(cm_r, coef_m.as<T>(), cb, hipMemcpyDeviceToHost));
    if (cm_i) HIP_TRY(copy_sync(cm_i, coef_m.as<T>() + ncoef, cb, hipMemcpyDeviceToHost));
    if (cv_r) HIP_TRY(copy_sync(cv_r, coef_v.as<T>(), cb, hipMemcpyDeviceToHost));
    if (cv_i) HIP_TRY(copy_sync(cv_i, coef_v.as<T>() + ncoef, cb, hipMemcpyDeviceToHost));
    if (t) *t = h_state->t;
    return CAL_OK;
  }

  int set_moments(const void* gm_r, const void* gm_i, const void* gv_r, const void* gv_i, const void* cm_r, const void* cm_i,
                  const void* cv_r, const void* cv_i, int64_t t) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem) return fail(CAL_ERR_STATE, "set_moments before set_problem");
    if (t < 0) return fail(CAL_ERR_INVALID, "set_moments: negative iteration count");
    if (gm_r) CAL_TRY(upload_rows(gm_r, gains_m.as<T>(), nants, 2, 0));
    if (gm_i) CAL_TRY(upload_rows(gm_i, gains_m.as<T>(), nants, 2, 1));
    if (gv_r) CAL_TRY(upload_rows(gv_r, gains_v.as<T>(), nants, 2, 0));
    if (gv_i) CAL_TRY(upload_rows(gv_i, gains_v.as<T>(), nants, 2, 1));
    const size_t cb = (size_t)ncoef * sizeof(T);
    if (cm_r) HIP_TRY(copy_sync(coef_m.as<T>(), cm_r, cb, hipMemcpyHostToDevice));
    if (cm_i) HIP_TRY(copy_sync(coef_m.as<T>() + ncoef, cm_i, cb, hipMemcpyHostToDevice));
    if (cv_r) HIP_TRY(copy_sync(coef_v.as<T>(), cv_r, cb, hipMemcpyHostToDevice));
    if (cv_i) HIP_TRY(copy_sync(coef_v.as<T>() + ncoef, cv_i, cb, hipMemcpyHostToDevice));
    // beta^t as the device keeps it: a running product, one factor per update (pow() differs from it in the last bits, and a
    // resumed fit must continue bit for bit).  Uses the betas of the optimizer set so far: set_optimizer comes first.
    double b1t = 1.0, b2t = 1.0, sched = 1.0;
    for (int64_t k = 0; k < t; ++k) {
      b1t *= opt.beta_1;
      b2t *= opt.beta_2;
    }
    if (opt.optimizer == CAL_OPT_NADAM && t > 0) {
      // Nadam's momentum schedule is a running product of pow() terms: rebuilt by the DEVICE's pow (advance_state's), whose last
      // bits differ from the host's -- a resumed fit continues bit for bit
      hipLaunchKernelGGL(nadam_sched_kernel, dim3(1), dim3(1), 0, stream, opt.beta_1, (long long)t, scal.as<double>());
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&sched, scal.p, sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    for (int sl = 0; sl < nslices; ++sl) {
      h_state[sl].t = t;
      h_state[sl].b1t = b1t;
      h_state[sl].b2t = b2t;
      h_state[sl].nadam_sched = sched;
    }
    return CAL_OK;
  }

  // ---- one pass of the hot path -------------------------------------------------------------------------------
  FusedArgs<T> fused_args() {
    FusedArgs<T> a{};
    a.tiles = tiles.as<T>();
    a.bl_tile = bl_tile.as<long long>();
    a.bl_ant = bl_ant.as<int2>();
    a.data_r = data_r.as<T>();
    a.data_i = data_i.as<T>();
    a.wgts = wgts.as<T>();
    a.gains = gains.as<T2>();
    a.c_r = coef.as<T>();
    a.c_i = coef.as<T>() + ncoef;
    a.items = items.as<Item>();
    a.q0 = q0.as<T2>();
    a.q1 = q1.as<T2>();
    a.gcp0_r = gcp0.as<T>();
    a.gcp0_i = gcp0.as<T>() + gcp_len;
    a.gcp1_r = gcp1.as<T>();
    a.gcp1_i = gcp1.p ? gcp1.as<T>() + gcp_len : nullptr;
    a.part = part.as<double>();
    a.model_r = nullptr;
    a.model_i = nullptr;
    a.state = st_cur();
    a.nslices = nslices;
    a.fpad = fpad;
    a.nbls = nbls;
    a.runs = runs.as<int2>();
    a.item_base = 0;
    a.stream_once = layout == CAL_LAYOUT_STREAM ? 1 : 0;
    a.members = members.as<Member>();
    a.heads = nheads > 0 ? heads.as<int>() : nullptr;
    return a;
  }
  // LDS buffer of gbar_G rows (MODE_GRAD); the narrowest tiles have the longest offset table
  static constexpr size_t kQLdsMax1 = TileCfg<T, FbSet<T>::fb_min>::q_lds_bytes(false);
  static constexpr size_t kQLdsMax2 = TileCfg<T, FbSet<T>::fb_min>::q_lds_bytes(true);
  template <int MODE> void launch_fused(const FusedArgs<T>& a0, bool with_reg) {
    FusedArgs<T> a = a0;
    // the passes with a multi-slice form leave the covered items to it (fused_basis_kernel would return at once for each of them)
    const int nsimple = ((MODE == MODE_LOSS || MODE == MODE_GRAD) && nheads > 0) ? nitems_plain : nitems_simple;
    if (nsimple > 0) {
      a.item_base = 0;
      constexpr bool kHasSmall = MODE == MODE_LOSS || MODE == MODE_GRAD;
      if (kHasSmall && small_loads) {
        if constexpr (kHasSmall) {
          if (with_reg)
            hipLaunchKernelGGL((fused_basis_kernel<T, MODE, true, kSmallLoads>), dim3(nsimple), dim3(kThreads), lds_bytes + (MODE == MODE_GRAD ? kQLdsMax2 : 0), stream, a);
          else
            hipLaunchKernelGGL((fused_basis_kernel<T, MODE, false, kSmallLoads>), dim3(nsimple), dim3(kThreads), lds_bytes + (MODE == MODE_GRAD ? kQLdsMax1 : 0), stream, a);
        }
      } else if (with_reg)
        hipLaunchKernelGGL((fused_basis_kernel<T, MODE, true>), dim3(nsimple), dim3(kThreads), lds_bytes + (MODE == MODE_GRAD ? kQLdsMax2 : 0), stream, a);
      else
        hipLaunchKernelGGL((fused_basis_kernel<T, MODE, false>), dim3(nsimple), dim3(kThreads), lds_bytes + (MODE == MODE_GRAD ? kQLdsMax1 : 0), stream, a);
    }
    if constexpr (MODE == MODE_LOSS || MODE == MODE_GRAD) {
      // baselines that share tiles (skipped by the launch above): one workgroup per set.  With the regulariser these kernels take
      // the two-pass form (loss pass: S; gradient pass: alpha of each member's slice from the state) -- enqueue_pass orders the passes
      if (nheads > 0) {
        if (nheads_mfma > 0) {
          if (with_reg && MODE == MODE_GRAD && heads_one_pass) hipLaunchKernelGGL((fused_multi_mfma_kernel<T, MODE_GRAD, 2>), dim3(mm_grid), dim3(kThreads), lds_multi_mfma_bytes, stream, a);
          else if (with_reg) hipLaunchKernelGGL((fused_multi_mfma_kernel<T, MODE, 1>), dim3(mm_grid), dim3(kThreads), lds_multi_mfma_bytes, stream, a);
          else hipLaunchKernelGGL((fused_multi_mfma_kernel<T, MODE, 0>), dim3(mm_grid), dim3(kThreads), lds_multi_mfma_bytes, stream, a);
        }
        if (nheads > nheads_mfma) {
          FusedArgs<T> b = a;
          b.heads = a.heads + mm_grid;
          if (with_reg) hipLaunchKernelGGL((fused_multi_kernel<T, MODE, true>), dim3(nheads - nheads_mfma), dim3(kThreads), lds_multi_bytes, stream, b);
          else hipLaunchKernelGGL((fused_multi_kernel<T, MODE, false>), dim3(nheads - nheads_mfma), dim3(kThreads), lds_multi_bytes, stream, b);
        }
      }
    }
    if (nitems > nitems_simple) {
      a.item_base = nitems_simple;
      if (with_reg)
        hipLaunchKernelGGL((fused_group_kernel<T, MODE, true>), dim3(nitems - nitems_simple), dim3(kThreads), lds_group_bytes, stream, a);
      else
        hipLaunchKernelGGL((fused_group_kernel<T, MODE, false>), dim3(nitems - nitems_simple), dim3(kThreads), lds_group_bytes, stream, a);
    }
  }
  // the dense pass: panels with more than four vector tiles (kernel instance with eight accumulator tiles per wave), then the rest
  template <bool GRAD> void launch_dense(Dense64Args m) {
    m.slot_map = mf_map.as<int>();
    hipLaunchKernelGGL((fused_dense64_kernel<GRAD>), dim3(mf_grid), dim3(kDenseThreads), (GRAD ? mf_lds_grad : mf_lds_loss)[0], stream, m);
  }
  template <bool GRAD> void launch_dense(MfmaArgs m) {
    m.slot_map = mf_map.as<int>();
    if (mf_split) {
      if (mf_split2) hipLaunchKernelGGL((fused_dense_split2_kernel<GRAD>), dim3(mf_grid), dim3(kDenseThreads), (GRAD ? mf_lds_grad : mf_lds_loss)[0], stream, m);
      else hipLaunchKernelGGL((fused_dense_split_kernel<GRAD>), dim3(mf_grid), dim3(kDenseThreads), (GRAD ? mf_lds_grad : mf_lds_loss)[0], stream, m);
      return;
    }
    hipLaunchKernelGGL((fused_dense_kernel<GRAD>), dim3(mf_grid), dim3(kDenseThreads), (GRAD ? mf_lds_grad : mf_lds_loss)[0], stream, m);
  }
  T* grad_c0() { return gc_direct ? gcp0.as<T>() : gc0.as<T>(); }
  T* grad_c1() { return gc_direct ? gcp1.as<T>() : gc1.as<T>(); }

  // enqueue: loss (+ gradients) of the current parameters; leaves loss in state, final gradients in comm[0] / grad_c0()
  // one_launch_tail: the caller follows up with step_tail_kernel (enqueue_update), which does everything behind the fused pass
  int enqueue_pass(bool grads, bool apply_update, int losses_cap, bool one_launch_tail = false) {
    const bool R = reg == CAL_REG_SUM;
    FusedArgs<T> a = fused_args();
    DevState* st = st_cur();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) {
      if (ev_used == ev_pool.size()) {
        hipEvent_t x, y;
        HIP_TRY(hipEventCreate(&x));
        HIP_TRY(hipEventCreate(&y));
        ev_pool.push_back({x, y});
      }
      e0 = ev_pool[ev_used].first;
      e1 = ev_pool[ev_used].second;
      ++ev_used;
      HIP_TRY(hipEventRecord(e0, stream));
    }
    // dense path; with the "sum" regulariser it runs twice: a loss-only pass yields S (hence alpha = 2 (S - P)), then the
    // gradient pass applies e = -2 w r + alpha w directly -- no second adjoint set, no combine kernels
    const bool use_mfma = mf_ok;
    const bool Rk = R && !use_mfma;  // the general kernel's two-adjoint-set form of the regulariser
    const bool two_pass = use_mfma && R && grads;
    if (use_mfma) {
      {
        typename std::conditional<std::is_same<T, float>::value, MfmaArgs, Dense64Args>::type m{};
        m.ops = mf_ops.as<T>();
        m.panels = mf_panels.as<PanelItem>();
        m.data_r = data_r.as<T>();
        m.data_i = data_i.as<T>();
        m.wgts = wgts.as<T>();
        m.gains = gains.as<T2>();
        m.c_r = coef.as<T>();
        m.c_i = coef.as<T>() + ncoef;
        m.q0 = q0.as<T2>();
        m.gc_r = grad_c0();
        m.gc_i = grad_c0() + ncoef;
        m.part = part.as<double>();
        m.state = st;
        m.nslices = nslices;
        m.fpad = fpad;
        m.nbls = nbls;
        m.use_alpha = 0;
        if (two_pass) {
          launch_dense<false>(m);
          hipLaunchKernelGGL((gain_grad_kernel<T, false>), dim3(nslices), dim3(256), 0, stream, q0.as<T2>(), q1.as<T2>(), gains.as<T2>(),
                             ant_ptr.as<int>(), ant_ent.as<int2>(), comm.as<T2>(), comm.as<T2>(), comm.as<T2>(), 0, fpad, part.as<double>(),
                             mf_npanels, scal.as<double>(), st, smap(true));
          if (comm_on()) CAL_TRY(all_reduce(scal.p, 4 * (size_t)nslices, CAL_XCHG_F64, CAL_XCHG_SUM));
          hipLaunchKernelGGL(alpha_kernel, dim3((nslices + 63) / 64), dim3(64), 0, stream, st, scal.as<double>(), nslices);
          m.use_alpha = 1;
        }
        if (grads) launch_dense<true>(m); else launch_dense<false>(m);
      }
    } else {
      if (grads && R && nheads > 0 && !heads_one_pass) {
        // baselines that share tiles + the "sum" regulariser: their multi-slice kernels need alpha = 2 (S - P) of every slice BEFORE
        // the gradient pass (no second adjoint set there): a loss pass over everything, the slices' sums, alpha -- then the gradients
        launch_fused<MODE_LOSS>(a, true);
        hipLaunchKernelGGL((gain_grad_kernel<T, false>), dim3(nslices), dim3(256), 0, stream, q0.as<T2>(), q1.as<T2>(), gains.as<T2>(),
                           ant_ptr.as<int>(), ant_ent.as<int2>(), comm.as<T2>(), comm.as<T2>(), comm.as<T2>(), 0, fpad, part.as<double>(),
                           nitems, scal.as<double>(), st, smap(false));
        if (comm_on()) CAL_TRY(all_reduce(scal.p, 4 * (size_t)nslices, CAL_XCHG_F64, CAL_XCHG_SUM));
        hipLaunchKernelGGL(alpha_kernel, dim3((nslices + 63) / 64), dim3(64), 0, stream, st, scal.as<double>(), nslices);
      }
      if (grads) launch_fused<MODE_GRAD>(a, R); else launch_fused<MODE_LOSS>(a, R);
    }
    const int n_parts = use_mfma ? mf_npanels : nitems;
    const SliceMap sm = smap(use_mfma);
    if (timing) HIP_TRY(hipEventRecord(e1, stream));
    if (one_launch_tail) {
      HIP_TRY(hipGetLastError());
      return CAL_OK;
    }
    const size_t gn = (size_t)nants * fpad;
    T2* r0 = comm.as<T2>();
    T2* r1 = r0 + gn;
    T2* r2 = r1 + gn;
    const bool fused_tail = apply_update && !Rk && tail_fits_one_launch();  // loop bookkeeping + update (+ partial-gradient sums) in ONE launch: enqueue_update
    if (grads) {
      if (!gc_direct && !use_mfma && !fused_tail) {
        hipLaunchKernelGGL(coeff_partial_reduce_kernel<T>, dim3((ncoef + 255) / 256), dim3(256), 0, stream, gcp0.as<T>(),
                           gcp0.as<T>() + gcp_len, gc0.as<T>(), gc0.as<T>() + ncoef, coef_grp.as<int>(), grp_coff.as<int>(),
                           grp_item_ptr.as<int>(), item_goff.as<int>(), ncoef, st, sm);
        if (Rk)
          hipLaunchKernelGGL(coeff_partial_reduce_kernel<T>, dim3((ncoef + 255) / 256), dim3(256), 0, stream, gcp1.as<T>(),
                             gcp1.as<T>() + gcp_len, gc1.as<T>(), gc1.as<T>() + ncoef, coef_grp.as<int>(), grp_coff.as<int>(),
                             grp_item_ptr.as<int>(), item_goff.as<int>(), ncoef, st, sm);
      }
      const int cpl = 16 / (int)sizeof(T2) > 0 ? 16 / (int)sizeof(T2) : 1;
      const int nb = nants * ((fpad + 64 * cpl - 1) / (64 * cpl)) + nslices;  // + one block per slice: its loss partial sums
      if (Rk)
        hipLaunchKernelGGL((gain_grad_kernel<T, true>), dim3(nb), dim3(256), 0, stream, q0.as<T2>(), q1.as<T2>(), gains.as<T2>(),
                           ant_ptr.as<int>(), ant_ent.as<int2>(), r0, r1, r2, nants, fpad, part.as<double>(), n_parts,
                           scal.as<double>(), st, sm);
      else
        hipLaunchKernelGGL((gain_grad_kernel<T, false>), dim3(nb), dim3(256), 0, stream, q0.as<T2>(), q1.as<T2>(), gains.as<T2>(),
                           ant_ptr.as<int>(), ant_ent.as<int2>(), r0, r1, r2, nants, fpad, part.as<double>(), n_parts,
                           scal.as<double>(), st, sm);
    } else {
      // loss only: just the partial-sum blocks of the gain kernel (nants = 0 -> no antenna work)
      hipLaunchKernelGGL((gain_grad_kernel<T, false>), dim3(nslices), dim3(256), 0, stream, q0.as<T2>(), q1.as<T2>(), gains.as<T2>(),
                         ant_ptr.as<int>(), ant_ent.as<int2>(), r0, r1, r2, 0, fpad, part.as<double>(), n_parts, scal.as<double>(), st, sm);
    }
    if (comm_on()) {
      // the one exchange step of the sharded fit: sum gain-gradient parts and loss scalars over ranks
      // (issued for a 1-rank communicator too, so the path can be exercised on a single GPU)
      const int gdt = sizeof(T) == 4 ? CAL_XCHG_F32 : CAL_XCHG_F64;
      if (nccl) NCCL_TRY(ncclGroupStart());
      int rc = CAL_OK;
      if (grads) rc = all_reduce(r0, (size_t)(Rk ? 3 : 1) * gn * 2, gdt, CAL_XCHG_SUM);
      if (rc == CAL_OK) rc = all_reduce(scal.p, 4 * (size_t)nslices, CAL_XCHG_F64, CAL_XCHG_SUM);
      if (nccl) NCCL_TRY(ncclGroupEnd());
      CAL_TRY(rc);
    }
    if (!fused_tail)
      hipLaunchKernelGGL(finalize_kernel, dim3((nslices + 63) / 64), dim3(64), 0, stream, st, scal.as<double>(), losses.as<double>(), losses_cap,
                         apply_update ? 1 : 0, nslices);
    if (grads && Rk) {
      hipLaunchKernelGGL(combine_gain_kernel<T>, dim3((int)((gn + 255) / 256)), dim3(256), 0, stream, r0, r1, r2, (int)gn, st, sm, fpad);
      hipLaunchKernelGGL(combine_coeff_kernel<T>, dim3((ncoef + 255) / 256), dim3(256), 0, stream, grad_c0(), grad_c0() + ncoef,
                         grad_c1(), grad_c1() + ncoef, ncoef, st, sm);
    }
    HIP_TRY(hipGetLastError());
    return CAL_OK;
  }

  // Small problems (a step of tens of microseconds: HERA-37, the tutorial) gain from one launch instead of three for the
  // tail of a step; with millions of parameters the per-block decision prologue of the fused kernel costs more than the two
  // kernel boundaries it saves (HERA-350: 105 us against 5 + 56 us), so those keep finalize_kernel + adam2_kernel.
  // (CAL_LAUNCH_KERNELS keeps every kernel its own launch: finalize_kernel + adam2_kernel at any size)
  // (... and LAMB's update is four launches of its own: per-variable norms sit between the moments and the parameters)
  bool tail_fits_one_launch() const {
    return 2LL * nants * fpad + 2LL * ncoef <= (1LL << 20) && launch_mode != CAL_LAUNCH_KERNELS && opt.optimizer != CAL_OPT_LAMB;
  }
  // Problems whose step is tens of microseconds: the whole tail as ONE launch (step_tail_kernel) -- no communicator (the
  // exchange sits between the reduction and the update), general kernels, and not when every kernel is asked to be its own launch
  // (... nor with the regulariser over baselines that share tiles: alpha is needed between that path's two passes)
  bool one_launch_tail() const {
    return !comm_on() && !mf_ok && tail_fits_one_launch() && launch_mode != CAL_LAUNCH_KERNELS && !(reg == CAL_REG_SUM && nheads > 0 && !heads_one_pass);
  }
  void launch_tail(const TailArgs<T>& a, unsigned grid, bool R) {
    if (R)
      hipLaunchKernelGGL((step_tail_kernel<T, true>), dim3(grid), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((step_tail_kernel<T, false>), dim3(grid), dim3(256), 0, stream, a);
  }
  int enqueue_tail(bool freeze_model, int losses_cap) {
    const bool R = reg == CAL_REG_SUM;
    TailArgs<T> a{};
    a.q0 = q0.as<T2>();
    a.q1 = q1.as<T2>();
    a.gains_in = gains.as<T2>();
    a.gains_out = gains_alt.as<T2>();
    a.gains_m = gains_m.as<T>();
    a.gains_v = gains_v.as<T>();
    a.gains_snap = gains_snap.p ? gains_snap.as<T>() : gains_alt.as<T>();  // only written when use_min found a new minimum
    a.ant_ptr = ant_ptr.as<int>();
    a.ant_ent = ant_ent.as<int2>();
    a.coef = AdamSet<T>{coef.as<T>(), gcp0.as<T>(), coef_m.as<T>(), coef_v.as<T>(), coef_snap.p ? coef_snap.as<T>() : coef.as<T>(),
                        freeze_model ? 0LL : 2LL * ncoef};
    a.coef_g1 = gcp1.as<T>();
    if (!gc_direct) {
      a.ps0 = PartialSum<T>{gcp0.as<T>(), gcp0.as<T>() + gcp_len, coef_grp.as<int>(), grp_coff.as<int>(), grp_item_ptr.as<int>(), item_goff.as<int>(), ncoef};
      a.ps1 = PartialSum<T>{gcp1.as<T>(), gcp1.p ? gcp1.as<T>() + gcp_len : nullptr, coef_grp.as<int>(), grp_coff.as<int>(), grp_item_ptr.as<int>(), item_goff.as<int>(), ncoef};
    }
    a.part = part.as<double>();
    a.nparts = nitems;
    a.nants = nants;
    a.fpad = fpad;
    const int cpl = 16 / (int)sizeof(T2) > 0 ? 16 / (int)sizeof(T2) : 1;
    a.nblk_gain = nants * ((fpad + 64 * cpl - 1) / (64 * cpl));
    a.in = st_cur();
    a.out = st_nxt();
    a.losses = losses.as<double>();
    a.losses_cap = losses_cap;
    a.M = smap(false);
    a.cblk_ptr = nslices > 1 ? slice_cblk.as<int>() : nullptr;
    a.ncoef = ncoef;
    const long long nblk_c = freeze_model ? 0 : h_slice_cblk[nslices];
    const unsigned grid = (unsigned)(a.nblk_gain + nblk_c);
    launch_tail(a, grid, R);
    st_par ^= 1;
    std::swap(gains.p, gains_alt.p);  // the buffer just written is the current one
    HIP_TRY(hipGetLastError());
    return CAL_OK;
  }
  int enqueue_update(bool freeze_model, int losses_cap) {
    DevState* st = st_cur();
    const long long gn = 2LL * nants * fpad;
    T* gsnap = gains_snap.p ? gains_snap.as<T>() : gains.as<T>();
    T* csnap = coef_snap.p ? coef_snap.as<T>() : coef.as<T>();
    const AdamSet<T> ga{gains.as<T>(), comm.as<T>(), gains_m.as<T>(), gains_v.as<T>(), gsnap, gn};
    const AdamSet<T> ca{coef.as<T>(), grad_c0(), coef_m.as<T>(), coef_v.as<T>(), csnap, freeze_model ? 0LL : 2LL * ncoef};
    const int nblk_a = (int)((ga.n + 255) / 256), nblk_b = (int)((ca.n + 255) / 256);
    const bool Rk = reg == CAL_REG_SUM && !mf_ok;
    if (!Rk && tail_fits_one_launch()) {
      // the common path: finalize + update (+ the partial coefficient-gradient sums of split groups) as one launch
      PartialSum<T> ps{};
      if (!gc_direct && !mf_ok && !freeze_model)
        ps = PartialSum<T>{gcp0.as<T>(), gcp0.as<T>() + gcp_len, coef_grp.as<int>(), grp_coff.as<int>(), grp_item_ptr.as<int>(),
                           item_goff.as<int>(), ncoef};
      const unsigned nb = (unsigned)std::max(1, std::min(nblk_a + nblk_b, 16384));
      hipLaunchKernelGGL((step_update_kernel<T>), dim3(nb), dim3(256), (size_t)nslices * sizeof(SliceStep<T>), stream, ga, ca, ps, st, st_nxt(),
                         scal.as<double>(), losses.as<double>(), losses_cap, smap(mf_ok), fpad, ncoef);
      st_par ^= 1;
      HIP_TRY(hipGetLastError());
      return CAL_OK;
    }
    if (opt.optimizer == CAL_OPT_LAMB) {
      // moments + u (in place of the gradients), per-variable norms, trust ratios, parameters.  (With a frozen model the
      // coefficient variables are not touched: their set is empty and their ratios are never read.)
      T* ua = comm.as<T>();
      T* ub = grad_c0();
      hipLaunchKernelGGL((lamb_moments_kernel<T>), dim3((unsigned)(nblk_a + nblk_b)), dim3(256), 0, stream, ga, ca, nblk_a, st, smap(mf_ok), fpad, ncoef, ua, ub);
      hipLaunchKernelGGL((lamb_norm_kernel<T>), dim3((unsigned)(lamb_nvar * kLambSeg)), dim3(256), 0, stream, lamb_vars.as<LambVar>(), gains.as<T>(), ua,
                         coef.as<T>(), ub, lamb_partial.as<double>());
      const double* glob = nullptr;
      if (comm_on() && !freeze_model) {
        // a coefficient variable's groups are spread over the ranks: its sums of squares are summed over them (a few doubles per slice)
        const size_t nglob = (size_t)2 * nslices * lamb_nv * 2;
        HIP_TRY(hipMemsetAsync(lamb_glob.p, 0, nglob * sizeof(double), stream));
        hipLaunchKernelGGL(lamb_fold_kernel, dim3((2 * lamb_ncvar + 63) / 64), dim3(64), 0, stream, lamb_partial.as<double>(), lamb_glob.as<double>(),
                           lamb_slot.as<int>(), 2 * nslices, lamb_ncvar, nslices * lamb_nv);
        CAL_TRY(all_reduce(lamb_glob.p, nglob, CAL_XCHG_F64, CAL_XCHG_SUM));
        glob = lamb_glob.as<double>();
      }
      hipLaunchKernelGGL(lamb_ratio_kernel, dim3((lamb_nvar + 63) / 64), dim3(64), 0, stream, lamb_partial.as<double>(), lamb_ratio.as<double>(), lamb_nvar,
                         glob, lamb_slot.as<int>(), 2 * nslices, lamb_ncvar, nslices * lamb_nv);
      hipLaunchKernelGGL((lamb_apply_kernel<T>), dim3((unsigned)(nblk_a + nblk_b)), dim3(256), 0, stream, ga, ca, nblk_a, st, smap(mf_ok), fpad, ncoef, ua, ub,
                         lamb_ratio.as<double>(), lamb_cvar_ptr.as<int>(), lamb_ncvar);
      HIP_TRY(hipGetLastError());
      return CAL_OK;
    }
    {
      constexpr long long per = 256LL * kAdamVec<T>;  // elements per block
      const int va = (int)((ga.n + per - 1) / per), vb = (int)((ca.n + per - 1) / per);
      hipLaunchKernelGGL((adam2_kernel<T>), dim3((unsigned)(va + vb)), dim3(256), 0, stream, ga, ca, va, st, smap(mf_ok), fpad, ncoef);
    }
    HIP_TRY(hipGetLastError());
    return CAL_OK;
  }

  void drop_graph() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    graph_exec = nullptr;
  }
  // kGraphSteps train steps (fused pass + one-launch tail each) captured once per configuration and replayed.  Everything a
  // step reads that changes between steps lives in device memory (loop state, parameters, moments); what is baked into the
  // captured launches -- buffer pointers, optimizer, frozen model, regulariser, loss-history capacity, which halves of the
  // double-buffered state and gains are current -- is the key.  kGraphSteps is even, so a replay leaves both double buffers
  // where it found them and the same graph serves the next replay.
  // `tail`: the two-launch form of small problems (fused pass + step_tail_kernel); else the kernels of a large problem's step (fused / dense
  // pass(es), gain_grad_kernel, finalize_kernel, the update): a host whose cores are busy elsewhere then pays ONE launch per `nsteps` steps
  // instead of four or five per step (measured on shared boxes: 2.4-3.2 ms per step of a 0.6-ms kernel while the host was contended).
  int replay_steps(bool freeze_model, int cap, bool tail, int nsteps) {
    const GraphKey key{opt.optimizer, freeze_model ? 1 : 0, reg, cap, st_par, tail ? 1 : 0, nsteps, gains.p, gains_snap.p, losses.p};
    if (!graph_exec || !(key == graph_key)) {
      drop_graph();
      hipGraph_t graph = nullptr;
      // enqueue_tail flips the double-buffer parities on the HOST for every captured step; nothing has run on the device until the
      // graph is launched, so every failure path below puts them back (an odd number of captured steps would otherwise leave the
      // host pointing at the stale halves)
      const int par0 = st_par;
      void* const gains0 = gains.p;
      void* const alt0 = gains_alt.p;
      auto undo = [&]() {
        st_par = par0;
        gains.p = gains0;
        gains_alt.p = alt0;
      };
      HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      int rc = CAL_OK;
      for (int k = 0; k < nsteps && rc == CAL_OK; ++k) {
        rc = enqueue_pass(true, true, cap, tail);
        if (rc == CAL_OK) rc = tail ? enqueue_tail(freeze_model, cap) : enqueue_update(freeze_model, cap);
      }
      const hipError_t e = hipStreamEndCapture(stream, &graph);
      if (rc != CAL_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        undo();
        return rc;
      }
      if (e != hipSuccess) {
        undo();
        return fail(CAL_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
      }
      const hipError_t ei = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) {
        graph_exec = nullptr;
        undo();
        return fail(CAL_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei));
      }
      undo();  // (kGraphSteps is even: the same values; said explicitly: the launch below is what moves the device)
      graph_key = key;
    }
    HIP_TRY(hipGraphLaunch(graph_exec, stream));
    return CAL_OK;
  }
  int set_launch_mode(int mode) override {
    if (mode != CAL_LAUNCH_AUTO && mode != CAL_LAUNCH_KERNELS && mode != CAL_LAUNCH_ONE_TAIL && mode != CAL_LAUNCH_GRAPH)
      return fail(CAL_ERR_INVALID, "set_launch_mode: unknown mode %d", mode);
    launch_mode = mode;
    return CAL_OK;
  }

  int push_state() {
    for (int t = 0; t < nslices; ++t) {
      DevState& h = h_state[t];
      h.lr = opt.learning_rate;
      h.beta1 = opt.beta_1;
      h.beta2 = opt.beta_2;
      h.eps = opt.epsilon;
      h.opt = opt.optimizer;
      h.nesterov = opt.nesterov;
      h.momentum = opt.momentum;
      h.rho = opt.rho;
      h.ftrl[0] = opt.optimizer == CAL_OPT_LAMB ? opt.weight_decay_rate : opt.learning_rate_power;
      h.ftrl[1] = opt.l1_regularization_strength;
      h.ftrl[2] = opt.l2_regularization_strength + opt.beta / (2.0 * opt.learning_rate);
      h.ftrl[3] = opt.l2_shrinkage_regularization_strength;
      h.reg = reg == CAL_REG_SUM;
      h.f32 = std::is_same<T, float>::value ? 1 : 0;
      h.prior_r = prior_r_t[t];
      h.prior_i = prior_i_t[t];
    }
    HIP_TRY(hipMemcpyAsync(st_cur(), h_state, (size_t)nslices * sizeof(DevState), hipMemcpyHostToDevice, stream));
    return CAL_OK;
  }
  int pull_state() {
    HIP_TRY(hipMemcpyAsync(h_state, st_cur(), (size_t)nslices * sizeof(DevState), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }
  void begin_pass_state() {  // a one-off pass (loss, gradients, model, initial coefficients): no slice is stopped
    for (int t = 0; t < nslices; ++t) h_state[t].done = h_state[t].done_after = 0;
  }
  int collect_timing() {
    for (size_t i = 0; i < ev_used; ++i) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
      t_total_ms += ms;
      ++t_launches;
    }
    ev_used = 0;
    return CAL_OK;
  }
  int ready() {
    if (!has_problem) return fail(CAL_ERR_STATE, "no problem set (cal_solver_set_problem)");
    if (!has_data) return fail(CAL_ERR_STATE, "no data set (cal_solver_set_data)");
    if (!has_gains || !has_coef) return fail(CAL_ERR_STATE, "gains and coefficients must both be set (cal_solver_set_params)");
    return CAL_OK;
  }

  int eval(bool grads, double* loss, void* gg_r, void* gg_i, void* gc_r, void* gc_i) override {
    HIP_TRY(hipSetDevice(device));
    CAL_TRY(ready());
    begin_pass_state();
    CAL_TRY(push_state());
    CAL_TRY(enqueue_pass(grads, false, 0));
    CAL_TRY(pull_state());
    if (timing) CAL_TRY(collect_timing());
    if (loss) {  // several time slices: the sum of their losses (cal_solver_get_slice_losses has each)
      double tot = 0;
      for (int t = 0; t < nslices; ++t) tot += h_state[t].loss;
      *loss = tot;
    }
    if (grads) {
      if (gg_r) CAL_TRY(download_rows(gg_r, comm.as<T>(), nants, 2, 0));
      if (gg_i) CAL_TRY(download_rows(gg_i, comm.as<T>(), nants, 2, 1));
      if (gc_r) HIP_TRY(copy_sync(gc_r, grad_c0(), (size_t)ncoef * sizeof(T), hipMemcpyDeviceToHost));
      if (gc_i) HIP_TRY(copy_sync(gc_i, grad_c0() + ncoef, (size_t)ncoef * sizeof(T), hipMemcpyDeviceToHost));
    }
    return CAL_OK;
  }

  int run(const cal_run_desc* r, double* losses_out, cal_run_result* res, bool per_slice) override {
    HIP_TRY(hipSetDevice(device));
    CAL_TRY(ready());
    if (!has_opt) return fail(CAL_ERR_STATE, "no optimizer set (cal_solver_set_optimizer)");
    if (!r || r->nsteps < 0) return fail(CAL_ERR_INVALID, "run: bad run description");
    if (!per_slice && nslices > 1)
      return fail(CAL_ERR_STATE, "run: the solver holds %d time slices, each with its own losses and stopping test: use cal_solver_run_slices", nslices);
    const int nres = per_slice ? nslices : 1;
    if (res) memset(res, 0, sizeof(*res) * nres);
    if (r->nsteps == 0) return CAL_OK;
    if (r->use_min && !gains_snap.p) {
      CAL_TRY(gains_snap.alloc(gains.bytes));
      CAL_TRY(coef_snap.alloc(coef.bytes));
    }
    const size_t lbytes = (size_t)nslices * r->nsteps * sizeof(double);
    if (r->record && losses.bytes < lbytes) CAL_TRY(losses.alloc(lbytes));
    if (!losses.p) CAL_TRY(losses.alloc((size_t)nslices * sizeof(double)));
    for (int t = 0; t < nslices; ++t) {
      DevState& h = h_state[t];
      h.done = h.done_after = 0;
      h.n_recorded = 0;
      h.nupdates = 0;
      h.improved = 0;
      h.nonfinite = 0;
      h.record = r->record ? 1 : 0;
      h.use_min = r->use_min ? 1 : 0;
      h.tol = r->tol;
    }
    CAL_TRY(push_state());
    // steps are enqueued in chunks; the device decides when the loop of a slice ends (finalize_kernel) and later steps of a
    // chunk fall through at once for it, so the host only synchronises once per chunk instead of once per step (:701)
    const int chunk = steps_per_sync;
    const int cap = r->record ? r->nsteps : 0;
    const bool tail1 = one_launch_tail();
    if (tail1 && !gains_alt.p) CAL_TRY(gains_alt.alloc(gains.bytes));
    // hipGraph replay of kGraphSteps train steps at a time: the steps of such problems are bound by launch latency.  Timed
    // runs (HIP events around every fused pass) issue their launches one by one.
    // Large problems replay too unless the step holds an exchange (a collective or a host callback is not captured): see replay_steps.
    // (capturing and instantiating the graph of a large problem's steps costs tens of milliseconds -- 40 for 16 steps of the dense path --, once
    // per shape of call: calls of fewer than kGraphMinSteps steps launch kernel by kernel, longer ones replay 8 steps at a time)
    constexpr int kGraphStepsLarge = 8, kGraphMinSteps = 256;
    // Of the large problems only the dense path replays in "auto": measured at HERA-350 (tools/graph_ab.py) its step is the same or 0.5 % shorter
    // from a graph, the streaming kernel's step 3-13 % LONGER (4.9 -> 5.1, 4.7 -> 5.3 ms on two boxes: what the passes leave in the caches
    // for the kernels behind them does not survive the graph's node boundaries).
    const bool replay = (tail1 || (!comm_on() && ((mf_ok && r->nsteps >= kGraphMinSteps) || launch_mode == CAL_LAUNCH_GRAPH))) && !timing &&
                        (launch_mode == CAL_LAUNCH_AUTO || launch_mode == CAL_LAUNCH_GRAPH);
    const int gsteps = tail1 ? kGraphSteps : std::min(kGraphStepsLarge, chunk & ~1);  // even: the double-buffered loop state ends a replay where it began
    int issued = 0;
    while (issued < r->nsteps) {
      const int n = std::min(chunk, r->nsteps - issued);
      const bool mark = timing && roctx().push;
      if (mark) {
        char label[96];
        snprintf(label, sizeof(label), "calamity: train steps [%d, %d)%s", issued, issued + n, r->record ? "" : " (unrecorded)");
        roctx().push(label);
      }
      int s = 0;
      if (replay) {
        for (; gsteps >= 2 && s + gsteps <= n; s += gsteps) CAL_TRY(replay_steps(r->freeze_model != 0, cap, tail1, gsteps));
      }
      for (; s < n; ++s) {
        CAL_TRY(enqueue_pass(true, true, cap, tail1));
        if (tail1) CAL_TRY(enqueue_tail(r->freeze_model != 0, cap)); else CAL_TRY(enqueue_update(r->freeze_model != 0, cap));
      }
      issued += n;
      const int prc = pull_state();
      if (mark) roctx().pop();
      CAL_TRY(prc);
      if (timing) CAL_TRY(collect_timing());
      bool all_over = true;
      for (int t = 0; t < nslices; ++t) all_over = all_over && (h_state[t].done || h_state[t].done_after || h_state[t].nonfinite);
      if (all_over) break;
    }
    int bad = -1;
    for (int t = 0; t < nslices; ++t) {
      const DevState& h = h_state[t];
      if (res && t < nres) {
        res[t].nrecorded = h.n_recorded;
        res[t].stopped = (h.done || h.done_after) && !h.nonfinite ? 1 : 0;
        res[t].nupdates = h.nupdates;
        res[t].nonfinite = h.nonfinite ? 1 : 0;
      }
      if (r->record && losses_out && h.n_recorded > 0)
        HIP_TRY(copy_sync(losses_out + (size_t)t * r->nsteps, losses.as<double>() + (size_t)t * cap,
                          (size_t)std::min(h.n_recorded, r->nsteps) * sizeof(double), hipMemcpyDeviceToHost));
      if (h.nonfinite && bad < 0) bad = t;
    }
    if (bad >= 0) {
      if (nslices > 1) return fail(CAL_ERR_NONFINITE, "loss of time slice %d became non-finite after %d updates", bad, h_state[bad].nupdates);
      return fail(CAL_ERR_NONFINITE, "loss became non-finite after %d updates", h_state[bad].nupdates);
    }
    return CAL_OK;
  }

  int model(void* mr, void* mi, bool with_gains) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem || !has_coef) return fail(CAL_ERR_STATE, "model: problem and coefficients must be set");
    if (with_gains && !has_gains) return fail(CAL_ERR_STATE, "data_model: the gains must be set");
    if (!mr || !mi) return fail(CAL_ERR_INVALID, "model: null output");
    const size_t rowbytes = (size_t)nbls * fpad * sizeof(T);
    if (model_buf.bytes < 2 * rowbytes) CAL_TRY(model_buf.alloc(2 * rowbytes));
    begin_pass_state();
    CAL_TRY(push_state());
    FusedArgs<T> a = fused_args();
    a.model_r = model_buf.as<T>();
    a.model_i = model_buf.as<T>() + (size_t)nbls * fpad;
    launch_fused<MODE_MODEL>(a, false);
    if (with_gains)
      hipLaunchKernelGGL(apply_gains_kernel<T>, dim3(grid_for((long long)nbls * fpad)), dim3(256), 0, stream, a.model_r, a.model_i, gains.as<T2>(),
                         bl_ant.as<int2>(), (long long)nbls, fpad);
    HIP_TRY(hipGetLastError());
    CAL_TRY(download_rows(mr, a.model_r, nbls, 1, 0));
    CAL_TRY(download_rows(mi, a.model_i, nbls, 1, 0));
    return CAL_OK;
  }

  int init_coeffs(const void* sr, const void* si) override {
    HIP_TRY(hipSetDevice(device));
    if (!has_problem || !has_data) return fail(CAL_ERR_STATE, "init_coeffs: problem and data (weights) must be set");
    if (!sr || !si) return fail(CAL_ERR_INVALID, "init_coeffs: null source");
    const size_t rowbytes = (size_t)nbls * fpad * sizeof(T);
    if (model_buf.bytes < 2 * rowbytes) CAL_TRY(model_buf.alloc(2 * rowbytes));
    T* s_r = model_buf.as<T>();
    T* s_i = s_r + (size_t)nbls * fpad;
    CAL_TRY(upload_rows(sr, s_r, nbls, 1, 0));
    CAL_TRY(upload_rows(si, s_i, nbls, 1, 0));
    begin_pass_state();
    CAL_TRY(push_state());
    FusedArgs<T> a = fused_args();
    a.data_r = s_r;
    a.data_i = s_i;
    launch_fused<MODE_INIT>(a, false);
    if (!gc_direct)
      hipLaunchKernelGGL(coeff_partial_reduce_kernel<T>, dim3((ncoef + 255) / 256), dim3(256), 0, stream, gcp0.as<T>(),
                         gcp0.as<T>() + gcp_len, gc0.as<T>(), gc0.as<T>() + ncoef, coef_grp.as<int>(), grp_coff.as<int>(),
                         grp_item_ptr.as<int>(), item_goff.as<int>(), ncoef, st_cur(), smap(false));
    HIP_TRY(hipGetLastError());
    // A^T b becomes the coefficient vector (orthonormal-column bases; the host applies the Gram solve otherwise)
    HIP_TRY(hipMemcpyAsync(coef.p, grad_c0(), 2 * (size_t)ncoef * sizeof(T), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    has_coef = true;
    return CAL_OK;
  }

  int synchronize() override {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(stream));
    return CAL_OK;
  }
  int timing_enable(int e) override {
    timing = e != 0;
    t_launches = 0;
    t_total_ms = 0;
    ev_used = 0;
    return CAL_OK;
  }
  int timing_get(cal_kernel_timing* out) override {
    if (!out) return fail(CAL_ERR_INVALID, "timing_get: null");
    if (!has_problem) return fail(CAL_ERR_STATE, "timing_get before set_problem");
    out->launches = t_launches;
    out->total_ms = t_total_ms;
    // B_step = s [F sum nvec (A) + 3 F Nbl (d_r, d_i, w) + 2 sum nvec (c) + 2 Na F (g)] + s [10 sum nvec + 10 Na F]
    const double s = sizeof(T);
    out->basis_bytes_per_launch = basis_bytes;
    out->algorithmic_bytes_per_launch =
        basis_bytes + s * (3.0 * nfreqs * nbls + 2.0 * ncoef + 2.0 * nants * nfreqs) + s * (10.0 * ncoef + 10.0 * nants * nfreqs);
    // forward A c and adjoint A^T gbar_v, complex x real: 4 + 4 flops per (channel, vector); the dense path's regularised step
    // runs the forward twice (loss-only pass for S, then the gradient pass) inside the timed region
    out->flops_per_launch = ((mf_ok && reg == CAL_REG_SUM) ? 12.0 : 8.0) * nfreqs * (double)ncoef;
    out->kernel_path = mf_ok ? ((mf_split || !std::is_same<T, float>::value) ? (mf_split && !mf_split2 ? CAL_PATH_DENSE_SPLIT1 : CAL_PATH_DENSE) : CAL_PATH_DENSE_F32) : CAL_PATH_GENERAL;
    // the dense kernels are written for two workgroups per CU (160 KB of LDS): a basis block of ~250 vectors needs more than
    // 80 KB for its coefficient panel + rings and runs one
    out->dense_wg_per_cu = mf_ok ? (mf_lds_grad[0] * 2 <= 160 * 1024 ? 2 : 1) : 0;
    return CAL_OK;
  }
  int memory_bytes(int64_t* b) override {
    if (!b) return fail(CAL_ERR_INVALID, "memory_bytes: null");
    const DevBuf* all[] = {&tiles, &bl_tile, &bl_ant, &items, &ant_ptr, &ant_ent, &coef_grp, &grp_coff, &grp_item_ptr, &item_goff,
                           &data_r, &data_i, &wgts, &gains, &gains_alt, &gains_m, &gains_v, &gains_snap, &coef, &coef_m, &coef_v, &coef_snap,
                           &q0, &q1, &comm, &scal, &gcp0, &gcp1, &gc0, &gc1, &part, &state, &losses, &scratch, &model_buf,
                           &mf_ops, &mf_panels, &mf_map, &members, &heads, &lamb_vars, &lamb_cvar_ptr, &lamb_partial, &lamb_ratio, &lamb_glob, &lamb_slot, &slice_coff, &slice_ipart_ptr, &slice_ipart_idx,
                           &slice_ppart_ptr, &slice_ppart_idx, &slice_cblk};
    int64_t n = 0;
    for (auto* d : all) n += (int64_t)d->bytes;
    *b = n;
    return CAL_OK;
  }
  int comm_init(const void* id, int rk, int nr) override {
    HIP_TRY(hipSetDevice(device));
    if (!id || nr < 1 || rk < 0 || rk >= nr) return fail(CAL_ERR_INVALID, "comm_init: bad rank/nranks");
    if (nccl) {
      (void)ncclCommDestroy(nccl);
      nccl = nullptr;
    }
    ncclUniqueId uid;
    static_assert(sizeof(ncclUniqueId) <= CAL_COMM_ID_BYTES, "unique id does not fit");
    memcpy(&uid, id, sizeof(uid));
    NCCL_TRY(ncclCommInitRank(&nccl, nr, uid, rk));
    hook = nullptr;
    return joined(rk, nr);
  }
  int set_exchange_hook(cal_exchange_fn fn, void* ctx, int rk, int nr) override {
    HIP_TRY(hipSetDevice(device));
    if (fn && (nr < 1 || rk < 0 || rk >= nr)) return fail(CAL_ERR_INVALID, "set_exchange_hook: bad rank/nranks");
    if (nccl) {
      (void)ncclCommDestroy(nccl);
      nccl = nullptr;
    }
    hook = fn;
    hook_ctx = ctx;
    if (!fn) {
      nranks = 1;
      rank = 0;
      return CAL_OK;
    }
    return joined(rk, nr);
  }
  // how many ranks actually take part in the exchange: every rank contributes 1 to an all-reduce (1 without a communicator)
  int comm_size(int* nranks_seen) override {
    HIP_TRY(hipSetDevice(device));
    if (!nranks_seen) return fail(CAL_ERR_INVALID, "comm_size: null");
    *nranks_seen = 1;
    if (!comm_on()) return CAL_OK;
    if (!agree_buf.p) CAL_TRY(agree_buf.alloc(4 * sizeof(int)));
    int one = 1;
    HIP_TRY(hipMemcpyAsync(agree_buf.p, &one, sizeof(int), hipMemcpyHostToDevice, stream));
    CAL_TRY(all_reduce(agree_buf.p, 1, CAL_XCHG_I32, CAL_XCHG_SUM));
    HIP_TRY(hipMemcpyAsync(&one, agree_buf.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *nranks_seen = one;
    return CAL_OK;
  }
  int joined(int rk, int nr) {
    nranks = nr;
    rank = rk;
    drop_graph();
    if (has_problem) {
      // a problem set before the communicator existed: agree now (fpad is rank-independent by construction; a rank
      // that chose the dense path falls back to the general kernel, which runs on the same buffers)
      CAL_TRY(agree_problem(CAL_OK));
    }
    return CAL_OK;
  }
};

namespace {
template <typename T>
int weighted_square_error(int64_t n, const void* const src[5], double* out) {
  hipStream_t st;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  T* dev = nullptr;
  double* part = nullptr;
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048));
  int rc = CAL_OK;
  auto step = [&](hipError_t e, const char* what) {
    if (rc == CAL_OK && e != hipSuccess) rc = fail(CAL_ERR_HIP, "cal_weighted_square_error: %s failed: %s", what, hipGetErrorString(e));
  };
  step(hipMalloc((void**)&dev, 5 * (size_t)n * sizeof(T)), "hipMalloc");
  step(hipMalloc((void**)&part, (size_t)(nblk + 1) * sizeof(double)), "hipMalloc");
  for (int k = 0; k < 5 && rc == CAL_OK; ++k) step(hipMemcpyAsync(dev + (size_t)k * n, src[k], (size_t)n * sizeof(T), hipMemcpyHostToDevice, st), "upload");
  if (rc == CAL_OK) {
    hipLaunchKernelGGL(square_error_kernel<T>, dim3(nblk), dim3(256), 0, st, dev, dev + n, dev + 2 * n, dev + 3 * n, dev + 4 * n, (long long)n, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part, nblk, part + nblk);
    step(hipGetLastError(), "launch");
    step(hipMemcpyAsync(out, part + nblk, sizeof(double), hipMemcpyDeviceToHost, st), "download");
    step(hipStreamSynchronize(st), "synchronize");
  }
  (void)hipFree(dev);
  (void)hipFree(part);
  (void)hipStreamDestroy(st);
  return rc;
}

}  // namespace

// ================================================================================================================
extern "C" {


#ifdef CAL_STAMP
int cal_debug_read_split_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(calk::g_split_stamps), sizeof(calk::g_split_stamps)); }
int cal_debug_read_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(calk::g_dense_stamps), sizeof(calk::g_dense_stamps)); }
#endif
const char* cal_last_error(void) { return g_err.c_str(); }
const char* cal_version(void) { return "calamity_hip 0.1 (gfx950)"; }

int cal_device_count(int* count) {
  if (!count) return fail(CAL_ERR_INVALID, "cal_device_count: null");
  *count = 0;
  HIP_TRY(hipGetDeviceCount(count));
  return CAL_OK;
}

int cal_device_info(int device, char* name, size_t name_len, int64_t* total_mem_bytes, int32_t* compute_units) {
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  if (name && name_len) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (total_mem_bytes) *total_mem_bytes = (int64_t)p.totalGlobalMem;
  if (compute_units) *compute_units = p.multiProcessorCount;
  return CAL_OK;
}

int cal_weighted_square_error(int device, int dtype, int64_t n, const void* model_r, const void* model_i, const void* data_r, const void* data_i,
                              const void* wgts, double* out) {
  if (!model_r || !model_i || !data_r || !data_i || !wgts || !out) return fail(CAL_ERR_INVALID, "cal_weighted_square_error: null argument");
  if (n < 0) return fail(CAL_ERR_INVALID, "cal_weighted_square_error: n = %lld", (long long)n);
  *out = 0.0;
  if (n == 0) return CAL_OK;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(CAL_ERR_HIP, "no usable HIP device (%s); this library has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(CAL_ERR_INVALID, "cal_weighted_square_error: device %d out of range [0, %d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  const void* const src[5] = {model_r, model_i, data_r, data_i, wgts};
  if (dtype == CAL_F32) return weighted_square_error<float>(n, src, out);
  if (dtype == CAL_F64) return weighted_square_error<double>(n, src, out);
  return fail(CAL_ERR_INVALID, "cal_weighted_square_error: unknown dtype %d", dtype);
}

int cal_device_stream_peak(int device, size_t bytes, int reps, double* read_gbps, double* copy_gbps) {
  if (bytes < (64u << 20) || reps < 1) return fail(CAL_ERR_INVALID, "cal_device_stream_peak: need >= 64 MiB and >= 1 repetition");
  HIP_TRY(hipSetDevice(device));
  const size_t n = bytes / 16;  // float4 elements
  void *src = nullptr, *dst = nullptr;
  float* sink = nullptr;
  hipEvent_t e0, e1;
  HIP_TRY(hipMalloc(&src, n * 16));
  if (hipMalloc(&dst, n * 16) != hipSuccess) { (void)hipFree(src); return fail(CAL_ERR_HIP, "cal_device_stream_peak: out of device memory"); }
  HIP_TRY(hipMalloc((void**)&sink, 1 << 20));
  HIP_TRY(hipMemset(src, 0, n * 16));
  HIP_TRY(hipMemset(dst, 0, n * 16));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  const int grid = p.multiProcessorCount * 8;
  double best_r = 0, best_c = 0;
  for (int r = 0; r < reps + 1; ++r) {  // first launch of each kernel is a warm-up
    float ms = 0;
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, 0, (const f4_t*)src, n, sink);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best_r = std::max(best_r, (double)(n * 16) / (ms * 1e-3) / 1e9);
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, 0, (const f4_t*)src, (f4_t*)dst, n);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best_c = std::max(best_c, (double)(2 * n * 16) / (ms * 1e-3) / 1e9);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  (void)hipFree(sink);
  if (read_gbps) *read_gbps = best_r;
  if (copy_gbps) *copy_gbps = best_c;
  return CAL_OK;
}

int cal_device_busy_clock_mhz(int device, double* mhz) {
  if (!mhz) return fail(CAL_ERR_INVALID, "cal_device_busy_clock_mhz: null");
  HIP_TRY(hipSetDevice(device));
  long long* out = nullptr;
  float* sink = nullptr;
  HIP_TRY(hipMalloc((void**)&out, 2 * sizeof(long long)));
  HIP_TRY(hipMalloc((void**)&sink, 64));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  int wall_khz = 0;
  HIP_TRY(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device));
  long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {  // the last repetition is measured at the clock the load has settled on
    hipLaunchKernelGGL(busy_clock_kernel, dim3(p.multiProcessorCount * 8), dim3(256), 0, 0, out, 400000, sink);
    HIP_TRY(hipDeviceSynchronize());
  }
  HIP_TRY(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  (void)hipFree(out);
  (void)hipFree(sink);
  if (h[1] <= 0 || wall_khz <= 0) return fail(CAL_ERR_HIP, "cal_device_busy_clock_mhz: no wall-clock counter");
  *mhz = (double)h[0] / ((double)h[1] / (wall_khz * 1e3)) / 1e6;
  return CAL_OK;
}

int cal_solver_create(cal_solver** out, int device, int dtype) {
  if (!out) return fail(CAL_ERR_INVALID, "cal_solver_create: null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(CAL_ERR_HIP, "no usable HIP device (%s); this library has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(CAL_ERR_INVALID, "cal_solver_create: device %d out of range [0, %d)", device, n);
  std::unique_ptr<cal_solver> s;
  if (dtype == CAL_F32) {
    auto* p = new SolverT<float>();
    s.reset(p);
    p->device = device;
    p->dtype = dtype;
    CAL_TRY(p->init());
  } else if (dtype == CAL_F64) {
    auto* p = new SolverT<double>();
    s.reset(p);
    p->device = device;
    p->dtype = dtype;
    CAL_TRY(p->init());
  } else {
    return fail(CAL_ERR_INVALID, "cal_solver_create: unknown dtype %d", dtype);
  }
  *out = s.release();
  return CAL_OK;
}

int cal_solver_destroy(cal_solver* s) {
  delete s;
  return CAL_OK;
}

#define NEED(s) \
  if (!(s)) return fail(CAL_ERR_INVALID, "%s: null solver handle", __func__)

int cal_solver_set_problem(cal_solver* s, const cal_problem_desc* d) { NEED(s); return s->set_problem(d); }
int cal_solver_set_data(cal_solver* s, const void* dr, const void* di, const void* w) { NEED(s); return s->set_data(dr, di, w); }
int cal_solver_set_regularization(cal_solver* s, int mode, double pr, double pi) { NEED(s); return s->set_regularization(mode, &pr, &pi, false); }
int cal_solver_set_regularization_slices(cal_solver* s, int mode, const double* pr, const double* pi) { NEED(s); return s->set_regularization(mode, pr, pi, true); }
int cal_solver_get_slice_losses(cal_solver* s, double* losses) { NEED(s); return s->get_slice_losses(losses); }
int cal_solver_set_optimizer(cal_solver* s, const cal_optimizer_desc* d) { NEED(s); return s->set_optimizer(d); }
int cal_solver_set_params(cal_solver* s, const void* g_r, const void* g_i, const void* c_r, const void* c_i) {
  NEED(s);
  return s->set_params(g_r, g_i, c_r, c_i);
}
int cal_solver_get_params(cal_solver* s, int which, void* g_r, void* g_i, void* c_r, void* c_i) {
  NEED(s);
  return s->get_params(which, g_r, g_i, c_r, c_i);
}
int cal_solver_get_moments(cal_solver* s, void* gm_r, void* gm_i, void* gv_r, void* gv_i, void* cm_r, void* cm_i, void* cv_r,
                           void* cv_i, int64_t* t) {
  NEED(s);
  return s->get_moments(gm_r, gm_i, gv_r, gv_i, cm_r, cm_i, cv_r, cv_i, t);
}
int cal_solver_set_moments(cal_solver* s, const void* gm_r, const void* gm_i, const void* gv_r, const void* gv_i, const void* cm_r,
                           const void* cm_i, const void* cv_r, const void* cv_i, int64_t t) {
  NEED(s);
  return s->set_moments(gm_r, gm_i, gv_r, gv_i, cm_r, cm_i, cv_r, cv_i, t);
}
int cal_solver_eval_loss(cal_solver* s, double* loss) { NEED(s); return s->eval(false, loss, nullptr, nullptr, nullptr, nullptr); }
int cal_solver_eval_grads(cal_solver* s, double* loss, void* gg_r, void* gg_i, void* gc_r, void* gc_i) {
  NEED(s);
  return s->eval(true, loss, gg_r, gg_i, gc_r, gc_i);
}
int cal_solver_run(cal_solver* s, const cal_run_desc* r, double* losses_out, cal_run_result* res) { NEED(s); return s->run(r, losses_out, res, false); }
int cal_solver_run_slices(cal_solver* s, const cal_run_desc* r, double* losses_out, cal_run_result* res) { NEED(s); return s->run(r, losses_out, res, true); }
int cal_solver_model(cal_solver* s, void* mr, void* mi) { NEED(s); return s->model(mr, mi, false); }
int cal_solver_data_model(cal_solver* s, void* mr, void* mi) { NEED(s); return s->model(mr, mi, true); }
int cal_solver_init_coeffs(cal_solver* s, const void* sr, const void* si) { NEED(s); return s->init_coeffs(sr, si); }
int cal_solver_synchronize(cal_solver* s) { NEED(s); return s->synchronize(); }
int cal_solver_set_launch_mode(cal_solver* s, int mode) { NEED(s); return s->set_launch_mode(mode); }
int cal_solver_timing_enable(cal_solver* s, int e) { NEED(s); return s->timing_enable(e); }
int cal_solver_timing_get(cal_solver* s, cal_kernel_timing* out) { NEED(s); return s->timing_get(out); }
int cal_solver_memory_bytes(cal_solver* s, int64_t* b) { NEED(s); return s->memory_bytes(b); }

int cal_comm_unique_id(void* id_out) {
  if (!id_out) return fail(CAL_ERR_INVALID, "cal_comm_unique_id: null");
  ncclUniqueId uid;
  NCCL_TRY(ncclGetUniqueId(&uid));
  memset(id_out, 0, CAL_COMM_ID_BYTES);
  memcpy(id_out, &uid, sizeof(uid));
  return CAL_OK;
}
int cal_solver_comm_init(cal_solver* s, const void* id, int rank, int nranks) { NEED(s); return s->comm_init(id, rank, nranks); }
int cal_solver_set_exchange_hook(cal_solver* s, cal_exchange_fn fn, void* ctx, int rank, int nranks) { NEED(s); return s->set_exchange_hook(fn, ctx, rank, nranks); }
int cal_solver_comm_size(cal_solver* s, int* nranks_seen) { NEED(s); return s->comm_size(nranks_seen); }

}  // extern "C"
