/* calamity_hip.h -- C-ABI of the MI355X (gfx950) gain + foreground gradient-descent fitter.
 *
 * The reference (aewallwi/calamity) has no FFI: its seam is the Python function
 *   fit_gains_and_foregrounds(...)            /root/reference/calamity/calibration.py:447-738
 * whose arithmetic is issued as TensorFlow ops (fg_model/data_model/mse/mse_chunked[_sum_regularized],
 * calibration.py:1587-1656; tf.GradientTape :664-666; tf.optimizers.* apply_gradients :667).  The entry
 * points below are what a ctypes binding for that seam needs; each one cites the reference code it
 * replaces.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *  - every function returns 0 on success and a negative cal_status on failure; cal_last_error() returns a
 *    thread-local message.  Nothing throws across the boundary.
 *  - all host pointers are caller-owned, C-contiguous, and are only read/written inside the call.
 *  - "real" arrays (void*) are float when the solver was created with CAL_F32 and double with CAL_F64.
 *  - the library owns every byte of device memory.  One handle is not re-entrant; distinct handles may be
 *    driven from distinct threads.
 *  - there is NO CPU fallback: without a usable HIP device cal_solver_create() fails.
 */
#ifndef CALAMITY_HIP_H
#define CALAMITY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cal_solver cal_solver;

enum cal_status {
  CAL_OK = 0,
  CAL_ERR_INVALID = -1, /* bad argument / inconsistent problem description */
  CAL_ERR_HIP = -2,     /* HIP runtime failure */
  CAL_ERR_RCCL = -3,    /* RCCL failure */
  CAL_ERR_STATE = -4,   /* call made in the wrong state (e.g. run before set_problem) */
  CAL_ERR_UNSUPPORTED = -5,
  CAL_ERR_NONFINITE = -6 /* loss became NaN/Inf; parameters are left as they were at that step */
};

enum cal_dtype { CAL_F32 = 0, CAL_F64 = 1 };                 /* dtype kwarg, calibration.py:464, :974 */
enum cal_optimizer { /* OPTIMIZERS, calibration.py:17-27: the whole table */
  CAL_OPT_ADAM = 0, CAL_OPT_ADAMAX = 1, CAL_OPT_SGD = 2, CAL_OPT_RMSPROP = 3, CAL_OPT_ADAGRAD = 4, CAL_OPT_NADAM = 5, CAL_OPT_ADADELTA = 6,
  CAL_OPT_FTRL = 7, CAL_OPT_LAMB = 8 /* tensorflow_addons.optimizers.LAMB, :26.  One trust ratio per VARIABLE (cal_problem_desc::grp_var);
                                        with a communicator attached the per-variable norms are summed over the ranks before the ratio */
};
enum cal_regularization { CAL_REG_NONE = 0, CAL_REG_SUM = 1 }; /* model_regularization, calibration.py:619-661 */
enum cal_layout {
  CAL_LAYOUT_STREAM = 0, /* every baseline owns its basis tiles in HBM (the reference's per-baseline tensor,
                            calibration.py:167-184, minus the zero padding): HBM-streaming kernel */
  CAL_LAYOUT_SHARED = 1  /* baselines alias the unique basis blocks (one per distinct delay, the operator_cache
                            of modeling.py:291-301): cache-resident basis */
};
enum cal_kernel_path {
  CAL_PATH_AUTO = 0,    /* dense kernel when the problem is eligible and large enough to fill the chip, else general */
  CAL_PATH_GENERAL = 1, /* register-direct streaming / group kernels (any dtype, layout, group shape) */
  CAL_PATH_DENSE = 2,   /* matrix-core kernel wherever the problem is eligible (SHARED layout, one baseline per fitting
                           group, basis_nvec <= 256, nfreqs > 64); CAL_ERR_UNSUPPORTED when it is not.  fp32: the split-bf16
                           kernel (six v_mfma_f32_32x32x16_bf16 per fp32 product block, four panels of a workgroup on one
                           operand image per unit of channels, read row-wise by the forward and transposed by the adjoint
                           product); fp64: v_mfma_f64_16x16x4_f64 */
  CAL_PATH_DENSE_F32 = 3, /* fp32 only: the dense kernel on v_mfma_f32_32x32x2_f32 (one panel per workgroup) that CAL_PATH_DENSE
                            ran before the split-bf16 kernel replaced it; kept for A/B measurements and as the accuracy yardstick */
  CAL_PATH_DENSE_SPLIT1 = 4 /* fp32 only: the first split-bf16 kernel (two packed operand streams through an LDS ring, coefficient
                            panels in LDS), which CAL_PATH_DENSE ran until the one-image form replaced it; kept for A/B measurements */
};

enum cal_launch_mode {
  CAL_LAUNCH_AUTO = 0,     /* the fastest form for the problem: large problems one launch per kernel; problems whose step is tens of
                              microseconds (no communicator, general kernels, at most 2^20 parameters) a two-launch step replayed
                              from a hipGraph, 16 steps per replay */
  CAL_LAUNCH_KERNELS = 1,  /* every kernel its own launch at every problem size (per-antenna reduction, loop bookkeeping, regulariser
                              fold, update: finalize_kernel + adam2_kernel, never the fused step_update_kernel / step_tail_kernel) */
  CAL_LAUNCH_ONE_TAIL = 2, /* small problems: fused pass + ONE tail launch per step, launched one by one (no graph) */
  CAL_LAUNCH_GRAPH = 3     /* as AUTO where the two-launch step applies */
};

/* Ragged description of one fit (replaces the zero-padded chunk tensors built by
 * tensorize_fg_model_comps_dict / tensorize_data, calibration.py:104-310).
 * A fitting group g shares ONE coefficient vector of basis_nvec[grp_basis[g]] complex numbers; its baselines are
 * bl in [grp_bl_start[g], grp_bl_start[g+1]).  Basis block u is row-major
 * [basis_nrowblk[u] * nfreqs][basis_nvec[u]] starting at basis_data + basis_offset[u] elements
 * ("Nfreqs x Ncomponents", modeling.py:288-289); baseline bl uses rows
 * [bl_rowblk[bl] * nfreqs, (bl_rowblk[bl] + 1) * nfreqs).  Consecutive baselines of a group with the same row block
 * (a redundant set, use_redundancy=True) share one forward / adjoint product on the device.
 * Limits: basis_nvec <= 896 (CAL_ERR_UNSUPPORTED beyond); all index arrays are validated (CAL_ERR_INVALID).
 * Device rows are padded to a multiple of min(128, next_pow2(nfreqs)) channels: a function of nfreqs alone, so every
 * rank of a sharded fit uses the same gain layout and all-reduce count. */
typedef struct cal_problem_desc {
  int32_t nants;
  int32_t nfreqs;
  int32_t ngrps;
  int32_t nbls;
  int32_t nbasis;
  const int64_t* basis_offset;   /* [nbasis + 1] */
  const int32_t* basis_nvec;     /* [nbasis] */
  const int32_t* basis_nrowblk;  /* [nbasis] */
  const void* basis_data;        /* real */
  const int32_t* grp_basis;      /* [ngrps] */
  const int32_t* grp_bl_start;   /* [ngrps + 1] */
  const int32_t* bl_ant0;        /* [nbls]  corr_inds[..][..][bl][0], calibration.py:176-177 */
  const int32_t* bl_ant1;        /* [nbls] */
  const int32_t* bl_rowblk;      /* [nbls] */
  int32_t layout;                /* cal_layout */
  int32_t kernel_path;           /* cal_kernel_path; with a communicator attached the ranks agree on one path */
  const int32_t* bl_alias;       /* [nbls] or NULL.  STREAM layout: baseline b reads the basis tiles of baseline bl_alias[b] (which
                                    owns its tiles: bl_alias of it is -1 or itself) instead of holding a copy -- the same physical
                                    baseline in the time slices one solver fits together (calibration.py:1160-1167 fits them one after
                                    another).  Both must be single-baseline groups on the same basis rows.  Such baselines are
                                    processed together: their tiles are read ONCE per pass for all of them. */
  int32_t nslices;               /* 0 or 1: one fit.  T > 1: the solver holds T independent fits -- the (polarization, time) slices that
                                    calibrate_and_model_tensor fits one after another (calibration.py:1160-1167) -- over the same kind of
                                    array: nants = T * (antennas of one slice), slice t owns antennas [t nants / T, (t + 1) nants / T), a
                                    baseline's two antennas and all baselines of a fitting group lie in ONE slice, groups are listed slice
                                    by slice, every slice has at least one group; T <= CAL_MAX_SLICES.  Each slice keeps its own loss,
                                    regulariser sums and priors, tolerance stop, use_min snapshot and optimizer iteration count
                                    (cal_solver_run_slices); a stopped slice's parameters freeze while the others go on. */
  int32_t reserved;
  const int32_t* grp_var;        /* [ngrps] or NULL: which optimizer VARIABLE a group's coefficients belong to -- the chunk of the reference's
                                    fg_r[chunk] / fg_i[chunk] tensors (calibration.py:596-603).  Only a layer-wise optimizer (LAMB: one trust
                                    ratio per variable) looks at it; groups of a variable are contiguous inside a time slice.  NULL: the
                                    coefficients of a slice are one variable per component. */
} cal_problem_desc;
#define CAL_MAX_SLICES 256

typedef struct cal_optimizer_desc { /* **opt_kwargs -> tf.optimizers.X(...), calibration.py:571; semantics: Keras OptimizerV2
                                     * (TensorFlow 2.4 - 2.10), formulae in fit_kernels.hpp: optimizer_step */
  int32_t optimizer;              /* cal_optimizer */
  double learning_rate;           /* Keras defaults: 1e-3 (SGD: 1e-2) */
  double beta_1;                  /* Adam, Adamax, Nadam: 0.9 */
  double beta_2;                  /* 0.999 */
  double epsilon;                 /* 1e-7 */
  double rho;                     /* RMSprop 0.9, Adadelta 0.95 */
  double momentum;                /* SGD, RMSprop: 0 */
  double initial_accumulator_value; /* Adagrad: 0.1 */
  int32_t nesterov;               /* SGD */
  int32_t reserved;
  /* Ftrl (initial_accumulator_value above: 0.1) */
  double learning_rate_power;                  /* -0.5 */
  double l1_regularization_strength;           /* 0 */
  double l2_regularization_strength;           /* 0 */
  double l2_shrinkage_regularization_strength; /* 0 */
  double beta;                                 /* 0 */
  double weight_decay_rate;                    /* LAMB: 0 (epsilon defaults to 1e-6 there) */
} cal_optimizer_desc;

typedef struct cal_run_desc { /* loop controls of fit_gains_and_foregrounds, calibration.py:457-461 */
  int32_t nsteps;           /* number of train steps to issue at most */
  int32_t record;           /* 0: unrecorded updates (profile steps :681-687 and the "graph build" step :693);
                               1: recorded steps of the main loop :699-717 */
  int32_t use_min;          /* :702-710 */
  int32_t freeze_model;     /* :598-603 */
  double tol;               /* :712 (only consulted when record != 0) */
} cal_run_desc;

typedef struct cal_run_result {
  int32_t nrecorded; /* losses written to losses_out by this call */
  int32_t stopped;   /* 1 if the tolerance test ended the loop */
  int32_t nupdates;  /* optimizer updates applied by this call */
  int32_t nonfinite; /* 1 if the loss became NaN/Inf (the call then returns CAL_ERR_NONFINITE) */
} cal_run_result;

typedef struct cal_kernel_timing { /* HIP-event timing of the dominant kernel of a pass: the fused basis-streaming kernel or the dense kernel (bench.py roofline) */
  int64_t launches;
  double total_ms;
  double algorithmic_bytes_per_launch; /* SURVEY.md 8(d) B_step figure restated for this problem */
  double basis_bytes_per_launch;
  double flops_per_launch;             /* 8 F sum nvec: forward A c and adjoint A^T gbar_v, complex x real (12: the dense path's
                                          regularised step times its loss-only pass too) */
  int32_t kernel_path;                 /* CAL_PATH_GENERAL or CAL_PATH_DENSE: the family the timed launches belong to */
  int32_t dense_wg_per_cu;             /* dense path: workgroups per CU its LDS footprint allows (2 is what the kernels are tuned for) */
} cal_kernel_timing;

const char* cal_last_error(void);
const char* cal_version(void);
int cal_device_count(int* count);
int cal_device_info(int device, char* name, size_t name_len, int64_t* total_mem_bytes, int32_t* compute_units);
/* Measured streaming peaks of the device (no reference counterpart; BASELINE.md section 3 asks for the roofline against a
 * stream kernel measured on the box next to the nominal 8 TB/s): a read-only sweep (16-byte non-temporal loads, the
 * access pattern of the fit's tile stream) and a copy, each over `bytes` of HBM, best of `reps` launches, in GB/s. */
int cal_device_stream_peak(int device, size_t bytes, int reps, double* read_gbps, double* copy_gbps);
/* Shader clock the device sustains while every CU is busy with vector FMAs for a few milliseconds (MHz): boxes of the
 * same model differ in the clock they hold under load, and compute-side kernel times scale with it. */
int cal_device_busy_clock_mhz(int device, double* mhz);
/* mse, calibration.py:1608-1609, on its own: sum over n samples of w ((d_r - m_r)^2 + (d_i - m_i)^2) for five host arrays of n
 * reals of type dtype (CAL_F32 / CAL_F64), evaluated on `device` (products in dtype as the reference's graph does, partial sums
 * in double, fixed order).  The fit itself never materialises a model: this is the reference's building block under its own name. */
int cal_weighted_square_error(int device, int dtype, int64_t n, const void* model_r, const void* model_i, const void* data_r,
                              const void* data_i, const void* wgts, double* out);

/* tf.device / GPU selection of read_calibrate_and_model_dpss, calibration.py:1741-1753, :1796-1804 */
int cal_solver_create(cal_solver** out, int device, int dtype);
int cal_solver_destroy(cal_solver* s);

/* tensorize_fg_model_comps_dict, calibration.py:104-190 (called once per dataset, :1143-1152) */
int cal_solver_set_problem(cal_solver* s, const cal_problem_desc* desc);
/* tensorize_data, calibration.py:193-310 (per pol/time): [nbls][nfreqs] real each; weights already normalised */
int cal_solver_set_data(cal_solver* s, const void* data_r, const void* data_i, const void* wgts);
/* model_regularization="sum": priors of calibration.py:619-625; mode = cal_regularization */
int cal_solver_set_regularization(cal_solver* s, int mode, double prior_r_sum, double prior_i_sum);
/* the same with one pair of priors per time slice (cal_problem_desc::nslices): prior_*_sum [nslices] */
int cal_solver_set_regularization_slices(cal_solver* s, int mode, const double* prior_r_sum, const double* prior_i_sum);
int cal_solver_set_optimizer(cal_solver* s, const cal_optimizer_desc* desc); /* also zeroes moments and t */

/* tf.Variable(g_r), ... calibration.py:596-603.  gains [nants][nfreqs]; coefficients flat in group order,
 * sum_g nvec_g reals each.  Any pointer may be NULL to leave that array untouched. */
int cal_solver_set_params(cal_solver* s, const void* g_r, const void* g_i, const void* c_r, const void* c_i);
/* .value() snapshots, calibration.py:706-710, :724-728.  which = 0: current parameters; 1: use_min snapshot */
int cal_solver_get_params(cal_solver* s, int which, void* g_r, void* g_i, void* c_r, void* c_i);
/* optimizer slots (checkpoint / resume; no counterpart in the reference): m and v (Adam) / u (Adamax).  A fit resumed with
 * set_params + set_moments (after set_optimizer, whose betas the bias corrections are rebuilt from) continues bit for bit.
 * The two slots per parameter, (m, v): Adam / Nadam first and second moment; Adamax (m, u); SGD (momentum accumulator, unused);
 * RMSprop (momentum accumulator, mean square); Adagrad (unused, accumulator); Adadelta (accumulated updates, accumulated gradients);
 * Ftrl (linear, accumulator); LAMB (first, second moment).  Several time slices (cal_problem_desc::nslices): t is the count every
 * slice shares; get_moments fails with CAL_ERR_STATE when the slices have applied different numbers of updates (they stop on their own). */
int cal_solver_get_moments(cal_solver* s, void* gm_r, void* gm_i, void* gv_r, void* gv_i, void* cm_r, void* cm_i,
                           void* cv_r, void* cv_i, int64_t* t);
int cal_solver_set_moments(cal_solver* s, const void* gm_r, const void* gm_i, const void* gv_r, const void* gv_i,
                           const void* cm_r, const void* cm_i, const void* cv_r, const void* cv_i, int64_t t);

/* loss_function() alone: mse_chunked / mse_chunked_sum_regularized, calibration.py:1612-1656 */
int cal_solver_eval_loss(cal_solver* s, double* loss); /* several time slices: the sum of their losses */
/* the loss of every time slice as of the last cal_solver_eval_loss / cal_solver_eval_grads: losses [nslices] */
int cal_solver_get_slice_losses(cal_solver* s, double* losses);
/* tape.gradient(loss, vars), calibration.py:664-666, without the update (parity tests) */
int cal_solver_eval_grads(cal_solver* s, double* loss, void* gg_r, void* gg_i, void* gc_r, void* gc_i);
/* train_step() x nsteps with the loop semantics of calibration.py:681-717; losses_out: [nsteps] doubles or NULL */
int cal_solver_run(cal_solver* s, const cal_run_desc* run, double* losses_out, cal_run_result* result);
/* The same loop for every time slice of the solver at once (cal_problem_desc::nslices; the time loop of calibration.py:1160-1167,
 * :1244-1269 as ONE batch): each train step advances every slice that has not stopped; slice t records its own losses
 * (losses_out [nslices][run->nsteps], row t holds results[t].nrecorded values), applies the tolerance test of :712-717 and the
 * use_min bookkeeping of :702-710 to ITS loss, and stops on its own -- its gains and coefficients then stay as they are.
 * cal_solver_get_params(which = 1) holds every slice's own minimum.  results: [nslices].  A non-finite loss stops that slice
 * only; the call then returns CAL_ERR_NONFINITE after the other slices have finished (results[t].nonfinite tells which). */
int cal_solver_run_slices(cal_solver* s, const cal_run_desc* run, double* losses_out, cal_run_result* results);
/* yield_fg_model_array, calibration.py:402-444, per baseline instead of a nants x nants cube: [nbls][nfreqs] */
int cal_solver_model(cal_solver* s, void* model_r, void* model_i);
/* data_model, calibration.py:1593-1605: the foreground model of every baseline times its antennas' current gains,
 * g_ant0 conj(g_ant1) (A c): [nbls][nfreqs] (gains and coefficients must be set) */
int cal_solver_data_model(cal_solver* s, void* model_r, void* model_i);
/* tensorize_fg_coeffs, calibration.py:828-913: per group least squares of src on the basis with samples of zero
 * weight zeroed; the result becomes the current coefficients.  src_*: [nbls][nfreqs] real. */
int cal_solver_init_coeffs(cal_solver* s, const void* src_r, const void* src_i);

int cal_solver_synchronize(cal_solver* s);
/* How cal_solver_run issues a train step (cal_launch_mode).  Every mode computes the same numbers, bit for bit: the python
 * loop of calibration.py:699-717 pays a host synchronisation per step (:701); here the choice is only how many launches a
 * step costs. */
int cal_solver_set_launch_mode(cal_solver* s, int mode);
int cal_solver_timing_enable(cal_solver* s, int enable);
int cal_solver_timing_get(cal_solver* s, cal_kernel_timing* out);
int cal_solver_memory_bytes(cal_solver* s, int64_t* device_bytes);

/* Baseline-sharded data parallelism (no counterpart in the reference, which is single-device,
 * calibration.py:1796-1804): one process per GPU, one RCCL all-reduce of the per-antenna gain gradients and the
 * loss scalars per step.  id: CAL_COMM_ID_BYTES bytes produced on rank 0 and shared out of band. */
#define CAL_COMM_ID_BYTES 128
int cal_comm_unique_id(void* id_out);
int cal_solver_comm_init(cal_solver* s, const void* id, int rank, int nranks);
/* The same exchange through a transport of the caller's (MPI, gloo, ...; and the vehicle of the two-ranks-on-one-GPU tests:
 * RCCL refuses two ranks on one device).  Wherever the library would call ncclAllReduce -- the set-up agreement, the gain
 * gradients and the loss scalars of every step -- it drains its stream, hands the callback the SAME buffer and element
 * count staged in pinned host memory, and expects it reduced in place over the nranks callers (0 = success), then copies it
 * back.  Replaces a communicator of cal_solver_comm_init and vice versa; fn = NULL detaches. */
enum cal_exchange_dtype { CAL_XCHG_F32 = 0, CAL_XCHG_F64 = 1, CAL_XCHG_I32 = 2 };
enum cal_exchange_op { CAL_XCHG_SUM = 0, CAL_XCHG_MIN = 1 };
typedef int (*cal_exchange_fn)(void* ctx, void* host_buf, int64_t count, int dtype, int op);
int cal_solver_set_exchange_hook(cal_solver* s, cal_exchange_fn fn, void* ctx, int rank, int nranks);
/* How many ranks take part in the solver's exchange, COUNTED by the exchange itself: every rank adds 1 in an all-reduce over the
 * communicator / hook (1 without either).  A launcher's rank count is a claim; this is what the data path sees (bench.py reports it). */
int cal_solver_comm_size(cal_solver* s, int* nranks_seen);

#ifdef __cplusplus
}
#endif
#endif /* CALAMITY_HIP_H */
