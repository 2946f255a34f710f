"""Thin object wrapper over the C-ABI handle (include/calamity_hip.h).  NumPy in, NumPy out."""
import ctypes as C

import numpy as np

from . import _lib
from .problem import FitProblem

# OPTIMIZERS of /root/reference/calamity/calibration.py:17-27 -- the whole table, "LAMB" being tensorflow_addons.optimizers.LAMB;
# any other name raises KeyError exactly like ``OPTIMIZERS[optimizer]`` at
# calibration.py:571.  Constructor arguments and defaults are those of tf.keras.optimizers.* (OptimizerV2, TF 2.4 - 2.10);
# an argument the optimizer does not take raises TypeError, as the Keras constructor would.
OPTIMIZERS = {"Adam": _lib.CAL_OPT_ADAM, "Adamax": _lib.CAL_OPT_ADAMAX, "SGD": _lib.CAL_OPT_SGD, "RMSprop": _lib.CAL_OPT_RMSPROP,
              "Adagrad": _lib.CAL_OPT_ADAGRAD, "Nadam": _lib.CAL_OPT_NADAM, "Adadelta": _lib.CAL_OPT_ADADELTA, "Ftrl": _lib.CAL_OPT_FTRL, "LAMB": _lib.CAL_OPT_LAMB}
_MOMENTS = dict(learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7)
_OPT_DEFAULTS = {
    "Adam": _MOMENTS,
    "Adamax": _MOMENTS,
    "Nadam": _MOMENTS,
    "SGD": dict(learning_rate=1e-2, momentum=0.0, nesterov=False),
    "RMSprop": dict(learning_rate=1e-3, rho=0.9, momentum=0.0, epsilon=1e-7),
    "Adagrad": dict(learning_rate=1e-3, initial_accumulator_value=0.1, epsilon=1e-7),
    "Adadelta": dict(learning_rate=1e-3, rho=0.95, epsilon=1e-7),
    "Ftrl": dict(learning_rate=1e-3, learning_rate_power=-0.5, initial_accumulator_value=0.1, l1_regularization_strength=0.0,
                 l2_regularization_strength=0.0, l2_shrinkage_regularization_strength=0.0, beta=0.0),
    "LAMB": dict(learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-6, weight_decay_rate=0.0),
}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipFitSolver:
    """One gain + foreground fit resident on one MI355X."""

    def __init__(self, dtype=np.float32, device=0):
        self._lib = _lib.load()
        self.dtype = np.dtype(dtype)
        if self.dtype == np.float32:
            code = _lib.CAL_F32
        elif self.dtype == np.float64:
            code = _lib.CAL_F64
        else:
            raise ValueError(f"dtype must be float32 or float64, got {dtype}")
        self._h = C.c_void_p()
        _lib.check(self._lib.cal_solver_create(C.byref(self._h), int(device), code))
        self.problem = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cal_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _real(self, a, shape=None):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None and a.shape != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
        return a

    # ---- problem -------------------------------------------------------------------------------------------
    def set_problem(self, prob: FitProblem, layout="stream", kernel_path="auto"):
        """``kernel_path``: "auto" (dense matrix-core kernel when eligible and large enough), "general", "dense", or
        "dense_f32" (fp32: the v_mfma_f32_32x32x2_f32 kernel the split-bf16 one replaced)."""
        prob.validate()
        basis = [np.ascontiguousarray(b, dtype=self.dtype) for b in prob.basis]
        sizes = np.asarray([b.size for b in basis], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        flat = np.concatenate([b.ravel() for b in basis]) if len(basis) > 1 else basis[0].ravel()
        nvec = np.asarray([b.shape[1] for b in basis], dtype=np.int32)
        nrb = np.asarray([b.shape[0] // prob.nfreqs for b in basis], dtype=np.int32)
        keep = [
            flat, offs, nvec, nrb,
            np.ascontiguousarray(prob.grp_basis, dtype=np.int32),
            np.ascontiguousarray(prob.grp_bl_start, dtype=np.int32),
            np.ascontiguousarray(prob.bl_ant0, dtype=np.int32),
            np.ascontiguousarray(prob.bl_ant1, dtype=np.int32),
            np.ascontiguousarray(prob.bl_rowblk, dtype=np.int32),
            None if getattr(prob, "bl_alias", None) is None else np.ascontiguousarray(prob.bl_alias, dtype=np.int32),
            # the reference's variables fg_r[chunk] / fg_i[chunk] (a layer-wise optimizer -- LAMB -- takes one trust ratio per variable)
            None if getattr(prob, "chunk_of_grp", None) is None else np.ascontiguousarray(prob.chunk_of_grp, dtype=np.int32),
        ]
        d = _lib.ProblemDesc(
            nants=prob.nants, nfreqs=prob.nfreqs, ngrps=prob.ngrps, nbls=prob.nbls, nbasis=len(basis),
            basis_offset=_ptr(offs), basis_nvec=_ptr(nvec), basis_nrowblk=_ptr(nrb), basis_data=_ptr(flat),
            grp_basis=_ptr(keep[4]), grp_bl_start=_ptr(keep[5]), bl_ant0=_ptr(keep[6]), bl_ant1=_ptr(keep[7]),
            bl_rowblk=_ptr(keep[8]), bl_alias=_ptr(keep[9]), nslices=int(getattr(prob, "nslices", 1) or 1), grp_var=_ptr(keep[10]),
            layout={"stream": _lib.CAL_LAYOUT_STREAM, "shared": _lib.CAL_LAYOUT_SHARED}[layout],
            kernel_path={"auto": _lib.CAL_PATH_AUTO, "general": _lib.CAL_PATH_GENERAL, "dense": _lib.CAL_PATH_DENSE,
                         "dense_f32": _lib.CAL_PATH_DENSE_F32, "dense_split1": _lib.CAL_PATH_DENSE_SPLIT1}[kernel_path],
        )
        _lib.check(self._lib.cal_solver_set_problem(self._h, C.byref(d)))
        self.problem = prob
        self.nants, self.nfreqs, self.nbls, self.ncoeffs = prob.nants, prob.nfreqs, prob.nbls, prob.ncoeffs
        self.nslices = int(getattr(prob, "nslices", 1) or 1)
        if prob.data_r is not None:
            self.set_data(prob.data_r, prob.data_i, prob.wgts)
        return self

    def set_data(self, data_r, data_i, wgts):
        shp = (self.nbls, self.nfreqs)
        a, b, c = self._real(data_r, shp), self._real(data_i, shp), self._real(wgts, shp)
        _lib.check(self._lib.cal_solver_set_data(self._h, _ptr(a), _ptr(b), _ptr(c)))

    def set_regularization(self, mode=None, prior_r_sum=0.0, prior_i_sum=0.0):
        """``prior_*_sum``: one number, or one per time slice of the solver (cal_solver_set_regularization_slices)."""
        code = _lib.CAL_REG_SUM if mode == "sum" else _lib.CAL_REG_NONE
        if np.ndim(prior_r_sum) == 0:
            _lib.check(self._lib.cal_solver_set_regularization(self._h, code, float(prior_r_sum), float(prior_i_sum)))
            return
        pr = np.ascontiguousarray(prior_r_sum, dtype=np.float64)
        pi = np.ascontiguousarray(prior_i_sum, dtype=np.float64)
        if pr.shape != (self.nslices,) or pi.shape != (self.nslices,):
            raise ValueError(f"expected {self.nslices} priors per component")
        _lib.check(self._lib.cal_solver_set_regularization_slices(self._h, code, _ptr(pr), _ptr(pi)))

    def set_optimizer(self, optimizer="Adamax", **opt_kwargs):
        opt_id = OPTIMIZERS[optimizer]  # KeyError for anything else, like calibration.py:571
        defaults = _OPT_DEFAULTS[optimizer]
        unknown = set(opt_kwargs) - set(defaults)
        if unknown:
            raise TypeError(f"Unexpected keyword argument(s) passed to optimizer: {sorted(unknown)}")
        kw = dict(defaults, **opt_kwargs)
        d = _lib.OptimizerDesc(opt_id, kw["learning_rate"], kw.get("beta_1", 0.9), kw.get("beta_2", 0.999), kw.get("epsilon", 1e-7),
                               kw.get("rho", 0.9), kw.get("momentum", 0.0), kw.get("initial_accumulator_value", 0.1),
                               int(bool(kw.get("nesterov", False))), 0, kw.get("learning_rate_power", -0.5),
                               kw.get("l1_regularization_strength", 0.0), kw.get("l2_regularization_strength", 0.0),
                               kw.get("l2_shrinkage_regularization_strength", 0.0), kw.get("beta", 0.0), kw.get("weight_decay_rate", 0.0))
        if optimizer == "Ftrl" and (kw["learning_rate_power"] > 0.0 or kw["initial_accumulator_value"] < 0.0):
            raise ValueError("Ftrl: learning_rate_power must be <= 0 and initial_accumulator_value >= 0 (as the Keras constructor checks)")
        _lib.check(self._lib.cal_solver_set_optimizer(self._h, C.byref(d)))

    # ---- parameters ----------------------------------------------------------------------------------------
    def set_params(self, g_r=None, g_i=None, c_r=None, c_i=None):
        gs, cs = (self.nants, self.nfreqs), (self.ncoeffs,)
        arrs = [None if a is None else self._real(a, s) for a, s in ((g_r, gs), (g_i, gs), (c_r, cs), (c_i, cs))]
        _lib.check(self._lib.cal_solver_set_params(self._h, *[_ptr(a) for a in arrs]))

    def get_params(self, which=0):
        g_r = np.empty((self.nants, self.nfreqs), dtype=self.dtype)
        g_i = np.empty_like(g_r)
        c_r = np.empty(self.ncoeffs, dtype=self.dtype)
        c_i = np.empty_like(c_r)
        _lib.check(self._lib.cal_solver_get_params(self._h, int(which), _ptr(g_r), _ptr(g_i), _ptr(c_r), _ptr(c_i)))
        return g_r, g_i, c_r, c_i

    def get_moments(self):
        g = [np.empty((self.nants, self.nfreqs), dtype=self.dtype) for _ in range(4)]
        c = [np.empty(self.ncoeffs, dtype=self.dtype) for _ in range(4)]
        t = C.c_int64(0)
        _lib.check(self._lib.cal_solver_get_moments(self._h, *[_ptr(a) for a in g + c], C.byref(t)))
        return dict(gm_r=g[0], gm_i=g[1], gv_r=g[2], gv_i=g[3], cm_r=c[0], cm_i=c[1], cv_r=c[2], cv_i=c[3], t=t.value)

    def set_moments(self, gm_r, gm_i, gv_r, gv_i, cm_r, cm_i, cv_r, cv_i, t):
        gs, cs = (self.nants, self.nfreqs), (self.ncoeffs,)
        arrs = [self._real(a, gs) for a in (gm_r, gm_i, gv_r, gv_i)] + [self._real(a, cs) for a in (cm_r, cm_i, cv_r, cv_i)]
        _lib.check(self._lib.cal_solver_set_moments(self._h, *[_ptr(a) for a in arrs], int(t)))

    # ---- compute -------------------------------------------------------------------------------------------
    def eval_loss(self):
        loss = C.c_double(0)
        _lib.check(self._lib.cal_solver_eval_loss(self._h, C.byref(loss)))
        return loss.value

    def slice_losses(self):
        """The loss of every time slice as of the last eval_loss / eval_grads."""
        out = np.zeros(self.nslices, dtype=np.float64)
        _lib.check(self._lib.cal_solver_get_slice_losses(self._h, _ptr(out)))
        return out

    def eval_grads(self):
        gg_r = np.empty((self.nants, self.nfreqs), dtype=self.dtype)
        gg_i = np.empty_like(gg_r)
        gc_r = np.empty(self.ncoeffs, dtype=self.dtype)
        gc_i = np.empty_like(gc_r)
        loss = C.c_double(0)
        _lib.check(self._lib.cal_solver_eval_grads(self._h, C.byref(loss), _ptr(gg_r), _ptr(gg_i), _ptr(gc_r), _ptr(gc_i)))
        return loss.value, gg_r, gg_i, gc_r, gc_i

    def run(self, nsteps, record=True, tol=1e-14, use_min=False, freeze_model=False):
        """``nsteps`` train steps at most.  Returns (recorded losses, stopped, nupdates)."""
        d = _lib.RunDesc(int(nsteps), int(bool(record)), int(bool(use_min)), int(bool(freeze_model)), float(tol))
        losses = np.zeros(max(int(nsteps), 1), dtype=np.float64)
        res = _lib.RunResult()
        _lib.check(self._lib.cal_solver_run(self._h, C.byref(d), _ptr(losses), C.byref(res)))
        return losses[: res.nrecorded], bool(res.stopped), res.nupdates

    def run_slices(self, nsteps, record=True, tol=1e-14, use_min=False, freeze_model=False):
        """The same loop for every time slice of the solver at once (cal_solver_run_slices): each slice records its own
        losses, applies the tolerance test and the use_min bookkeeping to its own loss and stops on its own.  Returns a
        list with one (recorded losses, stopped, nupdates) per slice.  A non-finite loss in any slice raises (code
        CAL_ERR_NONFINITE) after the others have finished."""
        d = _lib.RunDesc(int(nsteps), int(bool(record)), int(bool(use_min)), int(bool(freeze_model)), float(tol))
        losses = np.zeros((self.nslices, max(int(nsteps), 1)), dtype=np.float64)
        res = (_lib.RunResult * self.nslices)()
        _lib.check(self._lib.cal_solver_run_slices(self._h, C.byref(d), _ptr(losses), res))
        return [(losses[t, : res[t].nrecorded].copy(), bool(res[t].stopped), res[t].nupdates) for t in range(self.nslices)]

    def model(self):
        m_r = np.empty((self.nbls, self.nfreqs), dtype=self.dtype)
        m_i = np.empty_like(m_r)
        _lib.check(self._lib.cal_solver_model(self._h, _ptr(m_r), _ptr(m_i)))
        return m_r, m_i

    def data_model(self):
        """data_model (calibration.py:1593-1605): ``g_ant0 conj(g_ant1) (A c)`` of every baseline, ``[nbls, nfreqs]``."""
        m_r = np.empty((self.nbls, self.nfreqs), dtype=self.dtype)
        m_i = np.empty_like(m_r)
        _lib.check(self._lib.cal_solver_data_model(self._h, _ptr(m_r), _ptr(m_i)))
        return m_r, m_i

    def init_coeffs(self, src_r, src_i):
        shp = (self.nbls, self.nfreqs)
        a, b = self._real(src_r, shp), self._real(src_i, shp)
        _lib.check(self._lib.cal_solver_init_coeffs(self._h, _ptr(a), _ptr(b)))

    def synchronize(self):
        _lib.check(self._lib.cal_solver_synchronize(self._h))

    def set_launch_mode(self, mode="auto"):
        """How a train step is issued: "auto", "kernels" (every kernel its own launch), "one_tail" (two launches per step),
        "graph" (two-launch steps replayed from a hipGraph).  Same numbers in every mode."""
        ids = {"auto": _lib.CAL_LAUNCH_AUTO, "kernels": _lib.CAL_LAUNCH_KERNELS, "one_tail": _lib.CAL_LAUNCH_ONE_TAIL, "graph": _lib.CAL_LAUNCH_GRAPH}
        _lib.check(self._lib.cal_solver_set_launch_mode(self._h, ids[mode]))

    def timing_enable(self, enable=True):
        _lib.check(self._lib.cal_solver_timing_enable(self._h, int(bool(enable))))

    def timing_get(self):
        t = _lib.KernelTiming()
        _lib.check(self._lib.cal_solver_timing_get(self._h, C.byref(t)))
        return dict(launches=t.launches, total_ms=t.total_ms, algorithmic_bytes_per_launch=t.algorithmic_bytes_per_launch,
                    basis_bytes_per_launch=t.basis_bytes_per_launch, flops_per_launch=t.flops_per_launch,
                    kernel_path={_lib.CAL_PATH_GENERAL: "general", _lib.CAL_PATH_DENSE: "dense", _lib.CAL_PATH_DENSE_F32: "dense_f32", _lib.CAL_PATH_DENSE_SPLIT1: "dense_split1"}[t.kernel_path],
                    dense_wg_per_cu=t.dense_wg_per_cu)

    def memory_bytes(self):
        n = C.c_int64(0)
        _lib.check(self._lib.cal_solver_memory_bytes(self._h, C.byref(n)))
        return n.value

    def set_exchange_hook(self, all_reduce, rank: int, nranks: int):
        """Run the per-step exchange through ``all_reduce(array, op)`` instead of RCCL: ``array`` is a NumPy view (float32,
        float64 or int32) of the library's staging buffer, to be reduced IN PLACE over the ``nranks`` callers; ``op`` is
        "sum" or "min".  ``None`` detaches.  (cal_solver_set_exchange_hook)"""
        if all_reduce is None:
            self._hook = None
            _lib.check(self._lib.cal_solver_set_exchange_hook(self._h, None, None, 0, 1))
            return
        dtypes = {_lib.CAL_XCHG_F32: np.float32, _lib.CAL_XCHG_F64: np.float64, _lib.CAL_XCHG_I32: np.int32}

        def trampoline(ctx, buf, count, dtype, op):
            try:
                dt = np.dtype(dtypes[dtype])
                arr = np.frombuffer((C.c_char * (count * dt.itemsize)).from_address(buf), dtype=dt)
                all_reduce(arr, "min" if op == _lib.CAL_XCHG_MIN else "sum")
                return 0
            except Exception:  # noqa: BLE001 -- never unwind through the C frames: report failure to the library
                import traceback

                traceback.print_exc()
                return 1

        self._hook = _lib.EXCHANGE_FN(trampoline)  # keep the callback alive as long as the solver uses it
        _lib.check(self._lib.cal_solver_set_exchange_hook(self._h, C.cast(self._hook, C.c_void_p), None, int(rank), int(nranks)))

    def comm_size(self):
        """Ranks that take part in the exchange, counted by an all-reduce of ones over it (cal_solver_comm_size)."""
        n = C.c_int(0)
        _lib.check(self._lib.cal_solver_comm_size(self._h, C.byref(n)))
        return n.value

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(bytes(unique_id), _lib.CAL_COMM_ID_BYTES)
        _lib.check(self._lib.cal_solver_comm_init(self._h, buf, int(rank), int(nranks)))


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(_lib.CAL_COMM_ID_BYTES)
    _lib.check(_lib.load().cal_comm_unique_id(buf))
    return buf.raw
