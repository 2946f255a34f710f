"""ctypes binding of libcalamity_hip.so (include/calamity_hip.h).  No torch, no TensorFlow.

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C calamity_amd/csrc``.  Loading fails loudly
when it is missing: the product has no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcalamity_hip.so")  # the one shipped build; experiment harnesses (tools/) assign this attribute before load()

CAL_F32, CAL_F64 = 0, 1
CAL_OPT_ADAM, CAL_OPT_ADAMAX, CAL_OPT_SGD, CAL_OPT_RMSPROP, CAL_OPT_ADAGRAD, CAL_OPT_NADAM, CAL_OPT_ADADELTA, CAL_OPT_FTRL, CAL_OPT_LAMB = range(9)
CAL_REG_NONE, CAL_REG_SUM = 0, 1
CAL_LAYOUT_STREAM, CAL_LAYOUT_SHARED = 0, 1
CAL_PATH_AUTO, CAL_PATH_GENERAL, CAL_PATH_DENSE, CAL_PATH_DENSE_F32, CAL_PATH_DENSE_SPLIT1 = 0, 1, 2, 3, 4
CAL_LAUNCH_AUTO, CAL_LAUNCH_KERNELS, CAL_LAUNCH_ONE_TAIL, CAL_LAUNCH_GRAPH = 0, 1, 2, 3
CAL_COMM_ID_BYTES = 128
CAL_MAX_SLICES = 256
CAL_ERR_NONFINITE = -6


class ProblemDesc(C.Structure):
    _fields_ = [
        ("nants", C.c_int32),
        ("nfreqs", C.c_int32),
        ("ngrps", C.c_int32),
        ("nbls", C.c_int32),
        ("nbasis", C.c_int32),
        ("basis_offset", C.c_void_p),
        ("basis_nvec", C.c_void_p),
        ("basis_nrowblk", C.c_void_p),
        ("basis_data", C.c_void_p),
        ("grp_basis", C.c_void_p),
        ("grp_bl_start", C.c_void_p),
        ("bl_ant0", C.c_void_p),
        ("bl_ant1", C.c_void_p),
        ("bl_rowblk", C.c_void_p),
        ("layout", C.c_int32),
        ("kernel_path", C.c_int32),
        ("bl_alias", C.c_void_p),
        ("nslices", C.c_int32),
        ("reserved", C.c_int32),
        ("grp_var", C.c_void_p),
    ]


class OptimizerDesc(C.Structure):
    _fields_ = [
        ("optimizer", C.c_int32),
        ("learning_rate", C.c_double),
        ("beta_1", C.c_double),
        ("beta_2", C.c_double),
        ("epsilon", C.c_double),
        ("rho", C.c_double),
        ("momentum", C.c_double),
        ("initial_accumulator_value", C.c_double),
        ("nesterov", C.c_int32),
        ("reserved", C.c_int32),
        ("learning_rate_power", C.c_double),
        ("l1_regularization_strength", C.c_double),
        ("l2_regularization_strength", C.c_double),
        ("l2_shrinkage_regularization_strength", C.c_double),
        ("beta", C.c_double),
        ("weight_decay_rate", C.c_double),
    ]


class RunDesc(C.Structure):
    _fields_ = [
        ("nsteps", C.c_int32),
        ("record", C.c_int32),
        ("use_min", C.c_int32),
        ("freeze_model", C.c_int32),
        ("tol", C.c_double),
    ]


class RunResult(C.Structure):
    _fields_ = [("nrecorded", C.c_int32), ("stopped", C.c_int32), ("nupdates", C.c_int32), ("nonfinite", C.c_int32)]


class KernelTiming(C.Structure):
    _fields_ = [
        ("launches", C.c_int64),
        ("total_ms", C.c_double),
        ("algorithmic_bytes_per_launch", C.c_double),
        ("basis_bytes_per_launch", C.c_double),
        ("flops_per_launch", C.c_double),
        ("kernel_path", C.c_int32),
        ("dense_wg_per_cu", C.c_int32),
    ]


# every symbol include/calamity_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "cal_last_error": (C.c_char_p, []),
    "cal_version": (C.c_char_p, []),
    "cal_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "cal_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "cal_device_stream_peak": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cal_device_busy_clock_mhz": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "cal_solver_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int]),
    "cal_solver_destroy": (C.c_int, [_P]),
    "cal_solver_set_problem": (C.c_int, [_P, C.POINTER(ProblemDesc)]),
    "cal_solver_set_data": (C.c_int, [_P, _P, _P, _P]),
    "cal_solver_set_regularization": (C.c_int, [_P, C.c_int, C.c_double, C.c_double]),
    "cal_solver_set_regularization_slices": (C.c_int, [_P, C.c_int, _P, _P]),
    "cal_solver_set_optimizer": (C.c_int, [_P, C.POINTER(OptimizerDesc)]),
    "cal_solver_set_params": (C.c_int, [_P, _P, _P, _P, _P]),
    "cal_solver_get_params": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "cal_solver_get_moments": (C.c_int, [_P] + [_P] * 8 + [C.POINTER(C.c_int64)]),
    "cal_solver_set_moments": (C.c_int, [_P] + [_P] * 8 + [C.c_int64]),
    "cal_solver_eval_loss": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "cal_solver_eval_grads": (C.c_int, [_P, C.POINTER(C.c_double), _P, _P, _P, _P]),
    "cal_solver_get_slice_losses": (C.c_int, [_P, _P]),
    "cal_solver_run": (C.c_int, [_P, C.POINTER(RunDesc), _P, C.POINTER(RunResult)]),
    "cal_solver_run_slices": (C.c_int, [_P, C.POINTER(RunDesc), _P, C.POINTER(RunResult)]),
    "cal_solver_model": (C.c_int, [_P, _P, _P]),
    "cal_solver_data_model": (C.c_int, [_P, _P, _P]),
    "cal_weighted_square_error": (C.c_int, [C.c_int, C.c_int, C.c_int64, _P, _P, _P, _P, _P, C.POINTER(C.c_double)]),
    "cal_solver_init_coeffs": (C.c_int, [_P, _P, _P]),
    "cal_solver_synchronize": (C.c_int, [_P]),
    "cal_solver_set_launch_mode": (C.c_int, [_P, C.c_int]),
    "cal_solver_timing_enable": (C.c_int, [_P, C.c_int]),
    "cal_solver_timing_get": (C.c_int, [_P, C.POINTER(KernelTiming)]),
    "cal_solver_memory_bytes": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "cal_comm_unique_id": (C.c_int, [_P]),
    "cal_solver_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "cal_solver_set_exchange_hook": (C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    "cal_solver_comm_size": (C.c_int, [_P, C.POINTER(C.c_int)]),
}

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)  # cal_exchange_fn
CAL_XCHG_F32, CAL_XCHG_F64, CAL_XCHG_I32 = 0, 1, 2
CAL_XCHG_SUM, CAL_XCHG_MIN = 0, 1

_lib = None


class CalamityHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"calamity_hip error {code}: {message}")
        self.code = code


def load():
    """Load the shared library (once) and declare every prototype.  Raises if the build is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C calamity_amd/csrc`.  calamity_amd has no CPU fallback."
        )
    # RCCL between processes / devices shares buffers by IPC handles; hosts whose driver only supports dmabuf IPC need the legacy
    # mode off BEFORE the HSA runtime starts (first HIP call), else communicator set-up fails with "hipIpcGetMemHandle: invalid argument"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise CalamityHipError(code, load().cal_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(load().cal_device_count(C.byref(n)))
    return n.value


def device_info(device=0):
    name = C.create_string_buffer(256)
    mem = C.c_int64(0)
    cus = C.c_int32(0)
    check(load().cal_device_info(device, name, 256, C.byref(mem), C.byref(cus)))
    return dict(name=name.value.decode(), total_mem_bytes=mem.value, compute_units=cus.value)


def stream_peak(device=0, nbytes=4 << 30, reps=5):
    """Measured streaming peaks of the device in GB/s: (read-only sweep, copy)."""
    rd, cp = C.c_double(0.0), C.c_double(0.0)
    check(load().cal_device_stream_peak(device, nbytes, reps, C.byref(rd), C.byref(cp)))
    return rd.value, cp.value


def busy_clock_mhz(device=0):
    """Shader clock (MHz) the device holds while all CUs run vector FMAs."""
    mhz = C.c_double(0.0)
    check(load().cal_device_busy_clock_mhz(device, C.byref(mhz)))
    return mhz.value
