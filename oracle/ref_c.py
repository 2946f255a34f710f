"""ctypes wrapper of oracle/ref_c.c (TEST INFRASTRUCTURE; PARITY UNPINNED, see ref_numpy.py).

Takes the same FitProblem as the HIP solver and returns NumPy arrays, so tests can put the three implementations
(NumPy restatement, this C/OpenMP restatement, the HIP path) side by side.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        # ORACLE_REF_C_LIB: the sanitizer build (oracle/Makefile: libref_c_san.so), chosen by the test that runs under it
        path = os.environ.get("ORACLE_REF_C_LIB") or os.path.join(_HERE, "libref_c.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, os.path.basename(path)], check=True, capture_output=True)
        _LIB = C.CDLL(path)
    return _LIB


def _problem_struct(real):
    class P(C.Structure):
        _fields_ = [("nants", C.c_int), ("nfreqs", C.c_int), ("ngrps", C.c_int), ("nbls", C.c_int),
                    ("basis_off", C.c_void_p), ("basis_nvec", C.c_void_p), ("basis_data", C.c_void_p),
                    ("grp_basis", C.c_void_p), ("grp_bl_start", C.c_void_p), ("grp_coff", C.c_void_p),
                    ("bl_ant0", C.c_void_p), ("bl_ant1", C.c_void_p), ("bl_rowblk", C.c_void_p),
                    ("data_r", C.c_void_p), ("data_i", C.c_void_p), ("wgts", C.c_void_p)]
    return P


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class CRef:
    """The fit of one FitProblem in C: loss_grads() and fit() mirror HipFitSolver.eval_grads() / run()."""

    def __init__(self, p, dtype=np.float64, nthreads=0):
        self.lib = load()
        self.dtype = np.dtype(dtype)
        self.suf = "f32" if self.dtype == np.float32 else "f64"
        self.p = p
        self.nthreads = int(nthreads)
        keep = self._keep = {}
        keep["basis_nvec"] = np.asarray([b.shape[1] for b in p.basis], dtype=np.int32)
        sizes = np.asarray([b.size for b in p.basis], dtype=np.int64)
        keep["basis_off"] = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        keep["basis_data"] = np.concatenate([np.ascontiguousarray(b, dtype=self.dtype).ravel() for b in p.basis])
        for k in ("grp_basis", "grp_bl_start", "bl_ant0", "bl_ant1", "bl_rowblk"):
            keep[k] = np.ascontiguousarray(getattr(p, k), dtype=np.int32)
        keep["grp_coff"] = np.ascontiguousarray(p.grp_coff, dtype=np.int64)
        for k in ("data_r", "data_i", "wgts"):
            keep[k] = np.ascontiguousarray(getattr(p, k), dtype=self.dtype)
        S = _problem_struct(self.dtype)
        self.desc = S(p.nants, p.nfreqs, p.ngrps, p.nbls, *[_ptr(keep[k]) for k in (
            "basis_off", "basis_nvec", "basis_data", "grp_basis", "grp_bl_start", "grp_coff", "bl_ant0", "bl_ant1",
            "bl_rowblk", "data_r", "data_i", "wgts")])
        self.reg, self.prior = 0, (0.0, 0.0)

    def max_threads(self):
        return getattr(self.lib, "ref_max_threads_" + self.suf)()

    def set_regularization(self, mode, prior_r=0.0, prior_i=0.0):
        self.reg = 1 if mode == "sum" else 0
        self.prior = (float(prior_r), float(prior_i))

    def _cast(self, *arrs):
        return [np.ascontiguousarray(a, dtype=self.dtype).copy() for a in arrs]

    def loss_grads(self, g_r, g_i, c_r, c_i, grads=True):
        g_r, g_i, c_r, c_i = self._cast(g_r, g_i, c_r, c_i)
        loss = C.c_double()
        out = [np.empty_like(g_r), np.empty_like(g_i), np.empty_like(c_r), np.empty_like(c_i)] if grads else [None] * 4
        fn = getattr(self.lib, "ref_loss_grads_" + self.suf)
        rc = fn(C.byref(self.desc), _ptr(g_r), _ptr(g_i), _ptr(c_r), _ptr(c_i), self.reg, C.c_double(self.prior[0]),
                C.c_double(self.prior[1]), C.byref(loss), *[_ptr(o) if o is not None else None for o in out],
                self.nthreads)
        assert rc == 0
        return (loss.value, *out) if grads else loss.value

    def fit(self, g_r, g_i, c_r, c_i, nsteps, optimizer="Adam", learning_rate=1e-3, beta_1=0.9, beta_2=0.999,
            epsilon=1e-7, freeze_model=False, moments=None, t0=0):
        """nsteps updates from the given start; returns (g_r, g_i, c_r, c_i, losses, moments)."""
        g_r, g_i, c_r, c_i = self._cast(g_r, g_i, c_r, c_i)
        if moments is None:
            moments = [np.zeros_like(a) for a in (g_r, g_r, g_i, g_i, c_r, c_r, c_i, c_i)]
        mom = (C.c_void_p * 8)(*[_ptr(m) for m in moments])
        losses = np.zeros(nsteps)
        fn = getattr(self.lib, "ref_fit_" + self.suf)
        rc = fn(C.byref(self.desc), _ptr(g_r), _ptr(g_i), _ptr(c_r), _ptr(c_i), C.c_longlong(c_r.size), self.reg,
                C.c_double(self.prior[0]), C.c_double(self.prior[1]), {"Adam": 0, "Adamax": 1}[optimizer],
                C.c_double(learning_rate), C.c_double(beta_1), C.c_double(beta_2), C.c_double(epsilon), C.c_longlong(t0),
                int(nsteps), int(bool(freeze_model)), mom, _ptr(losses), self.nthreads)
        assert rc == 0
        return g_r, g_i, c_r, c_i, losses, moments
