"""The time loop of calibrate_and_model_tensor as ONE batch, on one or several GPUs.

The reference fits the (polarization, time) slices of a data set one after another on one device
(/root/reference/calamity/calibration.py:1160-1167, :1244-1269; device selection :1796-1804).  The slices are independent
fits over the same array and the same modeling components, so here ``SliceBatchFitter`` puts T of them into one solver per
device (``FitProblem.nslices``: every slice keeps its own gains, weights, rms scale, priors, loss history, tolerance stop and
use_min snapshot -- include/calamity_hip.h: cal_solver_run_slices) and, with several devices, gives each device a share of the
fitting groups of EVERY slice (``distributed.partition_groups``): coefficients are local to the device that owns the group,
gains are replicated, and each train step exchanges the per-antenna gain gradients and the per-slice loss scalars once
(RCCL between distinct devices; through the library's exchange hook between workers that share a device).

The caller's process stays the only one: a worker is a thread that drives one ``HipFitSolver`` (ctypes releases the GIL for
the duration of a library call, so the workers' launches and collectives run side by side).
"""
import threading

import numpy as np

from . import distributed
from .problem import FitProblem
from .solver import HipFitSolver, comm_unique_id


def replicate_slices(prob, nt, groups=None, share_tiles=True):
    """The (data-less) problem of ``nt`` time slices of ``prob`` -- optionally only the fitting groups ``groups`` of every
    slice (a device's share).  Slice ``t`` uses antennas ``[t * nants, (t + 1) * nants)``; its baselines read the basis tiles of
    slice 0's (``bl_alias``: the STREAM layout then holds one copy and processes the slices of a baseline together).
    Returns (FitProblem with ``nslices = nt``, baseline indices of one slice, coefficient indices of one slice)."""
    groups = np.arange(prob.ngrps) if groups is None else np.asarray(groups)

    def runs(starts, counts):  # concatenation of arange(start, start + count) without a python loop over 61 075 groups
        starts, counts = np.asarray(starts, dtype=np.int64), np.asarray(counts, dtype=np.int64)
        ends = np.cumsum(counts)
        return np.repeat(starts - (ends - counts), counts) + np.arange(ends[-1] if len(ends) else 0, dtype=np.int64)

    starts, nbl_g = prob.grp_bl_start[groups], np.diff(prob.grp_bl_start)[groups]
    bl = runs(starts, nbl_g)
    coff = prob.grp_coff
    cidx = runs(coff[groups], coff[groups + 1] - coff[groups])
    used = np.unique(prob.grp_basis[groups])
    remap = -np.ones(len(prob.basis), dtype=np.int64)
    remap[used] = np.arange(len(used))
    nb = len(bl)
    single = bool(np.all(nbl_g == 1))
    sub = FitProblem(
        nants=prob.nants * nt,
        nfreqs=prob.nfreqs,
        basis=[prob.basis[u] for u in used],
        grp_basis=np.tile(remap[prob.grp_basis[groups]].astype(np.int32), nt),
        grp_bl_start=np.concatenate([[0], np.cumsum(np.tile(nbl_g, nt))]).astype(np.int32),
        bl_ant0=np.concatenate([prob.bl_ant0[bl] + t * prob.nants for t in range(nt)]).astype(np.int32),
        bl_ant1=np.concatenate([prob.bl_ant1[bl] + t * prob.nants for t in range(nt)]).astype(np.int32),
        bl_rowblk=np.tile(prob.bl_rowblk[bl], nt).astype(np.int32),
        data_r=None,
        data_i=None,
        wgts=None,
        bl_alias=(np.concatenate([np.full(nb, -1, dtype=np.int32)] + [np.arange(nb, dtype=np.int32)] * (nt - 1))
                  if share_tiles and single and nt > 1 else None),
        nslices=nt,
        chunk_of_grp=None if prob.chunk_of_grp is None else np.tile(np.asarray(prob.chunk_of_grp)[groups], nt).astype(np.int32),
    )
    return sub, bl, cidx


class _HostExchange:
    """In-process all-reduce between worker threads (cal_solver_set_exchange_hook): every worker leaves a view of its staging
    buffer, all of them reduce the views in rank order -- the same arithmetic on every worker -- and each writes the result
    back into its own buffer."""

    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n

    def hook(self, rank):
        def all_reduce(arr, op):
            self.slots[rank] = arr
            self.barrier.wait(timeout=600)
            if op == "min":
                out = np.minimum.reduce([self.slots[r] for r in range(self.n)])
            else:
                out = self.slots[0].copy()
                for r in range(1, self.n):
                    out += self.slots[r]
            self.barrier.wait(timeout=600)
            arr[:] = out
            self.barrier.wait(timeout=600)

        return all_reduce

    def abort(self):
        self.barrier.abort()


class SliceBatchFitter:
    """``nt`` time slices of the data-less single-slice problem ``prob`` on ``devices`` (one worker per entry; an entry may
    repeat, the workers then share that GPU and exchange through host memory).  Arrays handed in and out are GLOBAL and
    slice-major: per-sample arrays ``[nt * nbls, nfreqs]``, gains ``[nt * nants, nfreqs]``, coefficients ``[nt * ncoeffs]``."""

    def __init__(self, prob, nt, dtype=np.float32, layout="shared", devices=(0,), kernel_path="auto", communicator_of_one=False):
        """``communicator_of_one``: with a single device, join a one-rank RCCL communicator anyway (from the worker thread, like the
        workers of several devices do) so that the exchange path -- the set-up agreement, an all-reduce per step -- runs on a
        one-GPU box; the numbers are those of the plain fit."""
        self.prob, self.nt, self.dtype = prob, int(nt), np.dtype(dtype)
        self.devices = [int(d) for d in devices]
        D = self.nworkers = len(self.devices)
        if D > 1:
            shares = distributed.partition_groups(prob.grp_nvec, prob.grp_basis, np.diff(prob.grp_bl_start), D)
            if any(len(s) == 0 for s in shares):
                raise ValueError(f"{prob.ngrps} fitting groups cannot be shared out over {D} devices")
        else:
            shares = [None]
        self.subs, self.rows, self.cidx = [], [], []
        for r in range(D):
            sub, bl, cidx = replicate_slices(prob, self.nt, shares[r])
            self.subs.append(sub)
            # rows / coefficients of this worker in the global slice-major arrays
            self.rows.append(np.concatenate([bl + t * prob.nbls for t in range(self.nt)]))
            self.cidx.append(np.concatenate([cidx + t * prob.ncoeffs for t in range(self.nt)]))
        self.solvers, self._host_exchange = [], None
        try:
            self._set_up(prob, layout, kernel_path, communicator_of_one)
        except BaseException:
            self.close()  # (the solvers created so far: their device memory goes back at once)
            raise

    def _set_up(self, prob, layout, kernel_path, communicator_of_one):
        D = self.nworkers
        for d in self.devices:  # (every device index is checked here, before any worker waits for a peer in the communicator set-up)
            self.solvers.append(HipFitSolver(dtype=self.dtype, device=d))
        if D == 1 and communicator_of_one:
            uid = comm_unique_id()
            self._each_threaded(lambda r, s: s.comm_init(uid, 0, 1))
        if D > 1:
            if len(set(self.devices)) == D:
                uid = comm_unique_id()
                self._each(lambda r, s: s.comm_init(uid, r, D))
            else:
                self._host_exchange = _HostExchange(D)
                for r, s in enumerate(self.solvers):
                    s.set_exchange_hook(self._host_exchange.hook(r), r, D)
        # (with a communicator set_problem ends in an agreement between the workers: concurrently)
        self._each(lambda r, s: s.set_problem(self.subs[r], layout=layout, kernel_path=kernel_path))
        self.nbls, self.ncoeffs, self.nants, self.nfreqs = prob.nbls * self.nt, prob.ncoeffs * self.nt, prob.nants * self.nt, prob.nfreqs

    def _each(self, fn):
        """fn(rank, solver) on every worker, side by side; the first exception is raised in the caller."""
        if self.nworkers == 1:
            return [fn(0, self.solvers[0])]
        return self._each_threaded(fn)

    def _each_threaded(self, fn):
        out, errs = [None] * self.nworkers, [None] * self.nworkers

        def work(r):
            try:
                out[r] = fn(r, self.solvers[r])
            except BaseException as e:  # noqa: BLE001 -- handed to the caller below
                errs[r] = e
                if self._host_exchange is not None:
                    self._host_exchange.abort()  # the others must not wait for this worker in an exchange

        threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(self.nworkers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out

    def close(self):
        for s in self.solvers:
            s.close()

    def memory_bytes(self):
        return max(s.memory_bytes() for s in self.solvers)

    # ---- pass-throughs on global arrays
    def _take(self, a, r):
        return a if self.nworkers == 1 else np.take(a, self.rows[r], axis=0)

    def set_data(self, data_r, data_i, wgts):
        self._each(lambda r, s: s.set_data(self._take(data_r, r), self._take(data_i, r), self._take(wgts, r)))

    def init_coeffs(self, src_r, src_i):
        self._each(lambda r, s: s.init_coeffs(self._take(src_r, r), self._take(src_i, r)))

    def set_params(self, g_r=None, g_i=None, c_r=None, c_i=None):
        def one(r, s):
            sel = (lambda c: None if c is None else (c if self.nworkers == 1 else c[self.cidx[r]]))
            s.set_params(g_r, g_i, sel(c_r), sel(c_i))

        self._each(one)

    def get_params(self, which=0):
        outs = self._each(lambda r, s: s.get_params(which))
        if self.nworkers == 1:
            return outs[0]
        c_r = np.empty(self.ncoeffs, dtype=self.dtype)
        c_i = np.empty(self.ncoeffs, dtype=self.dtype)
        for r, o in enumerate(outs):
            c_r[self.cidx[r]] = o[2]
            c_i[self.cidx[r]] = o[3]
        return outs[0][0], outs[0][1], c_r, c_i  # the gains are replicated

    def model(self):
        outs = self._each(lambda r, s: s.model())
        if self.nworkers == 1:
            return outs[0]
        m_r = np.empty((self.nbls, self.nfreqs), dtype=self.dtype)
        m_i = np.empty_like(m_r)
        for r, o in enumerate(outs):
            m_r[self.rows[r]] = o[0]
            m_i[self.rows[r]] = o[1]
        return m_r, m_i

    def set_regularization(self, mode=None, prior_r=None, prior_i=None):
        if mode == "sum":
            self._each(lambda r, s: s.set_regularization("sum", np.asarray(prior_r, dtype=np.float64), np.asarray(prior_i, dtype=np.float64)))
        else:
            self._each(lambda r, s: s.set_regularization(None))

    def set_optimizer(self, optimizer, **kw):
        self._each(lambda r, s: s.set_optimizer(optimizer, **kw))

    def timing_enable(self, on):
        self._each(lambda r, s: s.timing_enable(on))

    def timing_get(self):
        return self.solvers[0].timing_get()

    def run_slices(self, nsteps, **kw):
        """Per slice (recorded losses, stopped, nupdates); the loop decisions are taken on the all-reduced sums, so every
        worker reports the same."""
        return self._each(lambda r, s: s.run_slices(nsteps, **kw))[0]
