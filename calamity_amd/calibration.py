"""Drop-in counterparts of the fit path of /root/reference/calamity/calibration.py, running on the MI355X HIP
library (include/calamity_hip.h) instead of TensorFlow.

Same function names, argument meaning and error behaviour as the reference for the gain + foreground
gradient-descent fitter: ``calibrate_and_model_dpss`` (:1503-1584) -> ``calibrate_and_model_tensor`` (:963-1331)
-> ``fit_gains_and_foregrounds`` (:447-738).  Differences a user can observe:

* tensors are NumPy arrays, and the foreground components are kept ragged (``problem.FitProblem``: true vector counts,
  identical basis blocks stored once) instead of one zero-padded ``(nvecs, ngrps, nbls, nfreqs)`` tensor per chunk
  (:140-146, :167); ``fg_comps`` arguments accept either form;
* ``graph_mode`` / ``graph_args_dict`` are accepted and ignored (there is no tracing compiler on this path);
* ``n_profile_steps`` writes HIP-event kernel timings as JSON into ``profile_log_dir`` instead of a TF profile;
* optimizers: the whole ``OPTIMIZERS`` table (:17-27) -- "Adamax", "Adam", "SGD", "RMSprop", "Adagrad", "Adadelta", "Nadam", "Ftrl" with
  the Keras (OptimizerV2) semantics, defaults and constructor arguments, "LAMB" with those of tensorflow-addons; every other name
  raises ``KeyError`` like ``OPTIMIZERS[...]`` (:571).

There is no CPU fallback: without the HIP library / a GPU these functions raise.
"""
import argparse
import collections
import concurrent.futures
import copy
import datetime
import json
import os
import threading

import numpy as np

from . import cal_utils, modeling, utils
from .problem import FitProblem, coeffs_from_chunks, coeffs_to_chunks, problem_from_chunks
from .solver import OPTIMIZERS, HipFitSolver
from .utils import PBARS, echo
from .uvcompat import gain4, is_uvcal, is_uvdata, polstr2num, vis3


# ------------------------------------------------------------------------------------------------------------------
# layout: calibration.py:30-190
# ------------------------------------------------------------------------------------------------------------------
def chunk_fg_comp_dict_by_nbls(fg_model_comps_dict, use_redundancy=False, grp_size_threshold=5):
    """Group fitting groups into chunks keyed ``(nbl, maxvecs)`` -- behaviour of calibration.py:30-101.

    Without ``use_redundancy`` a fitting group whose redundant groups all have the same length (and that has fewer
    than ``grp_size_threshold`` of them) is split into one fitting group per redundant copy, all sharing the same
    modeling vectors (:69-81).
    """
    comps = dict(fg_model_comps_dict)
    if not use_redundancy:
        # The reference pops every group it splits and appends the pieces (:73-81) -- a single baseline is "split" into itself, i.e. moved
        # to the end.  Same order here (it fixes the layout of fg_model_comps, corr_inds and the coefficient arrays): the groups that stay
        # whole first, then the pieces in the order of their groups; without 61 075 pops and re-inserts for HERA-350's single baselines.
        kept, moved = {}, {}
        for fit_grp, vectors in comps.items():
            if len(fit_grp) == 1 and len(fit_grp[0]) == 1:
                moved[fit_grp] = vectors  # one baseline: its only piece is itself
                continue
            rlens = np.asarray([len(red_grp) for red_grp in fit_grp])
            if np.allclose(rlens, np.mean(rlens)) and len(rlens) < grp_size_threshold:
                for rednum in range(int(rlens[0])):
                    moved[tuple((red_grp[rednum],) for red_grp in fit_grp)] = vectors
            else:
                kept[fit_grp] = vectors
        comps = {**kept, **moved}
    by_nbl, maxvecs = {}, {}
    for fit_grp, vectors in comps.items():
        nbl = sum(len(red_grp) for red_grp in fit_grp)
        by_nbl.setdefault(nbl, []).append(fit_grp)
        maxvecs[nbl] = max(maxvecs.get(nbl, 0), vectors.shape[1])
    return {(nbl, maxvecs[nbl]): {k: comps[k] for k in grps} for nbl, grps in by_nbl.items()}


def tensorize_fg_model_comps_dict(
    fg_model_comps_dict,
    ants_map,
    nfreqs,
    use_redundancy=False,
    dtype=np.float32,
    notebook_progressbar=False,
    verbose=False,
    grp_size_threshold=5,
):
    """Modeling-component dictionary -> (ragged ``FitProblem`` without data, ``corr_inds``).

    Counterpart of calibration.py:104-190.  ``corr_inds[chunk][group][baseline] = (i, j)`` is built exactly like the
    reference (:169-188); the components are NOT expanded into the zero-padded ``(nvecs, ngrps, nbls, nfreqs)``
    tensor (:167): each group keeps its own vector count and arrays that are the same object in the dictionary
    (the operator cache of modeling.py:291-301) are stored once.
    """
    echo(f"{datetime.datetime.now()} Computing foreground components matrices...\n", verbose=verbose)
    chunked = chunk_fg_comp_dict_by_nbls(fg_model_comps_dict, use_redundancy=use_redundancy, grp_size_threshold=grp_size_threshold)
    basis, basis_id = [], {}
    grp_basis, grp_bl_start = [], [0]
    bl_ant0, bl_ant1, bl_rowblk = [], [], []
    corr_inds, chunk_of_grp, pos_in_chunk, chunk_shapes = [], [], [], []
    for cnum, (nbls, nvecs) in enumerate(chunked):
        corr_inds_chunk = []
        for grpnum, (modeling_grp, vectors) in enumerate(chunked[(nbls, nvecs)].items()):
            if vectors.shape[0] != len(modeling_grp) * nfreqs:
                raise ValueError(
                    f"modeling vectors of a fitting group with {len(modeling_grp)} redundant groups must have "
                    f"{len(modeling_grp) * nfreqs} rows, got {vectors.shape[0]}"
                )
            if id(vectors) not in basis_id:
                basis_id[id(vectors)] = len(basis)
                basis.append(np.ascontiguousarray(vectors, dtype=np.float64))
            grp_basis.append(basis_id[id(vectors)])
            corr_inds_grp = []
            for rgrpnum, red_grp in enumerate(modeling_grp):
                for ap in red_grp:
                    i, j = ants_map[ap[0]], ants_map[ap[1]]
                    corr_inds_grp.append((i, j))
                    bl_ant0.append(i)
                    bl_ant1.append(j)
                    bl_rowblk.append(rgrpnum)
            grp_bl_start.append(grp_bl_start[-1] + len(corr_inds_grp))
            corr_inds_chunk.append(corr_inds_grp)
            chunk_of_grp.append(cnum)
            pos_in_chunk.append(grpnum)
        corr_inds.append(corr_inds_chunk)
        chunk_shapes.append((nvecs, len(corr_inds_chunk), nbls))
    nbl_total = len(bl_ant0)
    fg_model_comps = FitProblem(
        nants=len(ants_map),
        nfreqs=int(nfreqs),
        basis=basis,
        grp_basis=np.asarray(grp_basis, dtype=np.int32),
        grp_bl_start=np.asarray(grp_bl_start, dtype=np.int32),
        bl_ant0=np.asarray(bl_ant0, dtype=np.int32),
        bl_ant1=np.asarray(bl_ant1, dtype=np.int32),
        bl_rowblk=np.asarray(bl_rowblk, dtype=np.int32),
        data_r=None,
        data_i=None,
        wgts=None,
        chunk_of_grp=np.asarray(chunk_of_grp, dtype=np.int32),
        pos_in_chunk=np.asarray(pos_in_chunk, dtype=np.int32),
        chunk_shapes=chunk_shapes,
    )
    assert nbl_total == fg_model_comps.nbls
    return fg_model_comps, corr_inds


# ------------------------------------------------------------------------------------------------------------------
# data / gains <-> arrays: calibration.py:193-399
# ------------------------------------------------------------------------------------------------------------------
def _baseline_rows(uvdata, prob, ants_map, time):
    """Row of every baseline of ``prob`` in the blt axis of ``uvdata`` at ``time``, and whether the data hold the pair in the
    reversed order (then the conjugate is wanted, calibration.py:263-278) -- by one (antenna, antenna) -> row table instead of a
    dictionary lookup per baseline.  KeyError if a pair is in the data in neither order, like ``_key2inds``."""
    cache = prob.__dict__.setdefault("_row_cache", {})
    key = (id(uvdata), float(time), int(getattr(uvdata, "Nblts", 0)))
    hit = cache.get(key)
    if hit is not None and hit[0] is uvdata:
        return hit[1], hit[2]
    tsel = np.where(np.isclose(np.asarray(uvdata.time_array), time, rtol=0.0, atol=1e-7))[0]
    nants = len(ants_map)
    lut_keys = np.fromiter(ants_map.keys(), dtype=np.int64, count=nants)
    lut_vals = np.fromiter(ants_map.values(), dtype=np.int64, count=nants)
    order = np.argsort(lut_keys)
    a1 = np.asarray(uvdata.ant_1_array)[tsel].astype(np.int64)
    a2 = np.asarray(uvdata.ant_2_array)[tsel].astype(np.int64)

    def to_index(a):  # antenna number -> index of ants_map (-1: not a gain antenna)
        pos = np.clip(np.searchsorted(lut_keys[order], a), 0, nants - 1)
        ok = lut_keys[order][pos] == a
        return np.where(ok, lut_vals[order][pos], -1)

    i1, i2 = to_index(a1), to_index(a2)
    ok = (i1 >= 0) & (i2 >= 0)
    table = np.full((nants, nants), -1, dtype=np.int64)
    table[i1[ok], i2[ok]] = tsel[ok]
    b0, b1 = np.asarray(prob.bl_ant0, dtype=np.int64), np.asarray(prob.bl_ant1, dtype=np.int64)
    rows = table[b0, b1]
    conj = rows < 0
    if conj.any():
        rows = np.where(conj, table[b1, b0], rows)
        if (rows < 0).any():
            n = int(np.where(rows < 0)[0][0])
            inv = {v: k for k, v in ants_map.items()}
            raise KeyError((inv[int(b0[n])], inv[int(b1[n])]))
    if len(cache) > 64:
        cache.clear()
    cache[key] = (uvdata, rows, conj)
    return rows, conj


def _time_ind(times, inds, time):
    return inds[np.where(np.isclose(np.asarray(times)[inds], time, rtol=0.0, atol=1e-7))[0][0]]


class _Baselines:
    """The antenna index pairs of ``corr_inds`` as flat arrays, in chunk -> group -> baseline order (= the ragged baseline order
    of the FitProblem built from the same dictionary)."""

    def __init__(self, corr_inds):
        pairs = [np.asarray(chunk, dtype=np.int64).reshape(-1, 2) for chunk in corr_inds]
        flat = np.concatenate(pairs) if len(pairs) else np.zeros((0, 2), dtype=np.int64)
        self.bl_ant0, self.bl_ant1 = flat[:, 0], flat[:, 1]
        self.shapes = [(len(chunk), len(chunk[0])) for chunk in corr_inds]


def _baselines_of(corr_inds):
    hit = _baselines_of.cache.get(id(corr_inds))
    if hit is None or hit[0] is not corr_inds:
        if len(_baselines_of.cache) > 16:
            _baselines_of.cache.clear()
        hit = (corr_inds, _Baselines(corr_inds))
        _baselines_of.cache[id(corr_inds)] = hit
    return hit[1]


_baselines_of.cache = {}


def _tensorize_flat(uvdata, bls, ants_map, polarization, time, data_scale_factor=1.0, weights=None, nsamples_in_weights=False,
                    dtype=np.float32, want_weights=True):
    """The arithmetic of tensorize_data (calibration.py:193-310) on flat ``(nbls, nfreqs)`` arrays in the baseline order of
    ``bls`` (a FitProblem or _Baselines): data of a pair that only exists in reversed order are conjugated (:263-278); weights
    are ``~flags`` (``* nsamples`` / ``* UVFlag.weights``) divided by their sum over ALL baselines and channels (:282-303).
    Returns (data_r, data_i, wgts or None)."""
    rows, conj = _baseline_rows(uvdata, bls, ants_map, time)
    pols = np.asarray(uvdata.polarization_array)
    polnum = polstr2num(polarization, x_orientation=uvdata.x_orientation)
    swap = {-7: -8, -8: -7, -3: -4, -4: -3}  # conjugating swaps the feeds of cross-hand products
    pind = int(np.where(pols == polnum)[0][0])
    pind_conj = np.where(pols == swap.get(polnum, polnum))[0]
    any_conj = bool(conj.any())
    if any_conj and len(pind_conj) == 0:
        raise KeyError(f"polarization {polarization}: the conjugate product is not in the data")
    if weights is not None and want_weights:
        wrows, _ = _baseline_rows(weights, bls, ants_map, time)
        wpol = int(np.where(np.asarray(weights.polarization_array) == polstr2num(polarization, x_orientation=weights.x_orientation))[0][0])
    nb, nf = len(rows), uvdata.Nfreqs
    d_r = np.empty((nb, nf), dtype=dtype)
    d_i = np.empty((nb, nf), dtype=dtype)
    w = np.empty((nb, nf), dtype=dtype) if want_weights else None
    vis, flg = vis3(np.asarray(uvdata.data_array)), vis3(np.asarray(uvdata.flag_array))
    nsm = vis3(np.asarray(uvdata.nsample_array)) if (nsamples_in_weights and want_weights) else None
    wts = vis3(np.asarray(weights.weights_array)) if (weights is not None and want_weights) else None
    row_sums = np.zeros(nb, dtype=np.float64)  # per baseline: the total below does not depend on how the rows are chunked

    def rows_chunk(lo, hi):
        # gather + scale + split of a run of baselines (memory-bound NumPy passes: one thread per run, utils.for_row_chunks)
        r, cj = rows[lo:hi], conj[lo:hi]
        # (row gathers with np.take from the polarization's plane: NumPy's fancy indexing with two index arrays around a
        # slice is an order of magnitude slower on complex data)
        data = np.take(vis[:, :, pind], r, axis=0)
        if any_conj and cj.any():  # pairs held in the reversed order: the conjugate product's column, conjugated
            pc = int(pind_conj[0])
            data[cj] = np.take(vis[:, :, pc], r[cj], axis=0)
        data /= data_scale_factor
        d_r[lo:hi] = data.real
        d_i[lo:hi] = data.imag
        if any_conj:
            d_i[lo:hi][cj] *= -1
        if not want_weights:
            return
        keep = ~np.take(flg[:, :, pind], r, axis=0)
        ns = np.take(nsm[:, :, pind], r, axis=0) if nsm is not None else None
        if any_conj and cj.any():
            pc = int(pind_conj[0])
            keep[cj] = ~np.take(flg[:, :, pc], r[cj], axis=0)
            if ns is not None:
                ns[cj] = np.take(nsm[:, :, pc], r[cj], axis=0)
        ww = keep.astype(dtype) if wts is None else np.take(wts[:, :, wpol], wrows[lo:hi], axis=0).astype(dtype) * keep
        if ns is not None:
            ww = ww * ns
        w[lo:hi] = ww
        row_sums[lo:hi] = np.sum(w[lo:hi], axis=1, dtype=np.float64)

    utils.for_row_chunks(rows_chunk, nb)
    if want_weights:
        wgtsum = float(np.sum(row_sums))
        w = (w / wgtsum).astype(dtype)
    return d_r, d_i, w


def tensorize_data(
    uvdata,
    corr_inds,
    ants_map,
    polarization,
    time,
    data_scale_factor=1.0,
    weights=None,
    nsamples_in_weights=False,
    dtype=np.float32,
):
    """UVData -> per-chunk ``(ngrps, nbls, nfreqs)`` arrays ``data_r, data_i, wgts`` -- calibration.py:193-310.

    Data of a pair that only exists in reversed order are conjugated (:263-278); weights are ``~flags``
    (``* nsamples`` / ``* UVFlag.weights``) divided by their sum over ALL baselines and channels (:282-303).
    """
    bls = _baselines_of(corr_inds)
    d_r, d_i, w = _tensorize_flat(uvdata, bls, ants_map, polarization, time, data_scale_factor=data_scale_factor, weights=weights,
                                  nsamples_in_weights=nsamples_in_weights, dtype=dtype)
    data_r, data_i, wgts, lo = [], [], [], 0
    nf = uvdata.Nfreqs
    for ngrps, nbls in bls.shapes:
        hi = lo + ngrps * nbls
        data_r.append(d_r[lo:hi].reshape(ngrps, nbls, nf))
        data_i.append(d_i[lo:hi].reshape(ngrps, nbls, nf))
        wgts.append(w[lo:hi].reshape(ngrps, nbls, nf))
        lo = hi
    return data_r, data_i, wgts


def tensorize_gains(uvcal, polarization, time, dtype=np.float32):
    """UVCal -> ``(Nants, Nfreqs)`` real and imaginary gain arrays -- calibration.py:369-399."""
    polnum = np.where(np.asarray(uvcal.jones_array) == polstr2num(polarization, x_orientation=uvcal.x_orientation))[0][0]
    gindt = np.where(np.isclose(uvcal.time_array, time, atol=1e-7, rtol=0.0))[0][0]
    g = gain4(uvcal.gain_array)[:, :, gindt, polnum]
    return np.ascontiguousarray(g.real, dtype=dtype), np.ascontiguousarray(g.imag, dtype=dtype)


def renormalize(uvdata_reference_model, uvdata_deconv, gains, polarization, time, additional_flags=None):
    """Remove the arbitrary amplitude of the deconvolved model and gains -- calibration.py:313-366 (the phase factor is
    computed by the reference but deliberately not applied, :359).  Modifies ``uvdata_deconv`` and ``gains``.
    ``scale = sqrt(nanmean |reference / deconvolved|^2)`` over the samples no flag array marks, non-finite ratios left out
    (:355-358) -- accumulated baseline by baseline on the host's cores instead of through three boolean-indexed copies of the slice."""
    polnum_data = np.where(uvdata_deconv.polarization_array == polstr2num(polarization, x_orientation=uvdata_deconv.x_orientation))[0][0]
    rows = np.where(np.isclose(uvdata_deconv.time_array, time, atol=1e-7, rtol=0.0))[0]
    dec, dflag = vis3(uvdata_deconv.data_array), vis3(uvdata_deconv.flag_array)
    ref, rflag = vis3(np.asarray(uvdata_reference_model.data_array)), vis3(np.asarray(uvdata_reference_model.flag_array))
    aflag = vis3(np.asarray(additional_flags)) if additional_flags is not None else None
    ssq = np.zeros(len(rows), dtype=np.float64)
    cnt = np.zeros(len(rows), dtype=np.int64)

    def chunk(lo, hi):
        r = rows[lo:hi]
        sel = ~(np.take(dflag[:, :, polnum_data], r, axis=0) | np.take(rflag[:, :, polnum_data], r, axis=0))
        if aflag is not None:
            sel &= ~np.take(aflag[:, :, polnum_data], r, axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.take(ref[:, :, polnum_data], r, axis=0) / np.take(dec[:, :, polnum_data], r, axis=0)
            p = ratio.real * ratio.real + ratio.imag * ratio.imag
        sel &= np.isfinite(ratio.real) & np.isfinite(ratio.imag)  # data_ratio[~isfinite] = nan, then nanmean (:355-358)
        ssq[lo:hi] = np.sum(np.where(sel, p, 0.0), axis=1)
        cnt[lo:hi] = np.count_nonzero(sel, axis=1)

    utils.for_row_chunks(chunk, len(rows))
    n = int(cnt.sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        scale_factor = np.sqrt(np.float64(ssq.sum()) / n) if n else np.float64("nan")

    def scale(lo, hi):
        dec[rows[lo:hi], :, polnum_data] *= scale_factor

    utils.for_row_chunks(scale, len(rows))
    polnum_gains = np.where(np.asarray(gains.jones_array) == polstr2num(polarization, x_orientation=uvdata_deconv.x_orientation))[0][0]
    gindt = np.where(np.isclose(gains.time_array, time, atol=1e-7, rtol=0.0))[0][0]
    gains.gain_array[..., gindt, polnum_gains] *= (scale_factor) ** -0.5


# ------------------------------------------------------------------------------------------------------------------
# the solver behind a set of components
# ------------------------------------------------------------------------------------------------------------------
def _as_problem(fg_comps, corr_inds, nants):
    """Accept the ragged FitProblem or the reference's list of zero-padded chunk tensors."""
    if isinstance(fg_comps, FitProblem):
        return fg_comps
    if corr_inds is None:  # antenna indices are irrelevant to the caller (coefficient initialisation)
        corr_inds = [[[(0, 0)] * np.asarray(c).shape[2] for _ in range(np.asarray(c).shape[1])] for c in fg_comps]
        nants = 1
    dummy = [np.zeros(np.asarray(c).shape[1:]) for c in fg_comps]
    prob = problem_from_chunks(nants, fg_comps, corr_inds, dummy, dummy, dummy)
    prob.data_r = prob.data_i = prob.wgts = None
    return prob


def _flatten(chunks, prob):
    """Per-chunk ``(ngrps, nbls, nfreqs)`` arrays -> ``(nbls_total, nfreqs)`` in the ragged baseline order."""
    return np.concatenate([np.asarray(c).reshape(-1, prob.nfreqs) for c in chunks])


# device selection of the file driver (calibration.py:1741-1753): which GPU the solvers are created on, and a cap on
# the device memory one fit may take
_DEVICE = {"index": None, "memory_limit_gib": None}


def _wake_device(devices=None):
    """Start the HIP runtime on the devices a fit is going to use, on a thread of its own: the first HIP call of a process costs
    0.2 s (runtime start-up, device properties), and the entry points have half a second of host work -- the modeling components
    -- in front of their first solver.  Nothing depends on the thread: the fit creates its solvers as ever (and fails there,
    loudly, when there is no usable device).  Returns the thread; the entry points join it before they return or raise (it has
    long finished by then), so that it never outlives their call."""
    from . import _lib

    def wake():
        try:
            for d in (range(_lib.device_count()) if devices == "all" else devices) or [_DEVICE["index"] or 0]:
                _lib.device_info(int(d))
                HipFitSolver(dtype=np.float32, device=int(d)).close()  # (a stream on the device: its queues exist afterwards)
        except Exception:  # noqa: BLE001 -- reported by the solver that needs the device
            pass

    t = threading.Thread(target=wake, daemon=True)
    t.start()
    return t


def get_solver(fg_model_comps, dtype=np.float32, layout=None, device=None):
    """The HipFitSolver that holds these components on the GPU (created once per component set and dtype)."""
    dtype = np.dtype(dtype)
    cache = fg_model_comps.__dict__.setdefault("_solvers", {})
    # (a caller's explicit choice -- calibrate_and_model_tensor(layout=..., devices=[...]) -- travels with the components)
    layout = layout or fg_model_comps.__dict__.get("_layout") or "shared"
    if device is None:
        device = fg_model_comps.__dict__.get("_device")
    if device is None:
        device = _DEVICE["index"] if _DEVICE["index"] is not None else 0
    # one solver per calling thread: a handle is not re-entrant, and concurrent fits of different (pol, time) slices
    # (calibrate_and_model_tensor, parallel_fits > 1) each need their own device buffers and stream
    key = (dtype.str, layout, device, threading.get_ident())
    if key not in cache:
        shell = copy.copy(fg_model_comps)
        shell.data_r = shell.data_i = shell.wgts = None
        solver = HipFitSolver(dtype=dtype, device=device)
        solver.set_problem(shell, layout=layout)
        limit = _DEVICE["memory_limit_gib"]
        if limit is not None and solver.memory_bytes() > limit * 2.0**30:
            used = solver.memory_bytes() / 2.0**30
            solver.close()
            raise MemoryError(f"the fit needs {used:.2f} GiB of device memory, gpu_memory_limit is {limit} GiB")
        cache[key] = solver
    return cache[key]


def _gram_factors(prob):
    """Per fitting-group Gram matrices ``sum_bl A_bl^T A_bl`` that are not the identity (DPSS columns are orthonormal,
    so for per-baseline DPSS this is empty).  Keyed by (basis, row blocks)."""
    cache = prob.__dict__.setdefault("_gram", None)
    if cache is not None:
        return cache
    F = prob.nfreqs
    keys, out = {}, {}
    for g in range(prob.ngrps):
        rbs = tuple(prob.bl_rowblk[prob.grp_bl_start[g] : prob.grp_bl_start[g + 1]].tolist())
        keys.setdefault((int(prob.grp_basis[g]), rbs), []).append(g)
    for (u, rbs), grps in keys.items():
        blk = prob.basis[u]
        gram = sum(blk[rb * F : (rb + 1) * F].T @ blk[rb * F : (rb + 1) * F] for rb in rbs)
        if not np.allclose(gram, np.eye(gram.shape[0]), atol=1e-9):
            out[(u, rbs)] = (gram, np.asarray(grps))
    prob.__dict__["_gram"] = out
    return out


def _init_coeffs(solver, prob, src_r, src_i):
    """tensorize_fg_coeffs for both components at once: ``A^T (src * [w != 0])`` on the GPU, then the (rarely needed)
    small Gram solves on the host.  Returns flat (c_r, c_i)."""
    solver.init_coeffs(src_r, src_i)
    _, _, c_r, c_i = solver.get_params()
    grams = _gram_factors(prob)
    if grams:
        c_r = c_r.astype(np.float64)
        c_i = c_i.astype(np.float64)
        coff = prob.grp_coff
        for (u, rbs), (gram, grps) in grams.items():
            idx = coff[grps][None, :] + np.arange(gram.shape[0])[:, None]
            c_r[idx] = np.linalg.solve(gram, c_r[idx])
            c_i[idx] = np.linalg.solve(gram, c_i[idx])
        solver.set_params(c_r=c_r, c_i=c_i)
    return c_r, c_i


def tensorize_fg_coeffs(data, wgts, fg_model_comps, notebook_progressbar=False, verbose=False, dtype=None):
    """Initial foreground coefficients of ONE real component by per-group linear least squares on ``data`` with
    zero-weight samples zeroed -- calibration.py:828-913.  Returns the reference's list of ``(nvecs, ngrps, 1, 1)``
    zero-padded arrays."""
    echo(f"{datetime.datetime.now()} Computing initial foreground coefficient guesses using linear-leastsq...\n", verbose=verbose)
    dtype = np.dtype(dtype or np.asarray(data[0]).dtype)
    prob = _as_problem(fg_model_comps, None, None)
    solver = get_solver(prob, dtype)
    src = _flatten(data, prob)
    zeros = np.zeros_like(src)
    solver.set_data(zeros, zeros, _flatten(wgts, prob))
    c_r, _ = _init_coeffs(solver, prob, src, zeros)
    echo(f"{datetime.datetime.now()} Finished initial foreground coefficient guesses...\n", verbose=verbose)
    return coeffs_to_chunks(prob, c_r, dtype)


def yield_fg_model_array(nants, nfreqs, fg_model_comps, fg_coeffs, corr_inds, dtype=np.float32):
    """Foreground model ``sum_k c_k A_k`` of ONE real component as a ``(nants, nants, nfreqs)`` float64 cube --
    calibration.py:402-444 (the A c product runs on the GPU)."""
    prob = _as_problem(fg_model_comps, corr_inds, nants)
    solver = get_solver(prob, dtype)
    c = coeffs_from_chunks(prob, fg_coeffs)
    solver.set_params(c_r=c, c_i=np.zeros_like(c))
    m_r, _ = solver.model()
    model = np.zeros((nants, nants, nfreqs))
    model[prob.bl_ant0, prob.bl_ant1] = m_r
    return model


# ------------------------------------------------------------------------------------------------------------------
# the fit loop: calibration.py:447-738
# ------------------------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------------------------
# The reference's graph functions under their own names and signatures (calibration.py:1587-1656), on NumPy arrays in the
# reference's zero-padded chunk layout -- fg_comps[chunk]: (nvecs, ngrps, nbls, nfreqs); fg_r / fg_i[chunk]: (nvecs, ngrps, 1, 1);
# data / weights[chunk]: (ngrps, nbls, nfreqs); ant0_inds / ant1_inds[chunk]: (ngrps, nbls).  Every one of them is evaluated by the
# HIP library (the fused kernels of the fit in their loss-only / model form); the fit itself never calls them -- it runs forward,
# adjoints and update without materialising a model.
_GRAPH_SOLVERS = []  # [(key, arrays kept alive, FitProblem)]: the last few chunk sets seen, so that repeated calls reuse the uploaded basis


def _fingerprint(a):
    """Shape, dtype and a strided checksum (at most ~4096 samples): enough to notice a component tensor that was modified in place
    between two calls, cheap beside the upload it guards."""
    a = np.asarray(a)
    flat = a.reshape(-1)
    step = max(1, flat.size // 4096)
    return (a.shape, a.dtype.str, float(np.sum(flat[::step], dtype=np.float64)), float(flat[-1]) if flat.size else 0.0)


def _graph_problem(fg_comps, ant0_inds, ant1_inds, nants):
    # (the component tensors by identity AND a content fingerprint -- they are kept alive below, but a caller may write into them --,
    # the small index arrays by content)
    key = (tuple((id(c), _fingerprint(c)) for c in fg_comps), tuple(np.asarray(a, dtype=np.int64).tobytes() for a in ant0_inds),
           tuple(np.asarray(a, dtype=np.int64).tobytes() for a in ant1_inds), int(nants))
    for k, _, prob in _GRAPH_SOLVERS:
        if k == key:
            return prob
    corr_inds = [[[(int(i), int(j)) for i, j in zip(r0, r1)] for r0, r1 in zip(np.asarray(a0), np.asarray(a1))] for a0, a1 in zip(ant0_inds, ant1_inds)]
    prob = _as_problem(list(fg_comps), corr_inds, int(nants))
    _GRAPH_SOLVERS.append((key, list(fg_comps), prob))
    while len(_GRAPH_SOLVERS) > 4:
        _, _, old = _GRAPH_SOLVERS.pop(0)
        for sv in old.__dict__.get("_solvers", {}).values():
            sv.close()
    return prob


def _graph_dtype(*arrays):
    return np.dtype(np.float64) if any(np.asarray(a).dtype == np.float64 for a in arrays) else np.dtype(np.float32)


def fg_model(fg_r, fg_i, fg_comps):
    """Foreground model of ONE chunk, ``v = sum_vec fg * fg_comps`` separately for re and im (calibration.py:1587-1590):
    ``(vr, vi)``, each ``(ngrps, nbls, nfreqs)``."""
    fg_comps = np.asarray(fg_comps)
    _, ngrps, nbls, nfreqs = fg_comps.shape
    zeros = np.zeros((ngrps, nbls), dtype=np.int64)
    prob = _graph_problem([fg_comps], [zeros], [zeros], 1)
    solver = get_solver(prob, _graph_dtype(fg_comps, fg_r))
    solver.set_params(None, None, coeffs_from_chunks(prob, [np.asarray(fg_r)]), coeffs_from_chunks(prob, [np.asarray(fg_i)]))
    vr, vi = solver.model()
    return vr.reshape(ngrps, nbls, nfreqs), vi.reshape(ngrps, nbls, nfreqs)


def data_model(g_r, g_i, fg_r, fg_i, fg_comps, ant0_inds, ant1_inds):
    """Model visibilities of ONE chunk: the foreground model times ``g_ant0 conj(g_ant1)`` (calibration.py:1593-1605):
    ``(model_r, model_i)``, each ``(ngrps, nbls, nfreqs)``."""
    fg_comps, g_r, g_i = np.asarray(fg_comps), np.asarray(g_r), np.asarray(g_i)
    _, ngrps, nbls, nfreqs = fg_comps.shape
    prob = _graph_problem([fg_comps], [ant0_inds], [ant1_inds], g_r.shape[0])
    solver = get_solver(prob, _graph_dtype(fg_comps, g_r))
    solver.set_params(g_r, g_i, coeffs_from_chunks(prob, [np.asarray(fg_r)]), coeffs_from_chunks(prob, [np.asarray(fg_i)]))
    m_r, m_i = solver.data_model()
    return m_r.reshape(ngrps, nbls, nfreqs), m_i.reshape(ngrps, nbls, nfreqs)


def mse(model_r, model_i, data_r, data_i, wgts):
    """``sum(w ((d_r - m_r)^2 + (d_i - m_i)^2))`` (calibration.py:1608-1609) -- cal_weighted_square_error on the device."""
    import ctypes as C

    from . import _lib

    dtype = _graph_dtype(model_r, data_r, wgts)
    arrs = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=dtype), np.shape(data_r))).ravel() for a in (model_r, model_i, data_r, data_i, wgts)]
    out = C.c_double(0.0)
    device = _DEVICE["index"] if _DEVICE["index"] is not None else 0
    _lib.check(_lib.load().cal_weighted_square_error(int(device), _lib.CAL_F64 if dtype == np.float64 else _lib.CAL_F32, arrs[0].size,
                                                     *[a.ctypes.data_as(C.c_void_p) for a in arrs], C.byref(out)))
    return dtype.type(out.value)


def _graph_loss(g_r, g_i, fg_r, fg_i, fg_comps, nchunks, data_r, data_i, wgts, ant0_inds, ant1_inds, dtype, priors=None):
    g_r = np.asarray(g_r)
    dtype = np.dtype(dtype)
    prob = _graph_problem(list(fg_comps)[:nchunks], list(ant0_inds)[:nchunks], list(ant1_inds)[:nchunks], g_r.shape[0])
    solver = get_solver(prob, dtype)
    solver.set_data(_flatten(list(data_r)[:nchunks], prob), _flatten(list(data_i)[:nchunks], prob), _flatten(list(wgts)[:nchunks], prob))
    solver.set_params(g_r, np.asarray(g_i), coeffs_from_chunks(prob, list(fg_r)[:nchunks]), coeffs_from_chunks(prob, list(fg_i)[:nchunks]))
    if priors is None:
        solver.set_regularization(None)
    else:
        solver.set_regularization("sum", float(priors[0]), float(priors[1]))
    return dtype.type(solver.eval_loss())


def mse_chunked(g_r, g_i, fg_r, fg_i, fg_comps, nchunks, data_r, data_i, wgts, ant0_inds, ant1_inds, dtype=np.float32):
    """The loss of the fit: ``sum_chunks mse(data_model(...))`` (calibration.py:1612-1620) -- one loss-only pass of the fused kernel."""
    return _graph_loss(g_r, g_i, fg_r, fg_i, fg_comps, nchunks, data_r, data_i, wgts, ant0_inds, ant1_inds, dtype)


def mse_chunked_sum_regularized(
    g_r,
    g_i,
    fg_r,
    fg_i,
    fg_comps,
    nchunks,
    data_r,
    data_i,
    wgts,
    ant0_inds,
    ant1_inds,
    prior_r_sum,
    prior_i_sum,
    dtype=np.float32,
):
    """``mse_chunked + (sum w m_r - prior_r_sum)^2 + (sum w m_i - prior_i_sum)^2`` (calibration.py:1623-1656)."""
    return _graph_loss(g_r, g_i, fg_r, fg_i, fg_comps, nchunks, data_r, data_i, wgts, ant0_inds, ant1_inds, dtype, priors=(prior_r_sum, prior_i_sum))


def fit_gains_and_foregrounds(
    g_r,
    g_i,
    fg_r,
    fg_i,
    data_r,
    data_i,
    wgts,
    fg_comps,
    corr_inds,
    use_min=False,
    tol=1e-14,
    maxsteps=10000,
    optimizer="Adamax",
    freeze_model=False,
    verbose=False,
    notebook_progressbar=False,
    dtype=np.float32,
    graph_mode=False,
    n_profile_steps=0,
    profile_log_dir="./logdir",
    sky_model_r=None,
    sky_model_i=None,
    model_regularization=None,
    graph_args_dict=None,
    **opt_kwargs,
):
    """Run the optimization loop that fits gains and foreground coefficients -- calibration.py:447-738.

    Same arguments and returns as the reference (arrays are NumPy).  Loop semantics kept: ``n_profile_steps``
    profiled steps and one more step are real, unrecorded updates (:681-693); recorded loss k is evaluated before
    update k (:700-701); ``use_min`` returns the parameters held right after the update of the lowest-loss step
    (:702-710); the loop ends when ``step >= 1 and |l_k - l_{k-1}| < tol`` (:712-717); ``freeze_model`` optimises
    gains only and returns ``fg_r, fg_i`` untouched (:598-603, :730-732).  The whole loop runs on the GPU; losses come
    back once at the end instead of once per step (:701).
    """
    echo(f"Using {str(dtype)} precision.")
    echo(f"{datetime.datetime.now()} Provided the following opt_kwargs")
    for k in opt_kwargs:
        echo(f"{k}: {opt_kwargs[k]}")
    OPTIMIZERS[optimizer]  # unknown optimizer -> KeyError, like calibration.py:571
    dtype = np.dtype(dtype)
    g_r = np.asarray(g_r)
    nants = g_r.shape[0]
    prob = _as_problem(fg_comps, corr_inds, nants)
    solver = get_solver(prob, dtype)
    w_flat = _flatten(wgts, prob)
    solver.set_data(_flatten(data_r, prob), _flatten(data_i, prob), w_flat)
    solver.set_params(g_r, np.asarray(g_i), coeffs_from_chunks(prob, fg_r), coeffs_from_chunks(prob, fg_i))
    echo(f"{datetime.datetime.now()} Performing gradient descent on {np.prod(g_r.shape)} complex gain parameters...", verbose=verbose)
    if not freeze_model:
        echo(f"Performing gradient descent on total of {prob.ncoeffs} complex foreground parameters", verbose=verbose)
    if model_regularization == "sum":
        # priors of calibration.py:619-625 (accumulated in float64 on the host)
        solver.set_regularization("sum", *_prior_sums(_flatten(sky_model_r, prob), _flatten(sky_model_i, prob), w_flat))
    else:
        solver.set_regularization(None)
    solver.set_optimizer(optimizer, **opt_kwargs)
    fit_history = {"loss": []}
    if n_profile_steps > 0:
        echo(f"{datetime.datetime.now()} Profiling with {n_profile_steps}. And writing output to {profile_log_dir}...")
        solver.timing_enable(True)
        solver.run(n_profile_steps, record=False, freeze_model=freeze_model)
        os.makedirs(profile_log_dir, exist_ok=True)
        with open(os.path.join(profile_log_dir, f"calamity_amd_profile_{datetime.datetime.now():%Y%m%d_%H%M%S_%f}.json"), "w") as f:
            json.dump(dict(n_profile_steps=n_profile_steps, fused_basis_kernel=solver.timing_get()), f)
        solver.timing_enable(False)
    echo(f"{datetime.datetime.now()} Building Computational Graph...\n", verbose=verbose)
    solver.run(1, record=False, freeze_model=freeze_model)  # the unrecorded step of calibration.py:693
    echo(f"{datetime.datetime.now()} Performing Gradient Descent...\n", verbose=verbose)
    losses, stopped, _ = solver.run(maxsteps, record=True, tol=tol, use_min=use_min, freeze_model=freeze_model)
    fit_history["loss"] = [dtype.type(l) for l in losses]
    if stopped:
        echo(f"Tolerance thresshold met with delta of {np.abs(losses[-1] - losses[-2]):.2e}. Terminating...\n ", verbose=verbose)
    g_r_opt, g_i_opt, c_r, c_i = solver.get_params(which=1 if (use_min and len(losses) > 0) else 0)
    if freeze_model:
        fg_r_opt, fg_i_opt = fg_r, fg_i
    else:
        fg_r_opt = coeffs_to_chunks(prob, c_r, dtype)
        fg_i_opt = coeffs_to_chunks(prob, c_i, dtype)
    min_loss = np.min(losses) if (use_min and len(losses)) else (losses[-1] if len(losses) else np.nan)
    echo(f"{datetime.datetime.now()} Finished Gradient Descent. MSE of {min_loss:.2e}...\n", verbose=verbose)
    return g_r_opt, g_i_opt, fg_r_opt, fg_i_opt, fit_history


# ------------------------------------------------------------------------------------------------------------------
# write-back: calibration.py:741-825, :1334-1350
# ------------------------------------------------------------------------------------------------------------------
def insert_model_into_uvdata_tensor(uvdata, time, polarization, ants_map, red_grps, model_r, model_i, scale_factor=1.0):
    """Insert ``(Nants, Nants, Nfreqs)`` model cubes back into a UVData object, conjugating pairs stored in reversed
    order and multiplying by ``scale_factor`` -- calibration.py:741-795.  Modifies ``uvdata``."""
    antpairs_data = set(uvdata.get_antpairs())
    polnum = np.where(uvdata.polarization_array == polstr2num(polarization, x_orientation=uvdata.x_orientation))[0][0]
    for red_grp in red_grps:
        for ap in red_grp:
            i, j = ants_map[ap[0]], ants_map[ap[1]]
            if ap in antpairs_data:
                dind = _time_ind(uvdata.time_array, uvdata.antpair2ind(ap), time)
                model = model_r[i, j] + 1j * model_i[i, j]
            else:
                dind = _time_ind(uvdata.time_array, uvdata.antpair2ind(ap[::-1]), time)
                model = model_r[i, j] - 1j * model_i[i, j]
            vis3(uvdata.data_array)[dind, :, polnum] = model * scale_factor


def insert_gains_into_uvcal(uvcal, time, polarization, gains_re, gains_im):
    """Insert ``(Nants, Nfreqs)`` gain arrays back into a UVCal object -- calibration.py:798-825."""
    polnum = np.where(np.asarray(uvcal.jones_array) == polstr2num(polarization, x_orientation=uvcal.x_orientation))[0][0]
    gindt = np.where(np.isclose(uvcal.time_array, time, atol=1e-7, rtol=0.0))[0][0]
    gain4(uvcal.gain_array)[:, :, gindt, polnum] = np.asarray(gains_re) + 1j * np.asarray(gains_im)


def flag_poltime(data_object, time, polarization):
    """Flag (and zero / set to unity) one polarization-time of a UVData or UVCal -- calibration.py:1334-1350."""
    if is_uvdata(data_object):
        bltsel = np.isclose(data_object.time_array, time, atol=1e-7, rtol=0.0)
        polnum = np.where(data_object.polarization_array == polstr2num(polarization, x_orientation=data_object.x_orientation))[0][0]
        data_object.flag_array[bltsel, ..., polnum] = True
        data_object.data_array[bltsel, ..., polnum] = 0.0
    elif is_uvcal(data_object):
        polnum = np.where(np.asarray(data_object.jones_array) == polstr2num(polarization, x_orientation=data_object.x_orientation))[0][0]
        gindt = np.where(np.isclose(data_object.time_array, time, atol=1e-7, rtol=0.0))[0][0]
        gain4(data_object.gain_array)[:, :, gindt, polnum] = 1.0
        gain4(data_object.flag_array)[:, :, gindt, polnum] = True
    else:
        raise ValueError("only supports data_object that is UVCal or UVData.")


# ------------------------------------------------------------------------------------------------------------------
# orchestration: calibration.py:963-1331, :1503-1584
# ------------------------------------------------------------------------------------------------------------------
def calibrate_and_model_tensor(
    uvdata,
    fg_model_comps_dict,
    gains=None,
    freeze_model=False,
    optimizer="Adamax",
    tol=1e-14,
    maxsteps=10000,
    include_autos=False,
    verbose=False,
    sky_model=None,
    dtype=np.float32,
    use_min=False,
    use_redundancy=False,
    notebook_progressbar=False,
    correct_resid=False,
    correct_model=True,
    weights=None,
    nsamples_in_weights=True,
    graph_mode=False,
    grp_size_threshold=5,
    n_profile_steps=0,
    profile_log_dir="./logdir",
    model_regularization="sum",
    init_guesses_from_previous_time_step=False,
    skip_threshold=0.5,
    use_model_snr_weights=False,
    parallel_fits=None,
    batch_slices=None,
    devices=None,
    layout=None,
    device_split=None,
    **opt_kwargs,
):
    """Simultaneous calibration and foreground fitting -- calibration.py:963-1331, same arguments, defaults and
    returns ``(model, resid, gains, fit_history)``.  See SURVEY.md Appendix A for the behaviours kept (the input
    ``uvdata`` is not modified; a supplied ``gains`` object IS modified in place and returned).

    Not in the reference (which loops over polarizations and times, one fit after the other on one device, :1160-1167):

    * ``batch_slices`` (default: on whenever the slices are independent, i.e. unless ``init_guesses_from_previous_time_step``):
      all unskipped (polarization, time) slices of the call -- ``batch_slices=N``: at most N at a time -- are fitted TOGETHER
      in one solver per device: one pass over the modeling components serves every slice, while each slice keeps its own
      rms scale, weight normalisation, priors, loss history, tolerance stop and use_min snapshot, exactly as in the loop
      (``batched.SliceBatchFitter``).  ``fit_history`` and every output equal those of the sequential loop (to rounding:
      the same kernels in the same order).  ``batch_slices=False`` keeps the loop.
    * ``devices``: GPUs to fit on: a list of device indices, or "all" for every visible GPU (a set-up that fails on "all" continues on one
      device with a RuntimeWarning).  Default ``None``: the ONE device of the process (``gpu_index`` of the file driver, else 0) -- a run
      spreads over several GPUs only when the caller says so.
    * ``device_split``: what several devices share out.  "slices": whole batches of slices go to different devices -- no exchange
      between them, every slice is fitted exactly as on one device (bit for bit); "groups": each device takes a share of the
      fitting groups of every slice, with one exchange of the gain gradients per step.  Default: "slices" when the call has at
      least as many batches as devices, else "groups".
    * ``layout``: "shared" (default; baselines alias the distinct basis blocks) or "stream" (every baseline owns its tiles).
    * ``parallel_fits`` (default 1): with ``batch_slices=False``, fits that many slices concurrently, each on its own
      solver and HIP stream.
    (Nothing here is steered by environment variables: layout, devices and concurrency are arguments.)"""
    antpairs_data = uvdata.get_antpairs()
    if not include_autos:
        antpairs_data = set([ap for ap in antpairs_data if ap[0] != ap[1]])
    if len(antpairs_data) != len(uvdata.get_antpairs()):
        uvdata = uvdata.select(inplace=False, bls=[ap for ap in antpairs_data])
    # (with nothing to drop the input object itself is used: it is only read from here on, and the reference's
    # select(inplace=False) copy of a gigabyte of visibilities bought nothing but that guarantee)
    resid = _blank_copy(uvdata, keep_flags=True)  # its visibilities are written at the end (_finish_outputs): data - gains x model
    model = _blank_copy(uvdata)
    red_grps = []
    for fit_grp in fg_model_comps_dict.keys():
        for red_grp in fit_grp:
            red_grps.append(red_grp)
    unity_gains = gains is None
    if gains is None:
        echo(f"{datetime.datetime.now()} Gains are None. Initializing gains starting with unity...\n", verbose=verbose)
        gains = cal_utils.blank_uvcal_from_uvdata(uvdata)
    if sky_model is None and model_regularization is not None:
        echo(f"{datetime.datetime.now()} Sky model is None. Initializing from data...\n", verbose=verbose)
        # data / (g_i conj(g_j)) with the initial gains (:1131-1136).  With the unity, unflagged gains built just above that is
        # the data themselves: the sky model is only read from here on, so the input object stands in for the gigabyte copy
        sky_model = uvdata if unity_gains else cal_utils.apply_gains(uvdata, gains)
    else:
        # the reference dereferences sky_model here even when it is None (calibration.py:1137-1138)
        sky_model = sky_model.select(inplace=False, bls=[ap for ap in antpairs_data])
    fit_history = {}
    ants_map = {ant: i for i, ant in enumerate(np.asarray(gains.ant_array).tolist())}
    fg_model_comps, corr_inds = tensorize_fg_model_comps_dict(
        fg_model_comps_dict=fg_model_comps_dict,
        ants_map=ants_map,
        dtype=dtype,
        nfreqs=sky_model.Nfreqs,
        verbose=verbose,
        notebook_progressbar=notebook_progressbar,
        use_redundancy=use_redundancy,
        grp_size_threshold=grp_size_threshold,
    )
    echo(f"{datetime.datetime.now()}Finished Converting Foreground Modeling Components to Tensors...\n", verbose=verbose)
    del fg_model_comps_dict
    prob = fg_model_comps
    if parallel_fits is None:
        parallel_fits = 1
    if init_guesses_from_previous_time_step:
        parallel_fits = 1
    times = np.unique(uvdata.time_array)
    if batch_slices is None:
        batch_slices = parallel_fits <= 1
    if init_guesses_from_previous_time_step:
        batch_slices = False  # every time starts from the previous one's result: a chain, not a batch
    if batch_slices:
        max_batch = _auto_batch(prob, dtype, layout) if batch_slices is True else max(1, min(int(batch_slices), _lib_max_slices()))
        fit_history = _fit_slices_batched(
            uvdata=uvdata, sky_model=sky_model, gains=gains, resid=resid, model=model, prob=prob, corr_inds=corr_inds, ants_map=ants_map,
            times=times, weights=weights, nsamples_in_weights=nsamples_in_weights, dtype=dtype, skip_threshold=skip_threshold,
            use_model_snr_weights=use_model_snr_weights, optimizer=optimizer, use_min=use_min, freeze_model=freeze_model, tol=tol,
            maxsteps=maxsteps, n_profile_steps=n_profile_steps, profile_log_dir=profile_log_dir, model_regularization=model_regularization,
            verbose=verbose, max_batch=max_batch, devices=devices, layout=layout, opt_kwargs=opt_kwargs,
            correct_model=correct_model, correct_resid=correct_resid, device_split=device_split,
        )
        return model, resid, gains, fit_history  # (every slice left _fit_slices_batched in its final state)
    if layout is not None:
        prob.__dict__["_layout"] = layout
    if devices is not None:
        prob.__dict__["_device"] = 0 if isinstance(devices, str) else int(list(devices)[0])  # the loop fits on one device

    def fit_slice(polnum, pol, time_index, time, carry):
        """One (polarization, time) slice: calibration.py:1167-1330.  ``carry`` holds the parameters handed from one time
        to the next when init_guesses_from_previous_time_step is set."""
        solver = get_solver(fg_model_comps, dtype)
        hist = None
        g_r, g_i, fg_r, fg_i = carry.get("g_r"), carry.get("g_i"), carry.get("fg_r"), carry.get("fg_i")
        echo(f"{datetime.datetime.now()} Working on time {time_index + 1} of {uvdata.Ntimes}...\n", verbose=verbose)
        bltsel = np.isclose(uvdata.time_array, time, atol=1e-7, rtol=0.0)
        frac_unflagged, rmsdata = _slice_stats(uvdata, bltsel, polnum)
        if frac_unflagged >= skip_threshold:
            echo(f"{datetime.datetime.now()} Tensorizing data...\n", verbose=verbose)
            data_r, data_i, wgts = tensorize_data(
                uvdata, corr_inds=corr_inds, ants_map=ants_map, polarization=pol, time=time, data_scale_factor=rmsdata,
                weights=weights, nsamples_in_weights=nsamples_in_weights, dtype=dtype,
            )
            if sky_model is not None:
                echo(f"{datetime.datetime.now()} Tensorizing sky model...\n", verbose=verbose)
                sky_model_r, sky_model_i, _ = tensorize_data(
                    sky_model, corr_inds=corr_inds, ants_map=ants_map, polarization=pol, time=time, data_scale_factor=rmsdata,
                    weights=weights, dtype=dtype,
                )
            else:
                sky_model_r, sky_model_i = None, None
            if carry["first_time"] or not init_guesses_from_previous_time_step:
                carry["first_time"] = False
                echo(f"{datetime.datetime.now()} Tensorizing Gains...\n", verbose=verbose)
                g_r, g_i = tensorize_gains(gains, dtype=dtype, time=time, polarization=pol)
                echo(f"{datetime.datetime.now()} Tensorizing Foreground coeffs...\n", verbose=verbose)
                # tensorize_fg_coeffs x 2 (calibration.py:1219-1233): one device pass gives both components
                w_flat = _flatten(wgts, prob)
                zeros = np.zeros_like(w_flat)
                solver.set_data(zeros, zeros, w_flat)
                c_r, c_i = _init_coeffs(solver, prob, _flatten(sky_model_r, prob), _flatten(sky_model_i, prob))
                fg_r = coeffs_to_chunks(prob, c_r, dtype)
                fg_i = coeffs_to_chunks(prob, c_i, dtype)
                if use_model_snr_weights:
                    m_r, m_i = solver.model()
                    w_new = (np.square(m_r.astype(np.float64)) + np.square(m_i.astype(np.float64))) * w_flat
                    w_new = w_new / np.sum(w_new)
                    start = 0
                    new_wgts = []
                    for w in wgts:
                        n = w.shape[0] * w.shape[1]
                        new_wgts.append(w_new[start : start + n].reshape(w.shape).astype(dtype))
                        start += n
                    wgts = new_wgts
            (g_r, g_i, fg_r, fg_i, hist) = fit_gains_and_foregrounds(
                g_r=g_r, g_i=g_i, fg_r=fg_r, fg_i=fg_i, data_r=data_r, data_i=data_i, wgts=wgts, fg_comps=fg_model_comps,
                corr_inds=corr_inds, optimizer=optimizer, use_min=use_min, freeze_model=freeze_model,
                notebook_progressbar=notebook_progressbar, verbose=verbose, tol=tol, dtype=dtype, maxsteps=maxsteps,
                graph_mode=graph_mode, n_profile_steps=n_profile_steps, profile_log_dir=profile_log_dir,
                sky_model_r=sky_model_r, sky_model_i=sky_model_i, model_regularization=model_regularization, **opt_kwargs,
            )
            # yield_fg_model_array x 2 + insert_model_into_uvdata_tensor (calibration.py:1271-1292) without the
            # nants x nants cubes: one A c pass for both components, rows written straight back
            solver.set_params(c_r=coeffs_from_chunks(prob, fg_r), c_i=coeffs_from_chunks(prob, fg_i))
            m_r, m_i = solver.model()
            _insert_model_rows(model, time, pol, ants_map, prob, m_r, m_i, scale_factor=rmsdata)
            insert_gains_into_uvcal(uvcal=gains, time=time, polarization=pol, gains_re=g_r, gains_im=g_i)
        else:
            echo(f"{datetime.datetime.now()}: Only {frac_unflagged * 100}-percent of data unflagged. Skipping...\n", verbose=verbose)
            flag_poltime(resid, time=time, polarization=pol)
            flag_poltime(gains, time=time, polarization=pol)
            flag_poltime(model, time=time, polarization=pol)
            pass  # the reference's fit_history[polnum] = "skipped!" (:1309) is overwritten at the end of the pol loop (:1321)
        if not freeze_model and model_regularization == "post_hoc" and np.any(~model.flag_array[bltsel]):
            renormalize(
                uvdata_reference_model=sky_model, uvdata_deconv=model, gains=gains, polarization=pol, time=time,
                additional_flags=uvdata.flag_array,
            )
        carry.update(g_r=g_r, g_i=g_i, fg_r=fg_r, fg_i=fg_i)
        return hist

    # one pool for the whole call: its threads (and with them their solvers, get_solver) serve every polarization
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=parallel_fits) if parallel_fits > 1 and len(times) > 1 else None
    for polnum, pol in enumerate(uvdata.get_pols()):
        echo(f"{datetime.datetime.now()} Working on pol {pol}, {polnum + 1} of {uvdata.Npols}...\n", verbose=verbose)
        fit_history_p = {}
        if pool is not None:
            futures = {ti: pool.submit(fit_slice, polnum, pol, ti, t, {"first_time": True}) for ti, t in enumerate(times)}
            for ti, fut in futures.items():
                hist = fut.result()
                if hist is not None:
                    fit_history_p[ti] = hist
        else:
            carry = {"first_time": True}
            for time_index, time in enumerate(times):
                hist = fit_slice(polnum, pol, time_index, time, carry)
                if hist is not None:
                    fit_history_p[time_index] = hist
        fit_history[polnum] = fit_history_p
    if pool is not None:
        pool.shutdown()
    return _finish_outputs(uvdata, model, resid, gains, fit_history, correct_model, correct_resid)


def _output_finisher(uvdata, model, resid, gains, correct_model, correct_resid):
    """``finish(pnum, time)``: residual and model of ONE (polarization, time) in the requested calibration state --
    calibration.py:1322-1331: ``model_with_gains = apply_gains(model, gains, inverse=True)``; ``resid = data - model_with_gains``,
    zero where the model (with gains) or the data are flagged; ``correct_resid``: ``resid / (g_i conj(g_j))``, flags or-ed with
    the gain flags; ``correct_model=False`` returns the model with gains.  The same arithmetic in the same order as those calls, in
    ONE pass over the slice's baseline-times (row chunks on the host's cores) instead of four passes and three container copies."""
    ants = np.asarray(gains.ant_array).astype(np.int64)
    order = np.argsort(ants)

    def ant_rows(a):
        a = np.asarray(a).astype(np.int64)
        pos = np.clip(np.searchsorted(ants[order], a), 0, len(ants) - 1)
        if not np.all(ants[order][pos] == a):
            raise KeyError(int(a[np.where(ants[order][pos] != a)[0][0]]))
        return order[pos]

    a0, a1 = ant_rows(uvdata.ant_1_array), ant_rows(uvdata.ant_2_array)
    gtimes = np.asarray(gains.time_array)
    utimes = np.unique(uvdata.time_array)
    tind = np.asarray([np.where(np.isclose(gtimes, t, rtol=0.0, atol=1e-7))[0][0] for t in utimes])
    gt = tind[np.searchsorted(utimes, np.asarray(uvdata.time_array))]
    udata, uflag = vis3(np.asarray(uvdata.data_array)), vis3(np.asarray(uvdata.flag_array))
    pols = list(uvdata.get_pols())

    def finish(pnum, time):
        # (the containers' arrays as they are NOW: the write-backs before this call may have attached new ones)
        mdata, mflag = vis3(model.data_array), vis3(model.flag_array)
        rdata, rflag = vis3(resid.data_array), vis3(resid.flag_array)
        garr, gflags = gain4(gains.gain_array), gain4(gains.flag_array)
        gindp = np.where(np.asarray(gains.jones_array) == polstr2num(pols[pnum], x_orientation=gains.x_orientation))[0][0]
        t = tind[np.where(np.isclose(utimes, time, rtol=0.0, atol=1e-7))[0][0]]
        gplane = np.ascontiguousarray(garr[:, :, t, gindp])
        fplane = np.ascontiguousarray(gflags[:, :, t, gindp])
        sel = np.where(gt == t)[0]
        contiguous = len(sel) == sel[-1] - sel[0] + 1

        def rows(lo, hi):
            r = slice(sel[0] + lo, sel[0] + hi) if contiguous else sel[lo:hi]
            gg = np.take(gplane, a0[r], axis=0)
            gg *= np.conj(np.take(gplane, a1[r], axis=0))
            gf = np.take(fplane, a0[r], axis=0) | np.take(fplane, a1[r], axis=0)
            mwg = mdata[r, :, pnum] * gg  # apply_gains(model, gains, inverse=True)
            mwf = mflag[r, :, pnum] | gf
            d = udata[r, :, pnum] - mwg
            d[mwf | uflag[r, :, pnum]] = 0.0
            if correct_resid:  # apply_gains(resid, gains)
                d /= gg
                rflag[r, :, pnum] |= gf
            rdata[r, :, pnum] = d
            if not correct_model:
                mdata[r, :, pnum] = mwg
                mflag[r, :, pnum] = mwf

        utils.for_row_chunks(rows, len(sel))

    finish.times, finish.npols = utimes, len(pols)
    return finish


def _finish_outputs(uvdata, model, resid, gains, fit_history, correct_model, correct_resid):
    """Every (polarization, time) through _output_finisher: the outputs of calibrate_and_model_tensor."""
    finish = _output_finisher(uvdata, model, resid, gains, correct_model, correct_resid)
    for pnum in range(finish.npols):
        for time in finish.times:
            finish(pnum, time)
    return model, resid, gains, fit_history


def _slice_stats(uvdata, bltsel, polnum):
    """(fraction of unflagged samples, rms of the unflagged visibilities) of one (polarization, time) -- calibration.py:1168-1182
    -- in one threaded pass over the slice's rows (the reference's boolean fancy indexing copies the slice twice)."""
    rows = np.where(bltsel)[0]
    vis, flg = vis3(np.asarray(uvdata.data_array)), vis3(np.asarray(uvdata.flag_array))
    cnt = np.zeros(len(rows), dtype=np.int64)
    ssq = np.zeros(len(rows), dtype=np.float64)

    def chunk(lo, hi):
        keep = ~np.take(flg[:, :, polnum], rows[lo:hi], axis=0)
        d = np.take(vis[:, :, polnum], rows[lo:hi], axis=0)
        p = d.real * d.real + d.imag * d.imag
        cnt[lo:hi] = np.count_nonzero(keep, axis=1)
        ssq[lo:hi] = np.sum(p * keep, axis=1)

    utils.for_row_chunks(chunk, len(rows))
    n = int(cnt.sum())
    frac = n / (uvdata.Nbls * uvdata.Nfreqs)
    return frac, (float(np.sqrt(ssq.sum() / n)) if n else float("nan"))


def _prior_sums(sky_r, sky_i, wgts):
    """``P_r, P_i = sum w * sky`` (calibration.py:619-625), accumulated in float64 baseline by baseline on the host's cores; the
    total does not depend on how the rows are chunked."""
    nb = len(wgts)
    pr, pi = np.zeros(nb, dtype=np.float64), np.zeros(nb, dtype=np.float64)

    def chunk(lo, hi):
        w64 = wgts[lo:hi].astype(np.float64)
        pr[lo:hi] = np.sum(sky_r[lo:hi] * w64, axis=1)
        pi[lo:hi] = np.sum(sky_i[lo:hi] * w64, axis=1)

    utils.for_row_chunks(chunk, nb)
    return float(pr.sum()), float(pi.sum())


def _lib_max_slices():
    from . import _lib

    return _lib.CAL_MAX_SLICES


def _auto_batch(prob, dtype, layout=None):
    """How many slices one batch may hold when the caller does not say: what fits a quarter of the host memory now available (five
    per-sample arrays per slice and their concatenation) and half a device's memory (about eight per-sample arrays per slice), at
    most the library's CAL_MAX_SLICES.  Tutorial-scale arrays batch hundreds of slices.  In the SHARED layout a slice of 10^7 samples
    or more goes alone: its step is bound by arithmetic, not by launches (HERA-350: 1.29 s per 1 000 steps and slice whether four
    slices are fitted together or one by one), and batches of one let the host work of neighbouring slices run under the fits
    (_fit_slices_batched).  The STREAM layout keeps batching at every size: its slices share the basis tiles they stream."""
    from . import _lib

    if (layout or "shared") == "shared" and float(prob.nbls) * prob.nfreqs >= 1.0e7:
        return 1
    per_array = float(prob.nbls) * prob.nfreqs * np.dtype(dtype).itemsize
    host_avail = 8.0e9
    try:
        with open("/proc/meminfo") as meminfo:
            for line in meminfo:
                if line.startswith("MemAvailable:"):
                    host_avail = float(line.split()[1]) * 1024.0
                    break
    except OSError:
        pass
    try:
        dev_total = float(_lib.device_info(_DEVICE["index"] or 0)["total_mem_bytes"])
    except Exception:  # noqa: BLE001 -- sizing only
        dev_total = 64.0e9
    n = min(0.25 * host_avail / (10.0 * per_array), 0.5 * dev_total / (8.0 * per_array))
    return int(max(1, min(n, _lib_max_slices())))


def _default_devices(nsamples_per_step=None):
    """Devices of a batched fit when the caller names none: the ONE device selected for the process (read_calibrate_and_model_dpss:
    calibration.py:1741-1753; device 0 otherwise).  Several devices are an explicit request -- ``devices=[0, 1, ...]`` or
    ``devices="all"`` -- because that path sets up an RCCL communicator between threads of this process (or none, with whole batches
    per device) that no default should stake a caller's run on before it has been exercised on multi-GPU hardware."""
    return [_DEVICE["index"] if _DEVICE["index"] is not None else 0]


def _resolve_devices(devices):
    """``devices`` keyword of the batched entry points: None (the process's device), "all" (every visible GPU) or a list of indices."""
    from . import _lib

    if devices is None:
        return _default_devices(), True
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError(f"devices={devices!r}: None, 'all' or a list of device indices")
        return list(range(max(1, _lib.device_count()))), True  # ("all" is a wish, not a list: set-up failure falls back to one device, aloud)
    return [int(d) for d in devices], False


def _batch_fitter(prob, nt, dtype, layout, devices):
    """The SliceBatchFitter of ``nt`` slices of these components (kept with them: a second call re-uses the device copy)."""
    from .batched import SliceBatchFitter

    cache = prob.__dict__.setdefault("_batch_fitters", {})
    key = (np.dtype(dtype).str, layout, tuple(devices), int(nt), threading.get_ident())
    if key not in cache:
        for k in [k for k in cache if k[:3] == key[:3] and k[4] == key[4]]:  # another batch size of the same call: free it first
            cache.pop(k).close()
        fitter = SliceBatchFitter(prob, nt, dtype=dtype, layout=layout, devices=devices)
        limit = _DEVICE["memory_limit_gib"]
        if limit is not None and fitter.memory_bytes() > limit * 2.0**30:
            used = fitter.memory_bytes() / 2.0**30
            fitter.close()
            raise MemoryError(f"the fit needs {used:.2f} GiB of device memory, gpu_memory_limit is {limit} GiB")
        cache[key] = fitter
    return cache[key]


def _fit_slices_batched(uvdata, sky_model, gains, resid, model, prob, corr_inds, ants_map, times, weights, nsamples_in_weights, dtype,
                        skip_threshold, use_model_snr_weights, optimizer, use_min, freeze_model, tol, maxsteps, n_profile_steps,
                        profile_log_dir, model_regularization, verbose, max_batch, devices, layout, opt_kwargs, correct_model=True,
                        correct_resid=False, device_split=None):
    """The pol x time loop of calibration.py:1160-1331 with the fits of all unskipped slices issued as batches: per slice
    exactly the host-side steps of the loop body (skip test :1173-1177, rms scale :1178-1182, tensorize :1184-1233, write-back
    :1271-1300, post-hoc renormalisation :1311-1319, residual and calibration state of the outputs :1322-1331), the gradient
    descent of :1244-1269 for up to ``max_batch`` slices at once with per-slice loop control.  Returns ``fit_history``; model,
    resid and gains are complete when it returns."""
    OPTIMIZERS[optimizer]  # unknown optimizer -> KeyError, like calibration.py:571
    dtype = np.dtype(dtype)
    layout = layout or "shared"
    pols = list(uvdata.get_pols())
    fit_history = {polnum: {} for polnum in range(len(pols))}
    todo = [dict(polnum=polnum, pol=pol, time_index=time_index, time=time) for polnum, pol in enumerate(pols) for time_index, time in enumerate(times)]
    finish = _output_finisher(uvdata, model, resid, gains, correct_model, correct_resid)
    devices, chose_devices = _resolve_devices(devices)
    cat = lambda parts_: parts_[0] if len(parts_) == 1 else np.concatenate(parts_)  # noqa: E731

    # One batch goes through three stages: prep (host: the slices' rows out of the containers), fit (device), post (host: the model
    # rows and gains back into the containers).  The stages of neighbouring batches overlap -- while the device fits batch k, one
    # host thread prepares batch k + 1 and another writes batch k - 1 back (NumPy and the library both release the GIL): with several
    # batches the host work of all but the first and the last disappears behind the fits.  The (polarization, time) slices of
    # different batches are disjoint parts of the containers, so the stages never touch the same rows.
    def prep(candidates):
        d_r, d_i, w, s_r, s_i, g_r, g_i = [], [], [], [], [], [], []
        batch = []
        for sl in candidates:
            # skip test and rms scale of the slice (:1168-1182)
            bltsel = np.isclose(uvdata.time_array, sl["time"], atol=1e-7, rtol=0.0)
            frac_unflagged, sl["rmsdata"] = _slice_stats(uvdata, bltsel, sl["polnum"])
            if frac_unflagged < skip_threshold:
                echo(f"{datetime.datetime.now()}: Only {frac_unflagged * 100}-percent of data unflagged. Skipping...\n", verbose=verbose)
                flag_poltime(resid, time=sl["time"], polarization=sl["pol"])
                flag_poltime(gains, time=sl["time"], polarization=sl["pol"])
                flag_poltime(model, time=sl["time"], polarization=sl["pol"])
                continue
            batch.append(sl)
            dr_t, di_t, w_t = _tensorize_flat(uvdata, prob, ants_map, sl["pol"], sl["time"], data_scale_factor=sl["rmsdata"], weights=weights,
                                              nsamples_in_weights=nsamples_in_weights, dtype=dtype)
            d_r.append(dr_t)
            d_i.append(di_t)
            w.append(w_t)
            if sky_model is uvdata:  # (the data stand in for the sky model: the same rows)
                s_r.append(dr_t)
                s_i.append(di_t)
            elif sky_model is not None:
                sr_t, si_t, _ = _tensorize_flat(sky_model, prob, ants_map, sl["pol"], sl["time"], data_scale_factor=sl["rmsdata"], dtype=dtype,
                                                want_weights=False)
                s_r.append(sr_t)
                s_i.append(si_t)
            a, b = tensorize_gains(gains, dtype=dtype, time=sl["time"], polarization=sl["pol"])
            g_r.append(a)
            g_i.append(b)
        echo(f"{datetime.datetime.now()} {len(batch)} of {len(candidates)} (polarization, time) slices tensorized.\n", verbose=verbose)
        if not batch:
            return dict(batch=batch)
        pri = None
        if model_regularization == "sum" and not use_model_snr_weights:  # (with model-SNR weights the priors wait for the new weights: fit)
            pri = np.asarray([_prior_sums(a, b, c) for a, b, c in zip(s_r, s_i, w)])
        return dict(batch=batch, w=cat(w), d_r=cat(d_r), d_i=cat(d_i), s_r=cat(s_r), s_i=cat(s_i), g_r=cat(g_r), g_i=cat(g_i), pri=pri)

    def fit(batch, arrs, on=None):
        nonlocal devices
        nt = len(batch)
        echo(f"{datetime.datetime.now()} Working on {nt} (polarization, time) slices together...\n", verbose=verbose)
        try:
            fitter = _batch_fitter(prob, nt, dtype, layout, devices if on is None else on)
        except Exception as err:  # noqa: BLE001
            if on is not None:
                raise
            # several devices were this function's own choice, not the caller's: when they cannot be set up together (communicator,
            # memory of a peer) the fit runs on the one selected device instead -- said aloud, never silently; an explicit devices=[...]
            # or a single device fails as it stands
            if not (chose_devices and len(devices) > 1):
                raise
            import warnings

            warnings.warn(f"calamity_amd: the fit could not be set up on devices {devices} ({type(err).__name__}: {err}); continuing on device {devices[0]}",
                          RuntimeWarning, stacklevel=2)
            devices = devices[:1]
            fitter = _batch_fitter(prob, nt, dtype, layout, devices)
        w_all, d_r, d_i, s_r, s_i = arrs["w"], arrs["d_r"], arrs["d_i"], arrs["s_r"], arrs["s_i"]
        # tensorize_fg_coeffs x 2 (calibration.py:1219-1233) for every slice: one device pass gives both components (the weights
        # it masks with are those of set_data; the sky model arrives as the pass's own source rows)
        fitter.set_data(d_r, d_i, w_all)
        fitter.init_coeffs(s_r, s_i)
        _, _, c_r, c_i = fitter.get_params()
        if _gram_factors(prob):
            c_r, c_i = _gram_solve(prob, c_r, c_i, nt)
            fitter.set_params(c_r=c_r, c_i=c_i)
        if use_model_snr_weights:
            m_r, m_i = fitter.model()
            w_new = (np.square(m_r.astype(np.float64)) + np.square(m_i.astype(np.float64))) * w_all
            for t in range(nt):  # renormalised slice by slice (:1235-1242)
                rows = slice(t * prob.nbls, (t + 1) * prob.nbls)
                w_new[rows] /= np.sum(w_new[rows])
            w_all = w_new.astype(dtype)
            fitter.set_data(d_r, d_i, w_all)
        fitter.set_params(arrs["g_r"], arrs["g_i"], c_r, c_i)
        if model_regularization == "sum":
            # priors of calibration.py:619-625, one pair per slice (accumulated in float64 on the host)
            nb = prob.nbls
            pri = arrs["pri"]
            if pri is None:
                pri = np.asarray([_prior_sums(s_r[t * nb : (t + 1) * nb], s_i[t * nb : (t + 1) * nb], w_all[t * nb : (t + 1) * nb]) for t in range(nt)])
            fitter.set_regularization("sum", pri[:, 0], pri[:, 1])
        else:
            fitter.set_regularization(None)
        fitter.set_optimizer(optimizer, **opt_kwargs)
        if n_profile_steps > 0:
            fitter.timing_enable(True)
            fitter.run_slices(n_profile_steps, record=False, freeze_model=freeze_model)
            os.makedirs(profile_log_dir, exist_ok=True)
            with open(os.path.join(profile_log_dir, f"calamity_amd_profile_{datetime.datetime.now():%Y%m%d_%H%M%S_%f}.json"), "w") as f:
                json.dump(dict(n_profile_steps=n_profile_steps, slices=nt, fused_basis_kernel=fitter.timing_get()), f)
            fitter.timing_enable(False)
        fitter.run_slices(1, record=False, freeze_model=freeze_model)  # the unrecorded step of calibration.py:693
        results = fitter.run_slices(maxsteps, record=True, tol=tol, use_min=use_min, freeze_model=freeze_model)
        cur = fitter.get_params(0)
        best = fitter.get_params(1) if use_min and any(len(r[0]) for r in results) else None
        if freeze_model:
            cm_r, cm_i = c_r, c_i  # the coefficients the fit was handed (:730-732)
        else:
            cm_r, cm_i = np.array(cur[2]), np.array(cur[3])
        gm_r, gm_i = np.array(cur[0]), np.array(cur[1])
        for t, res in enumerate(results):
            if best is not None and len(res[0]):  # the slice's own minimum (:702-710, :724-728)
                ga, gb = slice(t * prob.nants, (t + 1) * prob.nants), slice(t * prob.ncoeffs, (t + 1) * prob.ncoeffs)
                gm_r[ga], gm_i[ga] = best[0][ga], best[1][ga]
                if not freeze_model:
                    cm_r[gb], cm_i[gb] = best[2][gb], best[3][gb]
        # yield_fg_model_array x 2 + insert_model_into_uvdata_tensor (:1271-1292) for every slice from one A c pass
        fitter.set_params(c_r=cm_r, c_i=cm_i)
        m_r, m_i = fitter.model()
        echo(f"{datetime.datetime.now()} ... fitted.\n", verbose=verbose)
        return dict(results=results, m_r=m_r, m_i=m_i, gm_r=gm_r, gm_i=gm_i)

    def post(batch, out):
        for t, (sl, res) in enumerate(zip(batch, out["results"])):
            rows, ga = slice(t * prob.nbls, (t + 1) * prob.nbls), slice(t * prob.nants, (t + 1) * prob.nants)
            _insert_model_rows(model, sl["time"], sl["pol"], ants_map, prob, out["m_r"][rows], out["m_i"][rows], scale_factor=sl["rmsdata"])
            insert_gains_into_uvcal(uvcal=gains, time=sl["time"], polarization=sl["pol"], gains_re=out["gm_r"][ga], gains_im=out["gm_i"][ga])
            fit_history[sl["polnum"]][sl["time_index"]] = {"loss": [dtype.type(l) for l in res[0]]}
            if res[1]:
                echo(f"Tolerance thresshold met for time {sl['time_index']}. Terminating...\n ", verbose=verbose)
            if not freeze_model and model_regularization == "post_hoc":  # (:1311-1319)
                bltsel = np.isclose(uvdata.time_array, sl["time"], atol=1e-7, rtol=0.0)
                if np.any(~model.flag_array[bltsel]):
                    renormalize(uvdata_reference_model=sky_model, uvdata_deconv=model, gains=gains, polarization=sl["pol"], time=sl["time"],
                                additional_flags=uvdata.flag_array)
            finish(sl["polnum"], sl["time"])  # (:1322-1331)
        echo(f"{datetime.datetime.now()} {len(batch)} (polarization, time) slices written back.\n", verbose=verbose)

    batches = [todo[lo : lo + max_batch] for lo in range(0, len(todo), max_batch)]
    if device_split not in (None, "slices", "groups"):
        raise ValueError(f"device_split={device_split!r}: 'slices', 'groups' or None")
    D = len(devices)
    if D > 1 and (device_split == "slices" or (device_split is None and len(batches) >= D)):
        # Whole batches on different devices: device d fits batches d, d + D, ... on a thread of its own (its solvers live on that
        # thread), nothing is exchanged between devices, and every slice is fitted exactly as it would be on one device.  One prep
        # thread runs at most D + 1 batches ahead of the fits (a batch of HERA-350 rows is 1.3 GB); one thread writes back in order.
        permits, failed = threading.Semaphore(D + 1), threading.Event()

        def prep_when_allowed(candidates):
            permits.acquire()
            return dict(batch=[]) if failed.is_set() else prep(candidates)

        fit_threads = set()

        def fit_on(batch, holder, dev):
            fit_threads.add(threading.get_ident())
            try:
                if failed.is_set():  # (another batch has failed: the call is on its way out, no more maxsteps-long fits)
                    raise RuntimeError("an earlier batch of this call failed")
                return fit(batch, holder.pop(), on=[dev])  # the arrays (1.1-1.3 GB per HERA-350 slice) die with this frame
            except BaseException:
                failed.set()
                raise
            finally:
                permits.release()

        def post_of(batch, fitted):
            try:
                post(batch, fitted.result())
            except BaseException:
                failed.set()
                raise

        pools = [concurrent.futures.ThreadPoolExecutor(1) for _ in range(D + 2)]
        prep_pool, post_pool, fit_pools = pools[0], pools[1], pools[2:]
        try:
            # The prep futures are consumed front to back and dropped as they go: a Future keeps its result alive, and a list of
            # all of them held every batch's arrays until the call returned (60 times x 4 polarizations: ~240 GB of host memory).
            ahead = collections.deque(prep_pool.submit(prep_when_allowed, c) for c in batches)
            written, nfit = collections.deque(), 0
            while ahead:
                arrs = ahead.popleft().result()
                batch = arrs["batch"]
                if not batch:
                    permits.release()
                    continue
                fitted = fit_pools[nfit % D].submit(fit_on, batch, [arrs], devices[nfit % D])
                nfit += 1
                del arrs
                written.append(post_pool.submit(post_of, batch, fitted))
                while written and written[0].done():  # (a failed fit or write-back surfaces here, not after every other batch has been fitted)
                    written.popleft().result()
            while written:
                written.popleft().result()
        except BaseException:
            failed.set()
            for _ in range(len(batches) + D + 2):  # (nobody stays blocked behind the throttle)
                permits.release()
            raise
        finally:
            for pl in pools:
                pl.shutdown(wait=True)
            # the fitters of this call's fit threads (kept with the components, keyed by thread): their threads are gone, nobody
            # could reuse them -- their device memory goes back now
            cache = prob.__dict__.get("_batch_fitters", {})
            for key in [k for k in cache if k[4] in fit_threads]:
                cache.pop(key).close()
    elif len(batches) <= 1:
        for candidates in batches:
            arrs = prep(candidates)
            if arrs["batch"]:
                post(arrs["batch"], fit(arrs["batch"], arrs))
    else:
        with concurrent.futures.ThreadPoolExecutor(1) as prep_pool, concurrent.futures.ThreadPoolExecutor(1) as post_pool:
            ahead, written = prep_pool.submit(prep, batches[0]), []
            for k in range(len(batches)):
                arrs = ahead.result()
                if k + 1 < len(batches):
                    ahead = prep_pool.submit(prep, batches[k + 1])
                batch = arrs["batch"]
                if not batch:  # (every slice of it was skipped)
                    continue
                out = fit(batch, arrs)
                del arrs
                written.append(post_pool.submit(post, batch, out))
                for f in written[:-2]:  # (an exception of a write-back surfaces here, not at the end of the whole job)
                    f.result()
                written = written[-2:]
            for f in written:
                f.result()
    return fit_history


def _gram_solve(prob, c_r, c_i, nt=1):
    """The per-group Gram solves of _init_coeffs for ``nt`` slices of flat coefficients."""
    c_r = np.asarray(c_r, dtype=np.float64).copy()
    c_i = np.asarray(c_i, dtype=np.float64).copy()
    coff = prob.grp_coff
    for (u, rbs), (gram, grps) in _gram_factors(prob).items():
        for t in range(nt):
            idx = t * prob.ncoeffs + coff[grps][None, :] + np.arange(gram.shape[0])[:, None]
            c_r[idx] = np.linalg.solve(gram, c_r[idx])
            c_i[idx] = np.linalg.solve(gram, c_i[idx])
    return c_r, c_i


def _blank_copy(uvdata, keep_flags=False):
    """A copy of ``uvdata`` with all-zero visibilities and no flags (the reference deep-copies, then clears, :1113-1116;
    ``keep_flags``: the flags are copied), without copying the arrays that are about to be overwritten -- and without touching
    the input, which other holders of the object may be reading: the deep copy is told that the two big arrays are already
    copied (the memo maps them to placeholders), then fresh arrays are attached."""
    data, flags = uvdata.data_array, uvdata.flag_array
    hold_d, hold_f = np.zeros(0, dtype=np.asarray(data).dtype), np.zeros(0, dtype=bool)
    out = copy.deepcopy(uvdata, {id(data): hold_d, id(flags): hold_f})
    # (np.zeros: fresh zero pages from the operating system, touched when they are first written -- the write-back of a slice, which
    # runs beside the fits; zeros_like fills the gigabytes here and now)
    out.data_array = np.zeros(np.shape(data), dtype=np.asarray(data).dtype)
    out.flag_array = np.array(flags, copy=True) if keep_flags else np.zeros(np.shape(flags), dtype=bool)
    return out


def _insert_model_rows(uvdata, time, polarization, ants_map, prob, m_r, m_i, scale_factor):
    """insert_model_into_uvdata_tensor (calibration.py:741-795) from per-baseline rows instead of cubes: the row table of the
    time slice (``_baseline_rows``), then a single scatter; pairs the container holds in the reversed order get the conjugate."""
    polnum = np.where(uvdata.polarization_array == polstr2num(polarization, x_orientation=uvdata.x_orientation))[0][0]
    rows, conj = _baseline_rows(uvdata, prob, ants_map, time)
    out = vis3(uvdata.data_array)
    m_r, m_i = np.asarray(m_r), np.asarray(m_i)
    sign = np.where(conj, -1.0, 1.0)

    def chunk(lo, hi):
        out[rows[lo:hi], :, polnum] = (m_r[lo:hi] + 1j * sign[lo:hi, None] * m_i[lo:hi]) * scale_factor

    utils.for_row_chunks(chunk, len(rows))


def calibrate_and_model_dpss(
    uvdata,
    horizon=1.0,
    min_dly=0.0,
    offset=0.0,
    include_autos=False,
    verbose=False,
    red_tol=1.0,
    notebook_progressbar=False,
    fg_model_comps_dict=None,
    **fitting_kwargs,
):
    """Simultaneously solve for gains and model foregrounds with per-baseline DPSS vectors -- the kept entry point,
    calibration.py:1503-1584.  ``fg_model_comps_dict`` is accepted and ignored, as in the reference (:1564)."""
    waker = _wake_device(fitting_kwargs.get("devices"))
    try:
        dpss_model_comps_dict = modeling.yield_pbl_dpss_model_comps(
            uvdata, horizon=horizon, min_dly=min_dly, offset=offset, include_autos=include_autos, red_tol=red_tol,
            notebook_progressbar=notebook_progressbar, verbose=verbose,
        )
        (model, resid, gains, fitted_info) = calibrate_and_model_tensor(
            uvdata=uvdata, fg_model_comps_dict=dpss_model_comps_dict, include_autos=include_autos, verbose=verbose,
            notebook_progressbar=notebook_progressbar, **fitting_kwargs,
        )
    finally:
        waker.join()
    return model, resid, gains, fitted_info


def calibrate_and_model_mixed(
    uvdata,
    horizon=1.0,
    min_dly=0.0,
    offset=0.0,
    ant_dly=0.0,
    include_autos=False,
    verbose=False,
    red_tol=1.0,
    red_tol_freq=0.5,
    n_angle_bins=200,
    notebook_progressbar=False,
    use_redundancy=False,
    use_tensorflow_to_derive_modeling_comps=False,
    eigenval_cutoff=1e-10,
    dtype_matinv=np.float64,
    require_exact_angle_match=True,
    angle_match_tol=1e-3,
    grp_size_threshold=5,
    model_comps_dict=None,
    save_dict_to=None,
    **fitting_kwargs,
):
    """Gains + foregrounds with DPSS vectors for baselines without frequency redundancy and joint covariance
    eigenvectors for groups of baselines whose uv tracks overlap -- calibration.py:1353-1500 (same signature and
    returns).  ``use_tensorflow_to_derive_modeling_comps`` is accepted for compatibility; the eigenproblems run on the
    host either way."""
    fitting_grps, blvecs, _, _ = modeling.get_uv_overlapping_grps_conjugated(
        uvdata, red_tol=red_tol, include_autos=include_autos, red_tol_freq=red_tol_freq, n_angle_bins=n_angle_bins,
        notebook_progressbar=notebook_progressbar, require_exact_angle_match=require_exact_angle_match,
        angle_match_tol=angle_match_tol,
    )
    if model_comps_dict is None:
        waker = _wake_device(fitting_kwargs.get("devices"))
        try:
            freqs = uvdata.freq_array[0] if np.ndim(uvdata.freq_array) == 2 else uvdata.freq_array
            model_comps_dict = modeling.yield_mixed_comps(
                fitting_grps, blvecs, freqs, eigenval_cutoff=eigenval_cutoff, ant_dly=ant_dly, horizon=horizon, offset=offset,
                min_dly=min_dly, verbose=verbose, dtype=dtype_matinv, notebook_progressbar=notebook_progressbar,
                grp_size_threshold=grp_size_threshold,
            )
        finally:
            waker.join()
    if save_dict_to is not None:
        np.save(save_dict_to, model_comps_dict)
    (model, resid, gains, fitted_info) = calibrate_and_model_tensor(
        uvdata=uvdata, fg_model_comps_dict=model_comps_dict, include_autos=include_autos, verbose=verbose,
        notebook_progressbar=notebook_progressbar, use_redundancy=use_redundancy, **fitting_kwargs,
    )
    return model, resid, gains, fitted_info


# ------------------------------------------------------------------------------------------------------------------
# weights from autocorrelations, file / command-line driver: calibration.py:916-960, :1659-1942
# ------------------------------------------------------------------------------------------------------------------
def get_auto_weights(uvdata, delay_extent=25.0):
    """Inverse-variance weights from DPSS-smoothed autocorrelations -- calibration.py:916-960.

    Every autocorrelation spectrum is fitted (least squares on its unflagged channels) with the DPSS modes of a
    zero-length baseline widened by ``delay_extent`` ns; the weight of baseline (i, j) is ``1 / (auto_i auto_j)`` on its
    unflagged samples.  Returns a UVFlag-like object in flag mode whose ``weights_array`` holds the weights.
    """
    freqs = uvdata.freq_array[0] if np.ndim(uvdata.freq_array) == 2 else uvdata.freq_array
    dpss_components = modeling.yield_dpss_model_comps_bl_grp(0.0, freqs, offset=delay_extent)
    if type(uvdata).__module__.startswith("pyuvdata"):
        from pyuvdata import UVFlag

        data_weights = UVFlag(uvdata, mode="flag")
    else:
        from .uvcompat import SimpleUVFlag

        data_weights = SimpleUVFlag(uvdata, mode="flag")
    data_weights.weights_array = np.zeros(uvdata.data_array.shape)
    auto_fit = {}
    bls = uvdata.get_antpairpols()
    for bl in bls:
        if bl[0] == bl[1]:
            rows = []
            for ds, fs in zip(np.atleast_2d(uvdata.get_data(bl)), ~np.atleast_2d(uvdata.get_flags(bl))):
                coeffs = np.linalg.lstsq(dpss_components[fs], ds[fs].real, rcond=None)[0]
                rows.append(dpss_components @ coeffs)
            auto_fit[bl] = np.atleast_2d(np.asarray(rows))
    for bl in bls:
        smooth_weights = 1.0 / (auto_fit[bl[0], bl[0], bl[-1]] * auto_fit[bl[1], bl[1], bl[-1]])
        smooth_weights = smooth_weights * ~np.atleast_2d(uvdata.get_flags(bl))
        dinds = data_weights.antpair2ind(*bl[:2])
        polnum = np.where(data_weights.polarization_array == polstr2num(bl[-1], x_orientation=data_weights.x_orientation))[0][0]
        vis3(data_weights.weights_array)[dinds, :, polnum] = smooth_weights
    return data_weights


def _read_uvdata(files):
    from .uvcompat import read_container

    if is_uvdata(files):
        return files
    files = [files] if isinstance(files, str) else list(files)
    try:
        from pyuvdata import UVData
    except ImportError:
        if len(files) != 1:
            raise ImportError("reading several data files into one object needs pyuvdata")
        return read_container(files[0])
    uvd = UVData()
    uvd.read(files)
    return uvd


def _read_uvcal(files):
    from .uvcompat import read_container

    if is_uvcal(files):
        return files
    files = [files] if isinstance(files, str) else list(files)
    try:
        from pyuvdata import UVCal
    except ImportError:
        if len(files) != 1:
            raise ImportError("reading several gain files into one object needs pyuvdata")
        return read_container(files[0])
    uvc = UVCal()
    uvc.read_calfits(files)
    return uvc


def read_calibrate_and_model_dpss(
    input_data_files,
    input_model_files=None,
    input_gain_files=None,
    resid_outfilename=None,
    gain_outfilename=None,
    model_outfilename=None,
    fitted_info_outfilename=None,
    x_orientation="east",
    clobber=False,
    bllen_min=0.0,
    bllen_max=np.inf,
    bl_ew_min=0.0,
    ex_ants=None,
    select_ants=None,
    gpu_index=None,
    gpu_memory_limit=None,
    precision=32,
    use_autocorrs_in_weights=False,
    **calibration_kwargs,
):
    """File driver of the DPSS fit -- calibration.py:1659-1817 (same arguments, returns and output files).

    Inputs are paths (read with pyuvdata when it is installed; without it, containers written by this package) or the
    objects themselves.  ``gpu_index`` picks the MI355X the solvers are created on (default: device 0, all of them are
    visible); ``gpu_memory_limit`` [GiB] makes a fit that needs more device memory raise ``MemoryError`` instead of
    configuring an allocator pool.  As in the reference the baseline cuts are applied to the data only (:1767-1783
    select on ``uvd`` twice and never on the model) and ``fitted_info_outfilename`` is accepted but nothing is written.
    """
    uvd = _read_uvdata(input_data_files)
    weights = get_auto_weights(uvd) if use_autocorrs_in_weights else None
    utils.select_baselines(uvd, bllen_min=bllen_min, bllen_max=bllen_max, bl_ew_min=bl_ew_min, ex_ants=ex_ants, select_ants=select_ants)
    uvd_model = _read_uvdata(input_model_files) if input_model_files is not None else None
    if uvd_model is not None:
        utils.select_baselines(uvd, bllen_min=bllen_min, bllen_max=bllen_max, bl_ew_min=bl_ew_min)
    uvc = _read_uvcal(input_gain_files) if input_gain_files is not None else None
    dtype = {32: np.float32, 64: np.float64}[precision]
    saved = dict(_DEVICE)
    _DEVICE["index"] = gpu_index
    _DEVICE["memory_limit_gib"] = gpu_memory_limit
    try:
        model_fit, resid_fit, gains_fit, fit_info = calibrate_and_model_dpss(
            uvdata=uvd, sky_model=uvd_model, gains=uvc, dtype=dtype, weights=weights, **calibration_kwargs
        )
    finally:
        _DEVICE.update(saved)
    if resid_outfilename is not None:
        resid_fit.write_uvh5(resid_outfilename, clobber=clobber)
    if gain_outfilename is not None:
        gains_fit.x_orientation = x_orientation
        gains_fit.write_calfits(gain_outfilename, clobber=clobber)
    if model_outfilename is not None:
        model_fit.write_uvh5(model_outfilename, clobber=clobber)
    fit_info["calibration_kwargs"] = calibration_kwargs
    fit_info["calibration_kwargs"]["dtype"] = dtype
    return model_fit, resid_fit, gains_fit, fit_info


def input_output_parser():
    """Input / output / selection / device arguments (names, types and defaults of calibration.py:1820-1858)."""
    ap = argparse.ArgumentParser()
    sp = ap.add_argument_group("Input and Output Arguments.")
    sp.add_argument("--input_data_files", type=str, nargs="+", required=True, help="visibility files to fit")
    sp.add_argument("--input_model_files", type=str, nargs="+", help="sky-model files that fix the overall amplitude and phase")
    sp.add_argument("--input_gain_files", type=str, nargs="+", help="gain files to start from")
    sp.add_argument("--resid_outfilename", type=str, default=None, help="where to write the residual visibilities")
    sp.add_argument("--model_outfilename", type=str, default=None, help="where to write the foreground model")
    sp.add_argument("--gain_outfilename", type=str, default=None, help="where to write the fitted gains")
    # (the reference's default is the STRING "False", which is truthy, so its CLI always overwrites, :1831; here the flag
    # means what it says)
    sp.add_argument("--clobber", action="store_true", default=False, help="replace existing output files")
    sp.add_argument("--x_orientation", default="east", type=str, help="x_orientation recorded in the output gains")
    sp.add_argument("--bllen_min", default=0.0, type=float, help="shortest baseline kept [m]")
    sp.add_argument("--bllen_max", default=np.inf, type=float, help="longest baseline kept [m]")
    sp.add_argument("--bl_ew_min", default=0.0, type=float, help="baselines need an east-west extent above this [m]")
    sp.add_argument("--ex_ants", default=None, type=int, nargs="+", help="antennas to leave out")
    sp.add_argument("--select_ants", default=None, type=int, nargs="+", help="use only these antennas")
    sp.add_argument("--gpu_index", default=None, type=int, help="which GPU of the node to run on")
    sp.add_argument("--gpu_memory_limit", default=None, type=int, help="refuse fits that need more device memory than this [GiB]")
    sp.add_argument("--precision", default=32, type=int, help="32 or 64 bit arithmetic")
    return ap


def fitting_argparser():
    """General fitting arguments (names, types and defaults of calibration.py:1861-1930)."""
    ap = input_output_parser()
    sp = ap.add_argument_group("General Fitting Arguments.")
    sp.add_argument("--tol", type=float, default=1e-14, help="stop when the loss changes by less than this between steps")
    sp.add_argument("--optimizer", type=str, default="Adamax", help="Adam or Adamax")
    sp.add_argument("--maxsteps", type=int, default=10000, help="upper bound on recorded descent steps")
    sp.add_argument("--verbose", default=False, action="store_true", help="print progress")
    sp.add_argument("--use_min", default=False, action="store_true", help="return the parameters of the lowest loss seen instead of the last ones")
    sp.add_argument("--use_redundancy", default=False, action="store_true", help="one set of foreground coefficients per redundant group")
    sp.add_argument("--correct_model", default=True, action="store_true", help="divide the fitted gains out of the model")
    sp.add_argument("--correct_resid", default=False, action="store_true", help="calibrate the residuals with the fitted gains")
    sp.add_argument("--graph_mode", default=False, action="store_true", help="accepted for compatibility; nothing is traced on this path")
    sp.add_argument("--init_guesses_from_previous_time_step", default=False, action="store_true", help="warm-start each time from the previous one")
    sp.add_argument("--learning_rate", type=float, default=1e-2, help="optimizer step size")
    sp.add_argument("--red_tol", type=float, default=1.0, help="baselines closer than this are redundant [m]")
    sp.add_argument("--skip_threshold", type=float, default=0.5, help="flag a (time, polarization) whose unflagged fraction is below this")
    sp.add_argument("--model_regularization", type=str, default="post_hoc", help="sum, post_hoc or None")
    sp.add_argument("--nsamples_in_weights", default=False, action="store_true", help="multiply the weights by nsamples")
    sp.add_argument("--use_model_snr_weights", default=False, action="store_true", help="weight samples by the model's signal to noise")
    sp.add_argument("--use_autocorrs_in_weights", default=False, action="store_true", help="inverse-variance weights from the autocorrelations")
    return ap


def dpss_fit_argparser():
    """DPSS-specific arguments on top of the general ones (calibration.py:1933-1942)."""
    ap = fitting_argparser()
    sp = ap.add_argument_group("DPSS Specific Fitting Arguments.")
    sp.add_argument("--horizon", default=1.0, type=float, help="fraction of the horizon delay covered by the basis")
    sp.add_argument("--min_dly", default=0.0, type=float, help="smallest delay half-width of a basis [ns]")
    sp.add_argument("--offset", default=0.0, type=float, help="delay added beyond the horizon [ns]")
    return ap
