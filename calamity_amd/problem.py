"""Ragged (un-padded) host-side description of one gain + foreground fit.

This is the layout handed to the C-ABI (include/calamity_hip.h: cal_solver_set_problem).  It replaces the
reference's zero-padded ``(nvecs, ngrps, nbls, nfreqs)`` chunk tensors
(/root/reference/calamity/calibration.py:104-190): every fitting group keeps its true number of vectors and
identical basis blocks are stored once (``basis`` + ``grp_basis``).

Vocabulary follows the reference: a *fitting group* shares one coefficient vector; its basis has one
``nfreqs``-row block per *redundant group*; every baseline of the group points at the row block it uses.
"""
import hashlib
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class FitProblem:
    nants: int
    nfreqs: int
    basis: List[np.ndarray]  # unique blocks, each (nrowblk * nfreqs, nvec) C-contiguous real
    grp_basis: np.ndarray  # (ngrps,) int32 index into basis
    grp_bl_start: np.ndarray  # (ngrps + 1,) int32; baselines of group g are [start[g], start[g+1])
    bl_ant0: np.ndarray  # (nbls,) int32
    bl_ant1: np.ndarray  # (nbls,) int32
    bl_rowblk: np.ndarray  # (nbls,) int32 row block of the group's basis used by the baseline
    data_r: np.ndarray  # (nbls, nfreqs)
    data_i: np.ndarray
    wgts: np.ndarray
    sky_r: Optional[np.ndarray] = None
    sky_i: Optional[np.ndarray] = None
    # bookkeeping for the round trip back to the reference's chunk tensors
    chunk_of_grp: Optional[np.ndarray] = None
    pos_in_chunk: Optional[np.ndarray] = None
    chunk_shapes: List[tuple] = field(default_factory=list)  # (nvecs, ngrps, nbls) per chunk
    # (nbls,) int32 or None: baseline b reads the basis tiles of baseline bl_alias[b] (-1: its own) -- the same physical
    # baseline in several time slices fitted by one solver (distributed.batch_time_slices); cal_problem_desc::bl_alias
    bl_alias: Optional[np.ndarray] = None
    # number of independent time slices held together (cal_problem_desc::nslices): slice t owns antennas
    # [t * nants / nslices, (t + 1) * nants / nslices), its groups are contiguous; each slice has its own loss and loop state
    nslices: int = 1

    @property
    def ngrps(self):
        return len(self.grp_basis)

    @property
    def nbls(self):
        return len(self.bl_ant0)

    @property
    def grp_nvec(self):
        nvec_u = np.asarray([blk.shape[1] for blk in self.basis], dtype=np.int32)  # per distinct block, then one gather
        return nvec_u[np.asarray(self.grp_basis)]

    @property
    def grp_coff(self):
        """(ngrps + 1,) offsets of each group's coefficients in the flat coefficient vectors."""
        return np.concatenate([[0], np.cumsum(self.grp_nvec)]).astype(np.int64)

    @property
    def ncoeffs(self):
        return int(self.grp_coff[-1])

    def validate(self):
        assert self.grp_bl_start[0] == 0 and self.grp_bl_start[-1] == self.nbls
        assert np.all(np.diff(self.grp_bl_start) >= 1)
        for arr in (self.data_r, self.data_i, self.wgts):
            assert arr is None or arr.shape == (self.nbls, self.nfreqs)
        assert self.bl_ant0.min() >= 0 and self.bl_ant0.max() < self.nants
        assert self.bl_ant1.min() >= 0 and self.bl_ant1.max() < self.nants
        # every baseline's row block exists in its group's basis (one vectorised check: 61 075 groups at HERA-350)
        nrb_u = np.asarray([blk.shape[0] // self.nfreqs for blk in self.basis])
        assert all(blk.shape[0] % self.nfreqs == 0 for blk in self.basis)
        assert np.asarray(self.grp_basis).min() >= 0 and np.asarray(self.grp_basis).max() < len(self.basis)
        grp_of_bl = np.repeat(np.arange(self.ngrps), np.diff(self.grp_bl_start))
        rb = np.asarray(self.bl_rowblk)
        assert rb.min() >= 0 and np.all(rb < nrb_u[np.asarray(self.grp_basis)][grp_of_bl])


def problem_from_chunks(nants, fg_comps, corr_inds, data_r, data_i, wgts, sky_model_r=None, sky_model_i=None):
    """Reference chunk tensors -> FitProblem.

    ``fg_comps[c]`` is ``(nvecs, ngrps, nbls, nfreqs)`` zero-padded along nvecs
    (calibration.py:140-146, :167); a group's true vector count is the index of its first all-zero vector
    (the rule tensorize_fg_coeffs uses, calibration.py:886-892).  Identical blocks are stored once.
    """
    basis, basis_key = [], {}
    grp_basis, grp_bl_start = [], [0]
    bl_ant0, bl_ant1, bl_rowblk = [], [], []
    d_r, d_i, w, s_r, s_i = [], [], [], [], []
    chunk_of_grp, pos_in_chunk, chunk_shapes = [], [], []
    nfreqs = None
    for c, comps in enumerate(fg_comps):
        comps = np.asarray(comps)
        nvecs, ngrps, nbls, nfreqs = comps.shape
        chunk_shapes.append((nvecs, ngrps, nbls))
        for g in range(ngrps):
            blk = comps[:, g].reshape(nvecs, nbls * nfreqs)
            zero_rows = np.where(np.all(blk == 0.0, axis=1))[0]
            nvec = int(zero_rows.min()) if len(zero_rows) else nvecs
            if nvec == 0:
                raise ValueError(f"fitting group {g} of chunk {c} has no non-zero modeling vectors")
            # de-duplicate row blocks inside the group (redundant baselines share rows), then the block.
            rows, rowkey, rb = [], {}, []
            for b in range(nbls):
                sub = np.ascontiguousarray(comps[:nvec, g, b].T.astype(np.float64))  # (nfreqs, nvec)
                k = hashlib.sha1(sub.tobytes()).digest()
                if k not in rowkey:
                    rowkey[k] = len(rows)
                    rows.append(sub)
                rb.append(rowkey[k])
            full = np.ascontiguousarray(np.concatenate(rows, axis=0))
            k = (full.shape, hashlib.sha1(full.tobytes()).digest())
            if k not in basis_key:
                basis_key[k] = len(basis)
                basis.append(full)
            grp_basis.append(basis_key[k])
            for b in range(nbls):
                i, j = corr_inds[c][g][b]
                bl_ant0.append(i)
                bl_ant1.append(j)
                bl_rowblk.append(rb[b])
            grp_bl_start.append(grp_bl_start[-1] + nbls)
            chunk_of_grp.append(c)
            pos_in_chunk.append(g)
        d_r.append(np.asarray(data_r[c], dtype=np.float64).reshape(ngrps * nbls, nfreqs))
        d_i.append(np.asarray(data_i[c], dtype=np.float64).reshape(ngrps * nbls, nfreqs))
        w.append(np.asarray(wgts[c], dtype=np.float64).reshape(ngrps * nbls, nfreqs))
        if sky_model_r is not None:
            s_r.append(np.asarray(sky_model_r[c], dtype=np.float64).reshape(ngrps * nbls, nfreqs))
            s_i.append(np.asarray(sky_model_i[c], dtype=np.float64).reshape(ngrps * nbls, nfreqs))
    prob = FitProblem(
        nants=int(nants),
        nfreqs=int(nfreqs),
        basis=basis,
        grp_basis=np.asarray(grp_basis, dtype=np.int32),
        grp_bl_start=np.asarray(grp_bl_start, dtype=np.int32),
        bl_ant0=np.asarray(bl_ant0, dtype=np.int32),
        bl_ant1=np.asarray(bl_ant1, dtype=np.int32),
        bl_rowblk=np.asarray(bl_rowblk, dtype=np.int32),
        data_r=np.concatenate(d_r),
        data_i=np.concatenate(d_i),
        wgts=np.concatenate(w),
        sky_r=np.concatenate(s_r) if s_r else None,
        sky_i=np.concatenate(s_i) if s_i else None,
        chunk_of_grp=np.asarray(chunk_of_grp, dtype=np.int32),
        pos_in_chunk=np.asarray(pos_in_chunk, dtype=np.int32),
        chunk_shapes=chunk_shapes,
    )
    prob.validate()
    return prob


def _chunk_gather_index(prob):
    """Per chunk: (flat index into the chunk's (nvecs, ngrps) plane, position in the flat coefficient vector) of every true
    coefficient, built once per problem: re-chunking is then two fancy-indexed copies instead of a python loop over 61 075
    groups (0.4 s per call at HERA-350, eight calls per fit)."""
    cache = prob.__dict__.get("_chunk_index")
    if cache is not None:
        return cache
    if prob.chunk_of_grp is None:
        raise ValueError("problem was not built from chunk tensors")
    nvec = prob.grp_nvec.astype(np.int64)
    coff = prob.grp_coff
    grp_of_coef = np.repeat(np.arange(prob.ngrps), nvec)
    k = np.arange(prob.ncoeffs, dtype=np.int64) - coff[grp_of_coef]
    chunk = np.asarray(prob.chunk_of_grp)[grp_of_coef]
    pos = np.asarray(prob.pos_in_chunk, dtype=np.int64)[grp_of_coef]
    cache = []
    for c, (nvecs, ngrps, nbls) in enumerate(prob.chunk_shapes):
        sel = np.where(chunk == c)[0]
        cache.append((k[sel] * ngrps + pos[sel], sel))
    prob.__dict__["_chunk_index"] = cache
    return cache


def coeffs_from_chunks(prob, fg):
    """list of (nvecs, ngrps, 1, 1) chunk arrays -> flat ragged vector in group order."""
    out = np.zeros(prob.ncoeffs, dtype=np.float64)
    for (src, dst), arr in zip(_chunk_gather_index(prob), fg):
        out[dst] = np.asarray(arr).reshape(np.asarray(arr).shape[0], -1).ravel()[src]
    return out


def coeffs_to_chunks(prob, flat, dtype):
    """flat ragged vector -> list of zero-padded (nvecs, ngrps, 1, 1) arrays (the reference's fg_r / fg_i layout)."""
    flat = np.asarray(flat)
    out = []
    for (dst, src), (nvecs, ngrps, nbls) in zip(_chunk_gather_index(prob), prob.chunk_shapes):
        plane = np.zeros(nvecs * ngrps, dtype=dtype)
        plane[dst] = flat[src]
        out.append(plane.reshape(nvecs, ngrps, 1, 1))
    return out


def chunks_from_problem(prob, dtype=np.float64):
    """FitProblem -> the reference's chunk tensors (ONE chunk per distinct nbls, like
    chunk_fg_comp_dict_by_nbls, calibration.py:30-101).  Used by the parity tests to feed the oracle.

    Returns dict(fg_comps, corr_inds, data_r, data_i, wgts, sky_model_r, sky_model_i) and fills the
    chunk bookkeeping of ``prob`` in place.
    """
    nvec = prob.grp_nvec
    nbl_g = np.diff(prob.grp_bl_start)
    chunk_keys = []
    for n in nbl_g:
        if n not in chunk_keys:
            chunk_keys.append(int(n))
    out = dict(fg_comps=[], corr_inds=[], data_r=[], data_i=[], wgts=[], sky_model_r=[], sky_model_i=[])
    prob.chunk_of_grp = np.zeros(prob.ngrps, dtype=np.int32)
    prob.pos_in_chunk = np.zeros(prob.ngrps, dtype=np.int32)
    prob.chunk_shapes = []
    prob.__dict__.pop("_chunk_index", None)  # the re-chunking index of coeffs_to_chunks / coeffs_from_chunks follows the bookkeeping
    F = prob.nfreqs
    for c, nb in enumerate(chunk_keys):
        grps = np.where(nbl_g == nb)[0]
        V = int(nvec[grps].max())
        comps = np.zeros((V, len(grps), nb, F), dtype=dtype)
        ci = []
        sel = []
        for p, g in enumerate(grps):
            prob.chunk_of_grp[g], prob.pos_in_chunk[g] = c, p
            blk = prob.basis[prob.grp_basis[g]]
            b0 = prob.grp_bl_start[g]
            cg = []
            for b in range(nb):
                rb = prob.bl_rowblk[b0 + b]
                comps[: nvec[g], p, b] = blk[rb * F : (rb + 1) * F].T
                cg.append((int(prob.bl_ant0[b0 + b]), int(prob.bl_ant1[b0 + b])))
                sel.append(b0 + b)
            ci.append(cg)
        sel = np.asarray(sel)
        prob.chunk_shapes.append((V, len(grps), nb))
        out["fg_comps"].append(comps)
        out["corr_inds"].append(ci)
        for name, arr in (("data_r", prob.data_r), ("data_i", prob.data_i), ("wgts", prob.wgts)):
            if arr is not None:
                out[name].append(arr[sel].reshape(len(grps), nb, F).astype(dtype))
        if prob.sky_r is not None:
            out["sky_model_r"].append(prob.sky_r[sel].reshape(len(grps), nb, F).astype(dtype))
            out["sky_model_i"].append(prob.sky_i[sel].reshape(len(grps), nb, F).astype(dtype))
    if prob.sky_r is None:
        out["sky_model_r"] = out["sky_model_i"] = None
    return out
