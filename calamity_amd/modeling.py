"""Host-side DPSS basis construction (the producer of the per-baseline design matrices ``A_bl``).

Mirrors the interface of /root/reference/calamity/modeling.py:10-81 and :255-374.  The DPSS arithmetic of
the reference lives in the un-vendored third-party ``hera_filters.dspec.dpss_operator`` (modeling.py:294-300,
unpinned in setup.py:59-66); ``dpss_operator`` below restates its published algorithm with
``scipy.signal.windows.dpss``.  Mode-count parity with hera_filters is unpinned.
"""
import collections
import concurrent.futures
import datetime
import os
import threading

import numpy as np
from scipy.signal import windows

from . import simple_cov
from .utils import PBARS, echo, usable_cores


def _lapack_dstemr():
    """LAPACK's dstemr as a ctypes function (the pointer SciPy's own cython_lapack exports).  ctypes releases the GIL for the
    call, SciPy's f2py wrappers (scipy.linalg.eigh_tridiagonal, which scipy.signal.windows.dpss goes through) do not: the
    120 eigenproblems of a HERA-350 basis then really run side by side on the host's cores."""
    import ctypes as C

    from scipy.linalg import cython_lapack

    cap = cython_lapack.__pyx_capi__["dstemr"]
    C.pythonapi.PyCapsule_GetName.restype, C.pythonapi.PyCapsule_GetName.argtypes = C.c_char_p, [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype, C.pythonapi.PyCapsule_GetPointer.argtypes = C.c_void_p, [C.py_object, C.c_char_p]
    ptr = C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap))
    P, I, D = C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    # dstemr(jobz, range, n, d, e, vl, vu, il, iu, m, w, z, ldz, nzc, isuppz, tryrac, work, lwork, iwork, liwork, info)
    return C.CFUNCTYPE(None, P, P, I, D, D, D, D, I, I, I, D, D, I, I, I, I, D, I, I, I, I)(ptr)


_DSTEMR = None


def dpss_windows(M, NW, Kmax):
    """The first ``Kmax`` discrete prolate spheroidal sequences of length ``M`` and half bandwidth ``NW``, (Kmax, M), unit
    norm -- what ``scipy.signal.windows.dpss(M, NW, Kmax)`` returns (the function hera_filters builds its operator from), by
    the same route: the ``Kmax`` largest eigenpairs of the symmetric tridiagonal matrix that commutes with the sinc kernel
    (Slepian 1978), symmetric sequences with positive mean, antisymmetric ones starting with a positive lobe (Percival &
    Walden 1993, p. 379).  The eigenproblem is LAPACK dstemr called without the GIL (``_lapack_dstemr``); where that entry is
    not to be had, SciPy's function itself."""
    import ctypes as C

    global _DSTEMR
    if _DSTEMR is None:
        try:
            _DSTEMR = _lapack_dstemr()
        except Exception:  # noqa: BLE001 -- any SciPy without the capsule: fall back to its own (GIL-holding) path
            _DSTEMR = False
    if _DSTEMR is False:
        return windows.dpss(M, NW, Kmax)
    W = float(NW) / M
    nidx = np.arange(M, dtype=np.float64)
    d = ((M - 1 - 2 * nidx) / 2.0) ** 2 * np.cos(2 * np.pi * W)
    e = np.zeros(M)
    e[: M - 1] = nidx[1:] * (M - nidx[1:]) / 2.0
    n, il, iu, m = C.c_int(M), C.c_int(M - Kmax + 1), C.c_int(M), C.c_int(0)
    ldz, nzc, tryrac, info = C.c_int(M), C.c_int(Kmax), C.c_int(1), C.c_int(0)
    w = np.zeros(M)
    z = np.zeros((Kmax, M))  # column-major (M, Kmax) for LAPACK = row-major (Kmax, M): one eigenvector per row
    isuppz = np.zeros(2 * Kmax, dtype=np.int32)
    lwork, liwork = C.c_int(18 * M), C.c_int(10 * M)
    work, iwork = np.zeros(18 * M), np.zeros(10 * M, dtype=np.int32)
    zero = C.c_double(0.0)
    as_d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    as_i = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731
    _DSTEMR(b"V", b"I", C.byref(n), as_d(d), as_d(e), C.byref(zero), C.byref(zero), C.byref(il), C.byref(iu), C.byref(m), as_d(w), as_d(z),
            C.byref(ldz), C.byref(nzc), as_i(isuppz), C.byref(tryrac), as_d(work), C.byref(lwork), as_i(iwork), C.byref(liwork), C.byref(info))
    if info.value != 0 or m.value != Kmax:
        return windows.dpss(M, NW, Kmax)
    seqs = z[::-1].copy()  # ascending eigenvalues -> the most concentrated sequence first
    fix_even = seqs[::2].sum(axis=1) < 0
    seqs[::2][fix_even] *= -1
    thresh = max(1e-7, 1.0 / M)
    for i, s_ in enumerate(seqs[1::2]):
        if s_[s_ * s_ > thresh][0] < 0:
            seqs[2 * i + 1] *= -1
    return seqs


def _sinc_eigenvalues(vecs, dx, fw):
    """Rayleigh quotients ``v_k^T S v_k`` of unit-norm rows ``vecs`` (kmax, N) with the sinc kernel
    ``S[i, j] = 2 dx fw sinc(2 fw dx (i - j))`` -- the eigenvalue hera_filters' cut is applied to.  ``S`` is symmetric
    Toeplitz on a uniform grid, so ``v^T S v = sum_n acorr_v[n] t[|n|]``: one FFT autocorrelation per sequence instead
    of an N x N sinc evaluation and an N x N x kmax product (110 of the 160 ms a 1024-channel block used to cost)."""
    kmax, nf = vecs.shape
    spec = np.fft.rfft(vecs, n=2 * nf, axis=1)
    acorr = np.fft.irfft(spec.real**2 + spec.imag**2, n=2 * nf, axis=1)[:, :nf]  # lags 0 .. N-1
    t = 2.0 * dx * fw * np.sinc(2.0 * fw * dx * np.arange(nf))
    t[1:] *= 2.0  # lags -n and +n
    return acorr @ t


def dpss_operator(x, filter_centers, filter_half_widths, eigenval_cutoff, cache=None, xc=None):
    """Discrete prolate spheroidal design matrix on the grid ``x``.

    Stand-in for ``hera_filters.dspec.dpss_operator(x, filter_centers, filter_half_widths,
    eigenval_cutoff=..., cache=...)`` as called at modeling.py:294-300.  For each (centre, half width) the
    number of terms is the index of the last DPSS sequence whose sinc-kernel eigenvalue is at least the
    cutoff; the block is ``exp(2 pi i (x - xc) fc) * dpss(N, N dx fw, nterms)``.

    The eigenvalues fall off a cliff (about a decade per index) a few indices beyond ``2 N dx fw``: sequences are
    generated in growing batches until the batch reaches below the cutoff, instead of a fixed generous margin, and the
    eigenvalues are FFT Rayleigh quotients (``_sinc_eigenvalues``).  An eigenvalue within 1e-6 (relative) of the cutoff
    is re-evaluated against the dense kernel, so the count never hangs on FFT rounding.

    Returns (amat [N, sum nterms] complex, nterms list).
    """
    x = np.asarray(x, dtype=np.float64)
    nf = len(x)
    dx = np.abs(x[1] - x[0])
    if xc is None:
        xc = x[nf // 2]
    if cache is None:
        cache = {}
    key = ("dpss_operator", nf, float(dx), float(xc)) + tuple(
        (float(fc), float(fw), float(ec)) for fc, fw, ec in zip(filter_centers, filter_half_widths, eigenval_cutoff)
    )
    if key in cache:
        return cache[key]
    blocks, nterms = [], []
    for fc, fw, ec in zip(filter_centers, filter_half_widths, eigenval_cutoff):
        nw = nf * dx * fw
        margin = 24
        while True:
            kmax = int(min(nf, np.ceil(2.0 * nw) + margin))
            vecs = dpss_windows(nf, nw, kmax)  # (kmax, nf), unit norm rows
            eigvals = _sinc_eigenvalues(vecs, dx, fw)
            if kmax == nf or eigvals[-1] < ec * (1.0 - 1e-6):
                break
            margin *= 2
        close = np.where(np.abs(eigvals - ec) <= 1e-6 * ec)[0]
        if len(close):
            smat = np.sinc(2.0 * fw * (x[:, None] - x[None, :])) * 2.0 * dx * fw
            eigvals[close] = np.sum((vecs[close] @ smat) * vecs[close], axis=1)
        nt = int(np.max(np.where(eigvals >= ec)))
        nterms.append(nt)
        blocks.append(np.exp(2j * np.pi * (x[:, None] - xc) * fc) * vecs[:nt].T)
    out = (np.hstack(blocks), nterms)
    cache[key] = out
    return out


def dly_ns(length, horizon=1.0, min_dly=0.0, offset=0.0):
    """Delay half width in ns of a baseline of ``length`` metres -- modeling.py:293."""
    return float(np.ceil(max(min_dly, length / 0.3 * horizon + offset)))


def yield_dpss_model_comps_bl_grp(
    length, freqs, horizon=1.0, min_dly=0.0, offset=0.0, operator_cache=None, eigenval_cutoff=1e-10
):
    """Per-baseline DPSS modeling vectors, (Nfreqs, Ncomponents) real -- modeling.py:255-301.

    Baselines that round to the same delay share ONE ndarray through ``operator_cache`` (the C-ABI upload
    de-duplicates on object identity).
    """
    return _dpss_block(dly_ns(length, horizon=horizon, min_dly=min_dly, offset=offset), freqs, eigenval_cutoff, operator_cache)


# DPSS blocks outlive the call that built them: a block is a function of (channel count, first / last frequency, delay half width,
# eigenvalue cut) alone, and a pipeline calls calibrate_and_model_dpss once per file on the same array and band -- 0.5 s of
# eigen-decompositions per HERA-350 call that only the first call needs.  Bounded (least recently used first out); the arrays are
# read-only so that no caller can change what the next call gets.
_BLOCKS = collections.OrderedDict()
_BLOCKS_LOCK = threading.Lock()
_BLOCKS_MAX_BYTES = 2 << 30


def clear_dpss_block_cache():
    with _BLOCKS_LOCK:
        _BLOCKS.clear()


def _block_key(delay_ns, freqs, eigenval_cutoff):
    return ("bl_grp", delay_ns / 1e9, len(freqs), float(freqs[0]), float(freqs[-1]), eigenval_cutoff)


def _cached_block(delay_ns, freqs, eigenval_cutoff):
    """The block of this delay if an earlier call left it in the module's cache, else None."""
    with _BLOCKS_LOCK:
        blk = _BLOCKS.get(_block_key(delay_ns, freqs, eigenval_cutoff))
        if blk is not None:
            _BLOCKS.move_to_end(_block_key(delay_ns, freqs, eigenval_cutoff))
    return blk


def _dpss_block(delay_ns, freqs, eigenval_cutoff, operator_cache=None):
    """The real DPSS block of one delay half width (ns), cached by delay: modeling.py:291-301 -- within the call through
    ``operator_cache`` (baselines of one delay share ONE ndarray), across calls through the module's bounded cache."""
    if operator_cache is None:
        operator_cache = {}
    dly = delay_ns / 1e9
    key = _block_key(delay_ns, freqs, eigenval_cutoff)
    if key not in operator_cache:
        with _BLOCKS_LOCK:
            blk = _BLOCKS.get(key)
            if blk is not None:
                _BLOCKS.move_to_end(key)
        if blk is None:
            blk = np.ascontiguousarray(
                dpss_operator(
                    freqs, filter_centers=[0.0], filter_half_widths=[dly], eigenval_cutoff=[eigenval_cutoff], cache=operator_cache
                )[0].real
            )
            blk.setflags(write=False)
            with _BLOCKS_LOCK:
                _BLOCKS[key] = blk
                total = sum(b.nbytes for b in _BLOCKS.values())
                while total > _BLOCKS_MAX_BYTES and len(_BLOCKS) > 1:
                    total -= _BLOCKS.popitem(last=False)[1].nbytes
        operator_cache[key] = blk
    return operator_cache[key]


def get_redundant_grps_data(uvdata, remove_redundancy=False, tol=1.0, include_autos=False):
    """Redundant groups of the antenna pairs that carry data -- same arguments and return tuple as modeling.py:10-81:
    ``(antpairs, red_grps, vec_bin_centers, lengths)`` with one bin centre / length per returned group.

    pyuvdata's ``get_redundancies(include_conjugates=True)`` lists every pair of the ARRAY in the orientation of its
    redundancy convention; a pair counts as present when the data hold it in either orientation.  With
    ``remove_redundancy`` every kept pair becomes a group of its own that inherits the bin centre and length of the
    redundant set it came from (that is what the per-baseline DPSS delay is computed from, :293, :364).  The first
    element is always the empty set, as in the reference (its ``antpairs`` list is never filled, :47, :68)."""
    groups, centres, lengths, _ = uvdata.get_redundancies(use_antpos=True, include_conjugates=True, include_autos=include_autos, tol=tol)
    present = set()
    for i, j in uvdata.get_antpairs():
        present.add((i, j))
        present.add((j, i))
    out_grps, out_centres, out_lengths = [], [], []
    for bls, centre, length in zip(groups, centres, lengths):
        a1, a2 = uvdata.baseline_to_antnums(np.asarray(bls))  # (one call per redundant set, not per baseline)
        kept = [ap for ap in zip(np.atleast_1d(a1).tolist(), np.atleast_1d(a2).tolist()) if ap in present]
        for members in ([[ap] for ap in kept] if remove_redundancy else [kept] if kept else []):
            out_grps.append(members)
            out_centres.append(centre)
            out_lengths.append(length)
    return set(), out_grps, out_centres, out_lengths


def yield_pbl_dpss_model_comps(
    uvdata,
    horizon=1.0,
    min_dly=0.0,
    offset=0.0,
    include_autos=False,
    use_redundancy=False,
    red_tol=1.0,
    eigenval_cutoff=1e-10,
    notebook_progressbar=False,
    verbose=False,
):
    """Per-baseline DPSS modeling components keyed by fitting group -- modeling.py:304-374.

    Same keys and values as the reference (``((antpair, ...),) -> (Nfreqs, Ncomponents)`` real array, groups of one
    delay sharing ONE array object), built the other way round: the delay of every group first, then one DPSS block per
    DISTINCT delay -- independent eigenproblems, solved side by side on the host's cores (SciPy's LAPACK calls release
    the GIL) -- and only then the group -> block assignment.  HERA-350: 122 blocks for 61 075 groups."""
    _, red_grps, centres, _ = get_redundant_grps_data(
        uvdata, remove_redundancy=not use_redundancy, tol=red_tol, include_autos=include_autos
    )
    freqs = _freqs_of(uvdata)
    echo(f"{datetime.datetime.now()} Computing DPSS modeling vectors...\n", verbose=verbose)
    lengths = np.linalg.norm(np.asarray(centres, dtype=np.float64).reshape(len(red_grps), -1), axis=1) if len(red_grps) else np.zeros(0)
    # dly_ns (modeling.py:293) on the whole array of lengths: ceil(max(min_dly, length / 0.3 * horizon + offset))
    delays = np.ceil(np.maximum(min_dly, lengths / 0.3 * horizon + offset)).tolist()
    distinct = sorted(set(delays))

    def block_of(dly):
        # a private operator cache per task: the shared dict is only written from this thread, below
        return _dpss_block(dly, freqs, eigenval_cutoff, {})

    # blocks an earlier call of the process left in the module's cache need no thread pool (a second HERA-350 call: all 122)
    cache = {d: blk for d in distinct for blk in [_cached_block(d, freqs, eigenval_cutoff)] if blk is not None}
    todo = [d for d in distinct if d not in cache]
    workers = max(1, min(len(todo), usable_cores()))
    if workers > 1:
        with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
            blocks = list(PBARS[notebook_progressbar](pool.map(block_of, todo), total=len(todo), disable=not verbose))
    else:
        blocks = [block_of(d) for d in PBARS[notebook_progressbar](todo, disable=not verbose)]
    cache.update(zip(todo, blocks))
    return {(tuple(grp),): cache[dly] for grp, dly in zip(red_grps, delays)}


def _freqs_of(uvdata):
    return uvdata.freq_array[0] if np.ndim(uvdata.freq_array) == 2 else uvdata.freq_array


def get_uv_overlapping_grps_conjugated(
    uvdata,
    red_tol=1.0,
    include_autos=False,
    red_tol_freq=0.5,
    n_angle_bins=200,
    notebook_progressbar=False,
    require_exact_angle_match=True,
    angle_match_tol=1e-3,
):
    """Fitting groups of redundant groups whose uv tracks touch somewhere in the band -- modeling.py:84-252.

    Two redundant groups are *connected* when they fall in the same angular bin (and, by default, at the same angle to
    within ``angle_match_tol``), their ``|b| nu / c`` ranges overlap, and some pair of channels brings them within
    ``red_tol_freq`` wavelengths of each other in the uv plane (or of the conjugate point, in which case the second
    group is re-oriented).  Groups are then labelled greedily in order of (angle, length): an unlabelled group founds a
    fitting group and pulls in its unlabelled connections, a labelled one pulls its unlabelled connections into its
    parent.  Same return tuple as the reference: (fitting_grps, fitting_vec_centers, connections, grp_labels).

    Data structures (sets of tuples, insertion in bin / group order) are kept as in the reference so that ties are broken
    the same way.  The reference's re-orientation branch leaves the un-flipped key behind in its bookkeeping and would
    raise KeyError in the labelling loop; here the flipped key replaces it.
    """
    _, red_grps, vec_bin_centers, _ = get_redundant_grps_data(
        uvdata, include_autos=include_autos, tol=red_tol, remove_redundancy=False
    )
    red_grps = [list(g) for g in red_grps]
    vec_bin_centers = [np.asarray(v, dtype=np.float64) for v in vec_bin_centers]
    freqs = np.asarray(_freqs_of(uvdata), dtype=np.float64)
    fmin, fmax = freqs.min(), freqs.max()
    dangle = np.pi / n_angle_bins

    def angle(vbc):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.arctan(vbc[1] / vbc[0])

    bins = {i: [] for i in range(n_angle_bins)}
    for n, vbc in enumerate(vec_bin_centers):
        if np.abs(vbc[0]) > 0.0:
            b = int(np.min([np.round((angle(vbc) + np.pi / 2) / dangle), n_angle_bins - 2]))
        else:
            b = n_angle_bins - 1
        bins[b].append(n)

    def closest_approach(p0, p1):
        # smallest uv distance between any channel of track 0 and any channel of track 1 (wavelengths)
        u0, v0 = p0[0] * freqs / 3e8, p0[1] * freqs / 3e8
        u1, v1 = p1[0] * freqs / 3e8, p1[1] * freqs / 3e8
        best = np.inf
        for lo in range(0, len(freqs), 256):
            d = np.sqrt(np.abs(u0[None, :] - u1[lo : lo + 256, None]) ** 2.0 + (v0[None, :] - v1[lo : lo + 256, None]) ** 2.0)
            best = min(best, d.min())
        return best

    vbc_hash, connections = {}, {}
    for b in PBARS[notebook_progressbar](range(n_angle_bins), disable=True):
        nums = bins[b]
        for i, n0 in enumerate(nums):
            key0 = tuple(red_grps[n0])
            if key0 not in connections:
                connections[key0] = set({})
                vbc_hash[key0] = vec_bin_centers[n0]
            vbc0 = vec_bin_centers[n0]
            for n1 in nums[i + 1 :]:
                vbc1 = vec_bin_centers[n1]
                lo0, lo1 = fmin * np.linalg.norm(vbc0) / 3e8, fmin * np.linalg.norm(vbc1) / 3e8
                hi0, hi1 = fmax * np.linalg.norm(vbc0) / 3e8, fmax * np.linalg.norm(vbc1) / 3e8
                if not ((lo0 > lo1 and lo0 < hi1) or (lo1 > lo0 and lo1 < hi0)):
                    continue
                if require_exact_angle_match and not np.abs(angle(vbc0) - angle(vbc1)) <= angle_match_tol:
                    continue
                if closest_approach(vbc0, vbc1) <= red_tol_freq:
                    pass
                elif closest_approach(vbc0, -vbc1) <= red_tol_freq:
                    red_grps[n1] = [ap[::-1] for ap in red_grps[n1]]
                    vec_bin_centers[n1] = vbc1 = -vbc1
                else:
                    continue
                key1 = tuple(red_grps[n1])
                connections[key0].add(key1)
                if key1 not in connections:
                    connections[key1] = set({})
                    vbc_hash[key1] = vbc1
                connections[key1].add(key0)

    keys = list(vbc_hash)
    lengths = [np.linalg.norm(vbc_hash[k]) for k in keys]
    angles = [np.arccos(vbc_hash[k][0] / ln) for k, ln in zip(keys, lengths)]
    order = [keys[n] for n in sorted(range(len(keys)), key=lambda n: (angles[n], lengths[n]))]
    fitting_grps, grp_labels = {}, {}
    for red_grp in order:
        if red_grp not in grp_labels:
            fitting_grps[red_grp] = [red_grp]
            grp_labels[red_grp] = red_grp
        parent = grp_labels[red_grp]
        for connection in connections[red_grp]:
            if connection not in grp_labels:
                fitting_grps[parent].append(connection)
                grp_labels[connection] = parent
    fitting_grps = list(fitting_grps.values())
    fitting_vec_centers = [[vbc_hash[red_grp] for red_grp in fit_grp] for fit_grp in fitting_grps]
    return fitting_grps, fitting_vec_centers, connections, grp_labels


def yield_mixed_comps(
    fitting_grps,
    fitting_blvecs,
    freqs,
    eigenval_cutoff=1e-10,
    ant_dly=0.0,
    horizon=1.0,
    offset=0.0,
    min_dly=0.0,
    verbose=False,
    dtype=np.float64,
    notebook_progressbar=False,
    use_tensorflow=False,
    grp_size_threshold=5,
):
    """Modeling vectors for jointly fitted groups -- modeling.py:377-474.

    Fitting groups of at most ``grp_size_threshold`` redundant groups are split into per-redundant-group DPSS bases
    (keyed ``(red_grp,)``; note the reference passes ``ant_dly`` as the DPSS offset here, :447-455); larger ones get
    the leading eigenvectors of the analytic covariance, ``(Nfreqs * len(fit_grp), Ncomponents)``, one ``Nfreqs`` row
    block per redundant group.
    """
    operator_cache = {}
    modeling_vectors = {}
    for grpnum in PBARS[notebook_progressbar](range(len(fitting_grps)), disable=not verbose):
        fit_grp = tuple(fitting_grps[grpnum])
        blvecs = np.asarray(fitting_blvecs[grpnum], dtype=np.float64).reshape(-1, 3)
        bllens = np.linalg.norm(blvecs, axis=1)
        if len(fit_grp) <= grp_size_threshold:
            for red_grp, bllen in zip(fit_grp, bllens):
                modeling_vectors[(red_grp,)] = yield_dpss_model_comps_bl_grp(
                    freqs=freqs, length=bllen, offset=ant_dly, horizon=horizon, min_dly=min_dly,
                    operator_cache=operator_cache, eigenval_cutoff=eigenval_cutoff,
                )
        else:
            modeling_vectors[fit_grp] = simple_cov.yield_simple_multi_baseline_model_comps(
                blvecs=blvecs, ant_dly=ant_dly, offset=offset, min_dly=min_dly, horizon=horizon, dtype=dtype, freqs=freqs,
                eigenval_cutoff=eigenval_cutoff, verbose=verbose,
            )
    return modeling_vectors
