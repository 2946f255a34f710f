"""Host-side DPSS basis construction (the producer of the per-baseline design matrices ``A_bl``).

Mirrors the interface of /root/reference/calamity/modeling.py:10-81 and :255-374.  The DPSS arithmetic of
the reference lives in the un-vendored third-party ``hera_filters.dspec.dpss_operator`` (modeling.py:294-300,
unpinned in setup.py:59-66); ``dpss_operator`` below restates its published algorithm with
``scipy.signal.windows.dpss``.  Mode-count parity with hera_filters is unpinned.
"""
import datetime

import numpy as np
from scipy.signal import windows

from .utils import PBARS, echo


def dpss_operator(x, filter_centers, filter_half_widths, eigenval_cutoff, cache=None, xc=None):
    """Discrete prolate spheroidal design matrix on the grid ``x``.

    Stand-in for ``hera_filters.dspec.dpss_operator(x, filter_centers, filter_half_widths,
    eigenval_cutoff=..., cache=...)`` as called at modeling.py:294-300.  For each (centre, half width) the
    number of terms is the index of the last DPSS sequence whose sinc-kernel eigenvalue is at least the
    cutoff; the block is ``exp(2 pi i (x - xc) fc) * dpss(N, N dx fw, nterms)``.

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
        # eigenvalues fall off a cliff beyond 2 NW; a margin of 64 sequences covers cutoffs down to 1e-16.
        kmax = int(min(nf, np.ceil(2.0 * nw) + 64))
        vecs = windows.dpss(nf, nw, kmax)  # (kmax, nf), unit norm rows
        smat = np.sinc(2.0 * fw * (x[:, None] - x[None, :])) * 2.0 * dx * fw
        eigvals = np.sum((smat @ vecs.T) * vecs.T, axis=0)
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
    if operator_cache is None:
        operator_cache = {}
    dly = dly_ns(length, horizon=horizon, min_dly=min_dly, offset=offset) / 1e9
    key = ("bl_grp", dly, len(freqs), float(freqs[0]), float(freqs[-1]), eigenval_cutoff)
    if key not in operator_cache:
        operator_cache[key] = np.ascontiguousarray(
            dpss_operator(
                freqs, filter_centers=[0.0], filter_half_widths=[dly], eigenval_cutoff=[eigenval_cutoff], cache=operator_cache
            )[0].real
        )
    return operator_cache[key]


def get_redundant_grps_data(uvdata, remove_redundancy=False, tol=1.0, include_autos=False):
    """Antenna pairs organised in redundant groups -- modeling.py:10-81 (same return tuple)."""
    antpairs = []
    red_grps, vec_bin_centers, lengths, _ = uvdata.get_redundancies(
        use_antpos=True, include_conjugates=True, include_autos=include_autos, tol=tol
    )
    red_grps = [[uvdata.baseline_to_antnums(bl) for bl in red_grp] for red_grp in red_grps]
    ap_data = set(uvdata.get_antpairs())
    red_grps = [[ap for ap in red_grp if ap in ap_data or ap[::-1] in ap_data] for red_grp in red_grps]
    lengths = [length for length, red_grp in zip(lengths, red_grps) if len(red_grp) > 0]
    vec_bin_centers = [vbc for vbc, red_grp in zip(vec_bin_centers, red_grps) if len(red_grp) > 0]
    red_grps = [red_grp for red_grp in red_grps if len(red_grp) > 0]
    antpairs = set(antpairs)
    if remove_redundancy:
        red_grps_t, vec_bin_centers_t, lengths_t = [], [], []
        for red_grp, vbc, length in zip(red_grps, vec_bin_centers, lengths):
            for ap in red_grp:
                red_grps_t.append([ap])
                vec_bin_centers_t.append(vbc)
                lengths_t.append(length)
        red_grps, lengths, vec_bin_centers = red_grps_t, lengths_t, vec_bin_centers_t
    return antpairs, red_grps, vec_bin_centers, lengths


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
    """Per-baseline DPSS modeling components keyed by fitting group -- modeling.py:304-374."""
    operator_cache = {}
    _, red_grps, vec_bin_centers, _ = get_redundant_grps_data(
        uvdata, remove_redundancy=not (use_redundancy), tol=red_tol, include_autos=include_autos
    )
    fitting_grps = [(tuple(red_grp),) for red_grp in red_grps]
    modeling_vectors = {}
    freqs = uvdata.freq_array[0] if np.ndim(uvdata.freq_array) == 2 else uvdata.freq_array
    echo(f"{datetime.datetime.now()} Computing DPSS modeling vectors...\n", verbose=verbose)
    for grpnum in PBARS[notebook_progressbar](range(len(fitting_grps)), disable=not verbose):
        bllen = np.linalg.norm(vec_bin_centers[grpnum])
        modeling_vectors[fitting_grps[grpnum]] = yield_dpss_model_comps_bl_grp(
            freqs=freqs,
            length=bllen,
            offset=offset,
            horizon=horizon,
            min_dly=min_dly,
            operator_cache=operator_cache,
            eigenval_cutoff=eigenval_cutoff,
        )
    return modeling_vectors
