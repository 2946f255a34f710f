"""Seeded synthetic fit problems of the shapes in BASELINE.json / SURVEY.md section 8(d).

Hex-packed array (14.6 m spacing, first ``nants`` sites by radius), all ``i < j`` cross baselines, one
DPSS basis per unique rounded delay (modeling.py:293), truth ``c[k] ~ (N + iN)/(k+1)``,
``g = 1 + sigma (N + iN)``, data ``= g_i conj(g_j) (A c) + noise``, random flags, weights ``~flags / sum``
(calibration.py:282-303), data divided by the rms of the unflagged samples (calibration.py:1178-1190),
start at unity gains and ``c0 = A^T (d * [w != 0])`` (calibration.py:875-902; DPSS columns are orthonormal).
"""
import os

import numpy as np

from . import modeling
from .problem import FitProblem

CONFIGS = {
    # name: (nants, nfreqs, f0, df)
    "tutorial": (20, 64, 150e6, 200e3),
    "hera37": (37, 1024, 100e6, 100e6 / 1024),
    "hera350": (350, 1024, 100e6, 100e6 / 1024),
}


def hex_positions(nants, spacing=14.6):
    """First ``nants`` sites of a hexagonal lattice ordered by radius then angle."""
    n = int(np.ceil(np.sqrt(nants))) + 2
    pts = []
    for a in range(-n, n + 1):
        for b in range(-n, n + 1):
            x = spacing * (a + 0.5 * b)
            y = spacing * (np.sqrt(3.0) / 2.0) * b
            pts.append((np.hypot(x, y), np.arctan2(y, x), x, y))
    pts.sort(key=lambda p: (round(p[0], 6), p[1]))
    return np.asarray([(p[2], p[3], 0.0) for p in pts[:nants]])


def make_problem(
    nants,
    nfreqs,
    f0=100e6,
    df=None,
    seed=0,
    gain_sigma=0.1,
    noise_frac=1e-4,
    flag_frac=0.05,
    with_sky=False,
    max_bls=None,
    operator_cache=None,
    bl_sel=None,
    data_seed=None,
):
    """Build one (pol, time) fit.  Returns (FitProblem, truth dict, start dict).

    ``bl_sel`` (indices into the ``i < j`` baseline list, or a callable ``(nvec_per_baseline) -> indices``) keeps only
    some baselines -- the shard of one rank; gains always come from ``seed`` so every shard of a time slice sees the
    same antennas, while coefficients / noise / flags come from ``data_seed`` (default ``seed``).  Weights stay
    normalised by the sum over ALL baselines of the slice (calibration.py:300-303), not over the shard."""
    rng = np.random.default_rng(seed)
    if df is None:
        df = 100e6 / nfreqs
    freqs = f0 + df * np.arange(nfreqs)
    antpos = hex_positions(nants)
    i_idx, j_idx = np.triu_indices(nants, k=1)
    if max_bls is not None and len(i_idx) > max_bls:
        # bounded sample of the same workload: an evenly spaced subset keeps the nvec distribution
        sel = np.linspace(0, len(i_idx) - 1, max_bls).astype(np.int64)
        i_idx, j_idx = i_idx[sel], j_idx[sel]
    lengths = np.linalg.norm(antpos[i_idx] - antpos[j_idx], axis=1)
    dlys = np.asarray([modeling.dly_ns(L) for L in lengths])
    nbls_slice = len(i_idx)
    g_true = 1.0 + gain_sigma * (rng.standard_normal((nants, nfreqs)) + 1j * rng.standard_normal((nants, nfreqs)))
    if data_seed is not None:
        rng = np.random.default_rng(data_seed)
    if operator_cache is None:
        operator_cache = {}
    uniq, inv = np.unique(dlys, return_inverse=True)
    from concurrent.futures import ThreadPoolExecutor

    rep_len = [lengths[np.where(dlys == d)[0][0]] for d in uniq]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:  # LAPACK/BLAS release the GIL
        basis = list(ex.map(lambda L: modeling.yield_dpss_model_comps_bl_grp(L, freqs, operator_cache=operator_cache), rep_len))
    grp_basis = inv.astype(np.int32)
    nvec = np.asarray([basis[b].shape[1] for b in grp_basis])
    if bl_sel is not None:
        sel = np.asarray(bl_sel(nvec, grp_basis) if callable(bl_sel) else bl_sel, dtype=np.int64)
        i_idx, j_idx, grp_basis, nvec = i_idx[sel], j_idx[sel], grp_basis[sel], nvec[sel]
        used = np.unique(grp_basis)
        remap = -np.ones(len(basis), dtype=np.int64)
        remap[used] = np.arange(len(used))
        basis = [basis[u] for u in used]
        grp_basis = remap[grp_basis].astype(np.int32)
    nbls = len(i_idx)
    coff = np.concatenate([[0], np.cumsum(nvec)])
    # truth
    k_idx = np.concatenate([np.arange(n) for n in nvec])
    c_true = (rng.standard_normal(coff[-1]) + 1j * rng.standard_normal(coff[-1])) / (k_idx + 1.0)
    vis = np.empty((nbls, nfreqs), dtype=np.complex128)
    for u in range(len(basis)):
        bls = np.where(grp_basis == u)[0]
        idx = coff[bls][None, :] + np.arange(basis[u].shape[1])[:, None]  # (nvec, nb)
        cu = c_true[idx]
        vis[bls] = (basis[u] @ np.ascontiguousarray(cu.real)).T + 1j * (basis[u] @ np.ascontiguousarray(cu.imag)).T  # two real GEMMs (the basis is real)
    sig_rms = np.sqrt(np.mean(np.abs(vis) ** 2))
    data = g_true[i_idx] * np.conj(g_true[j_idx]) * vis
    data += noise_frac * sig_rms * (rng.standard_normal(data.shape) + 1j * rng.standard_normal(data.shape)) / np.sqrt(2.0)
    flags = rng.random((nbls, nfreqs)) < flag_frac
    wgts = (~flags).astype(np.float64)
    # global normalisation; with a shard the other ranks' baselines are flagged at the same rate in expectation
    wgts /= wgts.sum() * (nbls_slice / nbls)
    rms = np.sqrt(np.sum((data.real**2 + data.imag**2) * wgts) / wgts.sum())  # rms of the unflagged samples (wgts is uniform over them)
    data = data / rms
    prob = FitProblem(
        nants=nants,
        nfreqs=nfreqs,
        basis=basis,
        grp_basis=grp_basis,
        grp_bl_start=np.arange(nbls + 1, dtype=np.int32),
        bl_ant0=i_idx.astype(np.int32),
        bl_ant1=j_idx.astype(np.int32),
        bl_rowblk=np.zeros(nbls, dtype=np.int32),
        data_r=np.ascontiguousarray(data.real),
        data_i=np.ascontiguousarray(data.imag),
        wgts=wgts,
    )
    if with_sky:
        # sky_model=None path of calibrate_and_model_tensor: data / (g g*) with the initial (unity) gains.
        prob.sky_r, prob.sky_i = prob.data_r.copy(), prob.data_i.copy()
    # start: unity gains, c0 = A^T (d * mask)
    c0 = np.empty(coff[-1], dtype=np.complex128)
    dm = data * (~flags).astype(np.float64)
    for u in range(len(basis)):
        bls = np.where(grp_basis == u)[0]
        idx = coff[bls][None, :] + np.arange(basis[u].shape[1])[:, None]
        du = dm[bls].T
        c0[idx] = basis[u].T @ np.ascontiguousarray(du.real) + 1j * (basis[u].T @ np.ascontiguousarray(du.imag))  # (nvec, nb)
    truth = dict(c=c_true / rms, g=g_true, rms=rms, freqs=freqs, antpos=antpos, flags=flags)
    start = dict(
        g_r=np.ones((nants, nfreqs)),
        g_i=np.zeros((nants, nfreqs)),
        c_r=np.ascontiguousarray(c0.real),
        c_i=np.ascontiguousarray(c0.imag),
    )
    return prob, truth, start


def make_config(name, seed=None, **kwargs):
    nants, nfreqs, f0, df = CONFIGS[name]
    if seed is None:
        seed = list(CONFIGS).index(name)
    return make_problem(nants, nfreqs, f0=f0, df=df, seed=seed, **kwargs)


def add_redundant_group(prob, start, rng, nred=3):
    """Merge the first ``nred`` baselines that share basis 0 into ONE fitting group with shared
    coefficients (the ``use_redundancy`` / B > 1 case of calibration.py:173-184).  Test helper."""
    cand = np.where(prob.grp_basis == prob.grp_basis[0])[0][:nred]
    keep = np.setdiff1d(np.arange(prob.nbls), cand)
    order = np.concatenate([cand, keep])
    coff = prob.grp_coff
    nv = prob.grp_nvec[cand[0]]
    new_c_r = np.concatenate([start["c_r"][coff[cand[0]] : coff[cand[0]] + nv]] + [start["c_r"][coff[g] : coff[g + 1]] for g in keep])
    new_c_i = np.concatenate([start["c_i"][coff[cand[0]] : coff[cand[0]] + nv]] + [start["c_i"][coff[g] : coff[g + 1]] for g in keep])
    out = FitProblem(
        nants=prob.nants,
        nfreqs=prob.nfreqs,
        basis=prob.basis,
        grp_basis=np.concatenate([[prob.grp_basis[cand[0]]], prob.grp_basis[keep]]).astype(np.int32),
        grp_bl_start=np.concatenate([[0], len(cand) + np.arange(len(keep) + 1)]).astype(np.int32),
        bl_ant0=prob.bl_ant0[order],
        bl_ant1=prob.bl_ant1[order],
        bl_rowblk=np.zeros(prob.nbls, dtype=np.int32),
        data_r=prob.data_r[order],
        data_i=prob.data_i[order],
        wgts=prob.wgts[order],
        sky_r=None if prob.sky_r is None else prob.sky_r[order],
        sky_i=None if prob.sky_i is None else prob.sky_i[order],
    )
    out.validate()
    new_start = dict(start, c_r=new_c_r, c_i=new_c_i)
    return out, new_start


def make_uvdata(nants=6, nfreqs=64, ntimes=1, f0=100e6, df=400e3, seed=0, eor_db=-50.0, redundant=False, flag_frac=0.0,
                min_dly=2.0 / 0.3, offset=2.0 / 0.3, extent=60.0, future_shapes=False):
    """Small duck-typed UVData sets in the spirit of the reference's test fixtures (test_calibration.py:18-219):
    ``sky_model_projected`` -- smooth foregrounds projected onto each baseline's own DPSS basis, so the model is exactly
    representable (:144-154) -- and ``uvdata`` = projected sky + a flat-spectrum component ``eor_db`` below it
    (:183-193).  ``future_shapes``: arrays in the pyuvdata >= 3 layout (no spw axis).
    Returns (uvdata, sky_model_projected, dpss_vectors)."""
    from . import modeling
    from .uvcompat import SimpleUVData, vis3

    rng = np.random.default_rng(seed)
    if redundant:
        antpos = hex_positions(nants)
    else:
        antpos = np.concatenate([rng.uniform(-extent, extent, size=(nants, 2)), np.zeros((nants, 1))], axis=1)
    freqs = f0 + df * np.arange(nfreqs)
    times = 2458000.0 + 2.0 * np.arange(ntimes)
    antpairs = [(i, j) for i in range(nants) for j in range(i + 1, nants)]
    sky = SimpleUVData(antpos, antpairs, freqs, times, future_shapes=future_shapes)
    sky_vis = vis3(sky.data_array)  # (Nblts, Nfreqs, Npols) view for either layout
    # smooth-spectrum point sources: visibilities confined to the baseline's horizon delay
    nsrc = 12
    lmn = rng.uniform(-0.7, 0.7, size=(nsrc, 2))
    flux = rng.uniform(0.5, 2.0, size=(nsrc, 1)) * (freqs[None, :] / f0) ** rng.uniform(-1.0, -0.5, size=(nsrc, 1))
    for n in range(sky.Nblts):
        bvec = antpos[sky.ant_2_array[n]] - antpos[sky.ant_1_array[n]]
        tau = (lmn @ bvec[:2]) / 299792458.0
        sky_vis[n, :, 0] = np.sum(flux * np.exp(-2j * np.pi * tau[:, None] * freqs[None, :]), axis=0)
    dpss_vectors = modeling.yield_pbl_dpss_model_comps(sky, offset=offset, min_dly=min_dly)
    for ap in sky.get_antpairs():
        dinds = sky.antpair2ind(ap)
        key = ((ap,),) if ((ap,),) in dpss_vectors else ((ap[::-1],),)
        A = dpss_vectors[key]
        sky_vis[dinds, :, 0] = (A @ (sky_vis[dinds, :, 0] @ A).T).T
    uvd = copy_uvdata(sky)
    amp = np.sqrt(np.mean(np.abs(sky.data_array) ** 2)) * 10.0 ** (eor_db / 20.0)
    uvd.data_array = uvd.data_array + amp * (rng.standard_normal(uvd.data_array.shape) + 1j * rng.standard_normal(uvd.data_array.shape)) / np.sqrt(2)
    if flag_frac > 0:
        fl = rng.random(uvd.flag_array.shape) < flag_frac
        uvd.flag_array = fl
        sky.flag_array = fl.copy()
    return uvd, sky, dpss_vectors


def copy_uvdata(uvd):
    import copy

    return copy.deepcopy(uvd)


def merge_redundant_groups(prob, truth, start, tol=0.1):
    """Turn a one-group-per-baseline problem into its ``use_redundancy=True`` form (calibration.py:173-184 with redundant
    groups as fitting groups): baselines with the same vector (to ``tol`` metres, orientation i -> j) share ONE coefficient
    vector and one basis row block.  Coefficients of a group are those of its first member; the data are left as they are
    (no longer exactly representable -- fine for loss / gradient / throughput checks)."""
    antpos = truth["antpos"]
    vec = antpos[prob.bl_ant1] - antpos[prob.bl_ant0]
    key = np.round(vec[:, :2] / tol).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inv = np.asarray(inv).ravel()
    order = np.argsort(inv, kind="stable")
    counts = np.bincount(inv)
    coff = prob.grp_coff
    c_r = np.concatenate([start["c_r"][coff[g] : coff[g + 1]] for g in first])
    c_i = np.concatenate([start["c_i"][coff[g] : coff[g + 1]] for g in first])
    out = FitProblem(
        nants=prob.nants, nfreqs=prob.nfreqs, basis=prob.basis, grp_basis=prob.grp_basis[first].astype(np.int32),
        grp_bl_start=np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), bl_ant0=prob.bl_ant0[order], bl_ant1=prob.bl_ant1[order],
        bl_rowblk=np.zeros(prob.nbls, dtype=np.int32), data_r=prob.data_r[order], data_i=prob.data_i[order], wgts=prob.wgts[order],
        sky_r=None if prob.sky_r is None else prob.sky_r[order], sky_i=None if prob.sky_i is None else prob.sky_i[order],
    )
    out.validate()
    return out, dict(start, c_r=c_r, c_i=c_i)
