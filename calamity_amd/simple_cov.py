"""Analytic multi-baseline foreground covariance and its leading eigenvectors (host side, setup time).

Counterpart of /root/reference/calamity/simple_cov.py:7-189.  The covariance between sample (baseline a, channel i) and
(baseline b, channel j) is ``sinc(2 max(min_dly * dnu, horizon * |u_a(nu_i) - u_b(nu_j)| + offset * dnu)) *
sinc(2 ant_dly * dnu)`` with ``u = b nu / c`` in wavelengths and ``dnu = |nu_i - nu_j|`` in GHz (delays in ns); the
modeling vectors of a fitting group are the eigenvectors whose eigenvalue is at least ``eigenval_cutoff`` times the
largest, strongest first.  The reference's optional TensorFlow code path for the same arithmetic has no counterpart:
this is a one-off dense symmetric eigenproblem of size (Nbls * Nfreqs) on the host.
"""
import numpy as np


def simple_cov_matrix(blvecs, freqs, ant_dly=0.0, horizon=1.0, offset=0.0, min_dly=0.0, dtype=np.float64, **_ignored):
    """(Nbls * Nfreqs) square covariance, baseline-major rows -- simple_cov.py:7-107."""
    uvws = np.asarray(blvecs, dtype=dtype).reshape(-1, 3)
    freqs = np.asarray(freqs, dtype=dtype)
    # position of every (baseline, channel) sample in the uvw space, in wavelengths
    pts = (uvws[:, None, :] * (freqs[None, :, None] / 3e8)).reshape(-1, 3)
    sep = np.zeros((len(pts), len(pts)), dtype=dtype)
    for axis in range(3):
        sep += np.abs(pts[:, None, axis] - pts[None, :, axis]) ** 2.0
    sep = np.sqrt(sep) * horizon
    fvals = np.tile(freqs, len(uvws))
    dfg = np.abs(fvals[:, None] - fvals[None, :]) / 1e9
    sep += dfg * offset
    cmat = np.sinc(2 * np.maximum(min_dly * dfg, sep))
    cmat = cmat * np.sinc(2 * dfg * ant_dly)
    return cmat


def yield_simple_multi_baseline_model_comps(
    blvecs, freqs, ant_dly=0.0, horizon=1.0, offset=0.0, min_dly=0.0, dtype=np.float64, verbose=False, eigenval_cutoff=1e-10,
    **_ignored
):
    """(Nbls * Nfreqs, Ncomponents) eigenvectors above the cutoff, strongest first -- simple_cov.py:110-189."""
    cmat = simple_cov_matrix(blvecs, freqs, ant_dly=ant_dly, horizon=horizon, offset=offset, min_dly=min_dly, dtype=dtype)
    evals, evecs = np.linalg.eigh(cmat)
    keep = evals / evals[-1] >= eigenval_cutoff
    return evecs[:, keep][:, ::-1]
