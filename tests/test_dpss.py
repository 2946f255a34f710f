"""The hand-rolled DPSS route of calamity_amd/modeling.py against SciPy, at every (M, NW) the five BASELINE configurations use.

``modeling.dpss_windows`` replaces ``scipy.signal.windows.dpss`` (the function hera_filters' ``dspec.dpss_operator`` builds the
reference's design matrices from, modeling.py:294-300) with a direct LAPACK ``dstemr`` call that does not hold the GIL: same
tridiagonal matrix, same sign conventions.  Checked here: the sequences equal SciPy's to 1e-11 (up to the sign of sequences
whose defining sum is rounding noise), and the number of terms ``dpss_operator`` keeps -- decided from FFT Rayleigh quotients --
equals the count from the dense sinc kernel ``v_k^T S v_k >= cutoff`` it stands for."""
import numpy as np
import pytest
from scipy.signal import windows

from calamity_amd import modeling, synthetic


def _delays(config):
    nants, nfreqs, f0, df = synthetic.CONFIGS[config]
    antpos = synthetic.hex_positions(nants)
    i, j = np.triu_indices(nants, k=1)
    lengths = np.linalg.norm(antpos[i] - antpos[j], axis=1)
    return nfreqs, df, sorted(set(modeling.dly_ns(L) for L in lengths))


CASES = []
for _cfg in ("tutorial", "hera37", "hera350"):
    _nf, _df, _dl = _delays(_cfg)
    CASES += [(_cfg, _nf, _df, d) for d in _dl]
# hera37's delays are a subset of hera350's on the same grid: keep one of each (M, NW)
CASES = list({(nf, round(nf * df * d / 1e9, 12)): (cfg, nf, df, d) for cfg, nf, df, d in CASES}.values())
NCASES_ALL = len(CASES)
# SciPy's dpss costs ~1 s per 1024-channel block on one core: the CPU suite takes every third of the 1024-channel delays (both ends
# included) and all of the tutorial's; STRIDE = 1 runs all 120-odd (4 minutes)
STRIDE = 3
_big = [c for c in CASES if c[1] >= 512]
CASES = [c for c in CASES if c[1] < 512] + _big[::STRIDE] + ([_big[-1]] if (len(_big) - 1) % STRIDE else [])
_SCIPY = {}


def scipy_dpss(nf, nw, kmax):
    key = (nf, round(nw, 12), kmax)
    if key not in _SCIPY:
        _SCIPY[key] = windows.dpss(nf, nw, kmax)
    return _SCIPY[key]


def test_every_baseline_config_is_covered():
    assert len({c[0] for c in CASES}) >= 2 and NCASES_ALL >= 100 and len(CASES) >= 40  # 122 distinct delays at HERA-350 + the tutorial's


@pytest.mark.parametrize("chunk", range(4))
def test_dpss_windows_equal_scipy(chunk):
    for cfg, nf, df, dly in CASES[chunk::4]:
        nw = nf * df * dly / 1e9
        kmax = int(min(nf, np.ceil(2.0 * nw) + 48))
        ours = modeling.dpss_windows(nf, nw, kmax)
        ref = scipy_dpss(nf, nw, kmax)
        assert ours.shape == ref.shape == (kmax, nf)
        for k in range(kmax):
            d = min(np.max(np.abs(ours[k] - ref[k])), np.max(np.abs(ours[k] + ref[k])))
            assert d <= 1e-11, (cfg, dly, k, d)  # (two LAPACK routes to the same eigenvectors: dstemr here, stebz + stein in SciPy; 1e-12 typical)
        # the sign conventions agree wherever they are well defined (a symmetric sequence with a mean that is not rounding noise,
        # an antisymmetric one whose first lobe is not)
        for k in range(0, kmax, 2):
            if abs(ref[k].sum()) > 1e-8:
                assert np.max(np.abs(ours[k] - ref[k])) <= 1e-11, (cfg, dly, k)


@pytest.mark.parametrize("chunk", range(4))
def test_dpss_operator_term_count_equals_the_dense_kernel_rule(chunk):
    ec = 1e-10
    for cfg, nf, df, dly in CASES[chunk::4]:
        x = 100e6 + df * np.arange(nf)
        fw = dly / 1e9
        amat, nterms = modeling.dpss_operator(x, [0.0], [fw], [ec])
        nw = nf * df * fw
        kmax = int(min(nf, np.ceil(2.0 * nw) + 48))
        vecs = scipy_dpss(nf, nw, kmax)
        smat = np.sinc(2.0 * fw * (x[:, None] - x[None, :])) * 2.0 * df * fw
        eig = np.sum((vecs @ smat) * vecs, axis=1)
        want = int(np.max(np.where(eig >= ec)))
        assert nterms[0] == want, (cfg, dly, nterms[0], want)
        assert amat.shape == (nf, want) and np.allclose(amat.real.T @ amat.real, np.eye(want), atol=1e-9)


def test_dpss_blocks_are_kept_across_calls():
    """A block is a function of (channel count, band edges, delay half width, eigenvalue cut): a second call with a fresh
    ``operator_cache`` gets the SAME read-only array without an eigen-decomposition (calibrate_and_model_dpss once per file of a
    night: 0.5 s of set-up per HERA-350 call that only the first one pays); another band or cut builds its own."""
    from calamity_amd import modeling

    modeling.clear_dpss_block_cache()
    f = 100e6 + np.arange(256) * 100e6 / 256
    a = modeling.yield_dpss_model_comps_bl_grp(50.0, f, operator_cache={})
    b = modeling.yield_dpss_model_comps_bl_grp(50.0, f, operator_cache={})
    assert a is b and not a.flags.writeable
    c = modeling.yield_dpss_model_comps_bl_grp(50.0, f, operator_cache={}, eigenval_cutoff=1e-6)
    d = modeling.yield_dpss_model_comps_bl_grp(50.0, f + 1e6, operator_cache={})
    assert c is not a and d is not a and c.shape[1] <= a.shape[1]
    modeling.clear_dpss_block_cache()
    e = modeling.yield_dpss_model_comps_bl_grp(50.0, f, operator_cache={})
    assert e is not a and np.array_equal(e, a)
