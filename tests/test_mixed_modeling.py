"""Host-side producers of multi-baseline fitting groups: the analytic covariance, the
uv-overlap group finder and the mixed DPSS / covariance-eigenvector dictionary.

Two checks are known answers held by the reference's own tests:
* test_simple_cov.py:21-45 compares ``simple_cov_matrix`` of one baseline with a closed-form sinc product;
* test_modeling.py:21-32 lists the fitting groups of its 6-antenna fixture.  The fixture file itself (uvh5) cannot be
  read here (no h5py); its name says "Garray ... antenna_diameter2.0 fractional_spacing1.0 nant6 nf200 df100.000kHz
  f0100.000MHz", i.e. the order-6 Golomb ruler (marks 0 1 4 10 12 17) times 2 m on an east-west line, 200 channels of
  100 kHz from 100 MHz -- an inference, recorded as such; with that geometry the listed groups come out exactly.
"""
import numpy as np
import pytest

from calamity_amd import calibration, modeling, simple_cov
from calamity_amd.uvcompat import SimpleUVData


def golomb6():
    marks = np.array([0.0, 1.0, 4.0, 10.0, 12.0, 17.0]) * 2.0
    antpos = np.stack([marks, np.zeros(6), np.zeros(6)], axis=1)
    freqs = 100e6 + 100e3 * np.arange(200)
    antpairs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    return SimpleUVData(antpos, antpairs, freqs, np.array([2458000.0]))


@pytest.mark.parametrize("horizon, offset, min_dly, ant_dly", [(1.0, 20.0, 0.0, 0.0), (0.8, 123.0, 200.0, 0.0), (1.0, 0.0, 0.0, 2 / 0.3)])
def test_simple_cov_closed_form(horizon, offset, min_dly, ant_dly):
    blvecs = np.array([[2.0, 0.0, 0.0]])
    freqs = 100e6 + 100e3 * np.arange(200)
    fg0, fg1 = np.meshgrid(freqs, freqs)
    bldly = np.max([np.linalg.norm(blvecs[0]) * horizon / 0.3 + offset, min_dly])
    tcov = np.sinc(2 * bldly * (fg0 - fg1) / 1e9)
    if ant_dly > 0:
        tcov *= np.sinc(2 * (fg0 - fg1) / 1e9 * ant_dly)
    scov = simple_cov.simple_cov_matrix(blvecs, freqs, ant_dly=ant_dly, horizon=horizon, offset=offset, min_dly=min_dly, dtype=np.float64)
    assert np.allclose(scov, tcov)


def test_simple_cov_two_baselines_block_structure():
    freqs = 100e6 + 100e3 * np.arange(32)
    b = np.array([[14.6, 0.0, 0.0], [29.2, 0.0, 0.0]])
    c = simple_cov.simple_cov_matrix(b, freqs)
    assert c.shape == (64, 64) and np.allclose(c, c.T) and np.allclose(np.diag(c), 1.0)
    # cross block: |u_a(nu_i) - u_b(nu_j)| with u = b nu / c
    i, j = 3, 17
    sep = abs(14.6 * freqs[i] - 29.2 * freqs[j]) / 3e8
    assert np.isclose(c[i, 32 + j], np.sinc(2 * sep))
    vecs = simple_cov.yield_simple_multi_baseline_model_comps(b, freqs, eigenval_cutoff=1e-10)
    assert vecs.shape[0] == 64 and np.allclose(vecs.T @ vecs, np.eye(vecs.shape[1]), atol=1e-10)
    evals = np.linalg.eigvalsh(c)[::-1]
    assert vecs.shape[1] == np.count_nonzero(evals / evals[0] >= 1e-10)
    assert np.allclose(np.sum((c @ vecs) * vecs, axis=0), evals[: vecs.shape[1]])


def test_get_uv_overlapping_grps_conjugated_reference_known_answer():
    fitting_grps, fitting_vec_centers, connections, grp_labels = modeling.get_uv_overlapping_grps_conjugated(
        uvdata=golomb6(), red_tol_freq=0.5, n_angle_bins=200
    )
    assert fitting_grps == [
        [((0, 1),)],
        [((3, 4),)],
        [((1, 2),)],
        [((0, 2),)],
        [((4, 5),)],
        [((2, 3),), ((3, 5),), ((2, 4),), ((1, 3),), ((0, 3),), ((1, 4),), ((0, 4),), ((2, 5),)],
        [((1, 5),), ((0, 5),)],
    ]
    assert [len(c) for c in fitting_vec_centers] == [1, 1, 1, 1, 1, 8, 2]
    assert np.allclose(fitting_vec_centers[5][0], [12.0, 0.0, 0.0]) and np.allclose(fitting_vec_centers[6][1], [34.0, 0.0, 0.0])
    assert connections[((2, 3),)] == {((3, 5),)} and grp_labels[((2, 5),)] == ((2, 3),)


def test_group_finder_needs_matching_angles_and_conjugates():
    # two parallel baselines of overlapping length, one stored west-pointing in the data: still one fitting group;
    # a third at a different angle stays alone
    antpos = np.array([[0.0, 0.0, 0.0], [20.0, 0.0, 0.0], [44.0, 0.0, 0.0], [0.0, 21.0, 0.0]])
    freqs = 100e6 + 1e6 * np.arange(30)
    uvd = SimpleUVData(antpos, [(0, 1), (2, 1), (0, 3)], freqs, np.array([2458000.0]))
    grps, centers, _, _ = modeling.get_uv_overlapping_grps_conjugated(uvd, n_angle_bins=200)
    sizes = sorted(len(g) for g in grps)
    assert sizes == [1, 2]
    big = [g for g in grps if len(g) == 2][0]
    assert {rg[0] for rg in big} == {(0, 1), (1, 2)}


def test_yield_mixed_comps_threshold_and_shapes():
    uvd = golomb6()
    freqs = uvd.freq_array[0] if np.ndim(uvd.freq_array) == 2 else uvd.freq_array
    grps, centers, _, _ = modeling.get_uv_overlapping_grps_conjugated(uvd)
    comps = modeling.yield_mixed_comps(grps, centers, freqs, ant_dly=2.0 / 0.3, grp_size_threshold=5)
    big = tuple(grps[5])
    assert big in comps and comps[big].shape[0] == 8 * len(freqs)
    assert np.allclose(comps[big].T @ comps[big], np.eye(comps[big].shape[1]), atol=1e-9)
    # the pair and the singles fall back to per-redundant-group DPSS bases keyed (red_grp,)
    for g in grps[:5] + [grps[6]]:
        for red_grp in g:
            assert comps[(red_grp,)].shape[0] == len(freqs)
    assert sum(len(k) for k in comps) == 15
    # threshold 1 keeps the pair as a joint group (the reference's own fixtures use grp_size_threshold=1)
    comps1 = modeling.yield_mixed_comps(grps, centers, freqs, ant_dly=2.0 / 0.3, grp_size_threshold=1)
    assert tuple(grps[6]) in comps1 and comps1[tuple(grps[6])].shape[0] == 2 * len(freqs)


def test_mixed_dictionary_tensorizes_to_row_blocks():
    """Counterpart of test_calibration.py:285-338: every baseline of a joint group points at its own Nfreqs row block."""
    uvd = golomb6()
    freqs = uvd.freq_array[0] if np.ndim(uvd.freq_array) == 2 else uvd.freq_array
    grps, centers, _, _ = modeling.get_uv_overlapping_grps_conjugated(uvd)
    comps = modeling.yield_mixed_comps(grps, centers, freqs, ant_dly=2.0 / 0.3, grp_size_threshold=1)
    ants_map = {a: a for a in range(6)}
    prob, corr_inds = calibration.tensorize_fg_model_comps_dict(comps, ants_map, nfreqs=len(freqs), dtype=np.float64, grp_size_threshold=1)
    assert prob.nbls == 15
    seen = set()
    for g in range(prob.ngrps):
        blk = prob.basis[prob.grp_basis[g]]
        for b in range(prob.grp_bl_start[g], prob.grp_bl_start[g + 1]):
            ap = (int(prob.bl_ant0[b]), int(prob.bl_ant1[b]))
            seen.add(ap)
            fit = [k for k in comps if any(ap in rg for rg in k)][0]
            rnum = [n for n, rg in enumerate(fit) if ap in rg][0]
            assert prob.bl_rowblk[b] == rnum
            assert np.array_equal(blk[rnum * len(freqs) : (rnum + 1) * len(freqs)], comps[fit][rnum * len(freqs) : (rnum + 1) * len(freqs)])
    assert len(seen) == 15
