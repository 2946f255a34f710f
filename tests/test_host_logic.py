"""CPU tests of the host side: layout builders, UVData/UVCal plumbing, the C-ABI library's symbol table.
Modelled on the reference's unit tests (/root/reference/calamity/tests/test_calibration.py:222-463, :599-607)."""
import copy
import ctypes
import os
import re

import numpy as np
import pytest

from calamity_amd import _lib, cal_utils, calibration, modeling, problem, synthetic, uvcompat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=[False, True], ids=["spw_axis", "future_shapes"])
def sets(request):
    """Both pyuvdata array vintages: with the length-1 spw axis and without (pyuvdata >= 3)."""
    uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=48, ntimes=2, seed=1, future_shapes=request.param)
    return uvd, sky, vecs


@pytest.fixture
def gains(sets):
    return cal_utils.blank_uvcal_from_uvdata(sets[1])


def test_library_exports_every_declared_symbol():
    """Every function declared in include/calamity_hip.h is exported by the built library and bound by _lib."""
    header = open(os.path.join(ROOT, "include", "calamity_hip.h")).read()
    declared = set(re.findall(r"\b(cal_[a-z_0-9]+)\s*\(", header))
    declared -= {"cal_solver", "cal_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()  # build() must have run; raises otherwise
    for name in declared:
        assert hasattr(lib, name)
    assert lib.cal_version().decode().startswith("calamity_hip")
    # null handle -> error code + message, never a crash
    assert lib.cal_solver_synchronize(None) == -1
    assert b"null solver handle" in lib.cal_last_error()


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly instead of computing on the host."""
    n = ctypes.c_int(0)
    if _lib.load().cal_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    from calamity_amd.solver import HipFitSolver

    with pytest.raises(_lib.CalamityHipError):
        HipFitSolver()


def test_dpss_operator_properties():
    freqs = 100e6 + 400e3 * np.arange(64)
    A = modeling.yield_dpss_model_comps_bl_grp(30.0, freqs)
    assert A.shape[0] == 64 and np.allclose(A.T @ A, np.eye(A.shape[1]), atol=1e-10)
    # a longer baseline needs more modes; min_dly and offset widen the filter (modeling.py:293)
    assert modeling.yield_dpss_model_comps_bl_grp(120.0, freqs).shape[1] > A.shape[1]
    assert modeling.dly_ns(30.0) == 100.0 and modeling.dly_ns(30.0, offset=5.5) == 106.0 and modeling.dly_ns(3.0, min_dly=50.0) == 50.0
    # same rounded delay -> the SAME ndarray object through the operator cache (de-duplicated on upload)
    cache = {}
    a = modeling.yield_dpss_model_comps_bl_grp(30.0, freqs, operator_cache=cache)
    b = modeling.yield_dpss_model_comps_bl_grp(29.9, freqs, operator_cache=cache)
    assert a is b


def test_get_redundancies_hex():
    """7-element hexagon: 21 baselines in 9 redundant groups (3 orientations x {1, sqrt3, 2} spacings... known answer)."""
    pos = synthetic.hex_positions(7)
    uvd = uvcompat.SimpleUVData(pos, [(i, j) for i in range(7) for j in range(i + 1, 7)], np.linspace(1e8, 2e8, 8), [2458000.0])
    grps, centers, lengths, _ = uvd.get_redundancies(use_antpos=True, include_conjugates=True, include_autos=False, tol=1.0)
    assert sum(len(g) for g in grps) == 21
    assert len(grps) == 9
    for c in centers:  # east-positive orientation
        assert c[0] > -1e-9
    antpairs, red_grps, vbc, lens = modeling.get_redundant_grps_data(uvd, remove_redundancy=True)
    assert len(red_grps) == 21 and all(len(g) == 1 for g in red_grps)


def test_chunk_fg_comp_dict_by_nbls(sets):
    """test_calibration.py:274-278: per-baseline DPSS -> ONE chunk keyed (1, maxvecs)."""
    vecs = sets[2]
    chunked = calibration.chunk_fg_comp_dict_by_nbls(vecs)
    maxvecs = np.max([vecs[k].shape[1] for k in vecs])
    assert len(chunked) == 1 and list(chunked.keys())[0] == (1, maxvecs)
    # a redundant fitting group of 3 baselines is split into 3 single-baseline groups unless use_redundancy
    A = list(vecs.values())[0]
    d = {(((0, 1), (1, 2), (2, 3)),): A}
    assert list(calibration.chunk_fg_comp_dict_by_nbls(d).keys()) == [(1, A.shape[1])]
    assert list(calibration.chunk_fg_comp_dict_by_nbls(d, use_redundancy=True).keys()) == [(3, A.shape[1])]


def test_tensorize_fg_model_comps_dpss(sets, gains):
    """test_calibration.py:244-271: padded rows zero, non-padded rows equal the dictionary entries."""
    uvd, sky, vecs = sets
    ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
    comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, nfreqs=sky.Nfreqs, dtype=np.float64)
    padded = problem.chunks_from_problem(comps)["fg_comps"]
    ncomps = 0
    for cnum in range(len(corr_inds)):
        for gnum in range(len(corr_inds[cnum])):
            for blnum, bl in enumerate(corr_inds[cnum][gnum]):
                t = padded[cnum][:, gnum, blnum]
                d = vecs[((bl,),)].T
                assert np.allclose(d, t[: d.shape[0]]) and np.allclose(0.0, t[d.shape[0] :])
                ncomps += 1
    assert ncomps == len(vecs)
    # ragged <-> padded round trip
    prob2 = problem.problem_from_chunks(comps.nants, padded, corr_inds, *[[np.zeros(p.shape[1:]) for p in padded]] * 3)
    assert np.array_equal(prob2.grp_nvec, comps.grp_nvec) and np.array_equal(prob2.bl_ant0, comps.bl_ant0)
    c = np.arange(comps.ncoeffs, dtype=np.float64)
    assert np.array_equal(problem.coeffs_from_chunks(comps, problem.coeffs_to_chunks(comps, c, np.float64)), c)


def test_tensorize_gains(gains):
    """test_calibration.py:233-241."""
    for i, antnum in enumerate(gains.ant_array):
        gains.gain_array[i] *= antnum + 1.0
    g_r, g_i = calibration.tensorize_gains(gains, polarization="xx", time=gains.time_array[0], dtype=np.float64)
    assert g_r.dtype == np.float64 and g_i.dtype == np.float64
    for i, ant in enumerate(gains.ant_array):
        assert np.allclose(g_r[ant], ant + 1) and np.allclose(g_i[ant], 0.0)
    calibration.insert_gains_into_uvcal(gains, gains.time_array[1], "xx", 2 * g_r, g_i + 1)
    assert np.allclose(uvcompat.gain4(gains.gain_array)[:, :, 1, 0], 2 * g_r + 1j * (g_i + 1))


def test_tensorize_data_weights_and_conjugation(sets, gains):
    uvd, sky, vecs = sets
    ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
    comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, nfreqs=sky.Nfreqs)
    sky = copy.deepcopy(sky)
    uvcompat.vis3(sky.flag_array)[3, 5:9, 0] = True
    sky.nsample_array[:] = 2.0
    t0 = np.unique(sky.time_array)[0]
    d_r, d_i, w = calibration.tensorize_data(sky, corr_inds, ants_map, "xx", t0, data_scale_factor=3.0, dtype=np.float64)
    assert np.isclose(sum(x.sum() for x in w), 1.0)
    i, j = corr_inds[0][3][0]
    dind = sky.antpair2ind((i, j))
    if len(dind) == 0:
        dind = sky.antpair2ind((j, i))
    assert np.all(w[0][3, 0, :] >= 0)
    for g, grp in enumerate(corr_inds[0]):
        (a, b) = grp[0]
        inds = sky.antpair2ind((a, b))
        rev_order = len(inds) == 0  # the redundancy finder orients pairs east-positive; the data may hold (b, a)
        n = calibration._time_ind(sky.time_array, sky.antpair2ind((b, a)) if rev_order else inds, t0)
        row = np.conj(uvcompat.vis3(sky.data_array)[n, :, 0]) if rev_order else uvcompat.vis3(sky.data_array)[n, :, 0]
        assert np.allclose(d_r[0][g, 0] + 1j * d_i[0][g, 0], row / 3.0)
        assert np.all((w[0][g, 0] == 0) == uvcompat.vis3(sky.flag_array)[n, :, 0])
    # nsamples weights: the same after normalisation when nsamples is uniform
    _, _, w2 = calibration.tensorize_data(sky, corr_inds, ants_map, "xx", t0, nsamples_in_weights=True, dtype=np.float64)
    assert np.allclose(w2[0], w[0])
    # a pair stored in reversed order is conjugated (calibration.py:263-278)
    rev = uvcompat.SimpleUVData(sky.antenna_positions, [ap[::-1] for ap in sky.get_antpairs()], uvcompat.freqs_1d(sky), np.unique(sky.time_array),
                                data=np.conj(sky.data_array), flags=sky.flag_array, future_shapes=sky.future_array_shapes)
    r_r, r_i, _ = calibration.tensorize_data(rev, corr_inds, ants_map, "xx", t0, data_scale_factor=3.0, dtype=np.float64)
    assert np.allclose(r_r[0], d_r[0]) and np.allclose(r_i[0], d_i[0])
    # UVFlag-style weights
    uvf = uvcompat.SimpleUVFlag(sky)
    uvf.weights_array[:] = 0.5
    _, _, w3 = calibration.tensorize_data(sky, corr_inds, ants_map, "xx", t0, weights=uvf, dtype=np.float64)
    assert np.allclose(w3[0], w[0])


def test_insert_model_round_trip(sets, gains):
    """test_calibration.py:416-463 without the GPU: cubes built on the host go back into the UVData exactly."""
    uvd, sky, vecs = sets
    ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
    red_grps = modeling.get_redundant_grps_data(sky, remove_redundancy=True)[1]
    t0 = np.unique(sky.time_array)[0]
    nants, nfreqs = sky.Nants_data, sky.Nfreqs
    model_r = np.zeros((nants, nants, nfreqs))
    model_i = np.zeros_like(model_r)
    for red_grp in red_grps:
        for ap in red_grp:
            i, j = ants_map[ap[0]], ants_map[ap[1]]
            inds = sky.antpair2ind(ap)
            conj = len(inds) == 0
            n = calibration._time_ind(sky.time_array, sky.antpair2ind(ap[::-1]) if conj else inds, t0)
            row = np.conj(uvcompat.vis3(sky.data_array)[n, :, 0]) if conj else uvcompat.vis3(sky.data_array)[n, :, 0]
            model_r[i, j], model_i[i, j] = row.real / 7.0, row.imag / 7.0
    inserted = copy.deepcopy(sky)
    rng = np.random.default_rng(0)
    inserted.data_array = rng.standard_normal(inserted.data_array.shape) + 1j * rng.standard_normal(inserted.data_array.shape)
    calibration.insert_model_into_uvdata_tensor(inserted, t0, "xx", ants_map, red_grps, model_r, model_i, scale_factor=7.0)
    sel = np.isclose(sky.time_array, t0, rtol=0.0, atol=1e-7)
    assert np.allclose(inserted.data_array[sel], sky.data_array[sel])
    assert not np.allclose(inserted.data_array[~sel], sky.data_array[~sel])


def test_renormalize(sets, gains):
    """test_calibration.py:222-230."""
    sky = copy.deepcopy(sets[1])
    gains.gain_array *= (51.0 + 23j) ** -0.5
    ref = copy.deepcopy(sky)
    sky.data_array *= 51.0 + 23j
    for t in np.unique(sky.time_array):
        calibration.renormalize(ref, sky, gains, polarization="xx", time=t)
    assert np.allclose(np.abs(gains.gain_array), 1.0)
    assert np.allclose(np.abs(ref.data_array), np.abs(sky.data_array))


def test_flag_poltime(sets, gains):
    """test_calibration.py:599-607."""
    uvd = copy.deepcopy(sets[1])
    t = np.unique(uvd.time_array)[1]
    calibration.flag_poltime(uvd, time=t, polarization="xx")
    sel = np.isclose(uvd.time_array, t, rtol=0.0, atol=1e-7)
    assert np.all(uvd.flag_array[sel]) and not np.any(uvd.flag_array[~sel]) and np.all(uvd.data_array[sel] == 0)
    calibration.flag_poltime(gains, time=t, polarization="xx")
    assert np.all(uvcompat.gain4(gains.flag_array)[:, :, 1]) and not np.any(uvcompat.gain4(gains.flag_array)[:, :, 0])
    assert np.all(uvcompat.gain4(gains.gain_array)[:, :, 1] == 1.0)
    with pytest.raises(ValueError):
        calibration.flag_poltime(np.zeros(3), time=t, polarization="xx")


def test_apply_gains_round_trip(sets, gains):
    uvd = sets[0]
    rng = np.random.default_rng(3)
    gains.gain_array = gains.gain_array + 0.1 * (rng.standard_normal(gains.gain_array.shape) + 1j * rng.standard_normal(gains.gain_array.shape))
    cal = cal_utils.apply_gains(uvd, gains)
    back = cal_utils.apply_gains(cal, gains, inverse=True)
    assert np.allclose(back.data_array, uvd.data_array) and not np.allclose(cal.data_array, uvd.data_array)
    n = 4
    a0, a1 = uvd.ant_1_array[n], uvd.ant_2_array[n]
    expect = uvcompat.vis3(uvd.data_array)[n, :, 0] / (uvcompat.gain4(gains.gain_array)[a0, :, 0, 0] * np.conj(uvcompat.gain4(gains.gain_array)[a1, :, 0, 0]))
    assert np.allclose(uvcompat.vis3(cal.data_array)[n, :, 0], expect)
    uvcompat.gain4(gains.flag_array)[2, 7, 0, 0] = True
    fl = uvcompat.vis3(cal_utils.apply_gains(uvd, gains).flag_array)
    touched = (uvd.ant_1_array == 2) | (uvd.ant_2_array == 2)
    t0 = np.isclose(uvd.time_array, gains.time_array[0], rtol=0.0, atol=1e-7)
    assert np.all(fl[touched & t0, 7, 0]) and not np.any(fl[~touched, 7, 0])


def test_unknown_optimizer_is_a_keyerror():
    from calamity_amd.solver import OPTIMIZERS

    with pytest.raises(KeyError):
        OPTIMIZERS["NotAnOptimizer"]
    # calibration.py:17-27: the whole table
    assert set(OPTIMIZERS) == {"Adam", "Adamax", "SGD", "RMSprop", "Adagrad", "Adadelta", "Nadam", "Ftrl", "LAMB"}


def test_build_guard_rejects_a_spilling_dense_kernel():
    """calamity_amd/csrc/check_resources.py reads hipcc's -Rpass-analysis=kernel-resource-usage remarks in every build: a
    dense kernel that uses scratch must fail it (scratch traffic would break the counted waits of its operand ring); other
    kernels may spill."""
    import subprocess
    import sys

    script = os.path.join(ROOT, "calamity_amd", "csrc", "check_resources.py")

    def remarks(name, scratch, vspill, occ=2):
        return (f"./x.hpp:1:1: remark: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]\n"
                f"./x.hpp:1:1: remark:     VGPRs: 250 [-Rpass-analysis=kernel-resource-usage]\n"
                f"./x.hpp:1:1: remark:     ScratchSize [bytes/lane]: {scratch} [-Rpass-analysis=kernel-resource-usage]\n"
                f"./x.hpp:1:1: remark:     Occupancy [waves/SIMD]: {occ} [-Rpass-analysis=kernel-resource-usage]\n"
                f"./x.hpp:1:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n"
                f"./x.hpp:1:1: remark:     VGPRs Spill: {vspill} [-Rpass-analysis=kernel-resource-usage]\n")

    def run(text):
        return subprocess.run([sys.executable, script], input=text, capture_output=True, text=True)

    ok = run(remarks("_ZN4calk18fused_dense_kernelILb1EEEvNS_8MfmaArgsE", 0, 0) + remarks("_ZN4calk12adam2_kernelIfLi0EEEv", 16, 4))
    assert ok.returncode == 0, ok.stderr
    bad = run(remarks("_ZN4calk20fused_dense64_kernelILb1EEEvNS_11Dense64ArgsE", 8, 1))
    assert bad.returncode != 0 and "fused_dense64_kernel" in bad.stderr and "VGPRs Spill = 1" in bad.stderr
    assert run(remarks("_ZN4calk18fused_dense_kernelILb1EEEvNS_8MfmaArgsE", 0, 0, occ=1)).returncode != 0
    assert run(remarks("_ZN4calk12adam2_kernelIfLi0EEEv", 0, 0)).returncode != 0  # no dense kernel in the log at all
