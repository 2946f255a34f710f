"""GPU tests of the drop-in entry points, modelled on the reference's integration tests
(/root/reference/calamity/tests/test_calibration.py:350-463, :475-755).  Acceptance criteria are the reference's own:
rms(model) and rms(data) >= 100 x rms(resid) after <= 3000 steps (:593-596); gain recovery to 1e-4 and model to
1e-5 x rms in the freeze-model setting (:748-753); skipped times fully flagged / zero / unity gain (:633-640)."""
import copy
import glob

import numpy as np
import pytest

from calamity_amd import cal_utils, calibration, modeling, problem, synthetic, uvcompat
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def rms(x):
    return np.sqrt(np.mean(np.abs(x) ** 2.0))


# every drop-in test runs on both pyuvdata array vintages: with the length-1 spw axis (the layout the reference indexes,
# calibration.py:262-278) and without it (pyuvdata >= 3)
@pytest.fixture(scope="module", params=[False, True], ids=["spw_axis", "future_shapes"])
def sets(request):
    return synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=1, seed=2, future_shapes=request.param)


@pytest.fixture(scope="module", params=[False, True], ids=["spw_axis", "future_shapes"])
def sets_multitime(request):
    return synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=2, seed=3, future_shapes=request.param)


def randomized_gains(uvd, seed=0, sigma=1e-2):
    g = cal_utils.blank_uvcal_from_uvdata(uvd)
    rng = np.random.default_rng(seed)
    g.gain_array = g.gain_array + sigma * (rng.standard_normal(g.gain_array.shape) + 1j * rng.standard_normal(g.gain_array.shape))
    return g


def test_init_coeffs_and_model_round_trip(sets):
    """test_calibration.py:416-463 through the GPU: lstsq coefficients -> A c -> insert reproduces the projected sky."""
    uvd, sky, vecs = sets
    gains = cal_utils.blank_uvcal_from_uvdata(sky)
    ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
    red_grps = modeling.get_redundant_grps_data(sky, remove_redundancy=True)[1]
    comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, dtype=np.float64, nfreqs=sky.Nfreqs)
    rmsdata = rms(sky.data_array)
    t0 = sky.time_array[0]
    data_r, data_i, wgts = calibration.tensorize_data(sky, corr_inds, ants_map, polarization="xx", time=t0, dtype=np.float64, data_scale_factor=rmsdata)
    c_re = calibration.tensorize_fg_coeffs(data_r, wgts, comps)
    c_im = calibration.tensorize_fg_coeffs(data_i, wgts, comps)
    # against the oracle's restatement of calibration.py:828-913 on the padded tensors
    padded = problem.chunks_from_problem(comps)["fg_comps"]
    np.testing.assert_allclose(c_re[0], R.tensorize_fg_coeffs(data_r, wgts, padded)[0], atol=1e-9)
    nants, nfreqs = sky.Nants_data, sky.Nfreqs
    model_r = calibration.yield_fg_model_array(nants, nfreqs, comps, c_re, corr_inds, dtype=np.float64)
    model_i = calibration.yield_fg_model_array(nants, nfreqs, comps, c_im, corr_inds, dtype=np.float64)
    np.testing.assert_allclose(model_r, R.yield_fg_model_array(nants, nfreqs, padded, c_re, corr_inds), atol=1e-10)
    inserted = copy.deepcopy(sky)
    rng = np.random.default_rng(0)
    inserted.data_array = rng.standard_normal(inserted.data_array.shape) + 1j * rng.standard_normal(inserted.data_array.shape)
    calibration.insert_model_into_uvdata_tensor(inserted, t0, "xx", ants_map, red_grps, model_r, model_i, scale_factor=rmsdata)
    assert np.allclose(inserted.data_array, sky.data_array)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fit_gains_and_foregrounds_reference_signature(dtype):
    """The inner seam with the reference's own argument layout (zero-padded chunk tensors, calibration.py:447-473)
    against the oracle: same recorded losses and parameters after 25 Adamax steps (the reference's default)."""
    p, truth, start = synthetic.make_problem(8, 40, f0=150e6, df=400e3, seed=21, with_sky=True)
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, start["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, start["c_i"], np.float64)
    kw = dict(data_r=ch["data_r"], data_i=ch["data_i"], wgts=ch["wgts"], fg_comps=ch["fg_comps"], corr_inds=ch["corr_inds"],
              maxsteps=25, learning_rate=1e-2, sky_model_r=ch["sky_model_r"], sky_model_i=ch["sky_model_i"], model_regularization="sum")
    ref = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, **kw)
    out = calibration.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, dtype=dtype, **kw)
    tol = 1e-8 if dtype == np.float64 else 1e-3
    assert len(out[4]["loss"]) == 25 and out[4]["loss"][0].dtype == dtype
    np.testing.assert_allclose(np.asarray(out[4]["loss"], dtype=np.float64), ref[4]["loss"], rtol=max(tol, 1e-6))
    assert np.linalg.norm(out[0] - ref[0]) <= tol * np.linalg.norm(ref[0])
    assert out[2][0].shape == ref[2][0].shape
    assert np.linalg.norm(out[2][0] - ref[2][0]) <= tol * np.linalg.norm(ref[2][0])
    with pytest.raises(KeyError):
        calibration.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, optimizer="NotAnOptimizer", **kw)


@pytest.mark.parametrize("noweights, perfect_data, use_min", [(True, True, False), (True, False, False), (False, False, True)])
def test_calibrate_and_model_dpss(sets, noweights, perfect_data, use_min):
    """test_calibration.py:553-596."""
    uvd, sky, vecs = sets
    weights = None if noweights else uvcompat.SimpleUVFlag(sky)
    if perfect_data:
        data, g0 = sky, cal_utils.blank_uvcal_from_uvdata(sky)
    else:
        data, g0 = uvd, randomized_gains(sky)
    before = copy.deepcopy(data.data_array)
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=data, gains=g0, verbose=False, use_redundancy=False, sky_model=None,
        maxsteps=3000, tol=1e-10, correct_resid=True, correct_model=True, weights=weights, use_min=use_min,
    )
    assert rms(model.data_array) >= 1e2 * rms(resid.data_array)
    assert rms(uvd.data_array) >= 1e2 * rms(resid.data_array)
    assert len(fit_history) == 1 and len(fit_history[0]) == 1
    assert gains is g0  # a supplied gains object is modified in place and returned (calibration.py:1294-1300)
    assert np.array_equal(before, data.data_array)  # the input uvdata is never modified (:1111-1116)
    assert fit_history[0][0]["loss"][0].dtype == np.float32


def test_calibrate_and_model_dpss_multitime(sets_multitime):
    """test_calibration.py:475-516."""
    uvd, sky, vecs = sets_multitime
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=randomized_gains(sky), use_redundancy=False, sky_model=None,
        maxsteps=3000, tol=1e-10, correct_resid=True, correct_model=True, weights=None,
    )
    assert rms(model.data_array) >= 1e2 * rms(resid.data_array)
    assert rms(uvd.data_array) >= 1e2 * rms(resid.data_array)
    assert len(fit_history) == 1 and len(fit_history[0]) == 2


def test_init_guesses_from_previous_time_step():
    """calibration.py:1196-1233: with init_guesses_from_previous_time_step every time after the first starts from the gains
    and coefficients the previous time ended with instead of the input gains and the least-squares coefficients.  Two times
    holding the SAME visibilities: without the flag the two fits are identical; with it the second one starts where the
    first one stopped and the first is unchanged."""
    uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=2, seed=11)
    v = uvcompat.vis3(uvd.data_array)
    t0 = np.isclose(uvd.time_array, np.unique(uvd.time_array)[0], rtol=0.0, atol=1e-7)
    v[~t0] = v[t0]  # blt order is time-major in the synthetic sets: the same baselines in the same order
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, use_redundancy=False, sky_model=None, maxsteps=300, tol=0.0,
              optimizer="Adam", learning_rate=1e-2, model_regularization="post_hoc")
    def same_gains_at_both_times():
        g = randomized_gains(sky, seed=4)
        ga = uvcompat.gain4(g.gain_array)
        ga[:, :, 1] = ga[:, :, 0]
        return g

    _, _, _, cold = calibration.calibrate_and_model_dpss(gains=same_gains_at_both_times(), **kw)
    _, _, _, warm = calibration.calibrate_and_model_dpss(gains=same_gains_at_both_times(), init_guesses_from_previous_time_step=True, **kw)
    c0, c1 = np.asarray(cold[0][0]["loss"]), np.asarray(cold[0][1]["loss"])
    w0, w1 = np.asarray(warm[0][0]["loss"]), np.asarray(warm[0][1]["loss"])
    assert np.array_equal(c0, c1)             # same data, same start: the same fit twice
    assert np.array_equal(w0, c0)             # the first time does not know about the flag
    # the second fit starts from the first one's result -- its first recorded loss comes behind the unrecorded "graph build"
    # update (:693), a full-size first Adam step away from that result, and is still well below the cold start's
    assert w1[0] < 0.5 * c1[0]
    assert not np.array_equal(w1, c1)
    assert w1.min() <= 1.05 * w0.min()


def test_calibrate_and_model_dpss_flagged(sets_multitime):
    """test_calibration.py:610-653: a time with too little unflagged data is skipped: flagged, zero model, unity gains."""
    uvd, sky, vecs = sets_multitime
    uvd = copy.deepcopy(uvd)
    t_bad = np.unique(uvd.time_array)[1]
    sel = np.isclose(uvd.time_array, t_bad, rtol=0.0, atol=1e-7)
    uvcompat.vis3(uvd.flag_array)[sel, : int(0.8 * uvd.Nfreqs)] = True
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=randomized_gains(sky), sky_model=None, maxsteps=3000, tol=1e-10,
        correct_resid=True, correct_model=True, skip_threshold=0.5,
    )
    assert np.all(model.flag_array[sel]) and np.all(resid.flag_array[sel])
    assert np.all(model.data_array[sel] == 0.0) and np.all(resid.data_array[sel] == 0.0)
    assert np.all(uvcompat.gain4(gains.flag_array)[:, :, 1]) and np.allclose(uvcompat.gain4(gains.gain_array)[:, :, 1], 1.0)
    assert not np.any(model.flag_array[~sel])
    assert rms(model.data_array[~sel]) >= 1e2 * rms(resid.data_array[~sel])
    assert len(fit_history[0]) == 1 and 0 in fit_history[0]


def test_calibrate_and_model_dpss_redundant_and_options(tmp_path):
    """test_calibration.py:656-696: redundant array with use_redundancy (shared coefficients), "sum" regularisation,
    nsamples_in_weights, use_model_snr_weights, graph_mode accepted, profile steps write a log."""
    uvd, sky, vecs = synthetic.make_uvdata(nants=7, nfreqs=48, ntimes=1, seed=5, redundant=True, eor_db=-80.0)
    logdir = str(tmp_path / "logdir")
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=randomized_gains(sky), use_redundancy=True, sky_model=None,
        maxsteps=3000, tol=1e-10, correct_resid=True, correct_model=True, graph_mode=True, model_regularization="sum",
        nsamples_in_weights=True, use_model_snr_weights=True, n_profile_steps=2, profile_log_dir=logdir, learning_rate=1e-2,
    )
    assert rms(model.data_array) >= 1e2 * rms(resid.data_array)
    assert len(glob.glob(logdir + "/*")) > 0


def test_calibrate_and_model_dpss_dont_correct_resid(sets):
    """test_calibration.py:699-727."""
    uvd, sky, vecs = sets
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=randomized_gains(sky), sky_model=None, maxsteps=3000, tol=1e-10,
        correct_resid=False, correct_model=False,
    )
    assert rms(model.data_array) >= 1e2 * rms(resid.data_array)
    # uncorrected model = g g* x corrected model; resid = data - uncorrected model
    assert np.allclose(resid.data_array, uvd.data_array - model.data_array, atol=1e-10 * rms(uvd.data_array))


def test_calibrate_and_model_dpss_freeze_model(sets):
    """test_calibration.py:730-755: data = gains x sky; with the model frozen at the true sky the gain amplitudes are
    recovered to 1e-4 and the model to 1e-5 x rms."""
    uvd, sky, vecs = sets
    g_true = cal_utils.blank_uvcal_from_uvdata(sky)
    for i, antnum in enumerate(g_true.ant_array):
        g_true.gain_array[i] *= 1.0 + 0.02 * (antnum + 1.0)
    data = cal_utils.apply_gains(sky, g_true, inverse=True)
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=data, gains=None, sky_model=sky, freeze_model=True, maxsteps=10000, tol=1e-14,
        correct_resid=True, correct_model=True, dtype=np.float64, optimizer="Adam", learning_rate=1e-2,
        model_regularization=None,  # the "sum" prior is built from the sky model and would bias gains away from 1
    )
    assert rms(model.data_array) >= 1e2 * rms(resid.data_array)
    err = np.abs(np.abs(gains.gain_array) - np.abs(g_true.gain_array))
    # a harder variant than the reference's (next test): true gains 2-12 % off unity and unknown to the fit, which starts
    # at unity; channels where the sky is faint converge last
    assert np.median(err) <= 1e-4 and np.max(err) <= 2e-3
    assert np.allclose(model.data_array, sky.data_array, atol=1e-5 * rms(sky.data_array))


def test_calibrate_and_model_dpss_freeze_model_reference_setup(sets):
    """test_calibration.py:730-755 with the reference's own set-up: the data ARE the projected sky model (true gains 1), the
    fit starts from gains randomised by 1e-2 (:80-84), model frozen, defaults otherwise (Adamax, float32, "sum" prior,
    3000 steps, tol 1e-10).  The reference asserts |gains| against the very object the fit mutated (:753), which cannot
    fail; the bound it states, 1e-4 absolute on every gain amplitude, is asserted here against the TRUE gains."""
    uvd, sky, vecs = sets
    start = randomized_gains(sky, seed=5, sigma=1e-2)
    weights = uvcompat.SimpleUVFlag(sky, mode="flag")
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=sky, gains=start, use_redundancy=False, sky_model=sky, freeze_model=True,
        maxsteps=3000, tol=1e-10, correct_resid=True, correct_model=True, weights=weights,
    )
    assert gains is start  # mutated in place and returned (:1294-1300)
    assert rms(model.data_array) >= 1e2 * rms(resid.data_array)
    assert np.allclose(model.data_array, sky.data_array, atol=1e-5 * rms(model.data_array))
    err = np.abs(np.abs(uvcompat.gain4(gains.gain_array)[:, :, 0, 0]) - 1.0)  # (antenna, channel)
    # With the reference's settings the loop ends on its tolerance test after ~150 steps (|loss_k - loss_k-1| < 1e-10,
    # :712), well before the gains have converged to 1e-4 everywhere: 90 % of the gain amplitudes are within 5e-5, the
    # worst within 2.4e-4 (measured; the same in float64).  That is what the reference's set-up delivers ...
    assert len(fit_history[0][0]["loss"]) < 3000
    assert np.quantile(err, 0.9) <= 1e-4 and np.max(err) <= 5e-4
    assert len(fit_history) == 1 and len(fit_history[0]) == 1
    # ... and the bound it states, 1e-4 on EVERY gain amplitude, holds once the fit is allowed to converge (tol 1e-14,
    # float64, Adam lr 1e-2: 7e-7 measured)
    model, resid, gains2, _ = calibration.calibrate_and_model_dpss(
        min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=sky, gains=randomized_gains(sky, seed=5, sigma=1e-2), use_redundancy=False, sky_model=sky,
        freeze_model=True, maxsteps=10000, tol=1e-14, correct_resid=True, correct_model=True, weights=weights, dtype=np.float64,
        optimizer="Adam", learning_rate=1e-2,
    )
    assert np.allclose(np.abs(gains2.gain_array), 1.0, rtol=0.0, atol=1e-4)
    assert np.allclose(model.data_array, sky.data_array, atol=1e-5 * rms(model.data_array))


def test_calibrate_and_model_dpss_post_hoc_heavy_flags():
    """test_calibration.py:519-541: post_hoc renormalisation with heavy flagging produces no NaNs."""
    uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=1, seed=7, flag_frac=0.3)
    model, resid, gains, fit_history = calibration.calibrate_and_model_dpss(
        min_dly=4.0 / 0.3, offset=20.0, uvdata=uvd, gains=None, sky_model=None, maxsteps=200, tol=1e-10, correct_resid=True,
        correct_model=True, model_regularization="post_hoc",
    )
    assert np.all(np.isfinite(resid.data_array)) and np.all(np.isfinite(model.data_array)) and np.all(np.isfinite(gains.gain_array))


def test_read_calibrate_and_model_dpss(tmp_path, monkeypatch):
    """test_calibration.py:880-940: files in, files out, then the same through the argument parser with --precision 64
    and autocorrelation weights; plus the device-memory cap."""
    import os
    import sys

    uvd, sky, vecs = synthetic.make_uvdata(nants=5, nfreqs=48, ntimes=1, seed=4, redundant=True)
    # the driver derives weights from autocorrelations: add flat autos to the data set
    nants = uvd.antenna_positions.shape[0]
    pairs = uvd.get_antpairs() + [(a, a) for a in range(nants)]
    full = uvcompat.SimpleUVData(uvd.antenna_positions, pairs, uvcompat.freqs_1d(uvd), np.unique(uvd.time_array), x_orientation="east")
    for ap in uvd.get_antpairs():
        full.data_array[full.antpair2ind(ap)] = uvd.data_array[uvd.antpair2ind(ap)]
    for a in range(nants):
        full.data_array[full.antpair2ind((a, a))] = 50.0 + a
    data_path, gain_path = str(tmp_path / "data.uvh5"), str(tmp_path / "gains_input.calfits")
    full.write_uvh5(data_path)
    g = cal_utils.blank_uvcal_from_uvdata(uvd)
    g.x_orientation = "east"
    g.write_calfits(gain_path)
    outs = [str(tmp_path / n) for n in ("resid_fit.uvh5", "model_fit.uvh5", "gains_fit.calfits")]
    model, resid, gains, info = calibration.read_calibrate_and_model_dpss(
        input_data_files=data_path, input_model_files=data_path, input_gain_files=gain_path, resid_outfilename=outs[0],
        model_outfilename=outs[1], gain_outfilename=outs[2], maxsteps=50,
    )
    assert info["calibration_kwargs"]["dtype"] == np.float32
    # the strict east-west bound (|b_ew| > 0, utils.py:29) cuts the autos and any purely north-south baseline
    pos = uvd.antenna_positions
    assert set(model.get_antpairs()) == {ap for ap in uvd.get_antpairs() if abs(pos[ap[0], 0] - pos[ap[1], 0]) > 0}
    for fn in outs:
        assert os.path.exists(fn)
        os.remove(fn)
    monkeypatch.setattr(sys, "argv", [sys.argv[0], "--input_data_files", data_path, "--input_model_files", data_path,
                                      "--input_gain_files", gain_path, "--resid_outfilename", outs[0], "--model_outfilename", outs[1],
                                      "--gain_outfilename", outs[2], "--precision", "64", "--use_autocorrs_in_weights",
                                      "--maxsteps", "50", "--gpu_index", "0"])
    args = calibration.dpss_fit_argparser().parse_args()
    _, _, gains2, info = calibration.read_calibrate_and_model_dpss(**vars(args))
    assert info["calibration_kwargs"]["dtype"] == np.float64
    assert gains2.x_orientation == "east" and np.all(np.isfinite(gains2.gain_array))
    for fn in outs:
        assert os.path.exists(fn)
    back = uvcompat.read_container(outs[2])
    assert np.array_equal(back.gain_array, gains2.gain_array)
    # existing outputs are only replaced on request (the reference's --clobber defaults to a truthy string, :1831)
    with pytest.raises(IOError):
        calibration.read_calibrate_and_model_dpss(**vars(args))
    args.clobber = True
    calibration.read_calibrate_and_model_dpss(**vars(args))
    with pytest.raises(MemoryError):
        calibration.read_calibrate_and_model_dpss(input_data_files=data_path, maxsteps=5, gpu_memory_limit=1e-6)


def test_parallel_fits_equal_the_sequential_loop():
    """Four time slices fitted three at a time (one solver and stream per worker thread) give exactly what the sequential
    loop of calibration.py:1167 gives: the slices are independent and every kernel is deterministic."""
    import time as _time

    uvd, sky, vecs = synthetic.make_uvdata(nants=7, nfreqs=64, ntimes=4, seed=5, redundant=True, flag_frac=0.02)
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=400, tol=1e-12, correct_resid=True,
              correct_model=True, optimizer="Adam", learning_rate=1e-2)
    t0 = _time.perf_counter()
    m1, r1, g1, h1 = calibration.calibrate_and_model_dpss(**kw)
    t1 = _time.perf_counter()
    m2, r2, g2, h2 = calibration.calibrate_and_model_dpss(parallel_fits=3, **kw)
    t2 = _time.perf_counter()
    assert np.array_equal(m1.data_array, m2.data_array) and np.array_equal(r1.data_array, r2.data_array)
    assert np.array_equal(g1.gain_array, g2.gain_array) and np.array_equal(g1.flag_array, g2.flag_array)
    assert sorted(h1[0]) == sorted(h2[0]) == [0, 1, 2, 3]
    for ti in range(4):
        assert np.array_equal(h1[0][ti]["loss"], h2[0][ti]["loss"])
    print(f"sequential {t1 - t0:.2f} s, 3 at a time {t2 - t1:.2f} s")
