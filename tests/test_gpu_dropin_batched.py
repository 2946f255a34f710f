"""The kept entry point with its time loop as ONE batch (calibrate_and_model_tensor(batch_slices=...), devices=[...]).

The reference fits the (polarization, time) slices one after another (calibration.py:1160-1167, :1244-1269).  The batched
call must return what that loop returns: the same ``fit_history[pol][time]["loss"]`` (length included: every slice applies
the tolerance test of :712-717 to its own loss), gains, model and residual -- with slices that stop at different steps, a
skipped slice (:1173-1177, :1301-1309), use_min (:702-710), both regularisers -- to rounding (fp64: 1e-10), and a call that
shares the fitting groups out over two workers (here: two workers on the one GPU, exchanging through host memory) must equal
the one-worker call."""
import time as _time

import numpy as np
import pytest

from calamity_amd import calibration, synthetic, uvcompat

pytestmark = pytest.mark.gpu


def _five_times(seed=7, nants=7, nfreqs=64, ntimes=5, skip=3):
    """Five times of one array: different noise levels (so the fits stop at different steps), one time flagged away."""
    uvd, sky, vecs = synthetic.make_uvdata(nants=nants, nfreqs=nfreqs, ntimes=ntimes, seed=seed, redundant=True, flag_frac=0.02)
    rng = np.random.default_rng(seed)
    times = np.unique(uvd.time_array)
    amp = np.sqrt(np.mean(np.abs(uvd.data_array) ** 2))
    for k, t in enumerate(times):
        sel = np.isclose(uvd.time_array, t, atol=1e-7, rtol=0.0)
        noise = amp * 10.0 ** (-2.0 - 0.5 * k)
        shape = uvd.data_array[sel].shape
        uvd.data_array[sel] += noise * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))
    if skip is not None:
        sel = np.isclose(uvd.time_array, times[skip], atol=1e-7, rtol=0.0)
        uvd.flag_array[sel] = True
    return uvd, sky


def _same(a, b, rtol):
    assert np.linalg.norm(np.asarray(a) - np.asarray(b)) <= rtol * max(np.linalg.norm(np.asarray(b)), 1e-300)


def _equal_outputs(out1, out2, ntimes, rtol, skipped=()):
    (m1, r1, g1, h1), (m2, r2, g2, h2) = out1, out2
    assert sorted(h1) == sorted(h2)
    for pol in h1:
        assert sorted(h1[pol]) == sorted(h2[pol]) == [t for t in range(ntimes) if t not in skipped]
        for ti in h1[pol]:
            l1, l2 = np.asarray(h1[pol][ti]["loss"], dtype=np.float64), np.asarray(h2[pol][ti]["loss"], dtype=np.float64)
            assert len(l1) == len(l2), (pol, ti, len(l1), len(l2))
            np.testing.assert_allclose(l1, l2, rtol=rtol)
    _same(m1.data_array, m2.data_array, rtol)
    _same(r1.data_array, r2.data_array, rtol * 1e2)  # a difference of nearly equal numbers
    _same(g1.gain_array, g2.gain_array, rtol)
    assert np.array_equal(g1.flag_array, g2.flag_array) and np.array_equal(m1.flag_array, m2.flag_array) and np.array_equal(r1.flag_array, r2.flag_array)


@pytest.mark.parametrize("reg, use_min", [("sum", True), ("post_hoc", False)])
def test_batched_call_equals_the_loop(reg, use_min):
    uvd, sky = _five_times()
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=600, tol=3e-9, correct_resid=True,
              correct_model=True, optimizer="Adam", learning_rate=1e-2, dtype=np.float64, model_regularization=reg, use_min=use_min)
    loop = calibration.calibrate_and_model_dpss(batch_slices=False, **kw)
    batched = calibration.calibrate_and_model_dpss(**kw)  # the default
    _equal_outputs(loop, batched, 5, 1e-10, skipped=(3,))
    n = [len(loop[3][0][t]["loss"]) for t in sorted(loop[3][0])]
    assert len(set(n)) > 1 and min(n) < 600, n  # the slices did stop on their own, at different steps
    # the skipped time: fully flagged, zero model, unity gains (test_calibration.py:633-640)
    t3 = np.unique(uvd.time_array)[3]
    sel = np.isclose(uvd.time_array, t3, atol=1e-7, rtol=0.0)
    m, r, g, _ = batched
    assert np.all(m.flag_array[sel]) and np.all(r.flag_array[sel]) and np.all(m.data_array[sel] == 0)
    assert np.all(uvcompat.gain4(g.gain_array)[:, :, 3, 0] == 1.0) and np.all(uvcompat.gain4(g.flag_array)[:, :, 3, 0])
    # two at a time: batches of 2, 2 (the skipped slice takes no place)
    two = calibration.calibrate_and_model_dpss(batch_slices=2, **kw)
    _equal_outputs(loop, two, 5, 1e-10, skipped=(3,))


@pytest.mark.parametrize("layout", ["shared", "stream"])
def test_batched_float32_and_layouts(layout):
    uvd, sky = _five_times(seed=11, skip=None)
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=60, tol=1e-30, optimizer="Adamax",
              learning_rate=1e-2, dtype=np.float32, layout=layout, nsamples_in_weights=True, use_model_snr_weights=True)
    loop = calibration.calibrate_and_model_dpss(batch_slices=False, **kw)
    batched = calibration.calibrate_and_model_dpss(**kw)
    _equal_outputs(loop, batched, 5, 2e-4)


def test_two_workers_on_one_gpu_equal_one_worker():
    """devices=[0, 0]: the fitting groups of every slice are shared out over two workers (threads of this process, each with
    its own solver) that exchange gain gradients and per-slice loss sums every step -- against the one-worker call."""
    uvd, sky = _five_times(seed=13, nants=8)
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=300, tol=3e-9, correct_resid=True,
              optimizer="Adam", learning_rate=1e-2, dtype=np.float64, use_min=True)
    one = calibration.calibrate_and_model_dpss(devices=[0], **kw)
    two = calibration.calibrate_and_model_dpss(devices=[0, 0], **kw)
    _equal_outputs(one, two, 5, 1e-10, skipped=(3,))
    three = calibration.calibrate_and_model_dpss(devices=[0, 0, 0], model_regularization=None, **dict(kw, sky_model=sky))
    one_n = calibration.calibrate_and_model_dpss(devices=[0], model_regularization=None, **dict(kw, sky_model=sky))
    _equal_outputs(one_n, three, 5, 1e-10, skipped=(3,))


def test_batching_raises_the_step_rate_of_small_fits():
    """Tutorial scale (20 antennas x 64 channels): a step of one slice is bound by launch latency, sixteen slices per launch
    cost little more -- the per-slice step rate of a T = 16 call against a T = 1 call (>= 8x)."""
    from calamity_amd.batched import SliceBatchFitter

    p, _, start = synthetic.make_config("tutorial")
    rates = {}
    for nt in (1, 16):
        f = SliceBatchFitter(p, nt, dtype=np.float32, layout="shared", devices=[0])
        cat = lambda a: np.concatenate([a] * nt)  # noqa: E731
        f.set_data(cat(p.data_r), cat(p.data_i), cat(p.wgts))
        f.set_params(cat(start["g_r"]), cat(start["g_i"]), cat(start["c_r"]), cat(start["c_i"]))
        f.set_regularization(None)
        f.set_optimizer("Adam", learning_rate=1e-2)
        f.run_slices(64, record=False)
        best = 0.0
        for _ in range(3):
            t0 = _time.perf_counter()
            res = f.run_slices(2048, record=True, tol=0.0)
            dt = _time.perf_counter() - t0
            best = max(best, 2048 * nt / dt)
        assert all(len(r[0]) == 2048 for r in res)
        rates[nt] = best
        f.close()
    print(f"slice-steps/s: T=1 {rates[1]:.0f}, T=16 {rates[16]:.0f} ({rates[16] / rates[1]:.1f}x)")
    assert rates[16] >= 7.0 * rates[1], rates  # (measured 8.5-9x; the margin is for box-to-box spread)


def test_two_polarizations_batched():
    """Two polarizations x three times = six slices in one batch (fit_history keyed by polarization, then time), one of them
    skipped, against the loop."""
    uvd1, sky1, _ = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=3, seed=17, redundant=True, flag_frac=0.02)
    rng = np.random.default_rng(17)
    d1, f1 = uvcompat.vis3(uvd1.data_array), uvcompat.vis3(uvd1.flag_array)
    d2 = 0.6 * d1 * np.exp(0.3j) + 1e-3 * np.abs(d1).mean() * (rng.standard_normal(d1.shape) + 1j * rng.standard_normal(d1.shape))
    data = np.concatenate([d1, d2], axis=2)
    flags = np.concatenate([f1, rng.random(f1.shape) < 0.03], axis=2)
    t1 = np.unique(uvd1.time_array)[1]
    flags[np.isclose(uvd1.time_array, t1, atol=1e-7, rtol=0.0), :, 1] = True  # (yy, time 1): skipped
    uvd = uvcompat.SimpleUVData(uvd1.antenna_positions, uvd1.get_antpairs(), uvcompat.freqs_1d(uvd1), np.unique(uvd1.time_array),
                                pols=(-5, -6), data=data[:, None], flags=flags[:, None])
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=150, tol=1e-9, correct_resid=True,
              optimizer="Adam", learning_rate=1e-2, dtype=np.float64)
    loop = calibration.calibrate_and_model_dpss(batch_slices=False, **kw)
    batched = calibration.calibrate_and_model_dpss(**kw)
    (m1, r1, g1, h1), (m2, r2, g2, h2) = loop, batched
    assert sorted(h1) == sorted(h2) == [0, 1] and sorted(h2[0]) == [0, 1, 2] and sorted(h2[1]) == [0, 2]
    for pol in h1:
        for ti in h1[pol]:
            np.testing.assert_allclose(np.asarray(h2[pol][ti]["loss"], dtype=np.float64), np.asarray(h1[pol][ti]["loss"], dtype=np.float64), rtol=1e-10)
    _same(m2.data_array, m1.data_array, 1e-10)
    _same(g2.gain_array, g1.gain_array, 1e-10)
    _same(r2.data_array, r1.data_array, 1e-8)
    assert np.array_equal(g1.flag_array, g2.flag_array) and np.array_equal(m1.flag_array, m2.flag_array)


def test_batch_fitter_through_a_one_rank_rccl_communicator():
    """The batch fitter's exchange path over RCCL on a one-GPU box: a single worker joins a one-rank communicator from its own
    thread (what the workers of several devices do), so the set-up agreement and an RCCL all-reduce of the gain gradients and the
    per-slice sums run every step -- same losses and parameters as the plain batched fit, bit for bit."""
    from calamity_amd.batched import SliceBatchFitter

    p, _, start = synthetic.make_problem(9, 96, f0=150e6, df=400e3, seed=3, with_sky=True)
    nt = 3
    outs = []
    for comm in (False, True):
        f = SliceBatchFitter(p, nt, dtype=np.float64, layout="shared", devices=[0], communicator_of_one=comm)
        assert f.solvers[0].comm_size() == 1
        cat = lambda a: np.concatenate([a * (1.0 + 0.1 * t) for t in range(nt)])  # noqa: E731
        w = np.concatenate([p.wgts] * nt)
        f.set_data(cat(p.data_r), cat(p.data_i), w)
        f.set_params(np.concatenate([start["g_r"]] * nt), np.concatenate([start["g_i"]] * nt), cat(start["c_r"]), cat(start["c_i"]))
        pr = np.asarray([float(np.sum(p.sky_r * (1.0 + 0.1 * t) * p.wgts)) for t in range(nt)])
        pi = np.asarray([float(np.sum(p.sky_i * (1.0 + 0.1 * t) * p.wgts)) for t in range(nt)])
        f.set_regularization("sum", pr, pi)
        f.set_optimizer("Adamax", learning_rate=1e-2)
        f.run_slices(1, record=False)
        res = f.run_slices(30, record=True, tol=0.0, use_min=True)
        outs.append((res, f.get_params(0), f.get_params(1)))
        f.close()
    for t in range(nt):
        assert np.array_equal(outs[0][0][t][0], outs[1][0][t][0])
    for a, b in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("nworkers", [2, 3])
def test_lamb_over_several_workers(nworkers):
    """LAMB takes one trust ratio per VARIABLE (g_r, g_i, fg_r[chunk], fg_i[chunk]: calibration.py:596-603, :26).  With the fitting
    groups of a chunk shared out over several workers the two sums of squares of every (slice, chunk) are exchanged each step
    (a second, small all-reduce): same losses, gains and coefficients as ONE worker holding everything.  Three chunks, two time
    slices, workers sharing the one GPU (exchange through host memory)."""
    from calamity_amd.batched import SliceBatchFitter

    p, _, start = synthetic.make_problem(9, 96, f0=150e6, df=400e3, seed=5, with_sky=True)
    p.chunk_of_grp = (np.arange(p.ngrps) // 12).astype(np.int32)
    assert p.chunk_of_grp.max() == 2
    nt = 2
    cat = lambda a: np.concatenate([a * (1.0 + 0.2 * t) for t in range(nt)])  # noqa: E731
    outs = []
    for devices in ([0], [0] * nworkers):
        f = SliceBatchFitter(p, nt, dtype=np.float64, layout="stream", devices=devices)
        f.set_data(cat(p.data_r), cat(p.data_i), np.concatenate([p.wgts] * nt))
        f.set_params(np.concatenate([start["g_r"]] * nt), np.concatenate([start["g_i"]] * nt), cat(start["c_r"]), cat(start["c_i"]))
        f.set_regularization(None)
        f.set_optimizer("LAMB", learning_rate=2e-2, weight_decay_rate=1e-3)
        f.run_slices(1, record=False)
        res = f.run_slices(25, record=True, tol=0.0)
        outs.append((res, f.get_params()))
        f.close()
    for t in range(nt):
        np.testing.assert_allclose(outs[1][0][t][0], outs[0][0][t][0], rtol=1e-9)
        assert outs[0][0][t][0][-1] < 0.9 * outs[0][0][t][0][0]  # (it does descend)
    for a, b in zip(outs[1][1], outs[0][1]):
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b)


def test_a_failed_set_up_gives_its_solvers_back():
    """A batch fitter whose set-up fails (here: a device that does not exist) closes the solvers it had created and raises."""
    from calamity_amd import _lib
    from calamity_amd.batched import SliceBatchFitter

    p, _, _ = synthetic.make_problem(6, 32, f0=150e6, df=400e3, seed=1)
    with pytest.raises(_lib.CalamityHipError):
        SliceBatchFitter(p, 2, dtype=np.float32, devices=[0, 99])
    f = SliceBatchFitter(p, 2, dtype=np.float32, devices=[0])  # (and the device is as usable as before)
    assert f.memory_bytes() > 0
    f.close()


def test_batches_on_different_devices_equal_one_device_bit_for_bit():
    """device_split="slices": whole batches of slices go to different devices (here two and three workers on the one GPU, each on a
    thread of its own with its own solvers), nothing is exchanged, and every slice is fitted exactly as on one device: fit_history,
    gains, model and residual are IDENTICAL to the one-device call -- with a skipped time, slices that stop at different steps,
    use_min, the "sum" regulariser, batches of one and of two slices."""
    uvd, sky = _five_times(seed=21, nants=8)
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=200, tol=3e-9, correct_resid=True,
              optimizer="Adam", learning_rate=1e-2, dtype=np.float64, use_min=True)
    one = calibration.calibrate_and_model_dpss(devices=[0], batch_slices=1, **kw)
    for devices, per_batch in (([0, 0], 1), ([0, 0, 0], 1), ([0, 0], 2)):
        many = calibration.calibrate_and_model_dpss(devices=devices, device_split="slices", batch_slices=per_batch, **kw)
        _equal_outputs(one, many, 5, 0.0 if per_batch == 1 else 1e-10, skipped=(3,))
    # the default picks "slices" when there are at least as many batches as devices
    auto = calibration.calibrate_and_model_dpss(devices=[0, 0], batch_slices=1, **kw)
    _equal_outputs(one, auto, 5, 0.0, skipped=(3,))
    with pytest.raises(ValueError):
        calibration.calibrate_and_model_dpss(devices=[0, 0], device_split="times", **kw)


def test_devices_default_is_one_device_and_all_falls_back_aloud(monkeypatch):
    """devices=None fits on the process's ONE device however many GPUs are visible (several devices are an explicit request: a
    default must not stake a run on a communicator set-up it was never asked for); devices="all" on a node where the set-up of
    the visible devices fails together -- here a second device the library cannot open -- continues on one device with a
    RuntimeWarning and returns what the one-device call returns, bit for bit."""
    import warnings

    from calamity_amd import _lib

    uvd, sky = _five_times(ntimes=3, skip=None)
    kw = dict(min_dly=2.0 / 0.3, offset=2.0 / 0.3, uvdata=uvd, gains=None, sky_model=None, maxsteps=60, tol=1e-30, optimizer="Adam",
              learning_rate=1e-2, dtype=np.float64, model_regularization="sum")
    one = calibration.calibrate_and_model_dpss(devices=[0], **kw)
    monkeypatch.setattr(_lib, "device_count", lambda: 2)  # "two GPUs visible"
    assert calibration._resolve_devices(None) == ([0], True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the default path must not even try the second device
        default = calibration.calibrate_and_model_dpss(**kw)
    _equal_outputs(one, default, 3, 0.0)
    with pytest.warns(RuntimeWarning, match="could not be set up on devices"):
        fallback = calibration.calibrate_and_model_dpss(devices="all", device_split="groups", **kw)
    _equal_outputs(one, fallback, 3, 0.0)
    with pytest.raises(Exception):  # an explicit list fails as it stands
        calibration.calibrate_and_model_dpss(devices=[0, 1], device_split="groups", **kw)
