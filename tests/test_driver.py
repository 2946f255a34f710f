"""File / command-line driver pieces, CPU part: argument parsers (the reference's
test_calibration.py:758-765), baseline cuts (utils.py:13-37), autocorrelation weights (calibration.py:916-960) and the
container round trip used when pyuvdata is not installed."""
import sys

import numpy as np
import pytest

from calamity_amd import calibration, utils
from calamity_amd.uvcompat import SimpleUVData, read_container


def array_with_autos(nants=4, nfreqs=48, seed=0, ntimes=2):
    rng = np.random.default_rng(seed)
    antpos = np.array([[0.0, 0.0, 0.0], [14.6, 0.0, 0.0], [29.2, 0.0, 0.0], [7.3, 12.6, 0.0]])[:nants]
    freqs = 100e6 + 400e3 * np.arange(nfreqs)
    antpairs = [(i, j) for i in range(nants) for j in range(i, nants)]
    uvd = SimpleUVData(antpos, antpairs, freqs, 2458000.0 + np.arange(ntimes), x_orientation="east")
    bandpass = 1.0 + 0.1 * np.sin(2 * np.pi * (freqs - freqs[0]) / 120e6)  # 8 ns: inside the 25 ns fit
    level = 10.0 + rng.uniform(0, 5, size=nants)
    for n in range(uvd.Nblts):
        i, j = uvd.ant_1_array[n], uvd.ant_2_array[n]
        if i == j:
            uvd.data_array[n, 0, :, 0] = level[i] * bandpass
        else:
            uvd.data_array[n, 0, :, 0] = rng.standard_normal(nfreqs) + 1j * rng.standard_normal(nfreqs)
    return uvd, level, bandpass


def test_dpss_fit_argparser_defaults(monkeypatch):
    monkeypatch.setattr(sys, "argv", [sys.argv[0], "--input_data_files", "input.uvh5"])
    args = calibration.dpss_fit_argparser().parse_args()
    assert args.learning_rate == 1e-2
    assert args.tol == 1e-14
    assert args.maxsteps == 10000
    assert args.input_data_files == ["input.uvh5"]
    # CLI defaults that differ from the Python API's
    assert args.model_regularization == "post_hoc" and args.optimizer == "Adamax" and args.nsamples_in_weights is False
    assert args.precision == 32 and args.horizon == 1.0 and args.min_dly == 0.0 and args.offset == 0.0
    # every parsed name is an argument of the driver or of the fit
    import inspect

    accepted = set(inspect.signature(calibration.read_calibrate_and_model_dpss).parameters)
    accepted |= set(inspect.signature(calibration.calibrate_and_model_dpss).parameters)
    accepted |= set(inspect.signature(calibration.calibrate_and_model_tensor).parameters) | {"learning_rate"}
    assert set(vars(args)) <= accepted


def test_select_baselines():
    uvd, _, _ = array_with_autos()
    n0 = len(uvd.get_antpairs())
    cut = SimpleUVData.select(uvd, inplace=False)
    utils.select_baselines(cut, bllen_min=10.0, bllen_max=20.0)
    kept = set(cut.get_antpairs())
    assert kept == {(0, 1), (1, 2), (0, 3), (1, 3)} and len(uvd.get_antpairs()) == n0
    cut = SimpleUVData.select(uvd, inplace=False)
    utils.select_baselines(cut, bl_ew_min=10.0, ex_ants=[2])
    assert set(cut.get_antpairs()) == {(0, 1)}
    cut = SimpleUVData.select(uvd, inplace=False)
    utils.select_baselines(cut, select_ants=[0, 3])
    assert set(cut.get_antpairs()) == {(0, 3)}  # autos have no east-west extent: the strict bl_ew_min=0 cut drops them


def test_get_auto_weights():
    uvd, level, bandpass = array_with_autos()
    uvd.flag_array[uvd.antpair2ind(0, 1)[0], 0, 5:9, 0] = True
    w = calibration.get_auto_weights(uvd)
    assert w.weights_array.shape == uvd.data_array.shape
    d01 = w.antpair2ind(0, 1)
    expect = 1.0 / (level[0] * level[1] * bandpass**2)
    assert np.allclose(w.weights_array[d01[1], 0, :, 0], expect, rtol=1e-2)
    assert np.all(w.weights_array[d01[0], 0, 5:9, 0] == 0.0)
    assert np.allclose(np.delete(w.weights_array[d01[0], 0, :, 0], np.arange(5, 9)), np.delete(expect, np.arange(5, 9)), rtol=1e-2)
    # flagged channels of an autocorrelation are interpolated over by the smooth fit
    uvd.flag_array[uvd.antpair2ind(2, 2)[0], 0, 20:24, 0] = True
    uvd.data_array[uvd.antpair2ind(2, 2)[0], 0, 20:24, 0] = 1e6
    w2 = calibration.get_auto_weights(uvd)
    d12 = w2.antpair2ind(1, 2)
    assert np.allclose(w2.weights_array[d12[0], 0, :, 0], 1.0 / (level[1] * level[2] * bandpass**2), rtol=1e-2)


def test_container_round_trip(tmp_path):
    uvd, _, _ = array_with_autos()
    path = str(tmp_path / "data.uvh5")
    uvd.write_uvh5(path)
    with pytest.raises(IOError):
        uvd.write_uvh5(path)
    uvd.write_uvh5(path, clobber=True)
    back = read_container(path)
    assert np.array_equal(back.data_array, uvd.data_array) and back.get_antpairs() == uvd.get_antpairs()
