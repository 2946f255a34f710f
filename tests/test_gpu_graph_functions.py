"""The reference's graph functions under their own names -- fg_model, data_model, mse, mse_chunked, mse_chunked_sum_regularized
(/root/reference/calamity/calibration.py:1587-1656) -- evaluated by the HIP library on the reference's zero-padded chunk tensors,
against the NumPy restatement of the same lines (oracle/ref_numpy.py:27-87) on the same inputs.  fp64: 1e-12; fp32: 1e-5."""
import numpy as np
import pytest

from calamity_amd import calibration as cal
from oracle import ref_numpy as R
from test_gpu_parity import make_case, oracle_inputs, relnorm

pytestmark = pytest.mark.gpu


def chunk_inputs(dtype, redundant, seed=4):
    p, start = make_case(seed=seed, with_sky=True, redundant=redundant)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    cast = lambda xs: [np.asarray(x, dtype=dtype) for x in xs]  # noqa: E731
    t = dict(g_r=np.asarray(start["g_r"], dtype=dtype), g_i=np.asarray(start["g_i"], dtype=dtype), fg_r=cast(fg_r), fg_i=cast(fg_i),
             fg_comps=cast(ch["fg_comps"]), data_r=cast(ch["data_r"]), data_i=cast(ch["data_i"]), wgts=cast(ch["wgts"]), a0=a0, a1=a1,
             priors=R.prior_sums(ch["sky_model_r"], ch["sky_model_i"], ch["wgts"]))
    return t


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 1e-5)])
@pytest.mark.parametrize("redundant", [False, True])
def test_models_of_every_chunk(dtype, tol, redundant):
    t = chunk_inputs(dtype, redundant)
    f64 = lambda x: np.asarray(x, dtype=np.float64)  # noqa: E731
    for c in range(len(t["fg_comps"])):
        vr, vi = cal.fg_model(t["fg_r"][c], t["fg_i"][c], t["fg_comps"][c])
        wr, wi = R.fg_model(f64(t["fg_r"][c]), f64(t["fg_i"][c]), f64(t["fg_comps"][c]))
        assert vr.shape == wr.shape and vr.dtype == dtype
        assert relnorm(vr + 1j * vi, wr + 1j * wi) <= tol
        mr, mi = cal.data_model(t["g_r"], t["g_i"], t["fg_r"][c], t["fg_i"][c], t["fg_comps"][c], t["a0"][c], t["a1"][c])
        nr, ni = R.data_model(f64(t["g_r"]), f64(t["g_i"]), f64(t["fg_r"][c]), f64(t["fg_i"][c]), f64(t["fg_comps"][c]), t["a0"][c], t["a1"][c])
        assert mr.shape == nr.shape and relnorm(mr + 1j * mi, nr + 1j * ni) <= tol
        # mse of that model against the chunk's data: the device sum against the restatement's
        got = cal.mse(mr, mi, t["data_r"][c], t["data_i"][c], t["wgts"][c])
        want = R.mse(f64(mr), f64(mi), f64(t["data_r"][c]), f64(t["data_i"][c]), f64(t["wgts"][c]))
        assert abs(got - want) <= tol * abs(want)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 1e-5)])
@pytest.mark.parametrize("redundant", [False, True])
def test_chunked_losses(dtype, tol, redundant):
    t = chunk_inputs(dtype, redundant)
    n = len(t["fg_comps"])
    f64 = lambda xs: [np.asarray(x, dtype=np.float64) for x in xs]  # noqa: E731
    args = (t["g_r"], t["g_i"], t["fg_r"], t["fg_i"], t["fg_comps"], n, t["data_r"], t["data_i"], t["wgts"], t["a0"], t["a1"])
    ref_args = (np.float64(t["g_r"]), np.float64(t["g_i"]), f64(t["fg_r"]), f64(t["fg_i"]), f64(t["fg_comps"]), n, f64(t["data_r"]), f64(t["data_i"]),
                f64(t["wgts"]), t["a0"], t["a1"])
    got = cal.mse_chunked(*args, dtype=dtype)
    want = R.mse_chunked(*ref_args)
    assert isinstance(got, dtype) and abs(got - want) <= tol * abs(want)
    got = cal.mse_chunked_sum_regularized(*args, *t["priors"], dtype=dtype)
    want_reg = R.mse_chunked_sum_regularized(*ref_args, *t["priors"])
    assert abs(got - want_reg) <= tol * abs(want_reg)
    # priors away from the model's sums: the penalty terms are what is added
    far = (t["priors"][0] + 0.3, t["priors"][1] - 0.2)
    got = cal.mse_chunked_sum_regularized(*args, *far, dtype=dtype)
    want_far = R.mse_chunked_sum_regularized(*ref_args, *far)
    assert abs(got - want_far) <= tol * abs(want_far) and want_far > want
    # the chunk sets are cached by identity: a second call on the same tensors reuses the uploaded basis and agrees bit for bit
    assert cal.mse_chunked(*args, dtype=dtype) == cal.mse_chunked(*args, dtype=dtype)


def test_mse_edge_cases():
    z = np.zeros((0, 3, 8))
    assert cal.mse(z, z, z, z, z) == 0.0
    rng = np.random.default_rng(0)
    a = [rng.standard_normal((3, 2, 33)) for _ in range(4)]
    w = rng.random((3, 2, 33))
    w[1] = 0.0  # flagged samples carry no weight
    want = R.mse(*a, w)
    assert abs(cal.mse(*a, w) - want) <= 1e-13 * abs(want)
    big = [rng.standard_normal(1_000_003).astype(np.float32) for _ in range(4)]
    wb = rng.random(1_000_003).astype(np.float32)
    want = R.mse(*[np.float64(x) for x in big], np.float64(wb))
    assert abs(cal.mse(*big, wb) - want) <= 1e-5 * abs(want)
