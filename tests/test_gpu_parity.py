"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Tolerances: fp64 -- loss and gradients rel <= 1e-10, short trajectories rel <= 1e-8;
fp32 -- loss rel <= 1e-5, gradients norm-wise rel <= 1e-4, short trajectories <= 1e-3.
"""
import ctypes as C

import numpy as np
import pytest

from calamity_amd import problem, synthetic
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu

TOL = {np.float64: dict(loss=1e-10, grad=1e-10, traj=1e-8), np.float32: dict(loss=1e-5, grad=1e-4, traj=1e-3)}


def relnorm(a, b):
    a = np.asarray(a)
    a = a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def make_case(nants=9, nfreqs=40, seed=0, with_sky=False, redundant=False, perturb=True):
    p, truth, start = synthetic.make_problem(nants, nfreqs, f0=150e6, df=400e3, seed=seed, with_sky=with_sky)
    if redundant:
        p, start = synthetic.add_redundant_group(p, start, np.random.default_rng(seed), nred=3)
    rng = np.random.default_rng(seed + 17)
    if perturb:
        start["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
        start["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    return p, start


def oracle_inputs(p, start):
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, start["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, start["c_i"], np.float64)
    return ch, fg_r, fg_i


def make_solver(p, start, dtype, layout="stream", reg=False, kernel_path="auto"):
    from calamity_amd.solver import HipFitSolver

    s = HipFitSolver(dtype=dtype)
    s.set_problem(p, layout=layout, kernel_path=kernel_path)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    if reg:
        s.set_regularization("sum", np.sum(p.sky_r * p.wgts), np.sum(p.sky_i * p.wgts))
    return s


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("layout", ["stream", "shared"])
@pytest.mark.parametrize("reg", [False, True])
@pytest.mark.parametrize("redundant", [False, True])
def test_loss_and_gradients(dtype, layout, reg, redundant):
    p, start = make_case(seed=3, with_sky=reg, redundant=redundant)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    priors = R.prior_sums(ch["sky_model_r"], ch["sky_model_i"], ch["wgts"]) if reg else (None, None)
    loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(
        start["g_r"], start["g_i"], fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1, *priors
    )
    s = make_solver(p, start, dtype, layout, reg)
    tol = TOL[dtype]
    assert abs(s.eval_loss() - loss) <= tol["loss"] * abs(loss)
    l2, hg_r, hg_i, hc_r, hc_i = s.eval_grads()
    assert abs(l2 - loss) <= tol["loss"] * abs(loss)
    assert relnorm(hg_r, gg_r) <= tol["grad"]
    assert relnorm(hg_i, gg_i) <= tol["grad"]
    assert relnorm(hc_r, problem.coeffs_from_chunks(p, gf_r)) <= tol["grad"]
    assert relnorm(hc_i, problem.coeffs_from_chunks(p, gf_i)) <= tol["grad"]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("optimizer", ["Adam", "Adamax"])
@pytest.mark.parametrize("reg", [False, True])
def test_short_trajectory(dtype, optimizer, reg):
    """30 recorded steps (+1 unrecorded), same loop semantics as calibration.py:681-717."""
    p, start = make_case(seed=5, with_sky=reg, perturb=False)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    out = R.fit_gains_and_foregrounds(
        start["g_r"], start["g_i"], fg_r, fg_i, ch["data_r"], ch["data_i"], ch["wgts"], ch["fg_comps"], ch["corr_inds"],
        maxsteps=30, optimizer=optimizer, learning_rate=1e-2, sky_model_r=ch["sky_model_r"], sky_model_i=ch["sky_model_i"],
        model_regularization="sum" if reg else None,
    )
    s = make_solver(p, start, dtype, "stream", reg)
    s.set_optimizer(optimizer, learning_rate=1e-2)
    s.run(1, record=False)
    losses, stopped, nupd = s.run(30, record=True, tol=1e-14)
    assert len(losses) == 30 and not stopped and nupd == 30
    tol = TOL[dtype]
    np.testing.assert_allclose(losses, np.asarray(out[4]["loss"], dtype=np.float64), rtol=max(tol["traj"], 1e-7))
    g_r, g_i, c_r, c_i = s.get_params()
    g = g_r.astype(np.float64) + 1j * g_i
    assert relnorm(g, out[0] + 1j * out[1]) <= tol["traj"]
    assert relnorm(c_r, problem.coeffs_from_chunks(p, out[2])) <= tol["traj"]
    assert relnorm(c_i, problem.coeffs_from_chunks(p, out[3])) <= tol["traj"]


@pytest.mark.parametrize("dtype,path", [(np.float32, "dense"), (np.float32, "dense_split1"), (np.float32, "dense_f32"), (np.float64, "dense")])
@pytest.mark.parametrize("reg", [False, True])
def test_short_trajectory_dense_path(reg, dtype, path):
    """SHARED layout + one baseline per group -> the dense (matrix-core) path, fp32 and fp64 (two passes per step with the
    "sum" regulariser).  Problems this small normally take the general kernel; kernel_path="dense" asks for the dense one."""
    p, start = make_case(seed=6, nants=12, nfreqs=200, with_sky=reg, perturb=False)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    out = R.fit_gains_and_foregrounds(
        start["g_r"], start["g_i"], fg_r, fg_i, ch["data_r"], ch["data_i"], ch["wgts"], ch["fg_comps"], ch["corr_inds"],
        maxsteps=30, optimizer="Adam", learning_rate=1e-2, sky_model_r=ch["sky_model_r"], sky_model_i=ch["sky_model_i"],
        model_regularization="sum" if reg else None,
    )
    s = make_solver(p, start, dtype, "shared", reg, kernel_path=path)  # fp32 "dense": the split-bf16 kernel (one-image form), "dense_split1": its first form, "dense_f32": the fp32 MFMA kernel
    s.set_optimizer("Adam", learning_rate=1e-2)
    s.run(1, record=False)
    losses, stopped, nupd = s.run(30, record=True, tol=1e-14)
    tol = TOL[dtype]
    np.testing.assert_allclose(losses, np.asarray(out[4]["loss"], dtype=np.float64), rtol=tol["traj"])
    g_r, g_i, c_r, c_i = s.get_params()
    assert relnorm(g_r.astype(np.float64) + 1j * g_i, out[0] + 1j * out[1]) <= tol["traj"]
    assert relnorm(c_r, problem.coeffs_from_chunks(p, out[2])) <= tol["traj"]


def test_loop_controls_tol_usemin_freeze():
    p, start = make_case(seed=7, perturb=False)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    kw = dict(data_r=ch["data_r"], data_i=ch["data_i"], wgts=ch["wgts"], fg_comps=ch["fg_comps"], corr_inds=ch["corr_inds"],
              optimizer="Adam", learning_rate=5e-2)
    # tol stop: same number of recorded losses as the oracle
    ref = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, maxsteps=400, tol=1e-6, **kw)
    assert len(ref[4]["loss"]) < 400
    s = make_solver(p, start, np.float64)
    s.set_optimizer("Adam", learning_rate=5e-2)
    s.run(1, record=False)
    losses, stopped, nupd = s.run(400, record=True, tol=1e-6)
    assert stopped and len(losses) == len(ref[4]["loss"]) and nupd == len(losses)
    assert relnorm(s.get_params()[0], ref[0]) <= 1e-8
    # use_min: snapshot after the update of the lowest-loss step (large lr makes the loss non-monotonic)
    ref = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, maxsteps=60, use_min=True, **kw)
    s = make_solver(p, start, np.float64)
    s.set_optimizer("Adam", learning_rate=5e-2)
    s.run(1, record=False)
    losses, _, _ = s.run(60, record=True, use_min=True)
    np.testing.assert_allclose(losses, ref[4]["loss"], rtol=1e-8)
    g_r, g_i, c_r, c_i = s.get_params(which=1)
    assert relnorm(g_r, ref[0]) <= 1e-8 and relnorm(c_r, problem.coeffs_from_chunks(p, ref[2])) <= 1e-8
    # freeze_model: gains only
    ref = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, maxsteps=20, freeze_model=True, **kw)
    s = make_solver(p, start, np.float64)
    s.set_optimizer("Adam", learning_rate=5e-2)
    s.run(1, record=False, freeze_model=True)
    losses, _, _ = s.run(20, record=True, freeze_model=True)
    np.testing.assert_allclose(losses, ref[4]["loss"], rtol=1e-8)
    g_r, g_i, c_r, c_i = s.get_params()
    assert relnorm(g_r, ref[0]) <= 1e-8
    np.testing.assert_array_equal(c_r, start["c_r"])


def test_fp32_fit_stops_on_stagnation():
    """calibration.py:712 compares the float32 values loss.numpy() returns in a float32 fit: with the default tol = 1e-14
    such a fit stops when two consecutive float32 losses are EQUAL (the reference's usual way out of the loop).  Noisy data
    keep the loss of order one, where float32 resolves 1e-8: the double-accumulated losses the library records still
    differ by far more than tol at the stopping step, and a float64 fit of the same problem keeps going."""
    p, start = make_case(seed=11, perturb=True)
    rng = np.random.default_rng(5)
    p.data_r = p.data_r + 0.3 * rng.standard_normal(p.data_r.shape)
    p.data_i = p.data_i + 0.3 * rng.standard_normal(p.data_i.shape)
    s = make_solver(p, start, np.float32)
    s.set_optimizer("Adam", learning_rate=1e-2)
    s.run(1, record=False)
    losses, stopped, nupd = s.run(20000, record=True, tol=1e-14)
    assert stopped and 2 <= len(losses) < 20000 and nupd == len(losses)
    l32 = np.float32(losses)
    assert l32[-1] == l32[-2] and abs(losses[-1] - losses[-2]) > 1e-14
    assert not np.any(l32[1:-1] == l32[:-2])  # ... and it is the FIRST such pair
    s.close()
    s = make_solver(p, start, np.float64)
    s.set_optimizer("Adam", learning_rate=1e-2)
    s.run(1, record=False)
    l64, stopped64, _ = s.run(len(losses) + 50, record=True, tol=1e-14)
    assert not stopped64 and len(l64) == len(losses) + 50
    s.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_model_and_init_coeffs(dtype):
    p, start = make_case(seed=9, nants=8, nfreqs=52)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    s = make_solver(p, start, dtype)
    m_r, m_i = s.model()
    vr, vi = R.fg_model(fg_r[0], fg_i[0], ch["fg_comps"][0])
    tol = 1e-12 if dtype == np.float64 else 2e-6
    assert relnorm(m_r, vr.reshape(p.nbls, p.nfreqs)) <= tol and relnorm(m_i, vi.reshape(p.nbls, p.nfreqs)) <= tol
    # initial coefficients: tensorize_fg_coeffs (calibration.py:828-913)
    c0_r = problem.coeffs_from_chunks(p, R.tensorize_fg_coeffs(ch["data_r"], ch["wgts"], ch["fg_comps"]))
    c0_i = problem.coeffs_from_chunks(p, R.tensorize_fg_coeffs(ch["data_i"], ch["wgts"], ch["fg_comps"]))
    s.init_coeffs(p.data_r, p.data_i)
    _, _, c_r, c_i = s.get_params()
    tol = 1e-9 if dtype == np.float64 else 2e-5
    assert relnorm(c_r, c0_r) <= tol and relnorm(c_i, c0_i) <= tol


def test_split_items_small_problem_many_channels():
    """Few baselines x many channels: groups are split along tiles (partial coefficient gradients)."""
    p, start = make_case(seed=11, nants=5, nfreqs=700)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(
        start["g_r"], start["g_i"], fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1
    )
    s = make_solver(p, start, np.float64)
    l2, hg_r, hg_i, hc_r, hc_i = s.eval_grads()
    assert abs(l2 - loss) <= 1e-10 * abs(loss)
    assert relnorm(hg_r, gg_r) <= 1e-10 and relnorm(hc_r, problem.coeffs_from_chunks(p, gf_r)) <= 1e-10


def test_errors_are_reported():
    from calamity_amd import _lib
    from calamity_amd.solver import HipFitSolver

    s = HipFitSolver(dtype=np.float32)
    loss = C.c_double()
    with pytest.raises(_lib.CalamityHipError):
        _lib.check(s._lib.cal_solver_eval_loss(s._h, C.byref(loss)))  # no problem set yet
    p, start = make_case(seed=1)
    s.set_problem(p)
    with pytest.raises(KeyError):
        s.set_optimizer("NotAnOptimizer")
    bad = synthetic.make_problem(5, 32, seed=0)[0]
    bad.bl_ant1 = bad.bl_ant1.copy()
    bad.bl_ant1[0] = 99
    with pytest.raises((AssertionError, _lib.CalamityHipError)):
        s.set_problem(bad)


def test_single_rank_communicator_exercises_the_rccl_path():
    """A 1-rank RCCL communicator: the per-step all-reduce is issued (identity) and results are unchanged."""
    from calamity_amd.solver import comm_unique_id

    p, start = make_case(seed=13, with_sky=True)
    s0 = make_solver(p, start, np.float64, reg=True)
    s1 = make_solver(p, start, np.float64, reg=True)
    s1.comm_init(comm_unique_id(), 0, 1)
    for s in (s0, s1):
        s.set_optimizer("Adam", learning_rate=1e-2)
    l0, _, _ = s0.run(5, record=True)
    l1, _, _ = s1.run(5, record=True)
    np.testing.assert_array_equal(l0, l1)
    np.testing.assert_array_equal(s0.get_params()[0], s1.get_params()[0])
