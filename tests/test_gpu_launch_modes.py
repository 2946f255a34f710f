"""GPU tests of the launch forms of a train step (include/calamity_hip.h: cal_launch_mode) and of checkpoint / resume.

The reference's loop (/root/reference/calamity/calibration.py:699-717) pays a host synchronisation per step (:701).  Here
a step of a small problem is two launches replayed from a hipGraph; "kernels" issues every kernel on its own (per-antenna
reduction, bookkeeping, regulariser fold, update).  All forms must produce the SAME numbers, bit for bit: recorded
losses, parameters, use_min snapshots, the step a tolerance stop or a non-finite loss lands on.
"""
import numpy as np
import pytest

from calamity_amd import _lib
from test_gpu_parity import make_case, make_solver

pytestmark = pytest.mark.gpu

MODES = ["kernels", "one_tail", "graph"]


def run_fit(p, start, dtype, mode, optimizer="Adam", lr=2e-2, reg=False, nsteps=53, layout="stream", **run_kw):
    s = make_solver(p, start, dtype, layout=layout, reg=reg)
    s.set_launch_mode(mode)
    s.set_optimizer(optimizer, learning_rate=lr)
    s.run(1, record=False, freeze_model=run_kw.get("freeze_model", False))  # the unrecorded step of calibration.py:693
    losses, stopped, nupd = s.run(nsteps, record=True, **run_kw)
    cur = s.get_params()
    snap = s.get_params(which=1) if run_kw.get("use_min") else None
    s.close()
    return losses, stopped, nupd, cur, snap


def assert_same(a, b):
    assert a[1] == b[1] and a[2] == b[2]
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[3], b[3]):
        np.testing.assert_array_equal(x, y)
    if a[4] is not None:
        for x, y in zip(a[4], b[4]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("optimizer", ["Adam", "Adamax"])
@pytest.mark.parametrize("reg", [False, True])
@pytest.mark.parametrize("layout", ["stream", "shared"])
def test_trajectories_bit_identical_across_launch_modes(dtype, optimizer, reg, layout):
    """53 recorded steps = three graph replays of 16 + 5 single steps (+ the unrecorded one in front), use_min on."""
    p, start = make_case(seed=21, with_sky=reg, nants=10, nfreqs=72)
    ref = run_fit(p, start, dtype, "kernels", optimizer=optimizer, reg=reg, layout=layout, use_min=True, tol=0.0)
    assert len(ref[0]) == 53 and np.all(np.isfinite(ref[0])) and ref[0][-1] < ref[0][0]
    for mode in MODES[1:] + ["auto"]:
        assert_same(ref, run_fit(p, start, dtype, mode, optimizer=optimizer, reg=reg, layout=layout, use_min=True, tol=0.0))


@pytest.mark.parametrize("dtype,reg", [(np.float32, False), (np.float32, True), (np.float64, False)])
def test_dense_path_steps_replayed_from_a_graph_equal_kernel_by_kernel(dtype, reg):
    """The matrix-core path never takes the one-launch tail; since round 5 its steps (dense pass(es), gain_grad_kernel, finalize_kernel,
    the update) are replayed from a hipGraph as well -- 8 at a time, in calls of at least 256 steps ("auto") or always ("graph"; the streaming kernels
    of large problems replay only then).  300
    recorded steps with use_min: 37 replays + 4 single steps in "auto" and "graph", every kernel its own launch in "kernels": the
    same losses, parameters and snapshots bit for bit."""
    p, start = make_case(seed=33, with_sky=reg, nants=12, nfreqs=128)

    def fit(mode):
        s = make_solver(p, start, dtype, layout="shared", reg=reg, kernel_path="dense")
        assert s.timing_get()["kernel_path"] == "dense"
        s.set_launch_mode(mode)
        s.set_optimizer("Adam", learning_rate=5e-3)
        s.run(1, record=False)
        losses, stopped, nupd = s.run(300, record=True, use_min=True, tol=0.0)
        out = (losses, stopped, nupd, s.get_params(), s.get_params(which=1))
        s.close()
        return out

    ref = fit("kernels")
    assert len(ref[0]) == 300 and np.all(np.isfinite(ref[0])) and ref[0][-1] < ref[0][0]
    for mode in ("auto", "graph"):
        assert_same(ref, fit(mode))


def test_split_groups_and_redundant_group_across_launch_modes():
    """Groups cut into several work items (partial coefficient gradients summed in the tail) and a multi-baseline group."""
    p, start = make_case(seed=5, nants=7, nfreqs=300, redundant=True)
    ref = run_fit(p, start, np.float64, "kernels", nsteps=40, tol=0.0)
    for mode in MODES[1:]:
        assert_same(ref, run_fit(p, start, np.float64, mode, nsteps=40, tol=0.0))
    ref = run_fit(p, start, np.float32, "kernels", nsteps=40, tol=0.0, reg=False, freeze_model=True)
    for mode in MODES[1:]:
        assert_same(ref, run_fit(p, start, np.float32, mode, nsteps=40, tol=0.0, freeze_model=True))


def test_tolerance_stop_lands_on_the_same_step():
    p, start = make_case(seed=7, perturb=False)
    ref = run_fit(p, start, np.float64, "kernels", lr=5e-2, nsteps=400, tol=1e-6)
    assert ref[1] and 20 < len(ref[0]) < 400  # stopped inside a replay, not at its edge
    for mode in MODES[1:]:
        assert_same(ref, run_fit(p, start, np.float64, mode, lr=5e-2, nsteps=400, tol=1e-6))


def test_nonfinite_loss_stops_every_mode_at_the_same_update():
    """A float32 fit driven to overflow: CAL_ERR_NONFINITE after the same number of updates, parameters as they were."""
    p, start = make_case(seed=3)
    out = []
    for mode in MODES:
        s = make_solver(p, start, np.float32)
        s.set_launch_mode(mode)
        s.set_optimizer("Adam", learning_rate=3e37)
        with pytest.raises(_lib.CalamityHipError) as err:
            s.run(64, record=True, tol=0.0)
        assert err.value.code == _lib.CAL_ERR_NONFINITE
        out.append((str(err.value), s.get_params()))
        s.close()
    for msg, params in out[1:]:
        assert msg == out[0][0]
        for x, y in zip(params, out[0][1]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("optimizer", ["Adam", "Adamax", "Nadam"])  # (Nadam: its momentum schedule is rebuilt with the device's pow)
@pytest.mark.parametrize("mode", ["kernels", "graph"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_resume_from_parameters_and_moments(optimizer, mode, dtype):
    """Checkpoint / resume (no counterpart in the reference): 10 steps == 5 steps, get_params +
    get_moments, a NEW solver, set_params + set_moments, 5 more steps -- bitwise, for both update paths (the separate
    update kernel of large problems is covered in test_gpu_fullsize.py::test_hera350_resume_through_the_separate_update_kernel)."""
    p, start = make_case(seed=13, nants=8, nfreqs=64)

    def fresh():
        s = make_solver(p, start, dtype)
        s.set_launch_mode(mode)
        s.set_optimizer(optimizer, learning_rate=1e-2)
        return s

    s = fresh()
    l10, _, _ = s.run(10, record=True, tol=0.0)
    want = s.get_params()
    s.close()
    s = fresh()
    l5, _, _ = s.run(5, record=True, tol=0.0)
    params, moments = s.get_params(), s.get_moments()
    s.close()
    t = moments.pop("t")
    assert t == 5
    s = fresh()  # set_optimizer has zeroed moments and the iteration count
    s.set_params(*params)
    s.set_moments(**moments, t=t)
    l5b, _, _ = s.run(5, record=True, tol=0.0)
    got = s.get_params()
    s.close()
    np.testing.assert_array_equal(np.concatenate([l5, l5b]), l10)
    for x, y in zip(got, want):
        np.testing.assert_array_equal(x, y)


OPT_CASES = [
    ("SGD", dict(learning_rate=0.5)),
    ("SGD", dict(learning_rate=0.2, momentum=0.8)),
    ("SGD", dict(learning_rate=0.2, momentum=0.8, nesterov=True)),
    ("RMSprop", dict(learning_rate=1e-3)),
    ("RMSprop", dict(learning_rate=1e-3, rho=0.8, momentum=0.5)),
    ("Adagrad", dict(learning_rate=5e-2)),
    ("Adagrad", dict(learning_rate=5e-2, initial_accumulator_value=0.0)),
    ("Adadelta", dict(learning_rate=1.0, rho=0.9)),
    ("Nadam", dict(learning_rate=1e-2)),
    ("Ftrl", dict(learning_rate=0.5)),  # (Ftrl sets the parameter from its accumulators: a small rate leaves coefficients of 1e-5, all cancellation in float32)
    ("Ftrl", dict(learning_rate=5e-2, learning_rate_power=-0.3, l1_regularization_strength=1e-4, l2_regularization_strength=1e-3,
                  l2_shrinkage_regularization_strength=1e-3, beta=0.05, initial_accumulator_value=0.2)),
    ("LAMB", dict(learning_rate=1e-2)),
    ("LAMB", dict(learning_rate=1e-2, weight_decay_rate=1e-2, epsilon=1e-5)),
]


@pytest.mark.parametrize("optimizer,kw", OPT_CASES, ids=[f"{n}-{'-'.join(sorted(k))}" for n, k in OPT_CASES])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_other_keras_optimizers_follow_the_oracle(optimizer, kw, dtype):
    """The remaining OPTIMIZERS entries (calibration.py:17-27; Keras OptimizerV2 semantics) against the oracle's classes over the
    reference loop (one unrecorded update, then 25 recorded ones), and the same numbers from every launch form."""
    from calamity_amd import problem
    from oracle import ref_numpy as R
    from test_gpu_parity import TOL, oracle_inputs, relnorm

    p, start = make_case(seed=31, with_sky=True)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    ref = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, ch["data_r"], ch["data_i"], ch["wgts"], ch["fg_comps"],
                                      ch["corr_inds"], maxsteps=25, tol=0.0, optimizer=optimizer, sky_model_r=ch["sky_model_r"],
                                      sky_model_i=ch["sky_model_i"], model_regularization="sum", **kw)
    outs = []
    for mode in ("kernels", "graph"):
        s = make_solver(p, start, dtype, reg=True)
        s.set_launch_mode(mode)
        s.set_optimizer(optimizer, **kw)
        s.run(1, record=False)
        losses, _, nupd = s.run(25, record=True, tol=0.0)
        outs.append((losses, s.get_params()))
        s.close()
        assert nupd == 25
    tol = TOL[dtype]["traj"]
    losses, (g_r, g_i, c_r, c_i) = outs[0]
    np.testing.assert_allclose(losses, np.asarray(ref[4]["loss"], dtype=np.float64), rtol=tol)
    assert relnorm(g_r.astype(np.float64) + 1j * g_i, ref[0] + 1j * ref[1]) <= tol
    assert relnorm(c_r, problem.coeffs_from_chunks(p, ref[2])) <= tol and relnorm(c_i, problem.coeffs_from_chunks(p, ref[3])) <= tol
    np.testing.assert_array_equal(outs[1][0], losses)
    for x, y in zip(outs[1][1], outs[0][1]):
        np.testing.assert_array_equal(x, y)


def test_optimizer_arguments_are_checked_like_keras():
    p, start = make_case(seed=1)
    s = make_solver(p, start, np.float32)
    with pytest.raises(KeyError):
        s.set_optimizer("Lamb")
    with pytest.raises(ValueError):
        s.set_optimizer("Ftrl", learning_rate_power=0.5)  # Keras: learning_rate_power must be <= 0
    with pytest.raises(TypeError):
        s.set_optimizer("SGD", beta_1=0.9)  # not an argument of tf.keras.optimizers.SGD
    with pytest.raises(TypeError):
        s.set_optimizer("Adam", momentum=0.9)
    s.close()


def test_lamb_takes_one_trust_ratio_per_variable():
    """LAMB (tensorflow-addons, calibration.py:26) scales each VARIABLE's step by |var| / |update|: the reference's variables are
    g_r, g_i and one fg_r[chunk], fg_i[chunk] per chunk (:596-603).  A problem with two chunks (a three-baseline fitting group +
    single-baseline groups): six variables, against the oracle's LAMB over the reference loop; then three time slices of it in one
    solver (eighteen variables) against the slices fitted alone."""
    from calamity_amd import distributed, problem
    from calamity_amd.solver import HipFitSolver
    from oracle import ref_numpy as R
    from test_gpu_parity import oracle_inputs, relnorm

    p, start = make_case(seed=11, with_sky=True, redundant=True)
    ch, fg_r, fg_i = oracle_inputs(p, start)
    assert len(ch["fg_comps"]) == 2 and sorted(set(p.chunk_of_grp.tolist())) == [0, 1]
    kw = dict(learning_rate=2e-2, weight_decay_rate=1e-3)
    ref = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, ch["data_r"], ch["data_i"], ch["wgts"], ch["fg_comps"],
                                      ch["corr_inds"], maxsteps=20, tol=0.0, optimizer="LAMB", **kw)
    s = make_solver(p, start, np.float64)
    s.set_optimizer("LAMB", **kw)
    s.run(1, record=False)
    losses, _, _ = s.run(20, record=True, tol=0.0)
    g_r, g_i, c_r, c_i = s.get_params()
    s.close()
    np.testing.assert_allclose(losses, np.asarray(ref[4]["loss"], dtype=np.float64), rtol=1e-8)
    assert relnorm(g_r + 1j * g_i, ref[0] + 1j * ref[1]) <= 1e-8
    assert relnorm(c_r, problem.coeffs_from_chunks(p, ref[2])) <= 1e-8 and relnorm(c_i, problem.coeffs_from_chunks(p, ref[3])) <= 1e-8

    parts = []
    for t in range(3):
        q, st = make_case(seed=11 + t, redundant=True)
        problem.chunks_from_problem(q)  # (fills chunk_of_grp: the variables)
        parts.append((q, st))
    alone = []
    for q, st in parts:
        s = make_solver(q, st, np.float64)
        s.set_optimizer("LAMB", **kw)
        alone.append((s.run(12, record=True, tol=0.0)[0], s.get_params()))
        s.close()
    big, bstart = distributed.batch_time_slices(parts, per_slice=True)
    big.chunk_of_grp = np.concatenate([q.chunk_of_grp for q, _ in parts])
    s = HipFitSolver(dtype=np.float64)
    s.set_problem(big, layout="stream")
    s.set_params(bstart["g_r"], bstart["g_i"], bstart["c_r"], bstart["c_i"])
    s.set_optimizer("LAMB", **kw)
    res = s.run_slices(12, record=True, tol=0.0)
    bg_r = s.get_params()[0]
    s.close()
    for t, (q, _) in enumerate(parts):
        np.testing.assert_allclose(res[t][0], alone[t][0], rtol=1e-12)
        assert relnorm(bg_r[t * q.nants : (t + 1) * q.nants], alone[t][1][0]) <= 1e-12
