"""Several time slices in ONE solver with one loop state per slice (cal_problem_desc::nslices, cal_solver_run_slices).

The reference fits the (polarization, time) slices of a data set one after another (calibration.py:1160-1167, :1244-1269), each
with its own loss history, tolerance stop (:712-717) and use_min snapshot (:702-710).  A batched solver must reproduce exactly
that: every slice's recorded losses, stopping step, fitted gains / coefficients and minimum-loss snapshot equal those of the
slice fitted alone -- including slices that stop at different steps -- for every kernel family (general, dense fp32 / fp64,
group kernel), both layouts, with and without the "sum" regulariser (per-slice sums and priors), and for the launch forms.
Everything a slice computes runs in the same order as in its own solver, so fp64 results are compared at 1e-12."""
import numpy as np
import pytest

from calamity_amd import distributed, synthetic
from calamity_amd.problem import FitProblem

pytestmark = pytest.mark.gpu


def _slices(nt, nants=7, nfreqs=96, noise=(1e-4, 3e-2, 1e-3, 1e-2, 3e-3), with_sky=True):
    """nt independent (pol, time) slices of one array: same antennas and basis, different sky, gains, noise and flags."""
    cache = {}
    parts = []
    for t in range(nt):
        p, _, start = synthetic.make_problem(nants, nfreqs, f0=150e6, df=400e3, seed=100 + t, noise_frac=noise[t % len(noise)],
                                             with_sky=with_sky, operator_cache=cache)
        parts.append((p, start))
    # the same basis OBJECTS in every slice (the operator cache): batch_time_slices shares blocks on identity
    return parts


def _priors(p):
    return float(np.sum(p.sky_r * p.wgts)), float(np.sum(p.sky_i * p.wgts))


def _fit_alone(p, start, dtype, layout, kernel_path, reg, run_kw, optimizer="Adam", launch=None, **opt_kw):
    from calamity_amd.solver import HipFitSolver

    s = HipFitSolver(dtype=dtype)
    s.set_problem(p, layout=layout, kernel_path=kernel_path)
    if launch:
        s.set_launch_mode(launch)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    s.set_regularization("sum" if reg else None, *(_priors(p) if reg else (0.0, 0.0)))
    s.set_optimizer(optimizer, **opt_kw)
    s.run(1, record=False, freeze_model=run_kw.get("freeze_model", False))
    losses, stopped, nupd = s.run(**run_kw)
    cur = s.get_params(0)
    best = s.get_params(1) if run_kw.get("use_min") and len(losses) else None
    s.close()
    return losses, stopped, nupd, cur, best


def _fit_batched(parts, dtype, layout, kernel_path, reg, run_kw, optimizer="Adam", launch=None, **opt_kw):
    from calamity_amd.solver import HipFitSolver

    big, start = distributed.batch_time_slices(parts, per_slice=True)
    assert big.nslices == len(parts)
    s = HipFitSolver(dtype=dtype)
    s.set_problem(big, layout=layout, kernel_path=kernel_path)
    if launch:
        s.set_launch_mode(launch)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    if reg:
        pri = np.asarray([_priors(p) for p, _ in parts])
        s.set_regularization("sum", pri[:, 0], pri[:, 1])
    else:
        s.set_regularization(None)
    s.set_optimizer(optimizer, **opt_kw)
    s.run_slices(1, record=False, freeze_model=run_kw.get("freeze_model", False))
    res = s.run_slices(**run_kw)
    cur = s.get_params(0)
    best = s.get_params(1) if run_kw.get("use_min") else None
    s.close()
    return res, cur, best, big


def _split(big, parts, arrs):
    """Per-slice views of batched (g_r, g_i, c_r, c_i)."""
    na = parts[0][0].nants
    out, c0 = [], 0
    for t, (p, _) in enumerate(parts):
        nc = p.ncoeffs
        out.append((arrs[0][t * na : (t + 1) * na], arrs[1][t * na : (t + 1) * na], arrs[2][c0 : c0 + nc], arrs[3][c0 : c0 + nc]))
        c0 += nc
    return out


def _compare(parts, dtype, layout, kernel_path, reg, run_kw, rtol, **kw):
    res, cur, best, big = _fit_batched(parts, dtype, layout, kernel_path, reg, run_kw, **kw)
    cur_t = _split(big, parts, cur)
    best_t = _split(big, parts, best) if best is not None else None
    nsteps = []
    for t, (p, start) in enumerate(parts):
        losses, stopped, nupd, cur1, best1 = _fit_alone(p, start, dtype, layout, kernel_path, reg, run_kw, **kw)
        bl, bstopped, bnupd = res[t]
        assert len(bl) == len(losses), (t, len(bl), len(losses))
        assert bstopped == stopped and bnupd == nupd, (t, bstopped, stopped, bnupd, nupd)
        np.testing.assert_allclose(bl, losses, rtol=rtol, atol=0)
        for a, b in zip(cur_t[t], cur1):
            assert np.linalg.norm(a.astype(np.float64) - b) <= rtol * max(np.linalg.norm(b), 1e-30), t
        if best1 is not None:
            for a, b in zip(best_t[t], best1):
                assert np.linalg.norm(a.astype(np.float64) - b) <= rtol * max(np.linalg.norm(b), 1e-30), t
        nsteps.append(len(losses))
    return nsteps


@pytest.mark.parametrize("layout", ["shared", "stream"])
@pytest.mark.parametrize("reg", [False, True])
def test_slices_stop_on_their_own(layout, reg):
    """Five slices with different noise levels and a loose tolerance: they stop at different steps; every one equals its own fit."""
    parts = _slices(5)
    run_kw = dict(nsteps=400, record=True, tol=2e-7, use_min=True)
    n = _compare(parts, np.float64, layout, "general", reg, run_kw, 1e-12, learning_rate=1e-2)
    assert len(set(n)) > 1 and min(n) < 400, n  # the slices did stop at different steps


@pytest.mark.parametrize("launch", ["kernels", "one_tail", "graph"])
def test_slices_launch_forms(launch):
    parts = _slices(3)
    run_kw = dict(nsteps=70, record=True, tol=1e-30, use_min=True)
    _compare(parts, np.float64, "shared", "general", True, run_kw, 1e-12, launch=launch, optimizer="Adamax", learning_rate=1e-2)
    _compare(parts, np.float32, "stream", "general", False, run_kw, 2e-3, launch=launch, learning_rate=1e-2)


@pytest.mark.parametrize("dtype,path", [(np.float32, "dense"), (np.float32, "dense_split1"), (np.float32, "dense_f32"), (np.float64, "dense")])
@pytest.mark.parametrize("reg", [False, True])
def test_slices_dense_kernels(dtype, path, reg):
    """The matrix-core kernels (fp32: split-bf16 and the fp32 one it replaced; fp64): panels / super-panels never mix slices, the
    two-pass regulariser uses each slice's own alpha."""
    parts = _slices(3, nants=9, nfreqs=128)
    run_kw = dict(nsteps=40, record=True, tol=1e-30, use_min=False)
    _compare(parts, dtype, "shared", path, reg, run_kw, 1e-12 if dtype == np.float64 else 2e-3, learning_rate=1e-2)


def test_slices_frozen_model_and_other_optimizers():
    parts = _slices(3)
    _compare(parts, np.float64, "shared", "general", True, dict(nsteps=30, record=True, tol=1e-30, use_min=True, freeze_model=True), 1e-12,
             learning_rate=1e-2)
    for opt, kw in (("Nadam", dict(learning_rate=1e-2)), ("SGD", dict(learning_rate=1e-1, momentum=0.5)), ("RMSprop", dict(learning_rate=1e-3))):
        _compare(parts, np.float64, "stream", "general", False, dict(nsteps=25, record=True, tol=1e-30), 1e-12, optimizer=opt, **kw)


def test_slices_of_multi_baseline_groups():
    """Redundant sets as fitting groups (group kernel) in every slice."""
    from tests.test_gpu_shapes import random_problem

    parts = []
    for t in range(3):
        p, start = random_problem([5, 12, 30, 7], [3, 1, 20, 2], nants=9, nfreqs=200, seed=40 + t)
        parts.append((p, start))
    # the slices do not share basis objects here: every slice brings its own blocks
    run_kw = dict(nsteps=20, record=True, tol=1e-30, use_min=True)
    for layout in ("shared", "stream"):
        _compare(parts, np.float64, layout, "general", True, run_kw, 1e-12, learning_rate=1e-3)


def test_slice_losses_and_errors():
    from calamity_amd import _lib
    from calamity_amd.solver import HipFitSolver

    parts = _slices(3)
    big, start = distributed.batch_time_slices(parts, per_slice=True)
    s = HipFitSolver(dtype=np.float64)
    s.set_problem(big, layout="shared")
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    total = s.eval_loss()
    each = s.slice_losses()
    assert abs(each.sum() - total) <= 1e-14 * total
    for t, (p, st) in enumerate(parts):
        s1 = HipFitSolver(dtype=np.float64)
        s1.set_problem(p, layout="shared")
        s1.set_params(st["g_r"], st["g_i"], st["c_r"], st["c_i"])
        assert abs(s1.eval_loss() - each[t]) <= 1e-13 * each[t]
        s1.close()
    s.set_optimizer("Adam")
    with pytest.raises(_lib.CalamityHipError):  # one loss history per slice: the single-fit entry point refuses
        s.run(3)
    # a slice whose loss is not finite stops alone; the others finish; the call reports it
    g_r = start["g_r"].copy()
    g_r[parts[0][0].nants] = np.nan  # first antenna of slice 1
    s.set_params(g_r, start["g_i"], start["c_r"], start["c_i"])
    s.set_optimizer("Adam", learning_rate=1e-2)
    with pytest.raises(_lib.CalamityHipError) as err:
        s.run_slices(5)
    assert err.value.code == _lib.CAL_ERR_NONFINITE and "slice 1" in str(err.value)
    g = s.get_params(0)
    assert np.all(np.isfinite(g[0][: parts[0][0].nants])) and np.all(np.isfinite(g[0][2 * parts[0][0].nants :]))
    s.close()
    # slices must be listed slice by slice over disjoint antenna ranges
    bad = FitProblem(**{k: getattr(big, k) for k in ("nants", "nfreqs", "basis", "grp_basis", "grp_bl_start", "bl_ant0", "bl_ant1", "bl_rowblk",
                                                     "data_r", "data_i", "wgts")}, nslices=3)
    bad.bl_ant1 = bad.bl_ant1.copy()
    bad.bl_ant1[0] += parts[0][0].nants
    s = HipFitSolver(dtype=np.float64)
    with pytest.raises(_lib.CalamityHipError):
        s.set_problem(bad)
    s.close()


@pytest.mark.parametrize("layout", ["shared", "stream"])
def test_as_many_slices_as_the_library_takes(layout):
    """CAL_MAX_SLICES (256) time slices in ONE solver -- the batches the drop-in makes of tutorial-scale data -- each with its own
    noise level and tolerance stop: a sample of the slices against the same slices fitted alone; one slice more is refused."""
    from calamity_amd import _lib
    from calamity_amd.batched import replicate_slices
    from calamity_amd.solver import HipFitSolver

    nt = _lib.CAL_MAX_SLICES
    p, _, start = synthetic.make_problem(6, 32, f0=150e6, df=400e3, seed=9, with_sky=False)
    rng = np.random.default_rng(9)
    scale = 1.0 + 0.5 * rng.random(nt)                      # every slice its own sky level ...
    noise = 10.0 ** rng.uniform(-4, -1.5, nt)               # ... and noise: they stop at different steps
    d_r = np.concatenate([p.data_r * scale[t] + noise[t] * rng.standard_normal(p.data_r.shape) for t in range(nt)])
    d_i = np.concatenate([p.data_i * scale[t] + noise[t] * rng.standard_normal(p.data_i.shape) for t in range(nt)])
    w = np.concatenate([p.wgts] * nt)
    big, _, _ = replicate_slices(p, nt)
    run_kw = dict(record=True, tol=1e-9, use_min=False)
    s = HipFitSolver(dtype=np.float64)
    s.set_problem(big, layout=layout)
    s.set_data(d_r, d_i, w)
    s.set_params(np.concatenate([start["g_r"]] * nt), np.concatenate([start["g_i"]] * nt),
                 np.concatenate([start["c_r"] * scale[t] for t in range(nt)]), np.concatenate([start["c_i"] * scale[t] for t in range(nt)]))
    s.set_optimizer("Adam", learning_rate=2e-2)
    s.run_slices(1, record=False)
    res = s.run_slices(400, **run_kw)
    g_r = s.get_params()[0]
    s.close()
    nstop = sorted({len(r[0]) for r in res})
    assert len(nstop) > 8, nstop  # (the slices really stop on their own)
    one = HipFitSolver(dtype=np.float64)
    one.set_problem(FitProblem(**{k: getattr(p, k) for k in ("nants", "nfreqs", "basis", "grp_basis", "grp_bl_start", "bl_ant0", "bl_ant1", "bl_rowblk")},
                               data_r=None, data_i=None, wgts=None), layout=layout)
    nb = p.nbls
    for t in (0, 1, 77, 128, nt - 1):
        one.set_data(d_r[t * nb : (t + 1) * nb], d_i[t * nb : (t + 1) * nb], p.wgts)
        one.set_params(start["g_r"], start["g_i"], start["c_r"] * scale[t], start["c_i"] * scale[t])
        one.set_optimizer("Adam", learning_rate=2e-2)
        one.run(1, record=False)
        losses, stopped, _ = one.run(400, **run_kw)
        assert len(losses) == len(res[t][0]) and bool(stopped) == bool(res[t][1]), t
        np.testing.assert_allclose(res[t][0], losses, rtol=1e-11)
        assert np.linalg.norm(g_r[t * p.nants : (t + 1) * p.nants] - one.get_params()[0]) <= 1e-11 * np.linalg.norm(one.get_params()[0])
    one.close()
    too_many, _, _ = replicate_slices(p, nt + 1)
    s = HipFitSolver(dtype=np.float64)
    with pytest.raises(_lib.CalamityHipError):
        s.set_problem(too_many, layout=layout)
    s.close()
