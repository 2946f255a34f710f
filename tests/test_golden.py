"""Golden fixture tests/golden/fit_small.npz (restatement-derived, see tests/golden/make_golden.py): the oracle must
keep reproducing it (CPU), and the HIP path must match it through the C-ABI (GPU)."""
import os

import numpy as np
import pytest

from calamity_amd import problem
from oracle import ref_numpy as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fit_small.npz")
RUNS = {
    "adam_plain": dict(optimizer="Adam", model_regularization=None),
    "adamax_sum": dict(optimizer="Adamax", model_regularization="sum"),
    "adam_freeze": dict(optimizer="Adam", model_regularization=None, freeze_model=True),
    "adam_usemin": dict(optimizer="Adam", model_regularization=None, use_min=True, learning_rate=1e-1),
}


def load():
    g = dict(np.load(GOLD))
    basis, off = [], 0
    for shp in g["basis_shapes"]:
        n = int(shp[0] * shp[1])
        basis.append(g["basis_flat"][off : off + n].reshape(shp))
        off += n
    p = problem.FitProblem(
        nants=int(g["nants"]), nfreqs=int(g["nfreqs"]), basis=basis, grp_basis=g["grp_basis"], grp_bl_start=g["grp_bl_start"],
        bl_ant0=g["bl_ant0"], bl_ant1=g["bl_ant1"], bl_rowblk=g["bl_rowblk"], data_r=g["data_r"], data_i=g["data_i"], wgts=g["wgts"],
        sky_r=g["sky_r"], sky_i=g["sky_i"],
    )
    p.validate()
    return g, p


def test_oracle_reproduces_golden():
    g, p = load()
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, g["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, g["c_i"], np.float64)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    pri = R.prior_sums(ch["sky_model_r"], ch["sky_model_i"], ch["wgts"])
    assert np.isclose(pri[0], g["prior_r"], rtol=1e-13) and np.isclose(pri[1], g["prior_i"], rtol=1e-13)
    for tag, priors in (("plain", (None, None)), ("sum", pri)):
        loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(g["g_r"], g["g_i"], fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1, *priors)
        assert np.isclose(loss, g[f"{tag}_loss"], rtol=1e-12)
        np.testing.assert_allclose(gg_r, g[f"{tag}_gg_r"], rtol=1e-10, atol=1e-15)
        np.testing.assert_allclose(problem.coeffs_from_chunks(p, gf_i), g[f"{tag}_gc_i"], rtol=1e-10, atol=1e-15)
    kw = dict(data_r=ch["data_r"], data_i=ch["data_i"], wgts=ch["wgts"], fg_comps=ch["fg_comps"], corr_inds=ch["corr_inds"],
              maxsteps=10, learning_rate=1e-2, sky_model_r=ch["sky_model_r"], sky_model_i=ch["sky_model_i"])
    for tag, extra in RUNS.items():
        res = R.fit_gains_and_foregrounds(g["g_r"], g["g_i"], fg_r, fg_i, **dict(kw, **extra))
        np.testing.assert_allclose(res[4]["loss"], g[f"{tag}_loss_hist"], rtol=1e-11)
        np.testing.assert_allclose(res[0], g[f"{tag}_g_r"], rtol=1e-10)
        np.testing.assert_allclose(problem.coeffs_from_chunks(p, res[2]), g[f"{tag}_c_r"], rtol=1e-9, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 1e-4)])
@pytest.mark.parametrize("layout", ["stream", "shared"])
def test_hip_matches_golden(dtype, tol, layout):
    from calamity_amd.solver import HipFitSolver

    g, p = load()

    def rel(a, b):
        return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b)

    for tag, reg in (("plain", False), ("sum", True)):
        s = HipFitSolver(dtype=dtype)
        s.set_problem(p, layout=layout)
        s.set_params(g["g_r"], g["g_i"], g["c_r"], g["c_i"])
        if reg:
            s.set_regularization("sum", float(g["prior_r"]), float(g["prior_i"]))
        loss, gg_r, gg_i, gc_r, gc_i = s.eval_grads()
        assert abs(loss - g[f"{tag}_loss"]) <= max(tol, 1e-10) * abs(g[f"{tag}_loss"])
        assert rel(gg_r, g[f"{tag}_gg_r"]) <= tol and rel(gg_i, g[f"{tag}_gg_i"]) <= tol
        assert rel(gc_r, g[f"{tag}_gc_r"]) <= tol and rel(gc_i, g[f"{tag}_gc_i"]) <= tol
    for tag, extra in RUNS.items():
        extra = dict(extra)
        s = HipFitSolver(dtype=dtype)
        s.set_problem(p, layout=layout)
        s.set_params(g["g_r"], g["g_i"], g["c_r"], g["c_i"])
        if extra.get("model_regularization") == "sum":
            s.set_regularization("sum", float(g["prior_r"]), float(g["prior_i"]))
        s.set_optimizer(extra["optimizer"], learning_rate=extra.get("learning_rate", 1e-2))
        fz, um = extra.get("freeze_model", False), extra.get("use_min", False)
        s.run(1, record=False, freeze_model=fz)
        losses, _, _ = s.run(10, record=True, freeze_model=fz, use_min=um)
        np.testing.assert_allclose(losses, g[f"{tag}_loss_hist"], rtol=max(10 * tol, 1e-7))
        g_r, g_i, c_r, c_i = s.get_params(which=1 if um else 0)
        assert rel(g_r, g[f"{tag}_g_r"]) <= 10 * tol
        if fz:
            np.testing.assert_array_equal(c_r, g["c_r"].astype(dtype))
        else:
            assert rel(c_r, g[f"{tag}_c_r"]) <= 10 * tol


# ---- second fixture: the optimizers the first one does not cover + the graph functions (tests/golden/optimizers_small.npz) --------
GOLD2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optimizers_small.npz")


def opt_runs():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD2), "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.OPT_RUNS


def test_oracle_reproduces_the_optimizer_fixture():
    g, p = load()
    g2 = dict(np.load(GOLD2))
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, g["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, g["c_i"], np.float64)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    kw = dict(data_r=ch["data_r"], data_i=ch["data_i"], wgts=ch["wgts"], fg_comps=ch["fg_comps"], corr_inds=ch["corr_inds"], maxsteps=8, tol=0.0)
    for tag, (name, okw) in opt_runs().items():
        res = R.fit_gains_and_foregrounds(g["g_r"], g["g_i"], fg_r, fg_i, optimizer=name, **dict(kw, **okw))
        np.testing.assert_allclose(res[4]["loss"], g2[f"{tag}_loss_hist"], rtol=1e-11, err_msg=tag)
        np.testing.assert_allclose(res[0], g2[f"{tag}_g_r"], rtol=1e-10, err_msg=tag)
        np.testing.assert_allclose(problem.coeffs_from_chunks(p, res[3]), g2[f"{tag}_c_i"], rtol=1e-9, atol=1e-14, err_msg=tag)
    n = int(g2["nchunks"])
    assert n == len(ch["fg_comps"])
    args = (g["g_r"], g["g_i"], fg_r, fg_i, ch["fg_comps"], n, ch["data_r"], ch["data_i"], ch["wgts"], a0, a1)
    assert np.isclose(R.mse_chunked(*args), g2["mse_chunked"], rtol=1e-12)
    assert np.isclose(R.mse_chunked_sum_regularized(*args, float(g["prior_r"]), float(g["prior_i"])), g2["mse_chunked_sum_regularized"], rtol=1e-12)
    assert np.isclose(sum(float(g2[f"chunk{c}_mse"]) for c in range(n)), g2["mse_chunked"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 1e-4)])
def test_hip_matches_the_optimizer_fixture(dtype, tol):
    from calamity_amd import calibration as cal
    from calamity_amd.solver import HipFitSolver

    g, p = load()
    g2 = dict(np.load(GOLD2))
    ch = problem.chunks_from_problem(p)  # (also fills p.chunk_of_grp: LAMB's variables)

    def rel(a, b):
        return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b)

    for tag, (name, okw) in opt_runs().items():
        s = HipFitSolver(dtype=dtype)
        s.set_problem(p, layout="stream")
        s.set_params(g["g_r"], g["g_i"], g["c_r"], g["c_i"])
        s.set_optimizer(name, **okw)
        s.run(1, record=False)
        losses, _, _ = s.run(8, record=True, tol=0.0)
        np.testing.assert_allclose(losses, g2[f"{tag}_loss_hist"], rtol=max(10 * tol, 1e-7), err_msg=tag)
        g_r, g_i, c_r, c_i = s.get_params()
        s.close()
        # (Ftrl's coefficients pass through a cancellation: a looser bound on the float32 parameters, the losses above pin it)
        ptol = 10 * tol if not (name == "Ftrl" and dtype == np.float32) else 5e-2
        assert rel(g_r, g2[f"{tag}_g_r"]) <= ptol and rel(g_i, g2[f"{tag}_g_i"]) <= ptol, tag
        assert rel(c_r, g2[f"{tag}_c_r"]) <= ptol and rel(c_i, g2[f"{tag}_c_i"]) <= ptol, tag
    # the graph functions under the reference's names, on the chunk tensors
    cast = lambda xs: [np.asarray(x, dtype=dtype) for x in xs]  # noqa: E731
    fg_r, fg_i = cast(problem.coeffs_to_chunks(p, g["c_r"], np.float64)), cast(problem.coeffs_to_chunks(p, g["c_i"], np.float64))
    comps, d_r, d_i, w = cast(ch["fg_comps"]), cast(ch["data_r"]), cast(ch["data_i"]), cast(ch["wgts"])
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    gr, gi = np.asarray(g["g_r"], dtype=dtype), np.asarray(g["g_i"], dtype=dtype)
    n = int(g2["nchunks"])
    for c in range(n):
        vr, vi = cal.fg_model(fg_r[c], fg_i[c], comps[c])
        want = g2[f"chunk{c}_fg_model"]
        assert abs(vr.sum(dtype=np.float64) - want[0]) <= 10 * tol * want[2] and abs(vi.sum(dtype=np.float64) - want[1]) <= 10 * tol * want[3]
        mr, mi = cal.data_model(gr, gi, fg_r[c], fg_i[c], comps[c], a0[c], a1[c])
        assert rel(mr, g2[f"chunk{c}_data_model_r"]) <= tol and rel(mi, g2[f"chunk{c}_data_model_i"]) <= tol
        assert abs(cal.mse(mr, mi, d_r[c], d_i[c], w[c]) - g2[f"chunk{c}_mse"]) <= 20 * tol * g2[f"chunk{c}_mse"]
    args = (gr, gi, fg_r, fg_i, comps, n, d_r, d_i, w, a0, a1)
    assert abs(cal.mse_chunked(*args, dtype=dtype) - g2["mse_chunked"]) <= max(tol, 1e-10) * g2["mse_chunked"]
    assert abs(cal.mse_chunked_sum_regularized(*args, float(g["prior_r"]), float(g["prior_i"]), dtype=dtype) - g2["mse_chunked_sum_regularized"]) \
        <= max(tol, 1e-10) * g2["mse_chunked_sum_regularized"]
    assert abs(cal.mse_chunked_sum_regularized(*args, float(g["prior_r"]) + 0.25, float(g["prior_i"]) - 0.5, dtype=dtype)
               - g2["mse_chunked_sum_regularized_shifted"]) <= max(tol, 1e-10) * g2["mse_chunked_sum_regularized_shifted"]
