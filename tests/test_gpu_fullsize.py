"""Full-size GPU checks at the BASELINE.json configurations.

The padded NumPy oracle is only affordable up to HERA-37 x 1024 channels; at HERA-350 the checks are size-independent
properties: the two basis layouts and the two kernel families (general VALU kernel / MFMA kernel) must agree, the
analytic gradient must match a central finite difference of the loss along a random direction, the foreground model
is linear in the coefficients, and coefficients -> model -> least-squares initialisation is a round trip on an
orthonormal basis (calibration.py:828-913 with no flags).
"""
import numpy as np
import pytest

from calamity_amd import problem, synthetic
from oracle import ref_numpy as R

pytestmark = pytest.mark.gpu


def relnorm(a, b):
    a = np.asarray(a, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def solver_for(p, start, dtype, layout, kernel_path="auto"):
    from calamity_amd.solver import HipFitSolver

    s = HipFitSolver(dtype=dtype)
    s.set_problem(p, layout=layout, kernel_path=kernel_path)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    return s


def perturbed(p, start, seed):
    rng = np.random.default_rng(seed)
    out = dict(start)
    out["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    out["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    return out


def test_hera37_full_size_against_oracle():
    """BASELINE config 1 (HERA-37 hex, 1024 channels, fp64): loss and every gradient against the padded oracle."""
    p, truth, start = synthetic.make_config("hera37")
    start = perturbed(p, start, 1)
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, start["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, start["c_i"], np.float64)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(start["g_r"], start["g_i"], fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1)
    for layout in ("stream", "shared"):
        s = solver_for(p, start, np.float64, layout)
        l2, hg_r, hg_i, hc_r, hc_i = s.eval_grads()
        assert abs(l2 - loss) <= 1e-10 * abs(loss)
        assert relnorm(hg_r, gg_r) <= 1e-10 and relnorm(hg_i, gg_i) <= 1e-10
        assert relnorm(hc_r, problem.coeffs_from_chunks(p, gf_r)) <= 1e-10
        assert relnorm(hc_i, problem.coeffs_from_chunks(p, gf_i)) <= 1e-10
        s.close()
    # fp32, both kernel families, against the same oracle numbers
    for layout in ("stream", "shared"):
        s = solver_for(p, start, np.float32, layout)
        l2, hg_r, hg_i, hc_r, hc_i = s.eval_grads()
        assert abs(l2 - loss) <= 1e-5 * abs(loss)
        assert relnorm(hg_r, gg_r) <= 1e-4 and relnorm(hc_r, problem.coeffs_from_chunks(p, gf_r)) <= 1e-4
        s.close()


@pytest.fixture(scope="module")
def hera350():
    p, truth, start = synthetic.make_config("hera350", with_sky=True)
    return p, perturbed(p, start, 2)


def test_hera350_kernel_families_and_layouts_agree(hera350):
    """BASELINE config 2 at full size (61 075 baselines x 1024 channels): the fp64 general kernel is the reference; the fp64
    dense kernel (shared layout, v_mfma_f64_16x16x4) must agree with it to 1e-10, the fp32 dense kernel (v_mfma_f32_32x32x2)
    and the fp32 streaming kernel (per-baseline tiles) to fp32 accuracy."""
    p, start = hera350
    ref = solver_for(p, start, np.float64, "shared", kernel_path="general")
    l64, g64_r, g64_i, c64_r, c64_i = ref.eval_grads()
    ref.close()
    s = solver_for(p, start, np.float64, "shared", kernel_path="dense")
    assert abs(s.eval_loss() - l64) <= 1e-10 * abs(l64)
    ld, g_r, g_i, c_r, c_i = s.eval_grads()
    assert abs(ld - l64) <= 1e-10 * abs(l64)
    assert relnorm(g_r, g64_r) <= 1e-10 and relnorm(g_i, g64_i) <= 1e-10 and relnorm(c_r, c64_r) <= 1e-10 and relnorm(c_i, c64_i) <= 1e-10
    s.close()
    for layout in ("shared", "stream"):
        s = solver_for(p, start, np.float32, layout)
        assert abs(s.eval_loss() - l64) <= 1e-5 * abs(l64)
        l32, g_r, g_i, c_r, c_i = s.eval_grads()
        assert abs(l32 - l64) <= 1e-5 * abs(l64)
        assert relnorm(g_r, g64_r) <= 1e-4 and relnorm(g_i, g64_i) <= 1e-4
        assert relnorm(c_r, c64_r) <= 1e-4 and relnorm(c_i, c64_i) <= 1e-4
        s.close()


def test_hera350_gradient_is_the_derivative_of_the_loss(hera350):
    """Central finite difference of the fp64 loss along a random direction equals <grad, direction>, with and without the
    "sum" regulariser."""
    p, start = hera350
    rng = np.random.default_rng(5)
    d = dict(g_r=rng.standard_normal(start["g_r"].shape), g_i=rng.standard_normal(start["g_i"].shape),
             c_r=rng.standard_normal(start["c_r"].shape), c_i=rng.standard_normal(start["c_i"].shape))
    s = solver_for(p, start, np.float64, "shared")
    for reg in (False, True):
        if reg:
            s.set_regularization("sum", float(np.sum(p.sky_r * p.wgts)) * 0.9, float(np.sum(p.sky_i * p.wgts)) * 1.1)
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        _, g_r, g_i, c_r, c_i = s.eval_grads()
        slope = np.sum(g_r * d["g_r"]) + np.sum(g_i * d["g_i"]) + np.sum(c_r * d["c_r"]) + np.sum(c_i * d["c_i"])
        h = 1e-5
        s.set_params(*[start[k] + h * d[k] for k in ("g_r", "g_i", "c_r", "c_i")])
        lp = s.eval_loss()
        s.set_params(*[start[k] - h * d[k] for k in ("g_r", "g_i", "c_r", "c_i")])
        lm = s.eval_loss()
        assert np.isclose((lp - lm) / (2 * h), slope, rtol=1e-6)
    s.close()


def test_hera350_model_linearity_and_init_round_trip(hera350):
    p, start = hera350
    rng = np.random.default_rng(7)
    s = solver_for(p, start, np.float64, "shared")
    c1 = (rng.standard_normal(p.ncoeffs), rng.standard_normal(p.ncoeffs))
    c2 = (rng.standard_normal(p.ncoeffs), rng.standard_normal(p.ncoeffs))
    s.set_params(c_r=c1[0], c_i=c1[1])
    m1 = s.model()
    s.set_params(c_r=c2[0], c_i=c2[1])
    m2 = s.model()
    s.set_params(c_r=2.0 * c1[0] - 3.0 * c2[0], c_i=2.0 * c1[1] - 3.0 * c2[1])
    m3 = s.model()
    assert relnorm(m3[0], 2.0 * m1[0] - 3.0 * m2[0]) <= 1e-12 and relnorm(m3[1], 2.0 * m1[1] - 3.0 * m2[1]) <= 1e-12
    # coefficients -> A c -> A^T (A c) with unit weights returns the coefficients (orthonormal DPSS columns)
    s.set_data(m1[0], m1[1], np.ones((p.nbls, p.nfreqs)))
    s.init_coeffs(m1[0], m1[1])
    _, _, c_r, c_i = s.get_params()
    assert relnorm(c_r, c1[0]) <= 1e-9 and relnorm(c_i, c1[1]) <= 1e-9
    s.close()


def test_hera350_against_the_c_oracle(hera350):
    """BASELINE config 2 at full size against the C / OpenMP restatement (oracle/ref_c.c), which is independent of the HIP
    code and of the NumPy restatement it was checked against (tests/test_oracle_c.py): loss and every gradient in fp64,
    with and without the "sum" regulariser, then a short fp32 Adam trajectory on the streaming layout."""
    from oracle.ref_c import CRef

    p, start = hera350
    c = CRef(p, np.float64, nthreads=16)
    # every fp64 kernel family DIRECTLY against the C oracle at 1e-10 (the dense kernel used to be tied to it
    # only through the general kernel of the other layout): streaming general, shared general, shared dense (v_mfma_f64; with the
    # regulariser its two-pass form)
    refs = {}
    for layout, path in (("stream", "general"), ("shared", "general"), ("shared", "dense")):
        s = solver_for(p, start, np.float64, layout, kernel_path=path)
        for reg in (False, True):
            pr, pi = (float(np.sum(p.sky_r * p.wgts)) * 0.9, float(np.sum(p.sky_i * p.wgts)) * 1.1) if reg else (0.0, 0.0)
            s.set_regularization("sum" if reg else None, pr, pi)
            if reg not in refs:
                c.set_regularization("sum" if reg else None, pr, pi)
                refs[reg] = c.loss_grads(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
            loss, og_r, og_i, oc_r, oc_i = refs[reg]
            l2, hg_r, hg_i, hc_r, hc_i = s.eval_grads()
            assert abs(l2 - loss) <= 1e-10 * abs(loss), (layout, path, reg)
            assert abs(s.eval_loss() - loss) <= 1e-10 * abs(loss), (layout, path, reg)
            assert relnorm(hg_r, og_r) <= 1e-10 and relnorm(hg_i, og_i) <= 1e-10, (layout, path, reg)
            assert relnorm(hc_r, oc_r) <= 1e-10 and relnorm(hc_i, oc_i) <= 1e-10, (layout, path, reg)
        s.close()
    # fp32 streaming kernel, 5 Adam updates, against the fp64 C trajectory
    c.set_regularization(None)
    og_r, og_i, oc_r, oc_i, olosses, _ = c.fit(start["g_r"], start["g_i"], start["c_r"], start["c_i"], 5, optimizer="Adam", learning_rate=1e-2)
    s = solver_for(p, start, np.float32, "stream")
    s.set_optimizer("Adam", learning_rate=1e-2)
    losses, stopped, nupd = s.run(5, record=True, tol=0.0)
    g_r, g_i, c_r, c_i = s.get_params()
    assert np.allclose(losses, olosses, rtol=1e-4)
    assert relnorm(g_r, og_r) <= 1e-3 and relnorm(c_r, oc_r) <= 1e-3 and relnorm(c_i, oc_i) <= 1e-3
    s.close()


@pytest.mark.parametrize("optimizer", ["Adam", "Adamax"])
def test_hera350_resume_through_the_separate_update_kernel(hera350, optimizer):
    """Checkpoint / resume at a size whose update runs as finalize_kernel + adam2_kernel (more than 2^20 parameters; the
    one-launch tails of small problems are covered in test_gpu_launch_modes.py): 4 steps == 2 steps, get_params + get_moments,
    a new solver, set_params + set_moments, 2 more steps -- bitwise."""
    p, start = hera350

    def fresh():
        s = solver_for(p, start, np.float32, "shared")
        s.set_optimizer(optimizer, learning_rate=1e-2)
        return s

    s = fresh()
    l4, _, _ = s.run(4, record=True, tol=0.0)
    want = s.get_params()
    s.close()
    s = fresh()
    l2, _, _ = s.run(2, record=True, tol=0.0)
    params, moments = s.get_params(), s.get_moments()
    s.close()
    t = moments.pop("t")
    assert t == 2
    s = fresh()
    s.set_params(*params)
    s.set_moments(**moments, t=t)
    l2b, _, _ = s.run(2, record=True, tol=0.0)
    got = s.get_params()
    s.close()
    np.testing.assert_array_equal(np.concatenate([l2, l2b]), l4)
    for x, y in zip(got, want):
        np.testing.assert_array_equal(x, y)


def test_hera350_redundant_groups_against_the_c_oracle():
    """BASELINE config 5 (shared multi-baseline fitting groups): every redundant set of the hex array is one fitting group
    (up to 330 baselines share a coefficient vector and one basis row block).  Loss and gradients, fp64 and fp32, both
    layouts, with and without the regulariser, against the C restatement."""
    from oracle.ref_c import CRef

    p0, truth, start0 = synthetic.make_config("hera350", with_sky=True)
    p, start = synthetic.merge_redundant_groups(p0, truth, start0)
    assert p.ngrps < p0.ngrps / 10 and np.diff(p.grp_bl_start).max() > 100
    start = perturbed(p, start, 4)
    c = CRef(p, np.float64, nthreads=16)
    for reg in (False, True):
        pr, pi = (float(np.sum(p.sky_r * p.wgts)) * 0.9, float(np.sum(p.sky_i * p.wgts)) * 1.1) if reg else (0.0, 0.0)
        c.set_regularization("sum" if reg else None, pr, pi)
        ref = c.loss_grads(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        for dtype, tl, tg in ((np.float64, 1e-10, 1e-10), (np.float32, 1e-5, 1e-4)):
            for layout in ("stream", "shared"):
                s = solver_for(p, start, dtype, layout)
                if reg:
                    s.set_regularization("sum", pr, pi)
                out = s.eval_grads()
                assert abs(out[0] - ref[0]) <= tl * abs(ref[0]), (dtype, layout, reg)
                for a, b in zip(out[1:], ref[1:]):
                    assert relnorm(a, b) <= tg, (dtype, layout, reg)
                s.close()


def test_hera350_results_are_bitwise_reproducible(hera350):
    """No float atomics anywhere and fixed summation orders: repeated evaluations on the same solver must agree bit for
    bit, for every kernel family (general kernel on both layouts and precisions, dense kernel, group kernel).  Guards
    against synchronisation slips, which show up as run-to-run differences at this size first."""
    p, start = hera350
    truth = {"antpos": synthetic.hex_positions(p.nants)}
    cases = [(p, start, np.float32, "stream"), (p, start, np.float32, "shared"), (p, start, np.float64, "stream"), (p, start, np.float64, "shared")]
    pr, sr = synthetic.merge_redundant_groups(p, truth, start)
    cases.append((pr, sr, np.float32, "shared"))
    for prob, st, dtype, layout in cases:
        s = solver_for(prob, st, dtype, layout)
        s.set_regularization("sum", float(np.sum(prob.sky_r * prob.wgts)), float(np.sum(prob.sky_i * prob.wgts)))
        first = s.eval_grads()
        for _ in range(3):
            again = s.eval_grads()
            assert again[0] == first[0]
            for a, b in zip(again[1:], first[1:]):
                assert np.array_equal(a, b), (dtype, layout)
        s.set_regularization(None)
        first = s.eval_grads()
        for _ in range(3):
            again = s.eval_grads()
            assert again[0] == first[0] and all(np.array_equal(a, b) for a, b in zip(again[1:], first[1:])), (dtype, layout)
        s.close()
