"""BASELINE.json config 3 on the HIP path: HERA-350 x 1024 channels x 8 time slices, batched into ONE solver by
``distributed.batch_time_slices`` (slice t keeps its own gains: antenna index + t * nants).  The reference fits the times
one after another (calibration.py:1167), so the batched fit must equal 8 independent fits: loss, every gradient and a short
Adam trajectory are compared per slice with the C restatement (oracle/ref_c.c, fp64).

SHARED layout at full size: all 8 x 61 075 baselines alias the 120 unique basis blocks, i.e. every block is read once for
all the right-hand sides of all times ("Multiple times") and the dense matrix-core kernel runs.
STREAM layout (every (slice, baseline) owns its tiles: 8 x 24.9 GB at full size) on a bounded sample of the baselines."""
import numpy as np
import pytest

from calamity_amd import distributed as D
from calamity_amd import synthetic

pytestmark = pytest.mark.gpu
NT = 8


def relnorm(a, b):
    a = np.asarray(a, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def make_slices(max_bls):
    cache, parts = {}, []
    for t in range(NT):
        p, _, s = synthetic.make_config("hera350", seed=2 + 100 * t, operator_cache=cache, max_bls=max_bls)
        for k in ("data_r", "data_i", "wgts"):  # the fit runs in fp32: halve the host memory of 8 full-size slices
            setattr(p, k, getattr(p, k).astype(np.float32))
        rng = np.random.default_rng(1000 + t)
        s["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
        s["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
        parts.append((p, s))
    return parts


@pytest.mark.parametrize("layout,max_bls", [("shared", None), ("stream", 6000)])
def test_eight_batched_time_slices_equal_independent_fits(layout, max_bls):
    from calamity_amd.solver import HipFitSolver
    from oracle.ref_c import CRef

    parts = make_slices(max_bls)
    prob, start = D.batch_time_slices(parts)
    na, nb, nc = parts[0][0].nants, parts[0][0].nbls, parts[0][0].ncoeffs
    assert prob.nants == NT * na and prob.nbls == NT * nb and len(prob.basis) == len(parts[0][0].basis)
    s = HipFitSolver(dtype=np.float32)
    s.set_problem(prob, layout=layout)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    loss, g_r, g_i, c_r, c_i = s.eval_grads()
    s.set_optimizer("Adam", learning_rate=1e-2)
    nsteps = 5
    losses, _, nupd = s.run(nsteps, record=True, tol=0.0)
    assert nupd == nsteps
    fg_r, fg_i, fc_r, fc_i = s.get_params()
    s.close()

    ref_loss, ref_losses = 0.0, np.zeros(nsteps)
    for t, (p, st) in enumerate(parts):
        c = CRef(p, np.float64, nthreads=16)
        ga, ca = slice(t * na, (t + 1) * na), slice(t * nc, (t + 1) * nc)
        l, og_r, og_i, oc_r, oc_i = c.loss_grads(st["g_r"], st["g_i"], st["c_r"], st["c_i"])
        ref_loss += l
        assert relnorm(g_r[ga], og_r) <= 1e-4 and relnorm(g_i[ga], og_i) <= 1e-4, (layout, t)
        assert relnorm(c_r[ca], oc_r) <= 1e-4 and relnorm(c_i[ca], oc_i) <= 1e-4, (layout, t)
        tg_r, tg_i, tc_r, tc_i, tl, _ = c.fit(st["g_r"], st["g_i"], st["c_r"], st["c_i"], nsteps, optimizer="Adam", learning_rate=1e-2)
        ref_losses += tl
        assert relnorm(fg_r[ga], tg_r) <= 1e-3 and relnorm(fg_i[ga], tg_i) <= 1e-3, (layout, t)
        assert relnorm(fc_r[ca], tc_r) <= 1e-3 and relnorm(fc_i[ca], tc_i) <= 1e-3, (layout, t)
        del c
    # the recorded loss of the batched fit is the sum of the slices' losses
    assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss)
    assert np.allclose(losses, ref_losses, rtol=1e-4)


def test_every_share_of_the_eight_gpu_job_against_the_c_oracle():
    """The shares `bench.py --gpus 8` really runs (bench.build_sharded_job("hera350", r, 8, 8): 8 time slices x rank r's 1/8 of
    the baselines, fitting groups dealt round-robin, slices sharing tiles -> fused_multi_mfma_kernel), every rank r = 0 .. 7, on
    ONE GPU against the C restatement: loss of every slice, all gradients, fp32 tolerances.  Also: the shares are balanced
    (baselines and basis vectors per rank within 1 %), and the same job with one loop state per slice
    (batch_time_slices(per_slice=True)) reports each slice's own loss."""
    import bench
    from calamity_amd.solver import HipFitSolver
    from oracle.ref_c import CRef

    nbl, nvec = [], []
    for r in range(8):
        prob, start, na = bench.build_sharded_job("hera350", r, 8, 8, per_slice=True)
        nbl.append(prob.nbls // 8)
        nvec.append(prob.ncoeffs // 8)
        assert prob.nslices == 8 and prob.bl_alias is not None
        rng = np.random.default_rng(50 + r)
        start["g_r"] = 1.0 + 0.05 * rng.standard_normal(start["g_r"].shape)
        start["g_i"] = 0.05 * rng.standard_normal(start["g_i"].shape)
        s = HipFitSolver(dtype=np.float32)
        s.set_problem(prob, layout="stream")
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        loss, g_r, g_i, c_r, c_i = s.eval_grads()
        each = s.slice_losses()
        s.close()
        # the oracle sees the same batched problem as ONE fit (its loss is the sum of the slices'); per-slice losses from the slices' rows
        single = D.FitProblem(**{k: getattr(prob, k) for k in ("nants", "nfreqs", "basis", "grp_basis", "grp_bl_start", "bl_ant0", "bl_ant1", "bl_rowblk",
                                                                "data_r", "data_i", "wgts")})
        c = CRef(single, np.float64, nthreads=16)
        l, og_r, og_i, oc_r, oc_i = c.loss_grads(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        assert abs(loss - l) <= 1e-5 * abs(l), (r, loss, l)
        assert abs(each.sum() - loss) <= 1e-12 * abs(loss)
        for a, b in ((g_r, og_r), (g_i, og_i), (c_r, oc_r), (c_i, oc_i)):
            assert relnorm(a, b) <= 1e-4, r
        del c
        # slice 3 alone, from its own rows
        nb, nc = prob.nbls // 8, prob.ncoeffs // 8
        rows, cs, ants = slice(3 * nb, 4 * nb), slice(3 * nc, 4 * nc), slice(3 * na, 4 * na)
        one = D.FitProblem(nants=na, nfreqs=prob.nfreqs, basis=prob.basis, grp_basis=prob.grp_basis[rows], grp_bl_start=np.arange(nb + 1, dtype=np.int32),
                           bl_ant0=prob.bl_ant0[rows] - 3 * na, bl_ant1=prob.bl_ant1[rows] - 3 * na, bl_rowblk=prob.bl_rowblk[rows],
                           data_r=prob.data_r[rows], data_i=prob.data_i[rows], wgts=prob.wgts[rows])
        c = CRef(one, np.float64, nthreads=16)
        l3 = c.loss_grads(start["g_r"][ants], start["g_i"][ants], start["c_r"][cs], start["c_i"][cs])[0]
        assert abs(each[3] - l3) <= 1e-5 * abs(l3), (r, each[3], l3)
        del c, prob, start, single, one
    assert max(nbl) - min(nbl) <= 0.01 * min(nbl) and max(nvec) - min(nvec) <= 0.01 * min(nvec), (nbl, nvec)


@pytest.mark.parametrize("world,rank", [(2, 1), (4, 2)])
def test_a_share_of_the_two_and_four_gpu_jobs_against_the_c_oracle(world, rank):
    """The driver's scaling run also starts `bench.py --gpus 2` and `--gpus 4`: N time slices x a dealt 1 / N of every slice's
    baselines per rank (2 and 4 right-hand-side sets per shared tile instead of 8).  One rank's share of each, on a bounded sample
    of the baselines, against the C restatement: loss, every slice's own loss, all gradients, a short Adam trajectory."""
    import bench
    from calamity_amd.solver import HipFitSolver
    from oracle.ref_c import CRef

    prob, start, na = bench.build_sharded_job("hera350", rank, world, world, per_slice=True, max_bls=8000)
    assert prob.nslices == world and prob.bl_alias is not None
    rng = np.random.default_rng(7 + world)
    start["g_r"] = 1.0 + 0.05 * rng.standard_normal(start["g_r"].shape)
    start["g_i"] = 0.05 * rng.standard_normal(start["g_i"].shape)
    s = HipFitSolver(dtype=np.float32)
    s.set_problem(prob, layout="stream")
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    loss, g_r, g_i, c_r, c_i = s.eval_grads()
    each = s.slice_losses()
    s.set_optimizer("Adam", learning_rate=1e-2)
    res = s.run_slices(5, record=True, tol=0.0)
    fg_r, fg_i, fc_r, fc_i = s.get_params()
    s.close()
    single = D.FitProblem(**{k: getattr(prob, k) for k in ("nants", "nfreqs", "basis", "grp_basis", "grp_bl_start", "bl_ant0", "bl_ant1", "bl_rowblk",
                                                            "data_r", "data_i", "wgts")})
    c = CRef(single, np.float64, nthreads=16)
    l, og_r, og_i, oc_r, oc_i = c.loss_grads(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    assert abs(loss - l) <= 1e-5 * abs(l) and abs(each.sum() - loss) <= 1e-12 * abs(loss)
    for a, b in ((g_r, og_r), (g_i, og_i), (c_r, oc_r), (c_i, oc_i)):
        assert relnorm(a, b) <= 1e-4
    tg_r, tg_i, tc_r, tc_i, tl, _ = c.fit(start["g_r"], start["g_i"], start["c_r"], start["c_i"], 5, optimizer="Adam", learning_rate=1e-2)
    assert np.allclose(np.sum([r[0] for r in res], axis=0), tl, rtol=1e-4)
    for a, b in ((fg_r, tg_r), (fg_i, tg_i), (fc_r, tc_r), (fc_i, tc_i)):
        assert relnorm(a, b) <= 1e-3


def test_one_share_of_the_eight_gpu_job_in_double_precision():
    """--precision 64 of the reference (calibration.py:1857) on the job a rank of `bench.py --gpus 8 --dtype f64` runs: the time
    slices that share tiles go through fused_multi_mfma_kernel<double> (v_mfma_f64_16x16x4_f64, 8 members per head; HERA-350's
    blocks of 18 .. 204 vectors are every class of that kernel up to 13 tiles, the last one with the single sample buffer).  Loss,
    every slice's loss, all gradients and a short Adam trajectory against the fp64 C restatement at fp64 tolerances (the regularised
    two-pass form of the kernel: tests/test_gpu_shapes.py)."""
    import bench
    from calamity_amd.solver import HipFitSolver
    from oracle.ref_c import CRef

    prob, start, na = bench.build_sharded_job("hera350", 5, 8, 8, per_slice=True)
    assert prob.nslices == 8 and prob.bl_alias is not None
    rng = np.random.default_rng(77)
    start["g_r"] = 1.0 + 0.05 * rng.standard_normal(start["g_r"].shape)
    start["g_i"] = 0.05 * rng.standard_normal(start["g_i"].shape)
    single = D.FitProblem(**{k: getattr(prob, k) for k in ("nants", "nfreqs", "basis", "grp_basis", "grp_bl_start", "bl_ant0", "bl_ant1", "bl_rowblk",
                                                            "data_r", "data_i", "wgts")})
    c = CRef(single, np.float64, nthreads=16)
    s = HipFitSolver(dtype=np.float64)
    s.set_problem(prob, layout="stream")
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    loss, g_r, g_i, c_r, c_i = s.eval_grads()
    each = s.slice_losses()
    l, og_r, og_i, oc_r, oc_i = c.loss_grads(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    assert abs(loss - l) <= 1e-12 * abs(l) and abs(each.sum() - loss) <= 1e-12 * abs(loss), (loss, l)
    assert abs(s.eval_loss() - l) <= 1e-12 * abs(l)
    for a, b in ((g_r, og_r), (g_i, og_i), (c_r, oc_r), (c_i, oc_i)):
        assert relnorm(a, b) <= 1e-11
    s.set_optimizer("Adam", learning_rate=1e-2)
    res = s.run_slices(4, record=True, tol=0.0)
    got = s.get_params()
    ref = c.fit(start["g_r"], start["g_i"], start["c_r"], start["c_i"], 4, optimizer="Adam", learning_rate=1e-2)
    assert np.allclose(np.sum([r[0] for r in res], axis=0), ref[4], rtol=1e-10)
    for a, b in zip(got, ref[:4]):
        assert relnorm(a, b) <= 1e-8
    s.close()
