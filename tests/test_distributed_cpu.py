"""N > 1 path on CPU (gloo, world_size 2): the baseline partition, the exchange payload -- exactly what the library sends:
one exchange per step of the gain-gradient parts (one, or three with the "sum" regulariser) and four double scalars -- and the
replicated gain update, with the oracle as the per-rank compute engine (test infrastructure only).  The same algebra on the
real kernels, two ranks on one GPU: tests/test_gpu_exchange_hook.py."""
import os
import socket

import numpy as np
import pytest

from calamity_amd import distributed as D
from calamity_amd import problem, synthetic
from oracle import ref_numpy as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_args(p, start):
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, start["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, start["c_i"], np.float64)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    return ch, fg_r, fg_i, a0, a1


def _rank_main(rank, world, port, nsteps, reg, out_q):
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    p, truth, start = synthetic.make_problem(9, 32, f0=150e6, df=400e3, seed=4, with_sky=reg)
    priors = (float(np.sum(p.sky_r * p.wgts)), float(np.sum(p.sky_i * p.wgts))) if reg else (None, None)
    sp, ss = D.shard_problem(p, start, rank, world)
    ch, fg_r, fg_i, a0, a1 = _oracle_args(sp, ss)
    g_r, g_i = ss["g_r"].copy(), ss["g_i"].copy()
    opt_g, opt_c = R.Adam(learning_rate=1e-2), R.Adam(learning_rate=1e-2)
    spec = D.exchange_spec(p.nants, p.nfreqs, reg_sum=reg)
    losses = []
    for _ in range(nsteps):
        if not reg:
            loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(g_r, g_i, fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1)
            buf = torch.from_numpy(np.concatenate([np.stack([gg_r, gg_i], axis=-1).ravel(), [loss, 0.0, 0.0, 0.0]]))
            assert buf.numel() == spec["gain_grad_reals"] + spec["scalars_f64"]
            dist.all_reduce(buf)
            buf = buf.numpy()
            gg = buf[:-4].reshape(p.nants, p.nfreqs, 2)
            gg_r, gg_i, loss = gg[..., 0].copy(), gg[..., 1].copy(), buf[-4]
        else:
            # The regulariser couples the shards through S = sum w m, but every adjoint is LINEAR in e = e0 + alpha w with
            # alpha = 2 (S - P) known only after the reduction.  The library therefore sends, in ONE exchange, three gain-gradient
            # parts r0 | r1 | r2 (gradient = r0 + alpha r1 + conj(alpha) r2) and the scalars (chi^2, S_r, S_i, spare), and folds
            # them afterwards (calamity_hip.hip: enqueue_pass, combine_gain_kernel / combine_coeff_kernel).  Modelled here with
            # the oracle: its regularised gradient at three local priors gives the parts.
            m = [R.data_model(g_r, g_i, fg_r[c], fg_i[c], ch["fg_comps"][c], a0[c], a1[c]) for c in range(len(fg_r))]
            s_r_loc = sum(np.sum(mr * w) for (mr, _), w in zip(m, ch["wgts"]))
            s_i_loc = sum(np.sum(mi * w) for (_, mi), w in zip(m, ch["wgts"]))

            def grads_at(alpha_r, alpha_i):  # local gradient with alpha forced: prior = S_local - alpha / 2
                out = R.loss_and_grads(g_r, g_i, fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1,
                                       s_r_loc - alpha_r / 2.0, s_i_loc - alpha_i / 2.0)
                gg = out[1] + 1j * out[2]
                gf = [a + 1j * b for a, b in zip(out[3], out[4])]
                return out[0] - (alpha_r / 2.0) ** 2 - (alpha_i / 2.0) ** 2, gg, gf

            chi2, gg0, gf0 = grads_at(0.0, 0.0)
            _, gga, gfa = grads_at(1.0, 0.0)
            _, ggb, gfb = grads_at(0.0, 1.0)
            ga, gb = gga - gg0, ggb - gg0                      # d grad / d alpha_r, d grad / d alpha_i
            r1, r2 = (ga - 1j * gb) / 2.0, (ga + 1j * gb) / 2.0  # alpha r1 + conj(alpha) r2 = alpha_r ga + alpha_i gb
            gf1 = [a - z for a, z in zip(gfa, gf0)]
            for one, other, z in zip(gf1, gfb, gf0):             # coefficients need ONE complex part: d/d alpha_i = i d/d alpha_r
                np.testing.assert_allclose(other - z, 1j * one, rtol=1e-9, atol=1e-14)
            parts = np.concatenate([np.stack([x.real, x.imag], axis=-1).ravel() for x in (gg0, r1, r2)])
            assert parts.size == spec["gain_grad_reals"]
            buf = torch.from_numpy(np.concatenate([parts, [chi2, s_r_loc, s_i_loc, 0.0]]))
            assert buf.numel() == spec["gain_grad_reals"] + spec["scalars_f64"]
            dist.all_reduce(buf)  # the one exchange of the step
            buf = buf.numpy()
            q = buf[:-4].reshape(3, p.nants, p.nfreqs, 2)
            q = q[..., 0] + 1j * q[..., 1]
            alpha = 2.0 * ((buf[-3] - priors[0]) + 1j * (buf[-2] - priors[1]))
            gg = q[0] + alpha * q[1] + np.conj(alpha) * q[2]
            gg_r, gg_i = gg.real.copy(), gg.imag.copy()
            gf = [z + alpha * one for z, one in zip(gf0, gf1)]   # local: a group's coefficients live on one rank
            gf_r, gf_i = [x.real.copy() for x in gf], [x.imag.copy() for x in gf]
            loss = buf[-4] + (buf[-3] - priors[0]) ** 2 + (buf[-2] - priors[1]) ** 2
        losses.append(loss)
        opt_g.apply_gradients([(gg_r, g_r), (gg_i, g_i)])
        opt_c.apply_gradients(list(zip(gf_r, fg_r)) + list(zip(gf_i, fg_i)))
    nbl_g = np.diff(p.grp_bl_start)
    mine = D.partition_groups(p.grp_nvec, p.grp_basis, nbl_g, world)[rank]
    out_q.put((rank, losses, g_r, mine, problem.coeffs_from_chunks(sp, fg_r)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("reg", [False, True])
def test_two_rank_gloo_matches_single_process(reg):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp

    world, nsteps = 2, 6
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, nsteps, reg, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # single-process oracle
    p, truth, start = synthetic.make_problem(9, 32, f0=150e6, df=400e3, seed=4, with_sky=reg)
    ch, fg_r, fg_i, a0, a1 = _oracle_args(p, start)
    ref = R.fit_gains_and_foregrounds(
        start["g_r"], start["g_i"], fg_r, fg_i, ch["data_r"], ch["data_i"], ch["wgts"], ch["fg_comps"], ch["corr_inds"],
        maxsteps=nsteps - 1, optimizer="Adam", learning_rate=1e-2, sky_model_r=ch["sky_model_r"], sky_model_i=ch["sky_model_i"],
        model_regularization="sum" if reg else None,
    )
    # the reference loop has one unrecorded update first: its recorded losses are steps 1.. of the plain sequence
    np.testing.assert_allclose(results[0][1][1:], ref[4]["loss"], rtol=1e-10)
    np.testing.assert_allclose(results[0][1], results[1][1], rtol=1e-13)
    for rank, losses, g_r, mine, c_r in results:
        np.testing.assert_allclose(g_r, ref[0], rtol=1e-9, atol=1e-12)  # identical replicas
        coff = p.grp_coff
        expect = np.concatenate([problem.coeffs_from_chunks(p, ref[2])[coff[g] : coff[g + 1]] for g in mine])
        np.testing.assert_allclose(c_r, expect, rtol=1e-8, atol=1e-12)


def test_partition_balances_bytes_and_covers_all_groups():
    p, truth, start = synthetic.make_config("tutorial")
    nbl_g = np.diff(p.grp_bl_start)
    for world in (1, 2, 3, 8):
        parts = D.partition_groups(p.grp_nvec, p.grp_basis, nbl_g, world)
        allg = np.concatenate(parts)
        assert sorted(allg.tolist()) == list(range(p.ngrps))
        work = np.asarray([p.grp_nvec[x].sum() for x in parts], dtype=np.float64)
        assert work.max() <= 1.25 * work.mean() + p.grp_nvec.max()
        wsum = sum(D.select_groups(p, start, x)[0].wgts.sum() for x in parts)
        assert np.isclose(wsum, 1.0)  # weights keep their global normalisation


def test_batched_time_slices_equal_independent_fits():
    """Slices batched into one solver (antenna indices offset per slice) take exactly the updates of separate fits."""
    cache = {}
    parts = [synthetic.make_problem(6, 24, f0=150e6, df=400e3, seed=10 + t, operator_cache=cache)[::2] for t in range(3)]
    bp, bs = D.batch_time_slices(parts)
    assert bp.nants == 18 and len(bp.basis) == len(parts[0][0].basis)
    ch, fg_r, fg_i, a0, a1 = _oracle_args(bp, bs)
    kw = dict(maxsteps=4, optimizer="Adam", learning_rate=1e-2)
    joint = R.fit_gains_and_foregrounds(bs["g_r"], bs["g_i"], fg_r, fg_i, ch["data_r"], ch["data_i"], ch["wgts"], ch["fg_comps"], ch["corr_inds"], **kw)
    tot = np.zeros(4)
    for t, (p, s) in enumerate(parts):
        c, f_r, f_i, _, _ = _oracle_args(p, s)
        sep = R.fit_gains_and_foregrounds(s["g_r"], s["g_i"], f_r, f_i, c["data_r"], c["data_i"], c["wgts"], c["fg_comps"], c["corr_inds"], **kw)
        np.testing.assert_allclose(joint[0][t * 6 : (t + 1) * 6], sep[0], rtol=1e-12)
        tot += np.asarray(sep[4]["loss"])
    np.testing.assert_allclose(joint[4]["loss"], tot, rtol=1e-12)


def test_dealt_partition_balances_every_cost_at_hera350():
    """The 8 shares of the HERA-350 job (61 075 single-baseline groups, 122 delay classes): dealing the basis-and-size-ordered list
    gives every rank the same number of baselines and of basis vectors to well under 1 %, and every delay class to within one
    baseline -- whatever a step costs per group, the shares cost the same (round 3's contiguous cut: 16 009 against 4 310 baselines)."""
    from calamity_amd import modeling

    antpos = synthetic.hex_positions(350)
    i, j = np.triu_indices(350, k=1)
    dly = np.asarray([modeling.dly_ns(L) for L in np.linalg.norm(antpos[i] - antpos[j], axis=1)])
    classes, grp_basis = np.unique(dly, return_inverse=True)
    nvec = (np.ceil(2.0 * 1024 * 97656.25 * classes * 1e-9) + 8).astype(int)[grp_basis]  # about the DPSS term count of the delay class
    parts = D.partition_groups(nvec, grp_basis, np.ones(len(nvec)), 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(nvec)))
    nb = np.asarray([len(x) for x in parts])
    nv = np.asarray([nvec[x].sum() for x in parts], dtype=np.float64)
    assert nb.max() - nb.min() <= 1 and nv.max() / nv.min() <= 1.005, (nb, nv)
    for u in range(len(classes)):
        per = np.asarray([np.count_nonzero(grp_basis[x] == u) for x in parts])
        assert per.max() - per.min() <= 1
    old = D.partition_groups(nvec, grp_basis, np.ones(len(nvec)), 8, mode="contiguous")
    assert max(len(x) for x in old) > 3 * min(len(x) for x in old)  # what the cost-model-free deal replaced


def test_host_exchange_between_worker_threads():
    """The in-process all-reduce of batched.SliceBatchFitter's workers that share a device (the callback handed to
    cal_solver_set_exchange_hook): every worker ends with the same sum / minimum, computed in rank order; an aborting worker
    releases the others."""
    import threading

    from calamity_amd.batched import _HostExchange

    n = 3
    ex = _HostExchange(n)
    rng = np.random.default_rng(0)
    bufs = [rng.standard_normal(1000).astype(np.float32) for _ in range(n)]
    ints = [np.asarray([r + 2, 7 - r], dtype=np.int32) for r in range(n)]
    want = bufs[0].copy()
    for r in range(1, n):
        want += bufs[r]
    hooks = [ex.hook(r) for r in range(n)]

    def work(r):
        for _ in range(5):  # repeated exchanges: the slots are reused
            a = bufs[r].copy()
            hooks[r](a, "sum")
            assert np.array_equal(a, want)
        b = ints[r].copy()
        hooks[r](b, "min")
        assert b.tolist() == [2, 5]

    errs = []

    def guarded(r):
        try:
            work(r)
        except BaseException as e:  # noqa: BLE001 -- an assertion inside a thread must fail the test
            errs.append(e)
            ex.abort()

    threads = [threading.Thread(target=guarded, args=(r,), daemon=True) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
        assert not t.is_alive()
    assert not errs, errs
    # a worker that dies aborts the exchange: the others get an error instead of waiting for ever
    ex2 = _HostExchange(2)
    out = []

    def waiter():
        try:
            ex2.hook(0)(np.zeros(4, dtype=np.float32), "sum")
            out.append("returned")
        except threading.BrokenBarrierError:
            out.append("broken")

    t = threading.Thread(target=waiter, daemon=True)
    t.start()
    import time as _t

    _t.sleep(0.2)
    ex2.abort()
    t.join(timeout=10)
    assert out == ["broken"]
