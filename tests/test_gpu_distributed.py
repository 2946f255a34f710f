"""The multi-rank code path of bench.py / the library on ONE GPU: launched the way the driver launches N > 1
(``python -m torch.distributed.run --nproc-per-node 1 bench.py ...``) as a fresh child process, with ``--dist-rehearsal T``
so the single rank still partitions the groups, batches T time slices, rendezvouses over gloo, broadcasts the RCCL
unique id, creates a (one-rank) communicator BEFORE the problem is set, lets set_problem agree the kernel path and
steps-per-sync over the communicator and issues the grouped all-reduce every step.  A one-rank all-reduce is the
identity, so the recorded losses must equal, bit for bit, those of a plain solver without a communicator on the same
batched problem.  (The reference has no counterpart: calibration.py:1796-1804 is its only device code.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

STEPS, WARMUP, SLICES, MAX_BLS = 6, 2, 2, 4000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rehearse(layout, reg):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-rehearsal", str(SLICES),
           "--max-bls", str(MAX_BLS), "--steps", str(STEPS), "--warmup", str(WARMUP), "--layout", layout, "--reg", reg,
           "--no-cpu-baseline", "--no-shared"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("layout,reg", [("stream", "none"), ("stream", "sum"), ("shared", "none"), ("shared", "sum")])
def test_one_rank_rehearsal_under_torchrun_matches_plain_solver(layout, reg):
    out = _rehearse(layout, reg)
    assert out["n_gpus"] == 1 and out["steps"] == STEPS and out["warmup"] == WARMUP
    assert "rehearsal" in out["config"]["parallelism"]
    assert f"{SLICES} time slice(s)" in out["config"]["workload"]
    losses = np.asarray(out["extra"]["losses"])
    assert losses.shape == (STEPS,) and np.all(np.isfinite(losses)) and losses[-1] < losses[0]

    import bench
    from calamity_amd.solver import HipFitSolver

    # the same job in a plain solver: one loop state per time slice (each slice its own priors and loss), the job's loss their sum
    prob, start, _ = bench.build_sharded_job("hera350", 0, 1, SLICES, reg=reg == "sum", max_bls=MAX_BLS, per_slice=True)
    s = HipFitSolver(dtype=np.float32)
    s.set_problem(prob, layout=layout)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    if reg == "sum":
        nb = prob.nbls // SLICES
        s.set_regularization("sum", np.asarray([float(np.sum((prob.sky_r * prob.wgts)[t * nb : (t + 1) * nb])) for t in range(SLICES)]),
                             np.asarray([float(np.sum((prob.sky_i * prob.wgts)[t * nb : (t + 1) * nb])) for t in range(SLICES)]))
    s.set_optimizer("Adam", learning_rate=1e-2)
    s.run_slices(WARMUP, record=False)
    ref = np.sum([r[0] for r in s.run_slices(STEPS, record=True, tol=0.0)], axis=0)
    s.close()
    assert np.array_equal(losses, ref), (losses, ref)
    assert out["n_ranks_seen"] == 1


def test_two_self_launched_ranks_share_the_gpu_over_host_sockets():
    """`bench.py --gpus 2` WITHOUT a launcher: it starts its two ranks itself (fresh processes); with `--transport host` they share
    the one GPU of this box and exchange through the library's hook, so the whole N-rank path of the bench runs here: the dealt
    partition, the per-slice priors summed over ranks, one loop state per slice, the per-step exchange, the max-over-ranks timing,
    rank 0's line -- `n_ranks_seen` counted by the exchange itself.  The job (2 time slices, each slice's baselines over 2 ranks)
    is the job of the one-rank rehearsal with 2 slices: same losses to fp32 summation order."""
    common = ["--max-bls", str(MAX_BLS), "--steps", str(STEPS), "--warmup", str(WARMUP), "--layout", "stream", "--reg", "sum", "--no-cpu-baseline", "--no-shared"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host"] + common, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["n_ranks_seen"] == 2 and two["config"]["transport"] == "host"
    # the same job in ONE plain solver: both ranks' shares of every slice put together (slice-major), per-slice priors summed
    import bench
    from calamity_amd.problem import FitProblem
    from calamity_amd.solver import HipFitSolver

    shares = [bench.build_sharded_job("hera350", r, 2, 2, reg=True, max_bls=MAX_BLS, per_slice=True) for r in range(2)]
    na = shares[0][2]
    rows, coefs, gb, nb_r, nc_r = [], [], [], [], []
    for r, (p_r, st_r, _) in enumerate(shares):
        nb_r.append(p_r.nbls // 2)
        nc_r.append(p_r.ncoeffs // 2)
    off = [0, len(shares[0][0].basis)]
    cat = {k: [] for k in ("grp_basis", "bl_ant0", "bl_ant1", "bl_rowblk", "data_r", "data_i", "wgts", "sky_r", "sky_i")}
    c_r, c_i = [], []
    for t in range(2):
        for r, (p_r, st_r, _) in enumerate(shares):
            rs, cs = slice(t * nb_r[r], (t + 1) * nb_r[r]), slice(t * nc_r[r], (t + 1) * nc_r[r])
            cat["grp_basis"].append(p_r.grp_basis[rs] + off[r])
            for k in ("bl_ant0", "bl_ant1", "bl_rowblk", "data_r", "data_i", "wgts", "sky_r", "sky_i"):
                cat[k].append(getattr(p_r, k)[rs])
            c_r.append(st_r["c_r"][cs])
            c_i.append(st_r["c_i"][cs])
    cc = {k: np.concatenate(v) for k, v in cat.items()}
    nbl = len(cc["bl_ant0"])
    whole = FitProblem(nants=2 * na, nfreqs=shares[0][0].nfreqs, basis=list(shares[0][0].basis) + list(shares[1][0].basis),
                       grp_basis=cc["grp_basis"].astype(np.int32), grp_bl_start=np.arange(nbl + 1, dtype=np.int32), bl_ant0=cc["bl_ant0"],
                       bl_ant1=cc["bl_ant1"], bl_rowblk=cc["bl_rowblk"], data_r=cc["data_r"], data_i=cc["data_i"], wgts=cc["wgts"], nslices=2)
    s = HipFitSolver(dtype=np.float32)
    s.set_problem(whole, layout="stream")
    s.set_params(shares[0][1]["g_r"], shares[0][1]["g_i"], np.concatenate(c_r), np.concatenate(c_i))  # (the gains are the same on every rank)
    half = nbl // 2
    s.set_regularization("sum", np.asarray([float(np.sum((cc["sky_r"] * cc["wgts"])[t * half : (t + 1) * half])) for t in range(2)]),
                         np.asarray([float(np.sum((cc["sky_i"] * cc["wgts"])[t * half : (t + 1) * half])) for t in range(2)]))
    s.set_optimizer("Adam", learning_rate=1e-2)
    s.run_slices(WARMUP, record=False)
    b = np.sum([r_[0] for r_ in s.run_slices(STEPS, record=True, tol=0.0)], axis=0)
    s.close()
    a = np.asarray(two["extra"]["losses"])
    assert a.shape == b.shape == (STEPS,) and a[-1] < a[0]
    np.testing.assert_allclose(a, b, rtol=2e-5)
    # the same two ranks started the way the driver starts them: the torch launcher, whose own store listens on --master-port (the
    # ranks' socket rendezvous must not need that port: calamity_amd/rendezvous.py)
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host"] + common,
                              cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert launched.returncode == 0, launched.stdout[-2000:] + "\n" + launched.stderr[-4000:]
    lines = [ln for ln in launched.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, launched.stdout[-2000:]
    under = json.loads(lines[0])
    assert under["n_gpus"] == 2 and under["n_ranks_seen"] == 2
    np.testing.assert_allclose(np.asarray(under["extra"]["losses"]), b, rtol=2e-5)
    # a node with fewer GPUs than ranks is an error on every rank under the default transport -- never a one-GPU measurement
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    from calamity_amd import _lib

    if _lib.device_count() < 2:
        assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("split,gpus", [("slices", 2), ("groups", 2), ("groups", 1)])
def test_bench_times_the_products_own_multi_device_split(split, gpus):
    """`bench.py --gpus N --split slices|groups`: ONE process drives N workers through batched.SliceBatchFitter as
    calibration._fit_slices_batched does -- whole slices per device without an exchange, or every slice's groups shared with one
    all-reduce per step.  On this one-GPU box the workers share device 0 (host exchange between them); N = 1 with 'groups' joins a
    one-rank RCCL communicator from the worker thread.  The line names the split it timed; the steps optimise."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--split", split, "--max-bls", str(MAX_BLS), "--steps", "6", "--warmup", "2",
           "--layout", "shared"] + (["--split-workers-on-one-gpu"] if gpus > 1 else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == gpus and line["unit"] == "slice-steps/s" and f"device_split='{split}'" in line["config"]["parallelism"]
    assert line["extra"]["loss_last"] < line["extra"]["loss_first"]
    assert line["extra"]["ranks_the_exchange_spans"] == (gpus if split == "groups" else 1)
