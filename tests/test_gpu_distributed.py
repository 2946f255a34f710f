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
