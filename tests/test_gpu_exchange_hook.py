"""The multi-rank algebra on REAL kernels with two ranks on one GPU.

RCCL refuses two ranks on one device, so the exchange goes through cal_solver_set_exchange_hook: wherever the library would
call ncclAllReduce it hands the same buffers and counts to a callback, here a gloo all-reduce between two fresh child
processes that share the GPU (tests/_exchange_rank.py).  Baselines are sharded by whole fitting groups
(calamity_amd/distributed.py); gains are replicated.  Checked against the single-solver fit of the whole problem, fp64,
1e-10: recorded losses, the replicated gains (bit-identical on both ranks), every rank's coefficients -- for the general
and the dense (matrix-core) kernels, without and with the "sum" regulariser (three-part gain payload / the dense path's
extra scalar exchange); a shard that is too small for the dense kernels drags BOTH ranks to the general ones; a tolerance
stop lands on the same step on both.  (The reference is single-device: calibration.py:1796-1804.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_two_ranks(case, tmp_path):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"{case}_rank{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_exchange_rank.py"), "--case", case, "--rank", str(r),
                               "--port", str(port), "--out", outs[r]], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    logs = []
    try:
        for pr in procs:
            logs.append(pr.communicate(timeout=300)[0])
    finally:  # (whatever ends this -- a timeout here, pytest-timeout's alarm -- no rank is left behind holding the GPU)
        for q in procs:
            if q.poll() is None:
                q.kill()
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{logs[r][-4000:]}"
    return [np.load(o) for o in outs]


def relnorm(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300)


def reference(case):
    import _exchange_rank as X

    p, start, groups, opts, run, reg = X.build_case(case)
    if case.startswith("fallback"):
        opts = dict(opts, kernel_path="general")
    ref = X.fit(p, start, opts, run, X.priors(p) if reg else None)
    return p, start, groups, ref


def shard_coeffs(p, ref, groups, rank):
    from calamity_amd import distributed as D

    if groups is None:
        nbl_g = np.diff(p.grp_bl_start)
        groups_r = D.partition_groups(p.grp_nvec, p.grp_basis, nbl_g, 2)[rank]
    else:
        groups_r = groups[rank]
    coff = p.grp_coff
    idx = np.concatenate([np.arange(coff[g], coff[g + 1]) for g in groups_r])
    return ref["c_r"][idx], ref["c_i"][idx]


@pytest.mark.parametrize("case", ["general_none", "general_sum", "dense_none", "dense_sum", "tolstop_none", "fallback_none", "fallback_sum"])
def test_two_ranks_on_one_gpu_equal_the_single_solver_fit(case, tmp_path):
    p, start, groups, ref = reference(case)
    ranks = run_two_ranks(case, tmp_path)
    want_path = "dense" if case.startswith("dense") else "general"
    assert ref["path"] == want_path
    nrec = len(ref["losses"])
    if case.startswith("tolstop"):
        assert ref["stopped"] and 5 < nrec < 400
    for r, out in enumerate(ranks):
        assert str(out["path"]) == want_path, (case, r)  # fallback: rank 0's share alone would have taken the dense kernels
        assert int(out["nupd"]) == ref["nupd"] and bool(out["stopped"]) == bool(ref["stopped"]) and len(out["losses"]) == nrec, (case, r)
        np.testing.assert_allclose(out["losses"], ref["losses"], rtol=1e-10)
        assert relnorm(out["g_r"], ref["g_r"]) <= 1e-10 and relnorm(out["g_i"], ref["g_i"]) <= 1e-10, (case, r)
        c_r, c_i = shard_coeffs(p, ref, groups, r)
        assert relnorm(out["c_r"], c_r) <= 1e-10 and relnorm(out["c_i"], c_i) <= 1e-10, (case, r)
    # replicated gains: both ranks applied the identical update to the identical all-reduced gradient
    np.testing.assert_array_equal(ranks[0]["g_r"], ranks[1]["g_r"])
    np.testing.assert_array_equal(ranks[0]["g_i"], ranks[1]["g_i"])
    np.testing.assert_array_equal(ranks[0]["losses"], ranks[1]["losses"])
    # what was exchanged: the set-up agreement (4 ints, min), then per step the gain-gradient parts and 4 double scalars
    from calamity_amd import distributed as D

    reg = case.endswith("_sum")
    spec = D.exchange_spec(p.nants, 128, reg_sum=reg and want_path == "general")
    sizes, ops = ranks[0]["call_sizes"], [str(o) for o in ranks[0]["call_ops"]]
    assert sizes[0] == 4 and ops[0] == "min" and all(o == "sum" for o in ops[1:])
    per_step = [spec["gain_grad_reals"], 4] if not (reg and want_path == "dense") else [4, spec["gain_grad_reals"], 4]
    body = list(sizes[1:])
    assert len(body) % len(per_step) == 0 and body == per_step * (len(body) // len(per_step)), (case, body[:8])
