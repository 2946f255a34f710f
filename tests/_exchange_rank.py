"""One rank of a sharded fit whose exchange runs through cal_solver_set_exchange_hook over gloo (helper of
tests/test_gpu_exchange_hook.py; started as a fresh process per rank, two of them sharing the one GPU)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_case(name):
    """(full problem, start, per-rank group lists or None for the cost-balanced partition, solver options, run options)"""
    from calamity_amd import synthetic

    reg = name.endswith("_sum")
    if name.startswith("fallback"):
        # 70 antennas = 2415 baselines: rank 0's share (2100) is large enough for the dense kernels, rank 1's (315) is not
        p, _, start = synthetic.make_problem(70, 128, f0=150e6, df=400e3, seed=4, with_sky=reg)
        groups = [np.arange(0, 2100), np.arange(2100, p.ngrps)]
        return p, start, groups, dict(layout="shared", kernel_path="auto"), dict(nsteps=6, tol=0.0), reg
    p, _, start = synthetic.make_problem(12, 128, f0=150e6, df=400e3, seed=6, with_sky=reg)
    rng = np.random.default_rng(8)
    start["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    start["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    path = "dense" if name.startswith("dense") else "general"
    run = dict(nsteps=400, tol=1e-7) if name.startswith("tolstop") else dict(nsteps=12, tol=0.0)
    return p, start, None, dict(layout="shared", kernel_path=path), run, reg


def priors(p):
    return float(np.sum(p.sky_r * p.wgts)), float(np.sum(p.sky_i * p.wgts))


def fit(sub, start, opts, run, reg_priors, hook=None, rank=0, world=1):
    from calamity_amd.solver import HipFitSolver

    s = HipFitSolver(dtype=np.float64)
    if hook is not None:
        s.set_exchange_hook(hook, rank, world)  # before set_problem: the ranks then agree on the kernel family there
    s.set_problem(sub, **opts)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    if reg_priors is not None:
        s.set_regularization("sum", *reg_priors)
    s.set_optimizer("Adam", learning_rate=2e-2)
    s.run(1, record=False)
    losses, stopped, nupd = s.run(run["nsteps"], record=True, tol=run["tol"])
    g_r, g_i, c_r, c_i = s.get_params()
    path = s.timing_get()["kernel_path"]
    s.close()
    return dict(losses=losses, stopped=stopped, nupd=nupd, g_r=g_r, g_i=g_i, c_r=c_r, c_i=c_i, path=path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    from calamity_amd import _lib
    from calamity_amd import distributed as D

    _lib.load()  # our HIP runtime first, then torch (used for the gloo transport only)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{args.port}", rank=args.rank, world_size=args.world)
    p, start, groups, opts, run, reg = build_case(args.case)
    if groups is None:
        sub, sub_start = D.shard_problem(p, start, args.rank, args.world)
    else:
        sub, sub_start = D.select_groups(p, start, groups[args.rank])
    calls = []

    def all_reduce(arr, op):
        calls.append((arr.dtype.str, arr.size, op))
        t = torch.from_numpy(arr)  # shares the library's staging buffer: reduced in place
        dist.all_reduce(t, op=dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.SUM)

    out = fit(sub, sub_start, opts, run, priors(p) if reg else None, hook=all_reduce, rank=args.rank, world=args.world)
    np.savez(args.out, ncalls=len(calls), call_sizes=np.asarray([c[1] for c in calls]), call_ops=np.asarray([c[2] for c in calls]),
             call_dtypes=np.asarray([c[0] for c in calls]), **{k: np.asarray(v) for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
