#!/usr/bin/env python3
"""Kernel experiment harness: time the fused basis kernel of several library builds on the same problem, in one
process per build (the child assigns calamity_amd._lib.LIB_PATH before the library is loaded).  Usage: kbench.py [--config hera350] [--max-bls N] lib1.so lib2.so ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    from calamity_amd import _lib, synthetic

    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)  # experiment build instead of the shipped one
    from calamity_amd.solver import HipFitSolver

    dtype = np.float64 if args.dtype == "f64" else np.float32
    if args.slices:
        import bench
        prob, start, _ = bench.build_sharded_job(args.config, 0, args.slices, args.slices, max_bls=args.max_bls)
    elif args.cache and os.path.exists(args.cache):
        import pickle
        prob, start = pickle.load(open(args.cache, "rb"))
    else:
        prob, truth, start = synthetic.make_config(args.config, max_bls=args.max_bls, with_sky=True)
        if args.redundant:
            prob, start = synthetic.merge_redundant_groups(prob, truth, start)
        if args.cache:
            import pickle
            pickle.dump((prob, start), open(args.cache, "wb"), protocol=4)
    s = HipFitSolver(dtype=dtype)
    s.set_problem(prob, layout=args.layout)
    s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    if args.reg:
        src_r, src_i = (prob.sky_r, prob.sky_i) if prob.sky_r is not None else (0.9 * prob.data_r, 1.1 * prob.data_i)  # (a sharded job carries no sky: any prior will do for timing)
        s.set_regularization("sum", float(np.sum(src_r * prob.wgts)), float(np.sum(src_i * prob.wgts)))
    s.set_optimizer("Adam", learning_rate=1e-2)
    out = {}
    s.run(3, record=False)
    s.timing_enable(True)
    import time
    s.synchronize(); t0 = time.perf_counter()
    s.run(args.steps, record=True, tol=0.0)
    s.synchronize(); dt = time.perf_counter() - t0
    t = s.timing_get()
    out["grad_ms"] = t["total_ms"] / t["launches"]
    out["step_ms"] = dt / args.steps * 1e3
    out["grad_GBs"] = t["algorithmic_bytes_per_launch"] / out["grad_ms"] / 1e6
    s.timing_enable(True)
    for _ in range(args.steps):
        s.eval_loss()
    t = s.timing_get()
    out["loss_ms"] = t["total_ms"] / t["launches"]
    print("KBENCH " + json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hera350")
    ap.add_argument("--max-bls", type=int, default=None)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--layout", default="stream")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--reg", action="store_true")
    ap.add_argument("--slices", type=int, default=0, help="the per-rank job of an N-GPU run: N time slices x 1/N of the baselines, sharing tiles")
    ap.add_argument("--redundant", action="store_true", help="merge redundant baselines into shared-coefficient groups")
    ap.add_argument("--cache", default="/tmp/kbench_problem.pkl")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--lib", default=None, help="(child) the library build to load")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        child(args)
        sys.exit(0)
    for rnd in range(2):
        for lib in args.libs:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--lib", os.path.abspath(lib)] + [a for a in sys.argv[1:] if not a.endswith(".so")]
            r = subprocess.run(cmd, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("KBENCH ")]
            print(rnd, os.path.basename(lib), line[0][7:] if line else ("FAILED " + r.stderr[-400:]), flush=True)
