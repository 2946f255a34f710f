#!/usr/bin/env python3
"""Step rate of a batch of T time slices in one solver (cal_solver_run_slices) against T: tools/slices_bench.py [--config tutorial] [--T 1 4 16 64]."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calamity_amd import synthetic  # noqa: E402
from calamity_amd.batched import SliceBatchFitter  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="tutorial")
ap.add_argument("--T", type=int, nargs="+", default=[1, 4, 16, 64])
ap.add_argument("--steps", type=int, default=2048)
ap.add_argument("--layout", default="shared")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--reg", default="none")
args = ap.parse_args()
dtype = np.float32 if args.dtype == "f32" else np.float64
p, _, start = synthetic.make_config(args.config, with_sky=args.reg == "sum")
for nt in args.T:
    f = SliceBatchFitter(p, nt, dtype=dtype, layout=args.layout, devices=[0])
    cat = lambda a: np.concatenate([a] * nt)  # noqa: E731
    f.set_data(cat(p.data_r), cat(p.data_i), cat(p.wgts))
    f.set_params(cat(start["g_r"]), cat(start["g_i"]), cat(start["c_r"]), cat(start["c_i"]))
    if args.reg == "sum":
        pr, pi = float(np.sum(p.sky_r * p.wgts)), float(np.sum(p.sky_i * p.wgts))
        f.set_regularization("sum", np.full(nt, pr), np.full(nt, pi))
    else:
        f.set_regularization(None)
    f.set_optimizer("Adam", learning_rate=1e-2)
    f.run_slices(64, record=False)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        f.run_slices(args.steps, record=True, tol=0.0)
        best = max(best, args.steps * nt / (time.perf_counter() - t0))
    print(f"{args.config} {args.dtype} {args.layout} T={nt}: {best:.0f} slice-steps/s, {nt / best * 1e6:.1f} us per step", flush=True)
    f.close()
