#!/usr/bin/env python3
"""Step rate of the small configurations (launch-latency bound): tools/small_bench.py [--lib other.so] [--which notebook tutorial hera37]."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--which", nargs="+", default=["notebook", "tutorial", "hera37"])
ap.add_argument("--steps", type=int, default=5000)
ap.add_argument("--mode", default="auto")
args = ap.parse_args()
from calamity_amd import _lib, synthetic  # noqa: E402

if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from calamity_amd.solver import HipFitSolver  # noqa: E402

for name in args.which:
    if name == "notebook":
        p, _, st = synthetic.make_problem(15, 200, f0=100e6, df=100e3, seed=0)
        dtype, opt = np.float32, "Adamax"
    elif name == "tutorial":
        p, _, st = synthetic.make_config("tutorial")
        dtype, opt = np.float32, "Adam"
    else:
        p, _, st = synthetic.make_config("hera37")
        dtype, opt = np.float64, "Adam"
    s = HipFitSolver(dtype=dtype)
    s.set_problem(p, layout="shared")
    s.set_launch_mode(args.mode)
    s.set_params(st["g_r"], st["g_i"], st["c_r"], st["c_i"])
    s.set_optimizer(opt, learning_rate=1e-2)
    s.run(200, record=False)
    best = 0.0
    for _ in range(3):
        s.synchronize()
        t0 = time.perf_counter()
        s.run(args.steps, record=True, tol=0.0)
        s.synchronize()
        best = max(best, args.steps / (time.perf_counter() - t0))
    print(f"{name}: nvec {p.grp_nvec.min()}..{p.grp_nvec.max()}, {p.nbls} baselines x {p.nfreqs} ch, {np.dtype(dtype).name} {opt}: {best:.0f} steps/s ({1e6 / best:.1f} us)", flush=True)
    s.close()
