#!/usr/bin/env python3
"""Step time of ONE rank's share of `bench.py --gpus N` (N time slices x a dealt 1/N of every slice's baselines, the slices sharing
basis tiles) on a single GPU, without the exchange: what every GPU of an N-GPU run does between two all-reduces.
usage: python tools/share_bench.py [--ranks 2 4 8] [--steps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from calamity_amd.solver import HipFitSolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rank", type=int, default=0)
    args = ap.parse_args()
    for n in args.ranks:
        prob, start, _ = bench.build_sharded_job("hera350", args.rank % n, n, n, per_slice=True)
        s = HipFitSolver(dtype=np.float32)
        s.set_problem(prob, layout="stream")
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        s.set_optimizer("Adam", learning_rate=1e-2)
        s.run_slices(3, record=False, tol=0.0)
        s.timing_enable(True)
        s.synchronize()
        t0 = time.perf_counter()
        s.run_slices(args.steps, record=True, tol=0.0)
        s.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        tim = s.timing_get()
        print(json.dumps(dict(n_ranks=n, rank=args.rank % n, slices=n, baselines_per_slice=prob.nbls // n, ms_per_step=1e3 * dt,
                              kernel_ms=tim["total_ms"] / max(tim["launches"], 1), algorithmic_GB_per_launch=tim["algorithmic_bytes_per_launch"] / 1e9, predicted_slice_steps_per_s_without_exchange=n / dt,
                              device_memory_GB=s.memory_bytes() / 1e9)), flush=True)
        s.close()
        del prob, start


if __name__ == "__main__":
    main()
