#!/usr/bin/env python3
"""A/B of the two fp32 dense kernels of the SHARED layout: the split-bf16 kernel (kernel_path "dense", split_kernels.hpp) against the
v_mfma_f32_32x32x2_f32 kernel it replaced ("dense_f32", dense_kernels.hpp).
  * accuracy: loss and every gradient of both against the fp64 general kernel on the same problem (relative l2 error)
  * rate: HIP-event time of the gradient pass and wall time of an Adam step, with and without the "sum" regulariser
Usage: split_ab.py [--nants N --nfreqs F | --config hera350] [--steps K] [--no-time] [--paths dense,dense_f32]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from calamity_amd import synthetic  # noqa: E402
from calamity_amd.solver import HipFitSolver  # noqa: E402


def relnorm(a, b):
    a = np.asarray(a, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None)
    ap.add_argument("--nants", type=int, default=24)
    ap.add_argument("--nfreqs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--paths", default="dense,dense_f32")
    ap.add_argument("--reg", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--lib", default=None, help="an experiment build of the library (make variant) instead of the shipped one")
    args = ap.parse_args()
    if args.lib:
        from calamity_amd import _lib

        _lib.LIB_PATH = os.path.abspath(args.lib)
    t0 = time.time()
    if args.config:
        p, truth, start = synthetic.make_config(args.config, with_sky=True)
    else:
        p, truth, start = synthetic.make_problem(args.nants, args.nfreqs, f0=100e6, df=100e6 / args.nfreqs, seed=3, with_sky=True)
    rng = np.random.default_rng(5)
    start = dict(start)
    start["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    start["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    print(f"problem: nants {p.nants} nfreqs {p.nfreqs} nbls {p.nbls} sum nvec {int(p.grp_coff[-1])}  ({time.time() - t0:.1f} s)", flush=True)
    prior = (float(np.sum(p.sky_r * p.wgts)), float(np.sum(p.sky_i * p.wgts)))

    def solver(dtype, path):
        s = HipFitSolver(dtype=dtype)
        s.set_problem(p, layout="shared", kernel_path=path)
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        if args.reg:
            s.set_regularization("sum", *prior)
        return s

    ref = solver(np.float64, "general")
    l64, g64_r, g64_i, c64_r, c64_i = ref.eval_grads()
    ref.close()
    print(f"fp64 general: loss {l64:.12e}", flush=True)
    out = {"nbls": int(p.nbls), "reg": bool(args.reg)}
    for path in args.paths.split(","):
        s = solver(np.float32, path)
        l, g_r, g_i, c_r, c_i = s.eval_grads()
        rec = dict(loss_rel=abs(l - l64) / abs(l64), g_r=relnorm(g_r, g64_r), g_i=relnorm(g_i, g64_i), c_r=relnorm(c_r, c64_r), c_i=relnorm(c_i, c64_i))
        l2 = s.eval_loss()
        rec["loss_only_rel"] = abs(l2 - l64) / abs(l64)
        print(f"{path:10s} loss {l:.9e} rel {rec['loss_rel']:.2e} (loss-only pass {rec['loss_only_rel']:.2e})  grad g {rec['g_r']:.2e} {rec['g_i']:.2e}  grad c {rec['c_r']:.2e} {rec['c_i']:.2e}", flush=True)
        if not args.no_time:
            s.set_optimizer("Adam", learning_rate=1e-3)
            s.run(3, record=False)
            s.synchronize()
            t1 = time.perf_counter()
            s.run(args.steps, record=True, tol=0.0)
            s.synchronize()
            dt = time.perf_counter() - t1
            s.timing_enable(True)
            s.run(args.steps, record=True, tol=0.0)
            s.synchronize()
            t = s.timing_get()
            rec["pass_ms"] = t["total_ms"] / max(t["launches"], 1)
            rec["step_ms"] = dt / args.steps * 1e3
            rec["useful_TF"] = t["flops_per_launch"] / rec["pass_ms"] / 1e9
            rec["kernel_path"] = t["kernel_path"]
            print(f"{path:10s} timed region per step {rec['pass_ms']:.4f} ms ({rec['useful_TF']:.1f} useful TF)   Adam step {rec['step_ms']:.4f} ms = {1e3 / rec['step_ms']:.0f} steps/s  [{t['kernel_path']}]", flush=True)
        out[path] = rec
        s.close()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
