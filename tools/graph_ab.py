#!/usr/bin/env python3
"""Launch forms of a large problem's steps on one box: every kernel its own launch ("kernels") against replay from a hipGraph ("graph": forced;
"auto": calls of at least 256 steps), HERA-350, both layouts, several call lengths.  Same losses bit for bit; prints the time per step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from calamity_amd import synthetic
from calamity_amd.solver import HipFitSolver
p, truth, start = synthetic.make_config("hera350", with_sky=True)
for layout, lengths in (("shared", (64, 512)), ("stream", (48, 96, 256))):
    for n in lengths:
        for mode in ("kernels", "graph", "kernels", "graph"):
            s = HipFitSolver(dtype=np.float32)
            s.set_problem(p, layout=layout)
            s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
            s.set_optimizer("Adam", learning_rate=1e-3)
            s.set_launch_mode(mode)
            s.run(n, record=True, tol=0.0); s.synchronize()   # (creates the graph)
            t = time.perf_counter(); l, _, _ = s.run(n, record=True, tol=0.0); s.synchronize(); dt = time.perf_counter() - t
            print(layout, n, mode, f"{dt / n * 1e3:.4f} ms per step", f"last loss {l[-1]:.9e}", flush=True)
            s.close()
