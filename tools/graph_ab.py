import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from calamity_amd import synthetic
from calamity_amd.solver import HipFitSolver
p, truth, start = synthetic.make_config("hera350", with_sky=True)
for layout in ("shared", "stream"):
    for mode in ("kernels", "auto", "kernels", "auto"):
        s = HipFitSolver(dtype=np.float32)
        s.set_problem(p, layout=layout)
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        s.set_optimizer("Adam", learning_rate=1e-3)
        s.set_launch_mode(mode)
        n = 512 if layout == "shared" else 256
        s.run(n, record=True, tol=0.0); s.synchronize()   # (creates the graph)
        t = time.perf_counter(); l, _, _ = s.run(n, record=True, tol=0.0); s.synchronize(); dt = time.perf_counter() - t
        print(layout, mode, f"{dt / n * 1e3:.4f} ms per step", f"last loss {l[-1]:.9e}", flush=True)
        s.close()
