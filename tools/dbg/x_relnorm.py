import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calamity_amd import synthetic
from calamity_amd.solver import HipFitSolver
p, truth, start = synthetic.make_config("hera350", with_sky=True)
rng = np.random.default_rng(2)
start = dict(start)
start["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
start["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
def rn(a, b): return np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b)
ref = HipFitSolver(dtype=np.float64); ref.set_problem(p, layout="shared"); ref.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
r = ref.eval_grads(); ref.close()
for layout in ("shared", "stream"):
    s = HipFitSolver(dtype=np.float32); s.set_problem(p, layout=layout); s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
    o = s.eval_grads(); s.close()
    print(layout, "loss", abs(o[0]-r[0])/abs(r[0]), "g_r", rn(o[1], r[1]), "g_i", rn(o[2], r[2]), "c_r", rn(o[3], r[3]), "c_i", rn(o[4], r[4]))
