import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calamity_amd import synthetic
from calamity_amd.solver import HipFitSolver
import pickle
if os.path.exists("/tmp/kbench_problem.pkl"):
    p, start = pickle.load(open("/tmp/kbench_problem.pkl", "rb"))
else:
    p, truth, start = synthetic.make_config("hera350", with_sky=True)
    pickle.dump((p, start), open("/tmp/kbench_problem.pkl", "wb"), protocol=4)
s = HipFitSolver(dtype=np.float32); s.set_problem(p, layout="shared"); s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
vals = [s.eval_loss() for _ in range(8)]
print(os.environ.get("CALAMITY_HIP_LIB"), ["%.12e" % v for v in vals])
g = [s.eval_grads() for _ in range(3)]
print("grad norms", [float(np.linalg.norm(x[4])) for x in g])
