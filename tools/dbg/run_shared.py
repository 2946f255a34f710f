import sys, numpy as np, pickle, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calamity_amd import synthetic
from calamity_amd.solver import HipFitSolver
cache = "/tmp/kbench_problem.pkl"
if os.path.exists(cache):
    prob, start = pickle.load(open(cache, "rb"))
else:
    prob, truth, start = synthetic.make_config("hera350", with_sky=True)
    pickle.dump((prob, start), open(cache, "wb"), protocol=4)
s = HipFitSolver(dtype=np.float32)
s.set_problem(prob, layout=sys.argv[1] if len(sys.argv) > 1 else "shared")
s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
s.set_optimizer("Adam", learning_rate=1e-2)
s.run(3, record=False)
for _ in range(3): s.eval_loss()
