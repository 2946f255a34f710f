import sys, os
sys.path.insert(0, '.')
order = sys.argv[1]
import numpy as np
if order == "lib_first":
    from calamity_amd import _lib; _lib.load()
    import torch, torch.distributed as dist
else:
    import torch, torch.distributed as dist
    from calamity_amd import _lib; _lib.load()
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
from calamity_amd import synthetic
from calamity_amd.solver import HipFitSolver, comm_unique_id
p, t, s0 = synthetic.make_config("tutorial")
s = HipFitSolver(dtype=np.float32)
s.set_problem(p); s.set_params(s0["g_r"], s0["g_i"], s0["c_r"], s0["c_i"]); s.set_optimizer("Adam", learning_rate=1e-2)
s.comm_init(comm_unique_id(), 0, 1)
l, _, _ = s.run(5)
print(order, "ok", l[:2])
maps = open("/proc/self/maps").read()
print(sorted(set(line.split()[-1] for line in maps.splitlines() if "librccl" in line or "libamdhip64" in line)))
dist.destroy_process_group()
