import sys, os, numpy as np, pickle, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CALAMITY_HIP_LIB"] = os.path.join(ROOT, "calamity_amd/csrc/variants/" + (sys.argv[1] if len(sys.argv) > 1 else "lib_stamp.so"))
from calamity_amd import synthetic, _lib
from calamity_amd.solver import HipFitSolver
cache = "/tmp/kbench_problem.pkl"
if os.path.exists(cache):
    prob, start = pickle.load(open(cache, "rb"))
else:
    prob, truth, start = synthetic.make_config("hera350", with_sky=True)
    pickle.dump((prob, start), open(cache, "wb"), protocol=4)
s = HipFitSolver(dtype=np.float32)
s.set_problem(prob, layout="shared")
s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
s.set_optimizer("Adam", learning_rate=1e-2)
s.run(3, record=False)
lib = _lib.load()
buf = np.zeros((8, 8, 16, 4), dtype=np.int64)
print("rc", lib.cal_debug_read_stamps(buf.ctypes.data_as(C.c_void_p)))
for blk in (0, 3):
    t00 = buf[blk, 0, 0, 0]
    print("block", blk, "realtime ticks (100 MHz)", buf[blk,0,15,0]-buf[blk,0,14,0], "memtime", buf[blk,0,9,3]-buf[blk,0,0,3], "nvec", buf[blk,0,14,1], "NT", buf[blk,0,14,2], "roles", [int(buf[blk,w,14,3]) for w in range(4)])
    for wave in (0, 3, 4, 7):
        st = buf[blk, wave] - t00
        print(" wave", wave, "role", "M" if wave < 4 else "E")
        for k in range(10):
            a = st[k]
            if wave < 4:
                print(f"   tick {k}: start {a[0]:7d}  F {a[1]-a[0]:6d}  B {a[2]-a[1]:6d}  barrier-wait {a[3]-a[2]:6d}")
            else:
                print(f"   tick {k}: start {a[0]:7d}  process {a[1]-a[0]:6d}  request {a[2]-a[1]:6d}  barrier-wait {a[3]-a[2]:6d}")
