"""Diagnostic: per-phase cycle counts (forward MFMA, element-wise stage, adjoint MFMA) of the dense kernel, per panel and
wave.  Needs a library built with -DCAL_STAMP (make -C calamity_amd/csrc variant NAME=stamp EXTRA=-DCAL_STAMP);
usage on the GPU box: python tools/dense_stamps.py calamity_amd/csrc/variants/lib_stamp.so"""
import sys, os, numpy as np, pickle, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calamity_amd import synthetic, _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])  # the -DCAL_STAMP build
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
buf = np.zeros((4096, 4, 32), dtype=np.int64)
rc = lib.cal_debug_read_stamps(buf.ctypes.data_as(C.c_void_p))
print("rc", rc)
nv = np.sort(prob.grp_nvec)[::-1]
for pi in (0, 5, 300, 1000, 1500, 2000, 2048, 2500, 3000, 3500, 3800):
    nvec = int(nv[min(pi * 16, len(nv) - 1)])
    ngk, NT = (nvec + 7) // 8, (nvec + 31) // 32
    f, e, b, tot, pro, epi = buf[pi].mean(axis=0)[:6]
    print(f"panel {pi:4d} nvec~{nvec:3d}: F {f:8.0f} (ideal {8*ngk*256:6d})  E {e:8.0f}  B {b:8.0f} (ideal {8*NT*4*256:6d})  loop {tot:8.0f}  prologue {pro:7.0f}  epilogue {epi:7.0f}  per-wave spread loop {buf[pi][:,3].min()}..{buf[pi][:,3].max()}")
# the stamps are those of the LAST launch that touched each panel: the gradient pass of the last step
npan = int((buf[:, 0, 7] > 0).sum())
b = buf[:npan]
t0, t1 = b[:, :, 6].min(), b[:, :, 7].max()
life = (b[:, :, 7] - b[:, :, 6]).max(axis=1)
print(f"{npan} panels; first entry -> last exit {t1 - t0} ticks; sum of panel lifetimes {life.sum()} = {life.sum() / (t1 - t0):.1f} panels resident on average (512 slots)")
print(f"share of panel lifetime: prologue {b[:, :, 4].mean() / life.mean():.3f}  F {b[:, :, 0].mean() / life.mean():.3f}  E {b[:, :, 1].mean() / life.mean():.3f}  B {b[:, :, 2].mean() / life.mean():.3f}  epilogue {b[:, :, 5].mean() / life.mean():.3f}")
ent = np.sort(b[:, 0, 6] - t0); ex = np.sort(b[:, :, 7].max(axis=1) - t0)
for q in (0.5, 0.8, 0.9, 0.95, 0.99, 1.0):
    print(f"  {q:4.2f} of the panels entered by {ent[int(q * (npan - 1))]:9d}, exited by {ex[int(q * (npan - 1))]:9d}")
tag = sys.argv[2] if len(sys.argv) > 2 else "dense_stamps"
np.save(os.path.join(ROOT, "gpurun_out", tag + ".npy"), b)
