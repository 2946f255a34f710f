"""Diagnostic: per-phase s_memtime ticks of the split-bf16 dense kernel (split_kernels.hpp), per item and wave.  Needs a library built with
-DCAL_STAMP (make -C calamity_amd/csrc variant NAME=stamp EXTRA=-DCAL_STAMP); usage on the GPU box:
python tools/split_stamps.py calamity_amd/csrc/variants/lib_stamp.so [hera350]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from calamity_amd import _lib, synthetic  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from calamity_amd.solver import HipFitSolver  # noqa: E402

prob, truth, start = synthetic.make_config(sys.argv[2] if len(sys.argv) > 2 else "hera350", with_sky=True)
s = HipFitSolver(dtype=np.float32)
s.set_problem(prob, layout="shared")
s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
s.set_optimizer("Adam", learning_rate=1e-3)
s.run(3, record=False)
s.synchronize()
lib = _lib.load()
buf = np.zeros((4096, 4, 14), dtype=np.int64)
print("rc", lib.cal_debug_read_split_stamps(buf.ctypes.data_as(C.c_void_p)))
n = int((buf[:, 0, 7] > 0).sum())
b = buf[:n].astype(np.float64)
life = b[:, :, 7] - b[:, :, 6]
t0, t1 = b[:, :, 6].min(), b[:, :, 7].max()
print(f"{n} items; first entry -> last exit {t1 - t0:.0f} ticks; mean item lifetime {life.mean():.0f}; items resident on average {life.max(axis=1).sum() / (t1 - t0):.1f} (512 slots)")
names = ["F (incl. its waits)", "E", "B (incl. its waits)", "group wait + barrier", "coefficient wait", "sample wait"]
for k, nm in enumerate(names):
    print(f"  {nm:24s} {b[:, :, k].mean():10.0f} ticks  = {b[:, :, k].mean() / life.mean():.3f} of the lifetime")
for k, nm in ((10, "E: samples + gains there"), (11, "E: the loop"), (12, "E: tail (next samples)"), (13, "B: split of gbar_v")):
    print(f"  {nm:24s} {b[:, :, k].mean():10.0f} ticks  = {b[:, :, k].mean() / life.mean():.3f} of the lifetime")
print(f"  prologue + epilogue      {(life - b[:, :, 0] - b[:, :, 1] - b[:, :, 2]).mean():10.0f} ticks")
# where every item ran: XCD (HW_REG_XCC_ID), shader engine / array / CU (HW_REG_HW_ID); timelines per CU
hw = buf[:n, 0, 8]
xcc = buf[:n, 0, 9] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
e_item, x_item = b[:, :, 6].min(axis=1), b[:, :, 7].max(axis=1)
print(f"{len(np.unique(key))} CUs used, {len(np.unique(xcc))} XCDs")
for x in np.unique(xcc):
    m = xcc == x
    ks = np.unique(key[m])
    # the XCD's own clock: is it one time base?  spread of the CUs' first entries
    firsts = np.array([e_item[m & (key == k)].min() for k in ks])
    lasts = np.array([x_item[m & (key == k)].max() for k in ks])
    busy = np.array([(x_item - e_item)[m & (key == k)].sum() for k in ks])
    cnt = np.array([(m & (key == k)).sum() for k in ks])
    span = lasts.max() - firsts.min()
    print(f"  XCD {x}: {m.sum():4d} items on {len(ks):2d} CUs ({cnt.min()}..{cnt.max()} per CU); first entries spread {firsts.max() - firsts.min():9.0f}; span {span:9.0f} ticks; "
          f"CU busy (sum of lifetimes / 2 slots) mean {busy.mean() / 2:9.0f} max {busy.max() / 2:9.0f} -> {busy.mean() / 2 / span:.2f} of the span; CUs finish at {np.percentile(lasts - firsts.min(), [0, 25, 50, 75, 100]).round(0)}")
ent = np.sort(b[:, 0, 6] - t0)
ex = np.sort(b[:, :, 7].max(axis=1) - t0)
for q in (0.25, 0.5, 0.8, 0.9, 0.95, 0.99, 1.0):
    print(f"  {q:4.2f} of the items entered by {ent[int(q * (n - 1))]:9.0f}, exited by {ex[int(q * (n - 1))]:9.0f}")
order = np.argsort(-life.max(axis=1))
for i in list(order[:3]) + list(order[n // 2:n // 2 + 2]) + list(order[-2:]):
    print(f"  item {i:4d}: lifetime {life[i].max():8.0f}  F {b[i, :, 0].mean():8.0f} E {b[i, :, 1].mean():8.0f} B {b[i, :, 2].mean():8.0f} sync {b[i, :, 3].mean():8.0f} cw {b[i, :, 4].mean():7.0f} sw {b[i, :, 5].mean():7.0f}")
