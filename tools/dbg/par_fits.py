import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from calamity_amd import calibration, synthetic
uvd, sky, vecs = synthetic.make_uvdata(nants=37, nfreqs=256, ntimes=8, seed=1, redundant=True)
kw = dict(min_dly=2 / 0.3, offset=2 / 0.3, uvdata=uvd, sky_model=None, maxsteps=3000, tol=0.0, dtype=np.float32, optimizer="Adam", learning_rate=1e-2)
calibration.calibrate_and_model_dpss(**dict(kw, maxsteps=10))  # warm up (context, basis)
for n in (1, 2, 4, 8):
    t0 = time.perf_counter()
    calibration.calibrate_and_model_dpss(parallel_fits=n, **kw)
    print("parallel_fits", n, f"{time.perf_counter() - t0:.2f} s for 8 slices x 3000 steps of {uvd.Nbls} baselines x 256 channels", flush=True)
