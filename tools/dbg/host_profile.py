"""Where does the wall time of calibrate_and_model_dpss go on a mid-size array (host python vs device)?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from calamity_amd import calibration, synthetic

t0 = time.perf_counter()
uvd, sky, vecs = synthetic.make_uvdata(nants=int(sys.argv[1]) if len(sys.argv) > 1 else 61, nfreqs=256, ntimes=2, seed=1, redundant=True)
print("make_uvdata", time.perf_counter() - t0, "s;", uvd.Nbls, "baselines")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
model, resid, gains, hist = calibration.calibrate_and_model_dpss(uvdata=uvd, sky_model=None, maxsteps=1000, tol=0.0, dtype=np.float32,
                                                                 optimizer="Adam", learning_rate=1e-2, min_dly=2 / 0.3, offset=2 / 0.3)
pr.disable()
print("calibrate_and_model_dpss", time.perf_counter() - t0, "s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
