import numpy as np, sys, copy
sys.path.insert(0, '.')
from calamity_amd import synthetic, calibration, cal_utils, problem
from calamity_amd.solver import HipFitSolver
uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=1, seed=2)
gains = cal_utils.blank_uvcal_from_uvdata(sky)
ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, dtype=np.float64, nfreqs=sky.Nfreqs)
data_r, data_i, wgts = calibration.tensorize_data(sky, corr_inds, ants_map, polarization="xx", time=sky.time_array[0], dtype=np.float64, data_scale_factor=1.0)
sr = calibration._flatten(data_r, comps); si = calibration._flatten(data_i, comps); w = calibration._flatten(wgts, comps)
er = np.concatenate([comps.basis[comps.grp_basis[g]].T @ sr[g] for g in range(comps.ngrps)])
ei = np.concatenate([comps.basis[comps.grp_basis[g]].T @ si[g] for g in range(comps.ngrps)])
s = HipFitSolver(dtype=np.float64)
s.set_problem(copy.copy(comps), layout="shared")
z = np.zeros_like(sr)
s.set_data(z, z, w)
for it in range(4):
    s.init_coeffs(sr, z); a = s.get_params()
    s.init_coeffs(si, z); b = s.get_params()
    s.init_coeffs(sr, si); c = s.get_params()
    print(it, np.abs(a[2]-er).max(), np.abs(a[3]).max(), np.abs(b[2]-ei).max(), np.abs(c[2]-er).max(), np.abs(c[3]-ei).max())
