import numpy as np, sys, copy
sys.path.insert(0, '.')
from calamity_amd import synthetic, calibration, cal_utils, problem
uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=1, seed=2)
gains = cal_utils.blank_uvcal_from_uvdata(sky)
ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, dtype=np.float64, nfreqs=sky.Nfreqs)
data_r, data_i, wgts = calibration.tensorize_data(sky, corr_inds, ants_map, polarization="xx", time=sky.time_array[0], dtype=np.float64, data_scale_factor=1.0)
sr = calibration._flatten(data_r, comps); si = calibration._flatten(data_i, comps)
er = np.concatenate([comps.basis[comps.grp_basis[g]].T @ sr[g] for g in range(comps.ngrps)])
ei = np.concatenate([comps.basis[comps.grp_basis[g]].T @ si[g] for g in range(comps.ngrps)])
c_re = calibration.tensorize_fg_coeffs(data_r, wgts, comps)
snap = c_re[0].copy()
print("after first: err", np.abs(problem.coeffs_from_chunks(comps, c_re)-er).max())
c_im = calibration.tensorize_fg_coeffs(data_i, wgts, comps)
print("c_re changed:", np.abs(c_re[0]-snap).max(), "same obj", c_re[0] is c_im[0])
print("c_im err", np.abs(problem.coeffs_from_chunks(comps, c_im)-ei).max(), " c_re err now", np.abs(problem.coeffs_from_chunks(comps, c_re)-er).max())
print(comps.chunk_shapes, comps.chunk_of_grp, comps.pos_in_chunk)
