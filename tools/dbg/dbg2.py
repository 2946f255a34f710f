import numpy as np, sys
sys.path.insert(0, '.')
from calamity_amd import synthetic, calibration, cal_utils, problem
from oracle import ref_numpy as R
uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=1, seed=2)
gains = cal_utils.blank_uvcal_from_uvdata(sky)
ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, dtype=np.float64, nfreqs=sky.Nfreqs)
rmsdata = np.sqrt(np.mean(np.abs(sky.data_array)**2))
data_r, data_i, wgts = calibration.tensorize_data(sky, corr_inds, ants_map, polarization="xx", time=sky.time_array[0], dtype=np.float64, data_scale_factor=rmsdata)
src = calibration._flatten(data_r, comps)
expect = np.concatenate([comps.basis[comps.grp_basis[g]].T @ src[g] for g in range(comps.ngrps)])
c_re = calibration.tensorize_fg_coeffs(data_r, wgts, comps)
c_im = calibration.tensorize_fg_coeffs(data_i, wgts, comps)
got = problem.coeffs_from_chunks(comps, c_re)
print("gpu vs numpy", np.abs(got-expect).max())
padded = problem.chunks_from_problem(comps)["fg_comps"]
o = R.tensorize_fg_coeffs(data_r, wgts, padded)
print("oracle vs numpy", np.abs(problem.coeffs_from_chunks(comps, o)-expect).max())
print("grams", len(calibration._gram_factors(comps)))
A = comps.basis[0]; print("orth", np.abs(A.T@A-np.eye(A.shape[1])).max())
print(c_re[0][:3,0,0,0], o[0][:3,0,0,0], expect[:3])
