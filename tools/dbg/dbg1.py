import numpy as np, sys
sys.path.insert(0, '.')
from calamity_amd import synthetic, calibration, cal_utils, problem
from calamity_amd.solver import HipFitSolver
uvd, sky, vecs = synthetic.make_uvdata(nants=6, nfreqs=64, ntimes=1, seed=2)
gains = cal_utils.blank_uvcal_from_uvdata(sky)
ants_map = {ant: i for i, ant in enumerate(gains.ant_array)}
comps, corr_inds = calibration.tensorize_fg_model_comps_dict(vecs, ants_map, dtype=np.float64, nfreqs=sky.Nfreqs)
data_r, data_i, wgts = calibration.tensorize_data(sky, corr_inds, ants_map, polarization="xx", time=sky.time_array[0], dtype=np.float64, data_scale_factor=1.0)
src = calibration._flatten(data_r, comps); w = calibration._flatten(wgts, comps)
print("nvec", comps.grp_nvec, "basis ids", comps.grp_basis)
expect = np.concatenate([comps.basis[comps.grp_basis[g]].T @ src[g] for g in range(comps.ngrps)])
for layout in ("stream", "shared"):
    s = HipFitSolver(dtype=np.float64)
    import copy
    shell = copy.copy(comps)
    s.set_problem(shell, layout=layout)
    z = np.zeros_like(src)
    s.set_data(z, z, w)
    s.init_coeffs(src, z)
    c = s.get_params()[2]
    coff = comps.grp_coff
    bad = [g for g in range(comps.ngrps) if not np.allclose(c[coff[g]:coff[g+1]], expect[coff[g]:coff[g+1]], atol=1e-9)]
    print(layout, "bad groups", bad)
    if bad:
        g = bad[0]; print(c[coff[g]:coff[g]+5], expect[coff[g]:coff[g]+5])
