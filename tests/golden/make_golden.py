#!/usr/bin/env python3
"""Generates tests/golden/fit_small.npz and tests/golden/optimizers_small.npz.

RESTATEMENT-DERIVED, not reference-derived: the reference (TensorFlow + pyuvdata + hera_filters) cannot be imported in
the build container (ModuleNotFoundError: tensorflow -- an ordinary error), and its tests hold no golden numbers for
this path.  The expected values come from oracle/ref_numpy.py in float64, whose forward ops restate
/root/reference/calamity/calibration.py:1587-1656 one-for-one and whose adjoints are pinned against torch autograd and
finite differences (tests/test_oracle.py).  The fixture freezes those numbers so that later edits of the oracle or of
the HIP path cannot drift silently.

Cases: seeded 7-antenna x 24-channel per-baseline DPSS problem with 5 % flags, one redundant fitting group (shared
coefficients, B = 3); loss and gradients with and without the "sum" regulariser; 10-step Adam and Adamax trajectories
(loop semantics of calibration.py:681-717), freeze_model, use_min.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from calamity_amd import problem, synthetic  # noqa: E402
from oracle import ref_numpy as R  # noqa: E402


def build():
    p, truth, start = synthetic.make_problem(7, 24, f0=150e6, df=400e3, seed=42, with_sky=True)
    p, start = synthetic.add_redundant_group(p, start, np.random.default_rng(42), nred=3)
    rng = np.random.default_rng(43)
    start["g_r"] = 1.0 + 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    start["g_i"] = 0.05 * rng.standard_normal((p.nants, p.nfreqs))
    return p, start


def main():
    p, start = build()
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, start["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, start["c_i"], np.float64)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    out = dict(
        nants=p.nants, nfreqs=p.nfreqs,
        basis_flat=np.concatenate([b.ravel() for b in p.basis]), basis_shapes=np.asarray([b.shape for b in p.basis]),
        grp_basis=p.grp_basis, grp_bl_start=p.grp_bl_start, bl_ant0=p.bl_ant0, bl_ant1=p.bl_ant1, bl_rowblk=p.bl_rowblk,
        data_r=p.data_r, data_i=p.data_i, wgts=p.wgts, sky_r=p.sky_r, sky_i=p.sky_i,
        g_r=start["g_r"], g_i=start["g_i"], c_r=start["c_r"], c_i=start["c_i"],
    )
    pri = R.prior_sums(ch["sky_model_r"], ch["sky_model_i"], ch["wgts"])
    out["prior_r"], out["prior_i"] = pri
    for tag, priors in (("plain", (None, None)), ("sum", pri)):
        loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(start["g_r"], start["g_i"], fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1, *priors)
        out[f"{tag}_loss"] = loss
        out[f"{tag}_gg_r"], out[f"{tag}_gg_i"] = gg_r, gg_i
        out[f"{tag}_gc_r"] = problem.coeffs_from_chunks(p, gf_r)
        out[f"{tag}_gc_i"] = problem.coeffs_from_chunks(p, gf_i)
    kw = dict(data_r=ch["data_r"], data_i=ch["data_i"], wgts=ch["wgts"], fg_comps=ch["fg_comps"], corr_inds=ch["corr_inds"],
              maxsteps=10, learning_rate=1e-2, sky_model_r=ch["sky_model_r"], sky_model_i=ch["sky_model_i"])
    for tag, extra in (
        ("adam_plain", dict(optimizer="Adam", model_regularization=None)),
        ("adamax_sum", dict(optimizer="Adamax", model_regularization="sum")),
        ("adam_freeze", dict(optimizer="Adam", model_regularization=None, freeze_model=True)),
        ("adam_usemin", dict(optimizer="Adam", model_regularization=None, use_min=True, learning_rate=1e-1)),
    ):
        k = dict(kw, **extra)
        res = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, **k)
        out[f"{tag}_loss_hist"] = np.asarray(res[4]["loss"])
        out[f"{tag}_g_r"], out[f"{tag}_g_i"] = res[0], res[1]
        out[f"{tag}_c_r"] = problem.coeffs_from_chunks(p, res[2])
        out[f"{tag}_c_i"] = problem.coeffs_from_chunks(p, res[3])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fit_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


OPT_RUNS = {  # the rest of OPTIMIZERS (calibration.py:17-27): 8 recorded steps each, plain loss
    "sgd_nesterov": ("SGD", dict(learning_rate=5e-2, momentum=0.9, nesterov=True)),
    "rmsprop_momentum": ("RMSprop", dict(learning_rate=1e-2, momentum=0.5)),
    "adagrad": ("Adagrad", dict(learning_rate=5e-2)),
    "adadelta": ("Adadelta", dict(learning_rate=1.0)),
    "nadam": ("Nadam", dict(learning_rate=1e-2)),
    "ftrl_l1_l2": ("Ftrl", dict(learning_rate=0.5, l1_regularization_strength=1e-4, l2_regularization_strength=1e-3)),
    "lamb_decay": ("LAMB", dict(learning_rate=2e-2, weight_decay_rate=1e-3)),
}


def main_optimizers():
    """tests/golden/optimizers_small.npz: the same problem (read back from fit_small.npz's generator), the optimizers the first
    fixture does not cover, and the graph functions (calibration.py:1587-1656) on the chunk tensors: per-chunk sums of
    fg_model / data_model outputs, mse per chunk, mse_chunked, mse_chunked_sum_regularized."""
    # the problem AS STORED in fit_small.npz (the frozen fixture; regenerating it would move its basis in the 15th digit)
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fit_small.npz")))
    basis, off = [], 0
    for shp in g["basis_shapes"]:
        basis.append(g["basis_flat"][off : off + int(shp[0] * shp[1])].reshape(shp))
        off += int(shp[0] * shp[1])
    p = problem.FitProblem(nants=int(g["nants"]), nfreqs=int(g["nfreqs"]), basis=basis, grp_basis=g["grp_basis"], grp_bl_start=g["grp_bl_start"],
                           bl_ant0=g["bl_ant0"], bl_ant1=g["bl_ant1"], bl_rowblk=g["bl_rowblk"], data_r=g["data_r"], data_i=g["data_i"], wgts=g["wgts"],
                           sky_r=g["sky_r"], sky_i=g["sky_i"])
    start = {k: g[k] for k in ("g_r", "g_i", "c_r", "c_i")}
    ch = problem.chunks_from_problem(p)
    fg_r = problem.coeffs_to_chunks(p, start["c_r"], np.float64)
    fg_i = problem.coeffs_to_chunks(p, start["c_i"], np.float64)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    out = {}
    kw = dict(data_r=ch["data_r"], data_i=ch["data_i"], wgts=ch["wgts"], fg_comps=ch["fg_comps"], corr_inds=ch["corr_inds"], maxsteps=8, tol=0.0)
    for tag, (name, okw) in OPT_RUNS.items():
        res = R.fit_gains_and_foregrounds(start["g_r"], start["g_i"], fg_r, fg_i, optimizer=name, **dict(kw, **okw))
        out[f"{tag}_loss_hist"] = np.asarray(res[4]["loss"])
        out[f"{tag}_g_r"], out[f"{tag}_g_i"] = res[0], res[1]
        out[f"{tag}_c_r"] = problem.coeffs_from_chunks(p, res[2])
        out[f"{tag}_c_i"] = problem.coeffs_from_chunks(p, res[3])
    n = len(ch["fg_comps"])
    out["nchunks"] = n
    for c in range(n):
        vr, vi = R.fg_model(fg_r[c], fg_i[c], ch["fg_comps"][c])
        mr, mi = R.data_model(start["g_r"], start["g_i"], fg_r[c], fg_i[c], ch["fg_comps"][c], a0[c], a1[c])
        out[f"chunk{c}_fg_model"] = np.asarray([vr.sum(), vi.sum(), np.abs(vr).sum(), np.abs(vi).sum()])
        out[f"chunk{c}_data_model_r"], out[f"chunk{c}_data_model_i"] = mr, mi
        out[f"chunk{c}_mse"] = R.mse(mr, mi, ch["data_r"][c], ch["data_i"][c], ch["wgts"][c])
    pri = R.prior_sums(ch["sky_model_r"], ch["sky_model_i"], ch["wgts"])
    args = (start["g_r"], start["g_i"], fg_r, fg_i, ch["fg_comps"], n, ch["data_r"], ch["data_i"], ch["wgts"], a0, a1)
    out["mse_chunked"] = R.mse_chunked(*args)
    out["mse_chunked_sum_regularized"] = R.mse_chunked_sum_regularized(*args, *pri)
    out["mse_chunked_sum_regularized_shifted"] = R.mse_chunked_sum_regularized(*args, pri[0] + 0.25, pri[1] - 0.5)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "optimizers_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    # usage: make_golden.py fit | optimizers   (fit_small.npz is frozen: regenerate it only on purpose)
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    if which == "fit":
        main()
    elif which == "optimizers":
        main_optimizers()
    else:
        raise SystemExit("usage: make_golden.py fit | optimizers")
