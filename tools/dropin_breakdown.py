#!/usr/bin/env python3
"""End-to-end wall-clock split of the drop-in entry point at a BASELINE size:

    calibrate_and_model_dpss(SimpleUVData HERA-350 x 1024 channels x 1 time, maxsteps=N)

basis build / layout / tensorize / upload / initial coefficients / fit / model evaluation / write-back / residual, as
wall time of the host functions that do them (nested calls are attributed to the innermost timed function).  Prints one
JSON line.  Usage (GPU box): python tools/dropin_breakdown.py [--nants 350] [--nfreqs 1024] [--maxsteps 1000] [--host-only]
``--host-only`` stops in front of the first solver call (no GPU needed): the set-up part of the split."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from calamity_amd import batched, cal_utils, calibration, modeling, solver as solver_mod, synthetic  # noqa: E402
from calamity_amd.uvcompat import SimpleUVData  # noqa: E402

TIMES, CALLS = {}, {}
_LOCAL = threading.local()  # (one stack of open timers per thread: the batches' host stages run on threads of their own)


def timed(module, name, label=None):
    label = label or name
    fn = getattr(module, name)

    def wrapper(*a, **k):
        stack = _LOCAL.__dict__.setdefault("stack", [])
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            inner = stack.pop()
            TIMES[label] = TIMES.get(label, 0.0) + dt - inner
            CALLS[label] = CALLS.get(label, 0) + 1
            if stack:
                stack[-1] += dt

    setattr(module, name, wrapper)


class StopBeforeGpu(Exception):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nants", type=int, default=350)
    ap.add_argument("--nfreqs", type=int, default=1024)
    ap.add_argument("--maxsteps", type=int, default=1000)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reg", default="sum", help="model_regularization: sum (the Python API default) | post_hoc | none")
    ap.add_argument("--host-only", action="store_true")
    ap.add_argument("--ntimes", type=int, default=1)
    ap.add_argument("--verbose", action="store_true", help="the entry point's progress lines (time-stamped) on stderr-free stdout lines that do not start with {")
    ap.add_argument("--loop", action="store_true", help="the sequential time loop (batch_slices=False) instead of the batched call")
    ap.add_argument("--second-call", action="store_true", help="time the SECOND call of the process (a short first call runs before it)")
    args = ap.parse_args()
    dtype = np.float32 if args.dtype == "f32" else np.float64
    rng = np.random.default_rng(0)
    antpos = synthetic.hex_positions(args.nants)
    antpairs = [(i, j) for i in range(args.nants) for j in range(i + 1, args.nants)]
    freqs = np.linspace(100e6, 200e6, args.nfreqs, endpoint=False)
    t_build = time.perf_counter()
    uvd = SimpleUVData(antpos, antpairs, freqs, [2458000.0 + 0.01 * t for t in range(args.ntimes)])
    # smooth foregrounds (a few point sources) times random gains, 5 % flags: the content only matters for convergence
    a1, a2 = np.asarray(uvd.ant_1_array), np.asarray(uvd.ant_2_array)
    bvec = antpos[a2] - antpos[a1]
    vis = np.zeros((uvd.Nblts, args.nfreqs), dtype=np.complex128)
    for l, m, s in ((0.1, 0.05, 1.0), (-0.3, 0.2, 0.6), (0.45, -0.4, 0.8)):
        tau = (bvec[:, 0] * l + bvec[:, 1] * m) / 299792458.0
        vis += s * np.exp(-2j * np.pi * tau[:, None] * freqs[None, :])
    g = 1.0 + 0.05 * (rng.standard_normal((args.nants, args.nfreqs)) + 1j * rng.standard_normal((args.nants, args.nfreqs)))
    vis *= g[a1] * np.conj(g[a2])
    uvd.data_array[:] = vis.reshape(uvd.data_array.shape)
    uvd.flag_array[:] = (rng.random(uvd.flag_array.shape) < 0.05)
    t_build = time.perf_counter() - t_build

    timed(modeling, "yield_pbl_dpss_model_comps", "basis_build (DPSS blocks + redundancy grouping)")
    timed(calibration, "tensorize_fg_model_comps_dict", "layout (groups -> ragged problem)")
    timed(calibration, "tensorize_data", "tensorize_data (data, sky model -> rows)")
    timed(calibration, "tensorize_gains", "tensorize_gains")
    timed(calibration, "_init_coeffs", "initial coefficients (device A^T d + download)")
    timed(calibration, "_insert_model_rows", "write-back (model rows -> UVData)")
    timed(calibration, "insert_gains_into_uvcal", "write-back (gains -> UVCal)")
    timed(calibration, "fit_gains_and_foregrounds", "fit wrapper (flatten, set_params, history)")
    timed(calibration, "coeffs_to_chunks", "coefficient re-chunking")
    timed(calibration, "coeffs_from_chunks", "coefficient re-chunking")
    timed(cal_utils, "apply_gains", "apply_gains (sky model, model with gains)")
    timed(cal_utils, "blank_uvcal_from_uvdata", "blank_uvcal_from_uvdata")
    timed(calibration, "_blank_copy", "blank model container")
    timed(calibration, "_finish_outputs", "residual = data - gains x model, flags (calibration.py:1322-1331)")
    timed(batched, "replicate_slices", "batched problem description (replicate_slices)")
    H = solver_mod.HipFitSolver
    if args.host_only:
        def stop(*a, **k):
            raise StopBeforeGpu()
        H.__init__ = stop
    else:
        timed(H, "set_problem", "upload basis + layout tables (set_problem)")
        timed(H, "set_data", "upload data / weights (set_data)")
        timed(H, "set_params", "upload parameters")
        timed(H, "run", "FIT: device train steps (cal_solver_run)")
        timed(H, "run_slices", "FIT: device train steps (cal_solver_run)")
        timed(H, "init_coeffs", "initial coefficients (device A^T d + download)")
        timed(H, "model", "model evaluation A c (device) + download")
        timed(H, "get_params", "download parameters")
    kw = dict(maxsteps=args.maxsteps, tol=0.0, optimizer="Adam", learning_rate=1e-2, dtype=dtype,
              model_regularization=None if args.reg == "none" else args.reg)
    if args.loop:
        kw["batch_slices"] = False
    if args.verbose:
        kw["verbose"] = True
    if args.reg == "none":
        kw["sky_model"] = uvd  # the reference needs a sky model when there is no regularisation to build one for
    if args.second_call:
        # what every call after the first costs in one process (a night of files on one array and band): the DPSS blocks are kept by
        # modeling's cross-call cache, the HIP runtime is up -- the FIRST call runs here, untimed, and the split below is the second's
        calibration.calibrate_and_model_dpss(uvd, **dict(kw, maxsteps=2))
        TIMES.clear()
        CALLS.clear()
    t0 = time.perf_counter()
    nsteps = None
    try:
        model, resid, gains, hist = calibration.calibrate_and_model_dpss(uvd, **kw)
        nsteps = len(hist[0][0]["loss"])
        rms = lambda x: float(np.sqrt(np.mean(np.abs(x) ** 2)))  # noqa: E731
        quality = dict(rms_data=rms(uvd.data_array), rms_resid=rms(resid.data_array), loss_first=float(hist[0][0]["loss"][0]),
                       loss_last=float(hist[0][0]["loss"][-1]))
    except StopBeforeGpu:
        quality = None
    total = time.perf_counter() - t0
    accounted = sum(TIMES.values())
    out = dict(call="second call of the process (DPSS blocks cached, runtime up)" if args.second_call else "first call of the process",
               workload=f"calibrate_and_model_dpss ({'time loop' if args.loop else 'batched slices'}): {args.nants} antennas, {len(antpairs)} baselines x {args.nfreqs} channels x {args.ntimes} time(s), "
                        f"{np.dtype(dtype).name}, Adam lr 1e-2, model_regularization={args.reg}, maxsteps={args.maxsteps}",
               total_s=total, recorded_steps=nsteps, host_only=args.host_only,
               split_s={k: round(v, 4) for k, v in sorted(TIMES.items(), key=lambda kv: -kv[1])},
               calls=CALLS, other_s=round(total - accounted, 4),
               other_is="select / deep copies of the UVData containers, residual arithmetic, python glue",
               synthetic_uvdata_build_s=round(t_build, 3), quality=quality)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
