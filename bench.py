#!/usr/bin/env python3
"""bench.py -- Adam steps/s (+ chi^2 evals/s) of the gain + foreground fitter on synthetic HERA-350 x 1024-channel
per-baseline DPSS visibilities, with the HBM roofline of the fused basis-streaming kernel and a CPU baseline.

  python bench.py --gpus N --steps K --warmup W

A "step" is one train step of /root/reference/calamity/calibration.py:663-668 (loss + all adjoints + optimizer update
of gains and coefficients) for every time slice of the job.  N = 1: one time slice on one GPU.  N > 1 (launched by
torch.distributed.run, one rank per GPU): N time slices, every slice's baselines sharded over the N ranks, one RCCL
all-reduce of the per-antenna gain gradients + loss scalars per slice-step -- per-GPU work is constant (weak scaling)
and value = N slice-steps per wall step.  Prints ONE JSON line on rank 0.

With --gpus N > 1 and no launcher around it (no WORLD_SIZE in the environment) bench.py starts the N ranks ITSELF -- fresh
child processes, one per GPU, before this process has made a single HIP or RCCL call -- and relays rank 0's line: a run
that asks for N GPUs never measures fewer.  The line carries `n_ranks_seen`: the number of ranks counted by an all-reduce of
ones INSIDE the library's exchange (cal_solver_comm_size), beside `n_gpus`, the launcher's claim.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
MFMA_F64_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: half the fp32 matrix rate (128 FLOP/clk/CU x 256 CUs x 2.4 GHz)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16: 32768 flops per 32 cycles and SIMD)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="hera350", choices=["hera350", "hera37", "tutorial"])
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"])
    ap.add_argument("--layout", default=None, choices=["stream", "shared"],
                    help="basis layout; default: stream for hera350 (the headline: every baseline owns its tiles), shared (the product default) for the small configurations")
    ap.add_argument("--optimizer", default="Adam")
    ap.add_argument("--reg", default="none", choices=["none", "sum"])
    ap.add_argument("--max-bls", type=int, default=None, help="bounded sample of the baselines (debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shared", action="store_true", help="skip the extra shared-layout (MFMA) measurement")
    ap.add_argument("--extras", action="store_true", help="also time the tutorial-notebook and the redundant-groups configurations (they launch "
                    "the same kernels at other sizes, so a rocprofv3 --stats summary of such a run no longer averages the headline launches alone)")
    ap.add_argument("--cpu-sample-bls", type=int, default=192)
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host", "gloo"],
                    help="the exchange between the ranks: RCCL over xGMI (one rank per GPU: the measurement), or the library's exchange hook over "
                         "host sockets (ranks may then SHARE a GPU: a functional rehearsal of the N-rank path on a one-GPU box, not a measurement; "
                         "'gloo' is the old name of 'host')")
    ap.add_argument("--split", default=None, choices=["slices", "groups"],
                    help="time the PRODUCT's own multi-device form instead of one process per GPU: ONE process drives --gpus devices through "
                         "calamity_amd.batched.SliceBatchFitter exactly as calibrate_and_model_dpss does (calibration._fit_slices_batched) -- "
                         "'slices': whole time slices per device, no exchange (the default of a call with at least as many batches as devices); "
                         "'groups': every slice's fitting groups shared over the devices, one RCCL all-reduce per step between threads of this "
                         "process.  The job is --gpus time slices either way; --split-workers-on-one-gpu lets the workers share device 0")
    ap.add_argument("--split-workers-on-one-gpu", action="store_true", help="(with --split) all workers on device 0: a functional rehearsal on a one-GPU box")
    ap.add_argument("--dist-rehearsal", type=int, default=0, metavar="T",
                    help="with ONE rank (plain or under torch.distributed.run --nproc-per-node 1): take the multi-rank code path anyway -- "
                         "T batched time slices, group partition, socket rendezvous, unique id broadcast, RCCL communicator of one rank, "
                         "grouped all-reduce every step -- so that path is exercised on a one-GPU box")
    return ap.parse_args()


def host_info():
    """CPU model and core counts of the box the CPU baselines run on (north_star: 'core count stated')."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # the CPU time this process is actually allowed: the cgroup quota (a GPU box hands one GPU's job 16 of the host's cores
    # while nproc still shows all of them; 256 OpenMP threads on such a share ran 13x slower than 16)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(round(q / per)))
        except (OSError, ValueError):
            pass
    share = quota if quota is not None else min(usable, 16)  # no quota visible: the documented share of a one-GPU box
    return dict(cpu_model=model, cpu_count=os.cpu_count(), affinity_cores=usable, cgroup_quota_cores=quota, usable_cores=min(usable, share))


def kernel_source_hash():
    """sha256 over the kernel sources: ties a committed PMC summary to the code it was measured on."""
    import hashlib

    h = hashlib.sha256()
    for name in ("fit_kernels.hpp", "dense_kernels.hpp", "split_kernels.hpp", "split2_kernels.hpp", "dense64_kernels.hpp", "multi_mfma_kernels.hpp", "calamity_hip.hip"):
        with open(os.path.join(ROOT, "calamity_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_sharded_job(config, rank, world, ntimes, reg=False, max_bls=None, per_slice=False):
    """This rank's share of a job of ``ntimes`` time slices: the same 1/world of every slice's baselines (balanced by
    basis bytes), batched into ONE problem (slice t keeps its own gains: antenna index + t * nants), so one solver and
    one all-reduce per step serve all slices.  Returns (problem, start, nants of one slice)."""
    from calamity_amd import distributed as D
    from calamity_amd import synthetic

    cache = {}
    sel = lambda nvec, gb: D.partition_groups(nvec, gb, np.ones(len(nvec)), world)[rank]  # noqa: E731
    cfg_seed = list(synthetic.CONFIGS).index(config)
    parts = []
    for t in range(ntimes):
        p_t, _, s_t = synthetic.make_config(config, seed=cfg_seed + 100 * t, data_seed=100000 * (t + 1) + rank, bl_sel=sel,
                                            operator_cache=cache, with_sky=reg, max_bls=max_bls)
        parts.append((p_t, s_t))
    prob, start = D.batch_time_slices(parts, per_slice=per_slice)
    return prob, start, parts[0][0].nants


def cpu_baseline(prob, start, dtype, optimizer, sample_bls, reg):
    """Reference-faithful CPU port (oracle/ref_numpy.py: the op sequence of calibration.py:1587-1609 on the
    zero-padded (nvecs, ngrps, nbls, nfreqs) tensor + hand adjoints + Keras Adam), timed on a bounded sample."""
    from calamity_amd import problem as P
    from oracle import ref_numpy as R

    nb = prob.nbls
    sel = np.linspace(0, nb - 1, min(sample_bls, nb)).astype(np.int64)
    coff = prob.grp_coff
    sub = P.FitProblem(
        nants=prob.nants, nfreqs=prob.nfreqs, basis=prob.basis, grp_basis=prob.grp_basis[sel],
        grp_bl_start=np.arange(len(sel) + 1, dtype=np.int32), bl_ant0=prob.bl_ant0[sel], bl_ant1=prob.bl_ant1[sel],
        bl_rowblk=prob.bl_rowblk[sel], data_r=prob.data_r[sel], data_i=prob.data_i[sel], wgts=prob.wgts[sel],
    )
    c_r = np.concatenate([start["c_r"][coff[g] : coff[g + 1]] for g in sel])
    c_i = np.concatenate([start["c_i"][coff[g] : coff[g + 1]] for g in sel])
    ch = P.chunks_from_problem(sub, dtype=dtype)
    fg_r = P.coeffs_to_chunks(sub, c_r, dtype)
    fg_i = P.coeffs_to_chunks(sub, c_i, dtype)
    a0, a1 = R.ant_inds_from_corr_inds(ch["corr_inds"])
    g_r, g_i = start["g_r"].astype(dtype), start["g_i"].astype(dtype)
    opt = R.OPTIMIZERS[optimizer](learning_rate=1e-2)

    def step():
        loss, gg_r, gg_i, gf_r, gf_i = R.loss_and_grads(g_r, g_i, fg_r, fg_i, ch["fg_comps"], ch["data_r"], ch["data_i"], ch["wgts"], a0, a1)
        opt.apply_gradients([(gg_r, g_r), (gg_i, g_i)] + list(zip(gf_r, fg_r)) + list(zip(gf_i, fg_i)))
        return loss

    step()
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 50):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    # the sample holds len(sel)/nb of the job's baselines; the full job costs nb/len(sel) as much per step
    full_rate = 1.0 / (dt * nb / len(sel))
    padded_bytes = ch["fg_comps"][0].nbytes
    return dict(
        value=full_rate, unit="steps/s", cores=1, kind="port",
        sample=f"{len(sel)} of {nb} baselines (evenly spaced), {n} Adam steps of the NumPy restatement on the zero-padded "
               f"({ch['fg_comps'][0].shape[0]} x {len(sel)} x 1 x {prob.nfreqs}) tensor ({padded_bytes / 1e6:.0f} MB, {np.dtype(dtype).name}); "
               f"{dt * 1e3:.1f} ms per sample step, scaled by {nb}/{len(sel)}",
    )


def cpu_baseline_strong(prob, start, dtype, optimizer, reg, same_updates=0):
    """Strong CPU baseline: the C / OpenMP restatement (oracle/ref_c.c) -- un-padded, forward and
    adjoint fused per baseline, unique basis blocks shared, every host core -- on the FULL job for a bounded time."""
    from oracle.ref_c import CRef

    cores = host_info()["usable_cores"]  # every core this process may run on
    c = CRef(prob, dtype, nthreads=cores)
    if reg:
        c.set_regularization("sum", float(np.sum(prob.sky_r * prob.wgts)), float(np.sum(prob.sky_i * prob.wgts)))
    state = [start["g_r"], start["g_i"], start["c_r"], start["c_i"]]
    mom, n = None, 0
    g_r, g_i, c_r, c_i, ls, mom = c.fit(*state, 1, optimizer=optimizer, learning_rate=1e-2)
    losses = [float(ls[0])]  # pre-update loss of update k (k = 0: the start parameters)
    t0 = time.perf_counter()
    # at least as many updates as the GPU run applied (so the loss after the same number of updates can be put beside
    # the GPU's), then up to ~10 s
    while n < max(2, same_updates) or (time.perf_counter() - t0 < 10.0 and n < 50):
        g_r, g_i, c_r, c_i, ls, mom = c.fit(g_r, g_i, c_r, c_i, 1, optimizer=optimizer, learning_rate=1e-2, moments=mom, t0=n + 1)
        losses.append(float(ls[0]))
        n += 1
    dt = (time.perf_counter() - t0) / n
    return dict(value=1.0 / dt, unit="steps/s", cores=cores, kind="port",
                loss_before_update=losses[: same_updates + 1],
                sample=f"full job, {n} Adam steps of the C/OpenMP restatement (ragged, fused, shared unique basis blocks, "
                       f"{np.dtype(dtype).name}) on {cores} threads; {dt * 1e3:.0f} ms per step")


def ranks_sum_int(grp, vals):
    return [int(v) for v in grp.all_reduce(np.asarray(vals, dtype=np.int64), "sum")]


def ranks_sum_float(grp, vals):
    return np.asarray(grp.all_reduce(np.asarray(vals, dtype=np.float64), "sum"))


def dense_traffic(config, dtype):
    """HBM bytes per gradient launch of the dense kernel from the committed counter summary (tools/prof_dense.sh +
    tools/dense_pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes) -- only if it was measured on these kernel sources."""
    path = os.path.join(ROOT, "profiles", f"pmc_{config}_{'f32' if dtype == np.float32 else 'f64'}_shared.json")
    if not os.path.exists(path):
        return None, None
    pmc = json.load(open(path))
    if pmc.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{os.path.relpath(path, ROOT)} is stale (measured on other kernel sources): not reported"
    return pmc.get("gradient_pass", {}).get("hbm_bytes"), os.path.relpath(path, ROOT)


def split_issued_flops(prob):
    """bf16 flops the split-bf16 dense kernel ISSUES per gradient launch (split2_kernels.hpp): six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16
    block of the fp32 product, blocks padded to 32 vectors (forward and adjoint: the body is instantiated per number of 32-vector tiles),
    panels to 16 and super-panels to 64 baselines."""
    nv = np.asarray([b.shape[1] for b in prob.basis])
    nbl = np.bincount(prob.grp_basis[np.repeat(np.arange(prob.ngrps), np.diff(prob.grp_bl_start))], minlength=len(nv))
    fpad = -(-prob.nfreqs // 128) * 128
    waves = -(-nbl // 64) * 4
    mfma_per_wave = (fpad // 64) * (24 * -(-nv // 32) + 24 * -(-nv // 32))  # per pair of 32-channel blocks: 2 x (2 NT steps x 6) forward, 2 x (NT tiles x 2 steps x 6) adjoint
    return float(np.sum(waves * mfma_per_wave)) * 2.0 * 32 * 32 * 16


def dense_rooflines(prob, tim, kernel_ms, dtype, config=None):
    """Both bounds of the dense (shared-layout) kernel from its measured launch duration: the matrix pipe for
    8 F sum nvec flops, and HBM for the algorithmic bytes with every DISTINCT basis block counted once."""
    f32 = dtype == np.float32
    split = f32 and tim.get("kernel_path") == "dense"
    traffic, traffic_src = dense_traffic(config, dtype) if config else (None, None)
    peak = MFMA_F32_PEAK_TFLOPS if f32 else MFMA_F64_PEAK_TFLOPS
    flops = tim["flops_per_launch"]
    tf = flops / (kernel_ms * 1e-3) / 1e12
    uniq_bytes = float(sum(b.size for b in prob.basis)) * np.dtype(dtype).itemsize
    bytes_unique = tim["algorithmic_bytes_per_launch"] - tim["basis_bytes_per_launch"] + uniq_bytes
    gbs = bytes_unique / (kernel_ms * 1e-3) / 1e9
    out = {
        "kernel": ("fused_dense_split2_kernel<GRAD> (six v_mfma_f32_32x32x16_bf16 per fp32 product block: split-bf16 operands, one operand image read row-wise and transposed)" if split else
                   "fused_dense_kernel<GRAD> (v_mfma_f32_32x32x2_f32)") if f32 else "fused_dense64_kernel<GRAD> (v_mfma_f64_16x16x4_f64)",
        "kernel_ms": kernel_ms,
        # the USEFUL flops (8 F sum nvec: an fp32 product per (channel, vector, re | im), forward and adjoint) against the fp32 matrix peak:
        # the yardstick every fp32 dense kernel of this repository is held to, whatever instruction it issues
        "roofline_mfma": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "flops_per_launch": flops,
                          "traffic": traffic, "traffic_source": traffic_src},
        "roofline_hbm_unique_basis": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_launch": bytes_unique},
    }
    if split:
        issued = split_issued_flops(prob) * (1.5 if tim["flops_per_launch"] > 8.5 * prob.nfreqs * prob.ncoeffs else 1.0)  # (regularised step: + a loss-only pass)
        out["roofline_bf16_issued"] = {"bound": "mfma", "achieved": issued / (kernel_ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": issued / (kernel_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "flops_per_launch": issued,
                                       "note": "bf16 flops issued (6 per useful fp32 flop + padding) against the dense bf16 peak"}
    return out


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks here (one fresh process per GPU; nothing in THIS process has touched
    the GPU yet), relay rank 0's JSON line, exit with the worst return code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    # a rank that dies must not leave the others waiting for it in a rendezvous or a collective: the first non-zero exit ends them all
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(pr.poll() is None for pr in procs):
        if any(pr.poll() not in (None, 0) for pr in procs):
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
            break
        time.sleep(0.2)
    rcs = [pr.wait() for pr in procs]
    reader.join(timeout=10)
    out0 = out0[0] if out0 else ""
    if any(rcs):
        sys.stderr.write(f"bench.py: rank return codes {rcs} (launched {args.gpus} ranks on GPUs 0..{args.gpus - 1} of this node; the same run under a "
                         f"launcher: python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port {port} "
                         f"bench.py {' '.join(sys.argv[1:])})\n")
        raise SystemExit(max(abs(rc) for rc in rcs) or 1)
    sys.stdout.write(out0)
    sys.stdout.flush()


def run_split(args):
    """--split: the product's multi-device forms, one process, a thread per device (see parse())."""
    import threading

    from calamity_amd import _lib, synthetic
    from calamity_amd.batched import SliceBatchFitter

    if int(os.environ.get("RANK", "0")) != 0:
        return  # (under a launcher: rank 0 drives every device; the other ranks have nothing to do)
    _lib.load()
    N = args.gpus
    ndev = _lib.device_count()
    if N > ndev and not args.split_workers_on_one_gpu:
        raise SystemExit(f"--gpus {N} --split {args.split}: this node has {ndev} GPU(s) visible (--split-workers-on-one-gpu for a functional rehearsal)")
    devices = [0] * N if args.split_workers_on_one_gpu else list(range(N))
    dtype = {"f32": np.float32, "f64": np.float64, None: np.float32}[args.dtype]
    layout = args.layout or "shared"
    cache = {}
    cfg_seed = list(synthetic.CONFIGS).index(args.config)
    parts = [synthetic.make_config(args.config, seed=cfg_seed + 100 * t, data_seed=100000 * (t + 1), operator_cache=cache, max_bls=args.max_bls)
             for t in range(N)]
    prob = parts[0][0]
    cat = lambda key: np.concatenate([getattr(p[0], key) for p in parts])  # noqa: E731
    t_setup = time.perf_counter()
    if args.split == "groups":
        fitters = [SliceBatchFitter(prob, N, dtype=dtype, layout=layout, devices=devices, communicator_of_one=(N == 1))]
        fitters[0].set_data(cat("data_r"), cat("data_i"), cat("wgts"))
        fitters[0].set_params(np.concatenate([p[2]["g_r"] for p in parts]), np.concatenate([p[2]["g_i"] for p in parts]),
                              np.concatenate([p[2]["c_r"] for p in parts]), np.concatenate([p[2]["c_i"] for p in parts]))
    else:
        fitters = []
        for t, d in enumerate(devices):
            f = SliceBatchFitter(prob, 1, dtype=dtype, layout=layout, devices=[d])
            f.set_data(parts[t][0].data_r, parts[t][0].data_i, parts[t][0].wgts)
            f.set_params(parts[t][2]["g_r"], parts[t][2]["g_i"], parts[t][2]["c_r"], parts[t][2]["c_i"])
            fitters.append(f)
    for f in fitters:
        f.set_optimizer(args.optimizer, learning_rate=1e-2)
    t_setup = time.perf_counter() - t_setup

    def each(fn):  # every fitter on a thread of its own, as _fit_slices_batched's fit threads
        out, errs = [None] * len(fitters), []

        def work(i):
            try:
                out[i] = fn(fitters[i])
            except BaseException as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(fitters))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        return out

    def sync():
        for f in fitters:
            for sv in f.solvers:
                sv.synchronize()

    if args.warmup > 0:
        each(lambda f: f.run_slices(args.warmup, record=False, tol=0.0))
    sync()
    t0 = time.perf_counter()
    res = each(lambda f: f.run_slices(args.steps, record=True, tol=0.0))
    sync()
    dt = time.perf_counter() - t0
    losses = np.sum([np.sum([r[0] for r in rr], axis=0) for rr in res], axis=0)
    tim = fitters[0].timing_get()
    out = {
        "metric": "Adam steps/sec, HERA-350 1024ch DPSS; the product's multi-device split", "value": N * args.steps / dt, "unit": "slice-steps/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if dtype == np.float32 else "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: {prob.nants} antennas, {prob.nbls} baselines x {prob.nfreqs} channels per time slice, {N} time slices, "
                               f"optimizer {args.optimizer} lr 1e-2", "layout": layout, "kernel_path": tim["kernel_path"],
                   "parallelism": (f"ONE process, a thread per device, device_split='slices': whole time slices on {N} devices, no exchange"
                                   if args.split == "slices" else
                                   f"ONE process, a thread per device, device_split='groups': every slice's fitting groups shared over {N} devices, "
                                   f"one RCCL all-reduce of the gain gradients + loss scalars per step")
                                  + (" [all workers on device 0: functional rehearsal, not a measurement]" if args.split_workers_on_one_gpu else "")},
        "extra": {"loss_first": float(losses[0]), "loss_last": float(losses[-1]), "setup_s": t_setup,
                  # (a collective: every worker takes part)
                  "ranks_the_exchange_spans": fitters[0]._each(lambda r, sv: sv.comm_size())[0] if args.split == "groups" else 1},
    }
    print(json.dumps(out))
    for f in fitters:
        f.close()


def main():
    args = parse()
    if args.split:
        return run_split(args)
    if args.layout is None:
        args.layout = "stream" if args.config == "hera350" else "shared"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")
    from calamity_amd import _lib, synthetic
    from calamity_amd.solver import HipFitSolver, comm_unique_id

    # (no PyTorch anywhere in this file: the ranks meet over calamity_amd.rendezvous -- plain sockets on MASTER_ADDR:MASTER_PORT --
    # and the data path is RCCL inside the library)
    if args.transport == "gloo":
        args.transport = "host"  # (the old name of the host-side stand-in for ranks that share a GPU)
    _lib.load()
    ndev = _lib.device_count()
    if world > ndev and args.transport == "rccl":  # every rank sees this and leaves at once: none waits for a peer that cannot exist
        raise SystemExit(f"--gpus {args.gpus}: this node has {ndev} GPU(s) visible; one rank per GPU is the contract "
                         "(--transport host lets ranks share a GPU for a functional rehearsal)")

    dtype = {"f32": np.float32, "f64": np.float64, None: np.float64 if args.config == "hera37" else np.float32}[args.dtype]
    dist = None
    if args.dist_rehearsal and world != 1:
        raise SystemExit("--dist-rehearsal is a one-rank run")
    sharded = world > 1 or args.dist_rehearsal > 0  # the multi-rank code path (also taken by the one-rank rehearsal)
    if sharded:
        from calamity_amd.rendezvous import SocketGroup  # the RCCL id, a barrier, the max of the wall time: sockets, no torch

        dist = SocketGroup(rank=rank, world=world)

    t_setup = time.perf_counter()
    ntimes = world if world > 1 else max(args.dist_rehearsal, 1)
    solvers = []
    if not sharded:
        prob, truth, start = synthetic.make_config(args.config, max_bls=args.max_bls, with_sky=args.reg == "sum")
        full_nbls, full_ncoeffs, full_nants = prob.nbls, prob.ncoeffs, prob.nants
        s = HipFitSolver(dtype=dtype, device=0)
        s.set_problem(prob, layout=args.layout)
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        solvers.append(s)
    else:
        # N time slices; this rank owns the same 1/N of the baselines of every slice -> one all-reduce per step
        # (one loop state per time slice, as calibrate_and_model_tensor fits them: every slice records its own loss)
        prob, start, full_nants = build_sharded_job(args.config, rank, world, ntimes, reg=args.reg == "sum", max_bls=args.max_bls, per_slice=True)
        truth = None
        tot = ranks_sum_int(dist, [prob.nbls // ntimes, prob.ncoeffs // ntimes])
        full_nbls, full_ncoeffs = tot
        s = HipFitSolver(dtype=dtype, device=local_rank % ndev)
        # communicator first: set_problem then agrees the kernel path and the steps per host synchronisation over the ranks
        if args.transport == "rccl":
            uid = dist.broadcast(comm_unique_id() if rank == 0 else None, src=0)
            s.comm_init(uid, rank, world)
        else:
            # functional stand-in (ranks that share a GPU cannot form an RCCL communicator): the library's staging buffer, reduced in place
            s.set_exchange_hook(lambda arr, op: dist.all_reduce_inplace(arr, "min" if op == "min" else "sum"), rank, world)
        s.set_problem(prob, layout=args.layout)
        s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        solvers.append(s)
    # the ranks the library's exchange really spans (an all-reduce of ones over its communicator)
    n_ranks_seen = solvers[0].comm_size()
    if n_ranks_seen != world:
        raise SystemExit(f"the exchange spans {n_ranks_seen} ranks, the launcher started {world}")
    for s in solvers:
        if args.reg == "sum":
            if sharded:  # one pair of priors per time slice, summed over the ranks' shares
                nb = prob.nbls // ntimes
                pri = np.concatenate([[float(np.sum((prob.sky_r * prob.wgts)[t * nb : (t + 1) * nb])) for t in range(ntimes)],
                                      [float(np.sum((prob.sky_i * prob.wgts)[t * nb : (t + 1) * nb])) for t in range(ntimes)]])
                pri = ranks_sum_float(dist, pri)
                s.set_regularization("sum", pri[:ntimes], pri[ntimes:])
            else:
                s.set_regularization("sum", float(np.sum(prob.sky_r * prob.wgts)), float(np.sum(prob.sky_i * prob.wgts)))
        s.set_optimizer(args.optimizer, learning_rate=1e-2)
    t_setup = time.perf_counter() - t_setup

    def sync():
        for s in solvers:
            s.synchronize()
        if dist is not None:
            dist.barrier()

    def run_steps(n, record):
        if sharded:  # the recorded loss of the job: the sum of its slices' losses
            res = [s.run_slices(n, record=record, tol=0.0) for s in solvers][0]
            return np.sum([r[0] for r in res], axis=0) if record else []
        return [s.run(n, record=record, tol=0.0)[0] for s in solvers][0]

    run_steps(args.warmup, False) if args.warmup > 0 else None
    # Problems whose step is tens of microseconds are replayed from a hipGraph (two launches per step); HIP events around
    # every fused pass would force launch-by-launch issue, so for those the kernel duration is measured in a second pass
    # right after the timed one.  The headline keeps its events inside the timed region (they cost < 0.1 % of a 4.8-ms step).
    # (the shared-basis dense step is 0.75 ms: two events per pass cost it 14 % -- 0.86 ms -- so it is measured like the small problems)
    launch_bound = args.config != "hera350" or (args.layout == "shared" and not sharded)
    for s in solvers:
        s.timing_enable(not launch_bound)
    if launch_bound and not sharded and args.steps >= 16:
        run_steps(args.steps, True)  # more warm-up, of the timed call's own shape: the library captures a step graph on first use of a shape (small problems; large ones from 256 steps on)
    sync()
    t0 = time.perf_counter()
    timed_losses = run_steps(args.steps, True)
    sync()
    dt = time.perf_counter() - t0
    if launch_bound:
        for s in solvers:
            s.timing_enable(True)
        run_steps(min(args.steps, 200), False)
        sync()
    if dist is not None:
        dt = float(dist.all_reduce([dt], "max")[0])
    tim = solvers[0].timing_get()
    for s in solvers:
        s.timing_enable(False)
    # chi^2 evaluations per second (forward only), untimed part of the contract
    sync()
    t1 = time.perf_counter()
    nev = max(3, args.steps // 2)
    for _ in range(nev):
        for s in solvers:
            s.eval_loss()
    sync()
    chi2_rate = nev * ntimes / (time.perf_counter() - t1)

    # the SHARED layout of the same workload (baselines of one delay alias ONE basis block; dense MFMA path): measured
    # in the same run and reported beside the headline, against its own bounds
    shared = None
    if not sharded and args.layout == "stream" and not args.no_shared and args.reg == "none":
        s2 = HipFitSolver(dtype=dtype, device=0)
        s2.set_problem(prob, layout="shared")
        s2.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        s2.set_optimizer(args.optimizer, learning_rate=1e-2)
        def rate_and_kernel(sv):
            # steps per second from an un-instrumented run (as the product runs: two events per pass cost a 0.75-ms step 14 %), the kernel's
            # duration from a second run with HIP events around every pass
            sv.run(max(args.warmup, 1), record=False, tol=0.0)
            sv.run(args.steps, record=True, tol=0.0)  # (a call of the timed shape: the library captures its step graph on first use)
            sv.synchronize()
            t2 = time.perf_counter()
            sv.run(args.steps, record=True, tol=0.0)
            sv.synchronize()
            dt_plain = time.perf_counter() - t2
            sv.timing_enable(True)
            sv.synchronize()
            t2 = time.perf_counter()
            sv.run(args.steps, record=True, tol=0.0)
            sv.synchronize()
            dt_events = time.perf_counter() - t2
            tim_ = sv.timing_get()
            sv.timing_enable(False)
            return dt_plain, dt_events, tim_, tim_["total_ms"] / max(tim_["launches"], 1)

        dt2, dt2e, tim2, k2 = rate_and_kernel(s2)
        shared = dense_rooflines(prob, tim2, k2, dtype, args.config if args.max_bls is None else None)
        shared.update(steps_per_s=args.steps / dt2, ms_per_step=dt2 / args.steps * 1e3, ms_per_step_with_events=dt2e / args.steps * 1e3,
                      device_memory_GB=s2.memory_bytes() / 1e9)
        s2.close()
        if dtype == np.float32 and tim2["kernel_path"] == "dense":
            # the kernels this one replaced, on the same problem in the same run: the fp32 MFMA kernel (kernel_path "dense_f32") and the first
            # split-bf16 kernel with its two packed operand streams ("dense_split1")
            for path, key, label in (("dense_f32", "previous_f32_kernel", "fused_dense_kernel<GRAD> (v_mfma_f32_32x32x2_f32)"),
                                     ("dense_split1", "first_split_kernel", "fused_dense_split_kernel<GRAD> (split-bf16, two packed operand streams through an LDS ring)")):
                s2 = HipFitSolver(dtype=dtype, device=0)
                s2.set_problem(prob, layout="shared", kernel_path=path)
                s2.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
                s2.set_optimizer(args.optimizer, learning_rate=1e-2)
                dt2, dt2e, tim2, k2 = rate_and_kernel(s2)
                shared[key] = {"kernel": label, "kernel_ms": k2, "steps_per_s": args.steps / dt2, "ms_per_step": dt2 / args.steps * 1e3,
                               "ms_per_step_with_events": dt2e / args.steps * 1e3}
                s2.close()

    # what EVERY RANK of the driver's `--gpus 8` run executes per step, without the exchange: 8 time slices x that rank's 1/8 of the
    # baselines, the slices sharing basis tiles (fused_multi_mfma_kernel).  All eight shares are built and timed here, one after the
    # other on this GPU: the 8-GPU step is the slowest rank's step (+ the exchange), so the table says in advance what the scaling
    # run can reach and whether the partition (distributed.partition_groups: groups dealt round-robin) is balanced.
    rank_job = None
    if (rank == 0 and not sharded and args.layout == "stream" and not args.no_shared and args.reg == "none" and args.config == "hera350"
            and args.max_bls is None and dtype == np.float32):
        rows = []
        for r8 in range(8):
            rp, rstart, _ = build_sharded_job(args.config, r8, 8, 8)
            s3 = HipFitSolver(dtype=dtype, device=0)
            s3.set_problem(rp, layout="stream")
            s3.set_params(rstart["g_r"], rstart["g_i"], rstart["c_r"], rstart["c_i"])
            s3.set_optimizer(args.optimizer, learning_rate=1e-2)
            s3.run(max(min(args.warmup, 3), 1), record=False, tol=0.0)
            s3.timing_enable(True)
            s3.synchronize()
            t3 = time.perf_counter()
            s3.run(args.steps, record=True, tol=0.0)
            s3.synchronize()
            dt3 = time.perf_counter() - t3
            tim3 = s3.timing_get()
            k3 = tim3["total_ms"] / max(tim3["launches"], 1)
            rows.append({"rank": r8, "baselines_per_slice": rp.nbls // 8, "sum_nvec_per_slice": rp.ncoeffs // 8,
                         "ms_per_step": dt3 / args.steps * 1e3, "kernel_ms": k3,
                         "algorithmic_bytes_per_launch": tim3["algorithmic_bytes_per_launch"],
                         "achieved_GBps": tim3["algorithmic_bytes_per_launch"] / (k3 * 1e-3) / 1e9 if k3 > 0 else None,
                         "device_memory_GB": s3.memory_bytes() / 1e9})
            s3.close()
            del rp, rstart
        ms = [r["ms_per_step"] for r in rows]
        rank_job = {"what": "every rank's share of an 8-GPU job (8 time slices x 1/8 of each slice's baselines, fitting groups dealt round-robin over the "
                            "ranks; the slices share basis tiles), timed one after the other on this ONE GPU, no exchange",
                    "kernel": "fused_multi_mfma_kernel<MODE_GRAD> (v_mfma_f32_16x16x4_f32, 16 right-hand sides per tile)",
                    "shares": rows, "ms_per_step_max": max(ms), "ms_per_step_min": min(ms), "spread_max_over_min": max(ms) / min(ms),
                    "predicted_8gpu_slice_steps_per_s_without_exchange": 8.0 / (max(ms) * 1e-3),
                    # back-compatible summary of the slowest share
                    "ms_per_step": max(ms), "kernel_ms": max(r["kernel_ms"] for r in rows), "slice_steps_per_s": 8.0 / (max(ms) * 1e-3)}

    # measured streaming peaks of this box and, for orientation, the only configuration the
    # reference publishes a rate for (tutorial notebook: 15 antennas, 105 baselines x 200 channels, Adamax, 61.77 steps/s
    # on a P100) -- single-GPU runs only, after the solvers above have released their memory
    peaks, tutorial, redundant = None, None, None
    if rank == 0 and not sharded and not args.no_shared:
        for s_ in solvers:
            s_.synchronize()
        try:
            rd, cp = _lib.stream_peak(0, 4 << 30, 5)
            peaks = {"read_GBps": rd, "copy_GBps": cp, "how": "cal_device_stream_peak: 4 GiB, best of 5, 16-byte non-temporal loads / plain copy"}
            peaks["busy_shader_clock_MHz"] = _lib.busy_clock_mhz(0)
        except Exception as e:  # noqa: BLE001 -- a probe must not take the benchmark down
            peaks = {"error": str(e)}
    if rank == 0 and not sharded and args.extras:
        tp, _, tstart = synthetic.make_problem(15, 200, f0=100e6, df=100e3, seed=0)
        ts = HipFitSolver(dtype=np.float32)
        ts.set_problem(tp, layout="shared")
        ts.set_params(tstart["g_r"], tstart["g_i"], tstart["c_r"], tstart["c_i"])
        ts.set_optimizer("Adamax", learning_rate=1e-2)
        ts.run(200, record=False)
        best = 0.0
        for _ in range(3):  # (the first repetition behind the big runs of this process has come out at half the rate)
            ts.synchronize()
            t0 = time.perf_counter()
            ts.run(5000, record=True, tol=0.0)
            ts.synchronize()
            best = max(best, 5000 / (time.perf_counter() - t0))
        tutorial = {"steps_per_s": best, "config": f"{tp.nants} antennas, {tp.nbls} baselines x 200 channels, Adamax lr 1e-2, fp32",
                    "reference_published_steps_per_s": 61.77, "reference_hardware": "Tesla P100, TensorFlow eager (examples/Calamity_Tutorial.ipynb:1178)"}
        ts.close()
        # BASELINE config 5: the same array with every redundant set as ONE fitting group (shared coefficients)
        if args.config == "hera350" and args.max_bls is None:
            rp, rstart = synthetic.merge_redundant_groups(prob, truth, start)
            rs = HipFitSolver(dtype=dtype)
            rs.set_problem(rp, layout="shared")
            rs.set_params(rstart["g_r"], rstart["g_i"], rstart["c_r"], rstart["c_i"])
            rs.set_optimizer(args.optimizer, learning_rate=1e-2)
            rs.run(5, record=False)
            rs.synchronize()
            t0 = time.perf_counter()
            rs.run(200, record=True, tol=0.0)
            rs.synchronize()
            redundant = {"steps_per_s": 200 / (time.perf_counter() - t0), "config": f"{rp.ngrps} fitting groups (redundant sets, up to "
                         f"{int(np.diff(rp.grp_bl_start).max())} baselines each) over the same {rp.nbls} baselines, sum nvec = {rp.ncoeffs}"}
            rs.close()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = args.steps * ntimes / dt
        kern_ms = tim["total_ms"] / max(tim["launches"], 1)
        achieved = tim["algorithmic_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.config}_{'f32' if dtype == np.float32 else 'f64'}_{args.layout}.json")
        if not sharded and args.max_bls is None and args.reg == "none" and os.path.exists(pmc_path):
            # HBM bytes per launch of the same kernel on the same workload, from separate rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE passes (tools/pmc_summary.py; gfx950 corrections of MI355X_MICROARCH.md applied there).  The
            # summary records the hash of the kernel sources it was measured on: a stale one is not reported.
            pmc = json.load(open(pmc_path))
            if pmc.get("kernel_source_hash") == kernel_source_hash():
                for k, v in pmc["kernels"].items():
                    if "fused_basis_kernel<float, 1, false" in k or "fused_basis_kernel<double, 1, false" in k:  # the gradient pass, no regulariser
                        traffic, traffic_src = v["hbm_bytes"], os.path.relpath(pmc_path, ROOT)
            else:
                traffic_src = f"{os.path.relpath(pmc_path, ROOT)} is stale (measured on other kernel sources): not reported"
        cache_resident = tim["algorithmic_bytes_per_launch"] < 200e6  # the working set sits in the 256 MB Infinity Cache
        hbm_roofline = {
            "bound": "hbm" if not cache_resident else "launch latency (a {:.0f} MB working set lives in the 256 MB Infinity Cache: 'achieved' is cache "
                     "bandwidth, not an HBM figure; the step is bounded by the launches it takes)".format(tim["algorithmic_bytes_per_launch"] / 1e6),
            "kernel": ("fused_basis_kernel<MODE_GRAD>" if not (sharded and ntimes > 1 and args.layout == "stream") else
                       ("fused_multi_mfma_kernel<MODE_GRAD> (the slices of a rank share basis tiles: 16 right-hand sides per tile on "
                        "v_mfma_f32_16x16x4_f32; basis bytes counted once per unique tile)" if dtype == np.float32 else
                        "fused_multi_mfma_kernel<double, MODE_GRAD> (the slices of a rank share basis tiles: 16 right-hand sides per tile on "
                        "v_mfma_f64_16x16x4_f64; basis bytes counted once per unique tile)")),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel_ms": kern_ms,
            "algorithmic_bytes_per_launch": tim["algorithmic_bytes_per_launch"],
            "peak_measured": peaks,
            "frac_of_measured_read_peak": (achieved / peaks["read_GBps"]) if peaks and "read_GBps" in peaks else None,
        }
        if tim["kernel_path"] == "dense":
            # the shared layout's dense kernel is bound by the matrix pipe, not by HBM: lead with that roofline
            d = dense_rooflines(prob, tim, kern_ms, dtype, args.config if (not sharded and args.max_bls is None and args.reg == "none") else None)
            roofline = dict(d["roofline_mfma"], kernel=d["kernel"], kernel_ms=kern_ms, peak_measured=peaks,
                            hbm_unique_basis=d["roofline_hbm_unique_basis"])
        else:
            roofline = hbm_roofline
        out = {
            "metric": "Adam steps/sec (chi2 eval/sec in extra), " + {"hera350": "HERA-350", "hera37": "HERA-37", "tutorial": "tutorial-scale"}.get(args.config, args.config)
                      + f" {prob.nfreqs}ch DPSS; %HBM roofline",
            "value": value,
            "unit": "steps/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if dtype == np.float32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {full_nants} antennas, {full_nbls} baselines x {prob.nfreqs} channels, per-baseline DPSS "
                            f"(sum nvec = {full_ncoeffs}) per time slice, {ntimes} time slice(s), optimizer {args.optimizer} lr 1e-2, "
                            f"model_regularization {args.reg}",
                "layout": args.layout,
                "transport": args.transport if sharded else None,
                "parallelism": (f"every slice's baselines sharded over {world} ranks (one process each), one {'RCCL' if args.transport == 'rccl' else 'host-socket (exchange hook)'} all-reduce of the gain gradients + loss scalars per step"
                                + (" [one-rank rehearsal of the multi-rank path]" if world == 1 else "")) if sharded else "single GPU",
            },
            "roofline": roofline,
            "extra": {
                "chi2_evals_per_s": chi2_rate,
                # recorded (pre-update) losses of the timed steps: the steps did optimise; summed over the job's slices
                "loss_first": float(timed_losses[0]) if len(timed_losses) else None,
                "loss_last": float(timed_losses[-1]) if len(timed_losses) else None,
                "losses": [float(v) for v in timed_losses] if len(timed_losses) <= 64 else None,
                "updates_before_loss_last": args.warmup + len(timed_losses) - 1,
                "setup_s": t_setup,
                "device_memory_GB": solvers[0].memory_bytes() / 1e9 * len(solvers),
                "shared_layout": shared,
                "rank_of_8_job": rank_job,
                "tutorial_notebook_config": tutorial,
                "redundant_groups_config": redundant,
            },
        }
        if not args.no_cpu_baseline and not sharded:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(prob, start, dtype, args.optimizer, args.cpu_sample_bls, args.reg)
            nupd = args.warmup + args.steps - 1
            strong = cpu_baseline_strong(prob, start, dtype, args.optimizer, args.reg == "sum", same_updates=nupd if nupd <= 64 else 0)
            # the C restatement's loss after the same number of updates as the GPU's last recorded loss (outside the timed region)
            lb = strong.pop("loss_before_update")
            strong["loss_after_same_updates"] = lb[nupd] if len(lb) > nupd else None
            out["cpu_baseline"]["strong"] = strong
            out["cpu_baseline"]["host"] = host_info()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.close()


if __name__ == "__main__":
    main()
