"""Baseline-sharded data parallelism: one worker per GPU, one RCCL all-reduce per step.

The reference is single-device (/root/reference/calamity/calibration.py:1796-1804).  The fit shards naturally
(SURVEY.md section 8e): foreground coefficients belong to exactly one fitting group, so they live on the rank that owns
the group's baselines; gains are shared by all baselines, so they are replicated and their gradient -- a sum over
baselines -- is all-reduced together with the loss scalars (the payload of ``exchange_spec``).  Every rank then
applies the identical Adam update to its gain replica.
"""
import numpy as np

from .problem import FitProblem


def partition_groups(grp_nvec, grp_basis, grp_nbl, nranks, mode="deal"):
    """Assign whole fitting groups to ranks.  Returns a list of index arrays (one per rank, disjoint, covering all groups).

    ``mode="deal"`` (default): the groups, ordered by basis block and size, are dealt round-robin -- rank ``r`` takes entries
    ``r, r + nranks, ...`` of that list.  Every rank then holds the same number of groups (to one), the same number of
    baselines of every delay class (to one) and therefore the same basis bytes, sample bytes, vector-count classes and kernel
    mix: whatever a step costs per group -- tile bytes, per-sample bytes of every time slice, the per-class efficiency of the
    kernels -- the shares cost the same without a cost model.  (Round 3 cut the basis-ordered list into contiguous runs of
    equal ``sum nvec * nbl``: equal basis bytes, but at HERA-350 16 009 short baselines on rank 0 against 4 310 long ones on
    rank 7 -- and with the tiles shared by 8 time slices the per-sample traffic, not the basis, is most of a step.)
    Every rank touches every distinct basis block: 68 MB at HERA-350, read from cache in the shared layout.

    ``mode="contiguous"``: the round-3 rule (few distinct basis blocks per rank), kept for callers that want it."""
    grp_nvec = np.asarray(grp_nvec, dtype=np.float64)
    grp_basis = np.asarray(grp_basis)
    work = grp_nvec * np.asarray(grp_nbl)
    if mode == "deal":
        order = np.lexsort((np.arange(len(work)), -work, grp_basis))  # by basis, heaviest first, then index
        return [np.sort(order[r::nranks]) for r in range(nranks)]
    if mode != "contiguous":
        raise ValueError(f"unknown partition mode {mode!r}")
    order = np.argsort(grp_basis, kind="stable")
    csum = np.cumsum(work[order])
    total = csum[-1]
    cuts = [0]
    for r in range(1, nranks):
        cuts.append(int(np.searchsorted(csum, total * r / nranks, side="left")))
    cuts.append(len(order))
    for r in range(1, len(cuts)):  # never hand a rank an empty share while groups remain
        cuts[r] = max(cuts[r], min(cuts[r - 1] + 1, len(order)))
    return [np.sort(order[cuts[r] : cuts[r + 1]]) for r in range(nranks)]


def shard_problem(prob, start, rank, nranks):
    """The share of ``rank``: (FitProblem of its groups, start dict with its coefficients and the full gains).
    Weights are NOT renormalised: the loss of the job is the sum of the ranks' losses."""
    nbl_g = np.diff(prob.grp_bl_start)
    mine = partition_groups(prob.grp_nvec, prob.grp_basis, nbl_g, nranks)[rank]
    return select_groups(prob, start, mine)


def select_groups(prob, start, groups):
    groups = np.asarray(groups)
    bl = np.concatenate([np.arange(prob.grp_bl_start[g], prob.grp_bl_start[g + 1]) for g in groups])
    nbl_g = np.diff(prob.grp_bl_start)[groups]
    used = np.unique(prob.grp_basis[groups])
    remap = -np.ones(len(prob.basis), dtype=np.int64)
    remap[used] = np.arange(len(used))
    sub = FitProblem(
        nants=prob.nants,
        nfreqs=prob.nfreqs,
        basis=[prob.basis[u] for u in used],
        grp_basis=remap[prob.grp_basis[groups]].astype(np.int32),
        grp_bl_start=np.concatenate([[0], np.cumsum(nbl_g)]).astype(np.int32),
        bl_ant0=prob.bl_ant0[bl],
        bl_ant1=prob.bl_ant1[bl],
        bl_rowblk=prob.bl_rowblk[bl],
        data_r=prob.data_r[bl],
        data_i=prob.data_i[bl],
        wgts=prob.wgts[bl],
        sky_r=None if prob.sky_r is None else prob.sky_r[bl],
        sky_i=None if prob.sky_i is None else prob.sky_i[bl],
    )
    coff = prob.grp_coff
    sub_start = dict(
        g_r=start["g_r"],
        g_i=start["g_i"],
        c_r=np.concatenate([start["c_r"][coff[g] : coff[g + 1]] for g in groups]),
        c_i=np.concatenate([start["c_i"][coff[g] : coff[g + 1]] for g in groups]),
    )
    sub.validate()
    return sub, sub_start


def batch_time_slices(parts, per_slice=False):
    """Fit several independent time slices in ONE solver: slice ``t`` keeps its own gains by offsetting its antenna
    indices by ``t * nants`` (calibration.py:1167 loops over times; the fits are independent, Adam is element-wise, so
    the joint update equals the separate updates).  With ``per_slice`` the solver also keeps one loop state per slice
    (``FitProblem.nslices``: every slice records its own losses and stops on its own, ``HipFitSolver.run_slices``);
    without, the slices are one fit whose recorded loss is the sum of the slices' losses.
    ``parts``: list of (FitProblem, start) with identical nants / nfreqs.  Returns (FitProblem, start)."""
    nants, nfreqs = parts[0][0].nants, parts[0][0].nfreqs
    basis, key = [], {}
    grp_basis, grp_bl_start, a0, a1, rb, alias = [], [0], [], [], [], []
    # slices that hold the same baselines in the same order (the sharded multi-time job: every slice is the same share of the
    # array) share basis tiles: baseline b of slice t > 0 reads the tiles of baseline b of slice 0 (cal_problem_desc::bl_alias)
    same_bls = all(
        p.nbls == parts[0][0].nbls and np.array_equal(p.bl_ant0, parts[0][0].bl_ant0) and np.array_equal(p.bl_ant1, parts[0][0].bl_ant1)
        and np.array_equal(p.bl_rowblk, parts[0][0].bl_rowblk) and np.all(np.diff(p.grp_bl_start) == 1)
        and all(a is b for a, b in zip([p.basis[u] for u in p.grp_basis], [parts[0][0].basis[u] for u in parts[0][0].grp_basis]))
        for p, _ in parts
    )
    for t, (p, s) in enumerate(parts):
        assert p.nants == nants and p.nfreqs == nfreqs
        for u, blk in enumerate(p.basis):
            if id(blk) not in key:
                key[id(blk)] = len(basis)
                basis.append(blk)
        grp_basis.append(np.asarray([key[id(p.basis[u])] for u in p.grp_basis], dtype=np.int32))
        grp_bl_start.extend((p.grp_bl_start[1:] + grp_bl_start[-1]).tolist())
        a0.append(p.bl_ant0 + t * nants)
        a1.append(p.bl_ant1 + t * nants)
        rb.append(p.bl_rowblk)
        alias.append(np.full(p.nbls, -1, dtype=np.int32) if t == 0 else np.arange(p.nbls, dtype=np.int32))
    cat = lambda name: np.concatenate([getattr(p, name) for p, _ in parts])  # noqa: E731
    has_sky = parts[0][0].sky_r is not None
    out = FitProblem(
        nants=nants * len(parts),
        nfreqs=nfreqs,
        basis=basis,
        grp_basis=np.concatenate(grp_basis),
        grp_bl_start=np.asarray(grp_bl_start, dtype=np.int32),
        bl_ant0=np.concatenate(a0).astype(np.int32),
        bl_ant1=np.concatenate(a1).astype(np.int32),
        bl_rowblk=np.concatenate(rb).astype(np.int32),
        data_r=cat("data_r"),
        data_i=cat("data_i"),
        wgts=cat("wgts"),
        sky_r=cat("sky_r") if has_sky else None,
        sky_i=cat("sky_i") if has_sky else None,
        bl_alias=np.concatenate(alias) if same_bls and len(parts) > 1 else None,
        nslices=len(parts) if per_slice else 1,
    )
    start = dict(
        g_r=np.concatenate([s["g_r"] for _, s in parts]),
        g_i=np.concatenate([s["g_i"] for _, s in parts]),
        c_r=np.concatenate([s["c_r"] for _, s in parts]),
        c_i=np.concatenate([s["c_i"] for _, s in parts]),
    )
    out.validate()
    return out, start


def exchange_spec(nants, nfreqs, reg_sum=False):
    """What one step exchanges between ranks: the gain-gradient parts (interleaved re/im, one part
    without and three with the "sum" regulariser) in the fit dtype, plus four float64 scalars (loss, S_r, S_i, spare)."""
    return dict(gain_grad_reals=(3 if reg_sum else 1) * 2 * nants * nfreqs, scalars_f64=4)
