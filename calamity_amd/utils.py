"""echo / progress-bar helpers and the baseline cuts of the file driver -- /root/reference/calamity/utils.py."""
import numpy as np
import tqdm


def _notebook_tqdm(*args, **kwargs):
    import tqdm.notebook as tqdm_notebook

    return tqdm_notebook.tqdm(*args, **kwargs)


PBARS = {True: _notebook_tqdm, False: tqdm.tqdm}


def echo(message, verbose=True):
    if verbose:
        print(message)


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a GPU box hands a one-GPU job
    16 of the host's 256 cores)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return n


def for_row_chunks(fn, nrows, min_rows=2048):
    """``fn(lo, hi)`` over consecutive row ranges covering [0, nrows), on the host's cores.  The element-wise passes of the
    drop-in over (Nblts, Nfreqs) arrays -- gigabytes at HERA-350 -- are memory-bound NumPy ufuncs, which release the GIL: a
    thread per chunk makes them run side by side (1.5 s -> 0.3 s per pass on a 16-core share)."""
    import concurrent.futures

    nchunks = max(1, min(usable_cores(), nrows // max(1, min_rows)))
    if nchunks == 1:
        fn(0, nrows)
        return
    bounds = np.linspace(0, nrows, nchunks + 1).astype(np.int64)
    with concurrent.futures.ThreadPoolExecutor(max_workers=nchunks) as pool:
        list(pool.map(lambda k: fn(int(bounds[k]), int(bounds[k + 1])), range(nchunks)))


def select_baselines(uvdata, bllen_min=0.0, bllen_max=np.inf, bl_ew_min=0.0, ex_ants=None, select_ants=None):
    """Keep the baselines inside the length / east-west / antenna cuts, in place -- utils.py:13-37.

    Both length bounds are inclusive, the east-west bound is strict, as in the reference."""
    ex_ants = set([] if ex_ants is None else ex_ants)
    antpos, antnums = uvdata.get_ENU_antpos(pick_data_ants=True)
    select_ants = set(antnums) if select_ants is None else set(select_ants)
    posdict = {an: ap for an, ap in zip(antnums, antpos)}
    keep = []
    for ap in uvdata.get_antpairs():
        blvec = posdict[ap[0]] - posdict[ap[1]]
        bllen = np.linalg.norm(blvec)
        if (bllen_min <= bllen <= bllen_max and np.abs(blvec[0]) > bl_ew_min and ap[0] not in ex_ants and ap[1] not in ex_ants
                and ap[0] in select_ants and ap[1] in select_ants):
            keep.append(ap)
    uvdata.select(bls=keep, inplace=True)
