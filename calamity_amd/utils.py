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
