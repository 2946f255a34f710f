"""echo / progress-bar helpers -- /root/reference/calamity/utils.py:1-10 (select_baselines: out of scope)."""
import tqdm


def _notebook_tqdm(*args, **kwargs):
    import tqdm.notebook as tqdm_notebook

    return tqdm_notebook.tqdm(*args, **kwargs)


PBARS = {True: _notebook_tqdm, False: tqdm.tqdm}


def echo(message, verbose=True):
    if verbose:
        print(message)
