"""Unity-gain UVCal from UVData and gain application -- behaviour of /root/reference/calamity/cal_utils.py:7-105,
written against the duck-typed attribute surface (works with pyuvdata objects of either array vintage -- with or
without the length-1 spw axis -- and with
calamity_amd.uvcompat containers).  Host-side NumPy; nothing here is on the GPU hot path."""
import copy

import numpy as np

from . import uvcompat
from .utils import for_row_chunks


def _new_uvcal(uvdata):
    if isinstance(uvdata, uvcompat.SimpleUVData):
        return uvcompat.SimpleUVCal()
    try:  # a real pyuvdata object: build a real UVCal
        from pyuvdata import UVCal

        return UVCal()
    except Exception:
        return uvcompat.SimpleUVCal()


def blank_uvcal_from_uvdata(uvdata):
    """UVCal with the times, antennas, frequencies and jones of ``uvdata``: unity gains, no flags, convention
    "divide" (cal_utils.py:7-59)."""
    uvcal = _new_uvcal(uvdata)
    uvcal.Nfreqs = uvdata.Nfreqs
    uvcal.Njones = uvdata.Npols
    uvcal.Ntimes = uvdata.Ntimes
    uvcal.Nspws = uvdata.Nspws
    uvcal.history = ""
    uvcal.telescope_name = uvdata.telescope_name
    uvcal.telescope_location = uvdata.telescope_location
    uvcal.ant_array = np.asarray(sorted(set(np.asarray(uvdata.ant_1_array).tolist()).union(set(np.asarray(uvdata.ant_2_array).tolist()))))
    uvcal.Nants_data = uvdata.Nants_data
    uvcal.Nants_telescope = uvdata.Nants_telescope
    uvcal.antenna_names = uvdata.antenna_names
    uvcal.antenna_numbers = uvdata.antenna_numbers
    uvcal.antenna_positions = uvdata.antenna_positions
    uvcal.spw_array = uvdata.spw_array
    uvcal.freq_array = uvdata.freq_array
    uvcal.jones_array = uvdata.polarization_array
    uvcal.time_array = np.unique(uvdata.time_array)
    uvcal.integration_time = np.mean(uvdata.integration_time)
    uvcal.lst_array = np.unique(uvdata.lst_array)
    uvcal.gain_convention = "divide"  # always "divide" (cal_utils.py:43)
    # the gain arrays take the vintage of the data: with or without the length-1 spw axis
    if np.ndim(uvdata.data_array) == 4:
        shape = (len(uvcal.ant_array), uvcal.Nspws, uvcal.Nfreqs, uvcal.Ntimes, uvcal.Njones)
    else:
        shape = (len(uvcal.ant_array), uvcal.Nfreqs, uvcal.Ntimes, uvcal.Njones)
    uvcal.flag_array = np.zeros(shape, dtype=bool)
    uvcal.quality_array = np.zeros(shape, dtype=np.float64)
    uvcal.x_orientation = uvdata.x_orientation
    uvcal.gain_array = np.ones(shape, dtype=np.complex128)
    uvcal.cal_style = "redundant"
    uvcal.cal_type = "gain"
    uvcal.time_range = (uvcal.time_array.min() - uvcal.integration_time / 2.0, uvcal.time_array.max() + uvcal.integration_time / 2.0)
    uvcal.channel_width = np.median(np.diff(np.ravel(uvcal.freq_array))) if uvcal.Nfreqs > 1 else 1.0
    return uvcal


def apply_gains(uvdata, gains, inverse=False):
    """``data / (g_i conj(g_j))`` (or ``*`` when ``inverse``) for every baseline-time and polarization; flags are
    or-ed with the gain flags of both antennas (cal_utils.py:62-105).  Vectorised over baseline-times."""
    calibrated = copy.deepcopy(uvdata)
    ant_index = {int(a): n for n, a in enumerate(np.asarray(gains.ant_array).tolist())}
    a0 = np.asarray([ant_index[int(a)] for a in calibrated.ant_1_array])
    a1 = np.asarray([ant_index[int(a)] for a in calibrated.ant_2_array])
    gtimes = np.asarray(gains.time_array)
    tind = np.asarray([np.where(np.isclose(gtimes, t, rtol=0.0, atol=1e-7))[0][0] for t in np.unique(calibrated.time_array)])
    tmap = dict(zip(np.unique(calibrated.time_array).tolist(), tind.tolist()))
    gt = np.asarray([tmap[t] for t in calibrated.time_array.tolist()])
    data, flags = uvcompat.vis3(calibrated.data_array), uvcompat.vis3(calibrated.flag_array)
    garr, gflags = uvcompat.gain4(gains.gain_array), uvcompat.gain4(gains.flag_array)
    for pnum, pol in enumerate(uvdata.get_pols()):
        gindp = np.where(np.asarray(gains.jones_array) == uvcompat.polstr2num(pol, x_orientation=gains.x_orientation))[0][0]

        for t in np.unique(gt):
            # one (Nants, Nfreqs) gain plane per time: row gathers from a contiguous plane are an order of magnitude faster than
            # indexing the 4-D gain array with two index arrays and a slice between them
            gplane = np.ascontiguousarray(garr[:, :, t, gindp])
            fplane = np.ascontiguousarray(gflags[:, :, t, gindp])
            sel = np.where(gt == t)[0]
            contiguous = len(sel) == sel[-1] - sel[0] + 1

            def rows(lo, hi, pnum=pnum, sel=sel, gplane=gplane, fplane=fplane, contiguous=contiguous):
                r = slice(sel[0] + lo, sel[0] + hi) if contiguous else sel[lo:hi]
                gg = np.take(gplane, a0[r], axis=0)  # (np.take: fancy indexing of complex rows is 20x slower in NumPy 2.2)
                gg *= np.conj(np.take(gplane, a1[r], axis=0))
                if not inverse:
                    data[r, :, pnum] /= gg
                else:
                    data[r, :, pnum] *= gg
                flags[r, :, pnum] |= np.take(fplane, a0[r], axis=0) | np.take(fplane, a1[r], axis=0)

            for_row_chunks(rows, len(sel))
    return calibrated
