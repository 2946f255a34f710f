"""Minimal stand-ins for pyuvdata's UVData / UVCal / UVFlag.

pyuvdata is not installable in the build environment, so the drop-in entry points
(``calamity_amd.calibration.calibrate_and_model_dpss`` etc.) are written against the *attribute surface* the
reference touches.  Both array vintages are accepted everywhere, by shape: old pyuvdata
(>= 2.1.5) WITH the spw axis (``data_array[Nblts, 1, Nfreqs, Npols]``, ``gain_array[Nants, 1, Nfreqs, Ntimes, Njones]``,
the one the reference indexes) and pyuvdata >= 3 without it (``vis3`` / ``gain4`` / ``freqs_1d`` below).  Real pyuvdata
objects duck-type into the same code; these classes implement exactly that surface for tests, the
synthetic generator and users without pyuvdata.  They are containers, not a re-implementation of pyuvdata.
"""
import copy

import numpy as np

POLNUM2STR = {-5: "xx", -6: "yy", -7: "xy", -8: "yx", 1: "pI", 2: "pQ", 3: "pU", 4: "pV", -1: "rr", -2: "ll", -3: "rl", -4: "lr"}
POLSTR2NUM = {v: k for k, v in POLNUM2STR.items()}
_EN = {"east": {"e": "x", "n": "y"}, "north": {"e": "y", "n": "x"}}


def polstr2num(pol, x_orientation=None):
    """Subset of pyuvdata.utils.polstr2num used at calibration.py:294, :338, :363, :395, :782, :820."""
    if isinstance(pol, (int, np.integer)):
        return int(pol)
    p = pol.lower() if not pol.startswith("p") else pol
    if p in POLSTR2NUM:
        return POLSTR2NUM[p]
    if x_orientation is not None and set(p) <= {"e", "n"}:
        m = _EN[x_orientation.lower()]
        return POLSTR2NUM["".join(m[ch] for ch in p)]
    raise KeyError(f"Polarization {pol} cannot be converted to a polarization number.")


def polnum2str(num, x_orientation=None):
    return POLNUM2STR[int(num)]


def vis3(arr):
    """UVData-like array -> view (Nblts, Nfreqs, Npols): old pyuvdata (>= 2.1.5) carries a length-1 spw axis
    (Nblts, 1, Nfreqs, Npols), pyuvdata >= 3 ("future array shapes") does not.  A view: assignments reach the object."""
    return arr[:, 0] if np.ndim(arr) == 4 else arr


def gain4(arr):
    """UVCal-like array -> view (Nants, Nfreqs, Ntimes, Njones), with or without the length-1 spw axis."""
    return arr[:, 0] if np.ndim(arr) == 5 else arr


def freqs_1d(obj):
    f = np.asarray(obj.freq_array)
    return f[0] if f.ndim == 2 else f


def antnums_to_baseline(a1, a2):
    return 2048 * (int(a1) + 1) + (int(a2) + 1) + 2 ** 16


def baseline_to_antnums(bl):
    """One baseline number -> (ant1, ant2) ints; an array of them -> two arrays (as pyuvdata's method does)."""
    if np.ndim(bl) == 0:
        bl = int(bl) - 2 ** 16
        return (bl // 2048 - 1, bl % 2048 - 1)
    bl = np.asarray(bl, dtype=np.int64) - 2 ** 16
    return bl // 2048 - 1, bl % 2048 - 1


def _copy_array(a):
    """``a.copy()``; arrays of hundreds of megabytes (visibilities, nsamples at HERA-350) row chunk by row chunk on the host's cores."""
    if a.nbytes < (64 << 20) or a.ndim < 2 or not a.flags.c_contiguous:
        return a.copy()
    from .utils import for_row_chunks

    out = np.empty_like(a)

    def chunk(lo, hi):
        out[lo:hi] = a[lo:hi]

    for_row_chunks(chunk, a.shape[0])
    return out


class SimpleUVData:
    """Container with the UVData attributes and methods the calamity path uses."""

    def __init__(self, antpos, antpairs, freqs, times, pols=(-5,), data=None, flags=None, nsamples=None, antnums=None,
                 x_orientation=None, future_shapes=False):
        antpos = np.asarray(antpos, dtype=np.float64)
        self.antenna_numbers = np.arange(len(antpos)) if antnums is None else np.asarray(antnums)
        self.antenna_positions = antpos  # ENU, metres (pyuvdata stores ECEF; only differences are used here)
        self.antenna_names = [f"ant{n}" for n in self.antenna_numbers]
        self.Nants_telescope = len(antpos)
        self.telescope_name = "synthetic"
        self.telescope_location = np.zeros(3)
        times = np.atleast_1d(np.asarray(times, dtype=np.float64))
        antpairs = [tuple(int(x) for x in ap) for ap in antpairs]
        self.Nbls = len(antpairs)
        self.Ntimes = len(times)
        self.Nblts = self.Nbls * self.Ntimes
        # time-major ordering (all baselines of time 0, then time 1, ...)
        self.ant_1_array = np.tile(np.asarray([ap[0] for ap in antpairs]), self.Ntimes)
        self.ant_2_array = np.tile(np.asarray([ap[1] for ap in antpairs]), self.Ntimes)
        self.time_array = np.repeat(times, self.Nbls)
        self.lst_array = np.repeat(np.linspace(0.0, 1e-3, self.Ntimes, endpoint=False) if self.Ntimes > 1 else np.zeros(1), self.Nbls)
        self.integration_time = np.full(self.Nblts, 10.0)
        # future_shapes: the array layout of pyuvdata >= 3 (no spw axis, 1-D freq_array); default: the layout the
        # reference was written against
        self.future_array_shapes = bool(future_shapes)
        f1 = np.asarray(freqs, dtype=np.float64)
        self.freq_array = f1.copy() if future_shapes else f1[None, :]
        self.Nfreqs = len(f1)
        self.Nspws = 1
        self.spw_array = np.array([0])
        self.channel_width = float(np.median(np.diff(f1))) if self.Nfreqs > 1 else 1.0
        self.polarization_array = np.asarray(pols, dtype=int)
        self.Npols = len(self.polarization_array)
        self.x_orientation = x_orientation
        shape = (self.Nblts, self.Nfreqs, self.Npols) if future_shapes else (self.Nblts, 1, self.Nfreqs, self.Npols)
        self.data_array = np.zeros(shape, dtype=np.complex128) if data is None else np.asarray(data, dtype=np.complex128).reshape(shape)
        self.flag_array = np.zeros(shape, dtype=bool) if flags is None else np.asarray(flags, dtype=bool).reshape(shape)
        self.nsample_array = np.ones(shape, dtype=np.float64) if nsamples is None else np.asarray(nsamples, dtype=np.float64).reshape(shape)
        self._refresh()

    def __deepcopy__(self, memo):
        """Arrays are copied; the (antenna pair -> rows) index is shared: its arrays are never written, and copying 61 075 of
        them one by one cost more than copying the gigabyte of visibilities."""
        out = copy.copy(self)
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                out.__dict__[k] = memo[id(v)] if id(v) in memo else _copy_array(v)  # (memo: arrays the caller will replace anyway)
            elif isinstance(v, (list, tuple)):
                out.__dict__[k] = copy.copy(v)
        out._ap_index = dict(self._ap_index)
        return out

    def _refresh(self):
        self.baseline_array = np.asarray([antnums_to_baseline(a, b) for a, b in zip(self.ant_1_array, self.ant_2_array)])
        self.Nants_data = len(set(self.ant_1_array).union(set(self.ant_2_array)))
        self._ap_index = {}
        for n, ap in enumerate(zip(self.ant_1_array.tolist(), self.ant_2_array.tolist())):
            self._ap_index.setdefault(ap, []).append(n)
        self._ap_index = {k: np.asarray(v) for k, v in self._ap_index.items()}
        self.Nbls = len(self._ap_index)
        self.Ntimes = len(np.unique(self.time_array))
        self.Nblts = len(self.time_array)

    # ---- the methods the reference calls -----------------------------------------------------------------------
    def get_antpairs(self):
        return list(self._ap_index.keys())

    def get_antpairpols(self):
        return [ap + (p,) for ap in self.get_antpairs() for p in self.get_pols()]

    def get_pols(self):
        return [polnum2str(p, self.x_orientation) for p in self.polarization_array]

    def antpair2ind(self, ant1, ant2=None, ordered=True):
        ap = tuple(ant1) if ant2 is None else (ant1, ant2)
        return self._ap_index.get((int(ap[0]), int(ap[1])), np.asarray([], dtype=int))

    def _key2inds(self, key):
        """(blt inds of the pair, blt inds of the reversed pair, (pol ind, pol ind of the conjugate))."""
        a, b = int(key[0]), int(key[1])
        polnum = polstr2num(key[2], self.x_orientation) if len(key) > 2 else int(self.polarization_array[0])
        pi = np.where(self.polarization_array == polnum)[0]
        i1 = self.antpair2ind(a, b)
        i2 = self.antpair2ind(b, a) if a != b else np.asarray([], dtype=int)
        if len(i1) == 0 and len(i2) == 0:
            raise KeyError(f"Antenna pair ({a}, {b}) not found in data")
        # conjugating swaps the feed order of cross-hand pols (xy <-> yx); same index for the parallel hands
        swap = {-7: -8, -8: -7, -3: -4, -4: -3}
        pc = np.where(self.polarization_array == swap.get(polnum, polnum))[0]
        return i1, i2, (pi, pc)

    def get_data(self, key):
        i1, i2, (pi, pc) = self._key2inds(key)
        if len(i1):
            return vis3(self.data_array)[i1, :, pi[0]]
        return np.conj(vis3(self.data_array)[i2, :, pc[0]])

    def get_flags(self, key):
        i1, i2, (pi, pc) = self._key2inds(key)
        if len(i1):
            return vis3(self.flag_array)[i1, :, pi[0]]
        return vis3(self.flag_array)[i2, :, pc[0]]

    def baseline_to_antnums(self, bl):
        return baseline_to_antnums(bl)

    def antnums_to_baseline(self, a1, a2):
        return antnums_to_baseline(a1, a2)

    def get_ENU_antpos(self, pick_data_ants=False):
        if pick_data_ants:
            used = sorted(set(self.ant_1_array).union(set(self.ant_2_array)))
            sel = [int(np.where(self.antenna_numbers == a)[0][0]) for a in used]
            return self.antenna_positions[sel], np.asarray(used)
        return self.antenna_positions, self.antenna_numbers

    def select(self, bls=None, times=None, inplace=True):
        obj = self if inplace else copy.deepcopy(self)
        keep = np.ones(obj.Nblts, dtype=bool)
        if bls is not None:
            want = set((int(a), int(b)) for a, b in bls)
            keep &= np.asarray([(a, b) in want for a, b in zip(obj.ant_1_array.tolist(), obj.ant_2_array.tolist())])
        if times is not None:
            keep &= np.isin(obj.time_array, np.atleast_1d(times))
        for name in ("ant_1_array", "ant_2_array", "time_array", "lst_array", "integration_time", "data_array", "flag_array", "nsample_array"):
            setattr(obj, name, getattr(obj, name)[keep])
        obj._refresh()
        if not inplace:
            return obj

    def get_redundancies(self, tol=1.0, use_antpos=False, include_conjugates=False, include_autos=True, conjugate_bls=False):
        """Redundant baseline groups from the antenna positions, oriented east-positive (pyuvdata's u > 0 convention,
        v > 0 on ties).  Only the ``use_antpos=True`` form used at modeling.py:49 is implemented.
        Returns (groups of baseline numbers, vector bin centres, lengths, None)."""
        if not use_antpos:
            raise NotImplementedError("SimpleUVData.get_redundancies: only use_antpos=True is implemented")
        from scipy.spatial import cKDTree

        nums, pos = self.antenna_numbers, self.antenna_positions
        bls, vecs = [], []
        for aj in range(len(nums)):
            for ai in range(aj if include_autos else aj + 1, len(nums)):
                v = pos[ai] - pos[aj]  # uvw of baseline (aj, ai) = pos(ant2) - pos(ant1)
                a1, a2 = nums[aj], nums[ai]
                if v[0] < -1e-9 or (abs(v[0]) <= 1e-9 and v[1] < -1e-9) or (abs(v[0]) <= 1e-9 and abs(v[1]) <= 1e-9 and v[2] < 0):
                    v, a1, a2 = -v, a2, a1
                bls.append(antnums_to_baseline(a1, a2))
                vecs.append(v)
        vecs = np.asarray(vecs).reshape(-1, 3)
        tree = cKDTree(vecs)
        assigned = np.full(len(bls), -1)
        groups = []
        for n in range(len(bls)):
            if assigned[n] >= 0:
                continue
            members = [m for m in tree.query_ball_point(vecs[n], tol) if assigned[m] < 0]
            for m in members:
                assigned[m] = len(groups)
            groups.append(members)
        bl_groups = [[bls[m] for m in g] for g in groups]
        centers = [np.mean(vecs[g], axis=0) for g in groups]
        lengths = [float(np.linalg.norm(c)) for c in centers]
        return bl_groups, centers, lengths, None

    def __add__(self, other):
        """Concatenate along the blt axis (the reference's tests build multi-time sets this way)."""
        out = copy.deepcopy(self)
        for name in ("ant_1_array", "ant_2_array", "time_array", "lst_array", "integration_time", "data_array", "flag_array", "nsample_array"):
            setattr(out, name, np.concatenate([getattr(self, name), getattr(other, name)]))
        out._refresh()
        return out


class SimpleUVCal:
    """Container with the UVCal attributes the calamity path uses (cal_utils.py:7-59)."""

    def __init__(self):
        self.gain_convention = "divide"
        self.x_orientation = None

    @property
    def Nants_data(self):
        return len(self.ant_array)

    @Nants_data.setter
    def Nants_data(self, v):
        pass

    def __add__(self, other):
        out = copy.deepcopy(self)
        tax = np.ndim(self.gain_array) - 2  # the time axis: 3 with the spw axis, 2 without
        for name in ("gain_array", "flag_array", "quality_array"):
            setattr(out, name, np.concatenate([getattr(self, name), getattr(other, name)], axis=tax))
        out.time_array = np.concatenate([self.time_array, other.time_array])
        order = np.argsort(out.time_array)
        out.time_array = out.time_array[order]
        for name in ("gain_array", "flag_array", "quality_array"):
            setattr(out, name, np.take(getattr(out, name), order, axis=tax))
        out.Ntimes = len(out.time_array)
        return out


class SimpleUVFlag:
    """UVFlag(uvdata, mode="flag") surface used at calibration.py:287-296: weights_array, antpair2ind, get_antpairs,
    time_array, polarization_array, x_orientation."""

    def __init__(self, uvdata, mode="flag"):
        self.mode = mode
        self.ant_1_array = uvdata.ant_1_array.copy()
        self.ant_2_array = uvdata.ant_2_array.copy()
        self.time_array = uvdata.time_array.copy()
        self.polarization_array = uvdata.polarization_array.copy()
        self.x_orientation = uvdata.x_orientation
        self.flag_array = uvdata.flag_array.copy()
        self.weights_array = np.ones(uvdata.data_array.shape, dtype=np.float64)
        self._ap_index = {k: v.copy() for k, v in uvdata._ap_index.items()}

    def get_antpairs(self):
        return list(self._ap_index.keys())

    def antpair2ind(self, ant1, ant2=None):
        ap = tuple(ant1) if ant2 is None else (ant1, ant2)
        return self._ap_index.get((int(ap[0]), int(ap[1])), np.asarray([], dtype=int))


_KINDS = {}  # kind tag -> container class (filled below)


def _write_container(obj, path, clobber, kind):
    """Store a duck-typed container as an .npz archive (arrays as arrays, everything else as one JSON string) under
    exactly the given name.  No pickle: reading one of these files can never execute code."""
    import json
    import os

    if os.path.exists(path) and not clobber:
        raise IOError(f"{path} exists; use clobber=True to overwrite")
    arrays, meta = {}, {}
    for k, v in obj.__dict__.items():
        if k.startswith("_"):
            continue  # derived (rebuilt on read)
        if isinstance(v, np.ndarray):
            arrays[k] = v
        elif isinstance(v, (np.integer, np.floating, np.bool_)):
            meta[k] = v.item()
        elif isinstance(v, tuple):
            meta[k] = [x.item() if isinstance(x, np.generic) else x for x in v]
        else:
            meta[k] = v
    with open(path, "wb") as f:  # a file object: np.savez would append ".npz" to a bare name
        np.savez(f, __kind__=np.asarray(kind), __meta__=np.asarray(json.dumps(meta)), **arrays)


def read_container(path):
    """Read a container written by ``SimpleUVData.write_uvh5`` / ``SimpleUVCal.write_calfits`` (an .npz archive whatever
    its name; real uvh5 / calfits files need pyuvdata and are rejected here with a clear message)."""
    import json

    try:
        z = np.load(path, allow_pickle=False)
        kind = str(z["__kind__"])
    except Exception as e:  # not one of our archives (e.g. a real HDF5 / FITS file)
        raise IOError(f"{path} is not a calamity_amd container archive; reading uvh5 / calfits files needs pyuvdata ({e})")
    obj = object.__new__(_KINDS[kind])
    for k, v in json.loads(str(z["__meta__"])).items():
        setattr(obj, k, v)
    for k in z.files:
        if not k.startswith("__"):
            setattr(obj, k, z[k])
    if hasattr(obj, "_refresh"):
        obj._refresh()
    return obj


# The file driver (calibration.py:1659-1817) writes its outputs with these method names.  The duck-typed containers
# store themselves as .npz archives under whatever name they are given; real uvh5 / calfits I/O needs pyuvdata objects.
SimpleUVData.write_uvh5 = lambda self, path, clobber=False: _write_container(self, path, clobber, "uvdata")
SimpleUVCal.write_calfits = lambda self, path, clobber=False: _write_container(self, path, clobber, "uvcal")
_KINDS.update(uvdata=SimpleUVData, uvcal=SimpleUVCal)


def is_uvdata(obj):
    return hasattr(obj, "data_array") and hasattr(obj, "ant_1_array")


def is_uvcal(obj):
    return hasattr(obj, "gain_array") and hasattr(obj, "ant_array")
