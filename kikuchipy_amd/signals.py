"""A minimal `EBSD` holder with the three methods of the accelerated path.

NOT a re-implementation of kikuchipy's HyperSpy signal (out of scope): just
enough object surface - `data`, `static_background`, `xmap` (dictionary
rotations), the navigation/signal shapes - for the three methods to read like
the reference's:

* `EBSD.remove_static_background`   signals/ebsd.py:442-573
* `EBSD.remove_dynamic_background`  signals/ebsd.py:575-696
* `EBSD.dictionary_indexing`        signals/ebsd.py:1827-1984

Like the reference's methods, each call hands back host data (`self.data` is
replaced by the pre-processed array); callers that want the whole chain
static -> dynamic -> indexing to stay in HBM use the C ABI / `_lib.Context`
directly (`kpdi_remove_*_background` work in place on the resident patterns
that `kpdi_push_dictionary_chunk` then matches), as `bench.py --workload
config3` does.
"""

import numpy as np

from kikuchipy_amd import _lib
from kikuchipy_amd.indexing._dictionary_indexing import dictionary_indexing as _dictionary_indexing
from kikuchipy_amd.pattern import _pattern


class DictionaryXmap:
    """Stand-in for the `xmap` of a dictionary signal: one rotation
    (unit quaternion) per dictionary pattern."""

    def __init__(self, rotations, phase_name=""):
        self.rotations = np.asarray(rotations, dtype=np.float64).reshape(-1, 4)
        self.phase_name = phase_name

    @classmethod
    def empty(cls, shape):
        """Like `CrystalMap.empty((n,))`: identity rotations."""
        shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        q = np.zeros(shape + (4,))
        q[..., 0] = 1
        obj = cls(q.reshape(-1, 4))
        obj._shape = shape
        return obj

    @property
    def shape(self):
        return getattr(self, "_shape", (self.rotations.shape[0],))


class EBSD:
    def __init__(self, data, static_background=None, xmap=None, step_sizes=None, scan_unit="px",
                 device=0):
        self.data = data
        if np.ndim(data) < 2 or np.ndim(data) > 4:
            raise ValueError("EBSD data must have 0, 1 or 2 navigation axes and 2 signal axes")
        self.static_background = static_background
        self.xmap = xmap
        self.step_sizes = step_sizes
        self.scan_unit = scan_unit
        self._device = device
        self._ctx = None

    # ------------------------------------------------------------------ shapes
    @property
    def _navigation_shape_rc(self):
        return tuple(self.data.shape[:-2])

    @property
    def _signal_shape_rc(self):
        return tuple(self.data.shape[-2:])

    @property
    def navigation_size(self):
        return int(np.prod(self._navigation_shape_rc)) if self._navigation_shape_rc else 0

    def deepcopy(self):
        out = EBSD(np.array(self.data, copy=True),
                   None if self.static_background is None else np.array(self.static_background),
                   self.xmap, self.step_sizes, self.scan_unit, self._device)
        return out

    @property
    def context(self):
        if self._ctx is None:
            self._ctx = _lib.Context(self._device)
        return self._ctx

    # ------------------------------------------------------------------ pre-processing
    def remove_static_background(self, operation="subtract", static_bg=None, scale_bg=False,
                                 inplace=True):
        if static_bg is None:
            static_bg = self.static_background
            if not isinstance(static_bg, np.ndarray) and not hasattr(static_bg, "compute"):
                raise ValueError("`EBSD.static_background` is not a valid array")
        out = _pattern.remove_static_background(np.asarray(self.data), static_bg, operation, scale_bg,
                                                context=self.context)
        if inplace:
            self.data = out
            return None
        return EBSD(out, self.static_background, self.xmap, self.step_sizes, self.scan_unit, self._device)

    def remove_dynamic_background(self, operation="subtract", filter_domain="frequency", std=None,
                                  truncate=4.0, inplace=True):
        out = _pattern.remove_dynamic_background(np.asarray(self.data), operation, filter_domain, std,
                                                 truncate, context=self.context)
        if inplace:
            self.data = out
            return None
        return EBSD(out, self.static_background, self.xmap, self.step_sizes, self.scan_unit, self._device)

    # ------------------------------------------------------------------ indexing
    def dictionary_indexing(self, dictionary, metric="ncc", keep_n=20, n_per_iteration=None,
                            navigation_mask=None, signal_mask=None, rechunk=False, dtype=None, *,
                            comm=None, verbose=True):
        """See `kikuchipy_amd.dictionary_indexing`; `dictionary` is an `EBSD`
        with a 1-D navigation axis and an `xmap` of equal size."""
        dict_data = dictionary.data
        dict_nav = dictionary._navigation_shape_rc
        dict_size = int(np.prod(dict_nav)) if dict_nav else 0
        dict_xmap = dictionary.xmap
        sig_exp, sig_dict = self._signal_shape_rc, dictionary._signal_shape_rc
        if sig_exp != sig_dict:
            raise ValueError(
                f"Experimental {sig_exp} and dictionary {sig_dict} signal shapes must be identical"
            )
        if dict_xmap is None or dict_xmap.shape != (dict_size,) or len(dict_nav) != 1:
            raise ValueError(
                "Dictionary signal must have a non-empty `EBSD.xmap` attribute of equal"
                " size as the number of dictionary patterns, and both the signal and "
                "crystal map must have only one navigation dimension"
            )
        return _dictionary_indexing(
            self.data, dict_data, metric, keep_n, n_per_iteration, navigation_mask, signal_mask,
            rechunk, dtype, step_sizes=self.step_sizes, dictionary_rotations=dict_xmap.rotations,
            phase_name=dict_xmap.phase_name, scan_unit=self.scan_unit, device=self._device, comm=comm,
            verbose=verbose,
        )
