"""Minimal `EBSD` / `EBSDMasterPattern` holders with the methods of the accelerated path.

NOT a re-implementation of kikuchipy's HyperSpy signal (out of scope): just
enough object surface - `data`, `static_background`, `xmap` (dictionary
rotations), the navigation/signal shapes - for the three methods to read like
the reference's:

* `EBSD.remove_static_background`   signals/ebsd.py:442-573
* `EBSD.remove_dynamic_background`  signals/ebsd.py:575-696
* `EBSD.dictionary_indexing`        signals/ebsd.py:1827-1984
* `EBSDMasterPattern.get_patterns`  signals/ebsd_master_pattern.py:95-330
* `EBSD.refine_orientation` / `refine_projection_center` /
  `refine_orientation_projection_center`   signals/ebsd.py:1986-2700

Like the reference's methods, each call hands back host data (`self.data` is
replaced by the pre-processed array); callers that want the whole chain
static -> dynamic -> indexing to stay in HBM use the C ABI / `_lib.Context`
directly (`kpdi_remove_*_background` work in place on the resident patterns
that `kpdi_push_dictionary_chunk` then matches), as `bench.py --workload
config3` does.
"""

import warnings

import numpy as np

from kikuchipy_amd import _lib
from kikuchipy_amd.indexing._dictionary_indexing import dictionary_indexing as _dictionary_indexing
from kikuchipy_amd.pattern import _pattern
from kikuchipy_amd.simulations import DTYPE_RANGE, ProjectedDictionary


# a call that names no device spreads a refinement over every visible GPU from this many points on (one block of the
# points per GPU; below, the set-up of a group costs more than it saves)
REFINE_GROUP_MIN_POINTS = 2048
# ... and a background removal from this many patterns on (a block of the patterns per GPU, each over its own host link:
# the call is transfer-bound - patterns up, patterns down)
PREPROCESS_GROUP_MIN_POINTS = 16384


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

    @property
    def size(self):
        return int(self.rotations.shape[0])


class _Axes:
    """See `EBSD.axes_manager`."""

    def __init__(self, navigation_shape, signal_shape):
        self.navigation_shape = tuple(int(v) for v in navigation_shape)
        self.signal_shape = tuple(int(v) for v in signal_shape)

    navigation_dimension = property(lambda self: len(self.navigation_shape))
    signal_dimension = property(lambda self: len(self.signal_shape))
    navigation_size = property(lambda self: int(np.prod(self.navigation_shape)) if self.navigation_shape else 0)
    signal_size = property(lambda self: int(np.prod(self.signal_shape)))

    def __repr__(self):
        return f"<axes: navigation {self.navigation_shape} | signal {self.signal_shape}>"


class EBSD:
    def __init__(self, data, static_background=None, xmap=None, step_sizes=None, scan_unit="px",
                 device=None, devices=None, *, detector=None):
        """device: the GPU this signal's engine context lives on (None: GPU 0, and `dictionary_indexing`
        is free to shard the dictionary over every visible GPU); devices: "all" / a list of ids for
        `dictionary_indexing` (see `kikuchipy_amd.dictionary_indexing`); `detector`, `static_background`, `xmap`: the
        reference's custom attributes (signals/ebsd.py:188-199) - without a detector, one of the signal's shape."""
        self.data = data
        self._detector = None
        ndim = data.ndim if hasattr(data, "ndim") else np.ndim(data)  # lazy data is not touched
        if ndim < 2 or ndim > 4:
            raise ValueError("EBSD data must have 0, 1 or 2 navigation axes and 2 signal axes")
        self._static_background = static_background  # (the constructor does not check: signals/ebsd.py:198-199)
        self._xmap = xmap
        self.step_sizes = step_sizes
        self.scan_unit = scan_unit
        self._device = device
        self._devices = devices
        self._ctx = None
        self._groups = {}  # device ids -> _lib.Group, kept from call to call (its communicator is made once)
        if detector is not None:
            self.detector = detector

    @property
    def detector(self):
        """The detector - sample geometry (signals/ebsd.py:203-223): by default a detector of the signal's shape with
        the reference's default projection centre; setting one checks it against the signal as the reference does
        (signals/util/_detector.py:28-59)."""
        if self._detector is None:
            from kikuchipy_amd.detectors import EBSDDetector

            self._detector = EBSDDetector(shape=self._signal_shape_rc)
        return self._detector

    @property
    def xmap(self):
        """Crystal map of the signal (signals/ebsd.py:225-245): anything with a `shape`; setting one whose shape is not
        the navigation shape raises as the reference does (signals/util/_crystal_map.py:55-59)."""
        return self._xmap

    @xmap.setter
    def xmap(self, value):
        shape = getattr(value, "shape", None)
        nav = self._navigation_shape_rc
        if value is not None and shape is not None and tuple(shape) != nav and tuple(shape) != (nav or (1,)):
            raise ValueError(
                f"Crystal map shape {tuple(shape)} and signal's navigation shape {nav} must be the same "
                "(see EBSD.axes_manager)"
            )
        self._xmap = value

    @property
    def static_background(self):
        """Static background pattern (signals/ebsd.py:247-266); one of another data type or shape is set with the
        reference's warning (`remove_static_background` then refuses it)."""
        return self._static_background

    @static_background.setter
    def static_background(self, value):
        if value is not None:
            if getattr(value, "dtype", None) != self.data.dtype:
                warnings.warn("Background pattern has different data type from patterns")
            if tuple(getattr(value, "shape", ())) != self._signal_shape_rc:
                warnings.warn("Background pattern has different shape from patterns")
        self._static_background = value

    @detector.setter
    def detector(self, value):
        if tuple(value.shape) != self._signal_shape_rc:
            raise ValueError(f"Detector shape {value.shape} must be equal to the signal shape {self._signal_shape_rc}.")
        if value.navigation_shape != (1,) and value.navigation_shape != self._navigation_shape_rc:
            raise ValueError(
                "Detector must have exactly one projection center (PC), or one PC per pattern in an array of shape "
                "equal to signal's navigation shape + (3,)."
            )
        self._detector = value

    # ------------------------------------------------------------------ engines
    def close(self):
        """Destroy the engines this signal keeps from call to call (its own context, its groups over several GPUs -
        their host threads and communicators); they are made again when needed.  `with EBSD(...) as s:` calls it."""
        for g in self._groups.values():
            g.close()
        self._groups = {}
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

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

    @property
    def axes_manager(self):
        """The four numbers scripts written for kikuchipy read from HyperSpy's axes manager around this path
        (`s.axes_manager.signal_shape[::-1]`, `sim.axes_manager.navigation_size // 10` in the reference's
        doc/tutorials/pattern_matching.ipynb): shapes in HyperSpy's (x, y) order - the REVERSE of the array's - and
        sizes.  Nothing else of the signal model is here (out of scope)."""
        return _Axes(self._navigation_shape_rc[::-1], self._signal_shape_rc[::-1])

    def deepcopy(self):
        out = EBSD(np.array(self.data, copy=True),
                   None if self.static_background is None else np.array(self.static_background),
                   self.xmap, self.step_sizes, self.scan_unit, self._device, self._devices)
        out._detector = None if self._detector is None else self._detector.deepcopy()
        if hasattr(self, "original_metadata"):
            out.original_metadata = self.original_metadata
        return out

    @property
    def context(self):
        if self._ctx is None:
            self._ctx = _lib.Context(self._device or 0)
        return self._ctx

    def _member_contexts(self, devices, min_points):
        """One engine context per GPU for work that splits over the map's points (pre-processing, refinement): the
        members of this signal's group over `devices` (or over every visible GPU when nothing was named and the map has
        `min_points` points or more), None = this signal's own context."""
        ids = _lib.resolve_devices(devices if devices is not None else self._devices)
        if ids is None and self._device is None and self.navigation_size >= min_points:
            ids = _lib.default_devices()
        if ids is None or len(ids) < 2:
            return None
        key = tuple(ids)
        if key not in self._groups:
            self._groups[key] = _lib.make_engine(devices=ids)
        return self._groups[key].members

    def _engine(self, devices, comm, dictionary_size):
        """The engine a `dictionary_indexing` call of this signal runs on: its own context, or a `Group`
        over several GPUs (kept on the signal from call to call)."""
        from kikuchipy_amd.indexing._dictionary_indexing import pick_devices

        device, ids = pick_devices(self._device, devices if devices is not None else self._devices, comm,
                                   max(self.navigation_size, 1), dictionary_size)
        if ids is not None and len(ids) == 1:
            device, ids = ids[0], None
        if ids is None:
            return device, (self.context if device == (self._device or 0) else _lib.Context(device))
        key = tuple(ids)
        if key not in self._groups:
            self._groups[key] = _lib.make_engine(devices=ids)
        return ids[0], self._groups[key]

    # ------------------------------------------------------------------ pre-processing
    def _like(self, data):
        """A new signal around `data` with this one's custom attributes (the reference carries `detector`,
        `static_background` and `xmap` over, signals/ebsd.py:564-573, :686-696)."""
        out = EBSD(data, self.static_background, self.xmap, self.step_sizes, self.scan_unit, self._device, self._devices)
        out._detector = None if self._detector is None else self._detector.deepcopy()
        if hasattr(self, "original_metadata"):
            out.original_metadata = self.original_metadata
        return out

    def remove_static_background(self, operation="subtract", static_bg=None, scale_bg=False, show_progressbar=None,
                                 inplace=True, lazy_output=None, *, devices=None):
        """signals/ebsd.py:442-573, with the reference's parameters in the reference's order.  `show_progressbar`
        is accepted and has nothing to show (the whole signal is one kernel launch per GPU); `lazy_output=True`
        (only with `inplace=False`, as there) returns an ordinary signal - the result is computed on the device either
        way, this package has no lazy experimental signal."""
        if lazy_output and inplace:
            raise ValueError("'lazy_output=True' requires 'inplace=False'")
        if static_bg is None:
            static_bg = self.static_background
            if not isinstance(static_bg, np.ndarray) and not hasattr(static_bg, "compute"):
                raise ValueError("`EBSD.static_background` is not a valid array")
        contexts = self._member_contexts(devices, PREPROCESS_GROUP_MIN_POINTS)
        out = _pattern.remove_static_background(np.asarray(self.data), static_bg, operation, scale_bg,
                                                context=None if contexts else self.context, contexts=contexts)
        if inplace:
            self.data = out
            return None
        return self._like(out)

    def remove_dynamic_background(self, operation="subtract", filter_domain="frequency", std=None, truncate=4.0,
                                  show_progressbar=None, inplace=True, lazy_output=None, *, devices=None):
        """signals/ebsd.py:575-696; `show_progressbar` / `lazy_output` as in `remove_static_background`."""
        if lazy_output and inplace:
            raise ValueError("'lazy_output=True' requires 'inplace=False'")
        contexts = self._member_contexts(devices, PREPROCESS_GROUP_MIN_POINTS)
        out = _pattern.remove_dynamic_background(np.asarray(self.data), operation, filter_domain, std,
                                                 truncate, context=None if contexts else self.context, contexts=contexts)
        if inplace:
            self.data = out
            return None
        return self._like(out)

    # ------------------------------------------------------------------ refinement
    def _refine(self, mode, xmap, detector, master_pattern, energy, navigation_mask, signal_mask,
                pseudo_symmetry_ops, method, method_kwargs, trust_region, initial_step, rtol, maxeval, compute,
                verbose, comm=None, devices=None):
        from kikuchipy_amd.indexing._refinement import refine

        # the points are independent: with several GPUs (named, or all of them for a map worth it) each refines a block
        contexts = self._member_contexts(devices, REFINE_GROUP_MIN_POINTS) if comm is None else None
        return refine(mode, np.asarray(self.data), _rotations_of(xmap), detector, master_pattern, energy,
                      navigation_mask, signal_mask, pseudo_symmetry_ops, method, method_kwargs, trust_region,
                      initial_step, rtol, maxeval, context=None if contexts else self.context, verbose=verbose, comm=comm,
                      compute=compute, contexts=contexts, is_in_data=getattr(xmap, "is_in_data", None),
                      xmap_shape=getattr(xmap, "shape", None) if hasattr(xmap, "is_in_data") else None)

    def refine_orientation(self, xmap, detector, master_pattern, energy=None, navigation_mask=None,
                           signal_mask=None, pseudo_symmetry_ops=None, method="minimize", method_kwargs=None,
                           trust_region=None, initial_step=None, rtol=1e-4, maxeval=None, compute=True,
                           rechunk=True, chunk_kwargs=None, *, verbose=True, comm=None, devices=None):
        """signals/ebsd.py:1986-2185.  `xmap`: anything with `.rotations`
        (e.g. the result of `dictionary_indexing`) or a quaternion array.
        Returns a `RefinementResult` (`rotations`, `scores`, `num_evals`,
        `pseudo_symmetry_index`); with `compute=False` a `DeferredRefinement`, finished by
        `kikuchipy_amd.indexing.compute_refine_orientation_results` (the reference: a lazy Dask array)."""
        out = self._refine("ori", xmap, detector, master_pattern, energy, navigation_mask, signal_mask,
                           pseudo_symmetry_ops, method, method_kwargs, trust_region, initial_step, rtol, maxeval,
                           compute, verbose, comm, devices)
        return out[0] if compute else out  # compute=False: a DeferredRefinement (compute_refine_orientation_results)

    def refine_projection_center(self, xmap, detector, master_pattern, energy=None, navigation_mask=None,
                                 signal_mask=None, method="minimize", method_kwargs=None, trust_region=None,
                                 initial_step=None, rtol=1e-4, maxeval=None, compute=True, rechunk=True,
                                 chunk_kwargs=None, *, verbose=True, comm=None, devices=None):
        """signals/ebsd.py:2187-2390.  Returns `(scores, new_detector, num_evals)`
        like the reference."""
        out = self._refine("pc", xmap, detector, master_pattern, energy, navigation_mask, signal_mask, None,
                           method, method_kwargs, trust_region, initial_step, rtol, maxeval, compute, verbose, comm, devices)
        if not compute:
            return out  # a DeferredRefinement (compute_refine_projection_center_results)
        res, det = out
        return res.scores, det, res.num_evals

    def refine_orientation_projection_center(self, xmap, detector, master_pattern, energy=None,
                                             navigation_mask=None, signal_mask=None, pseudo_symmetry_ops=None,
                                             method="minimize", method_kwargs=None, trust_region=None,
                                             initial_step=None, rtol=1e-4, maxeval=None, compute=True,
                                             rechunk=True, chunk_kwargs=None, *, verbose=True, comm=None, devices=None):
        """signals/ebsd.py:2392-2700.  Returns `(RefinementResult, new_detector)`."""
        return self._refine("ori_pc", xmap, detector, master_pattern, energy, navigation_mask, signal_mask,
                            pseudo_symmetry_ops, method, method_kwargs, trust_region, initial_step, rtol, maxeval,
                            compute, verbose, comm, devices)

    # ------------------------------------------------------------------ indexing
    def dictionary_indexing(self, dictionary, metric="ncc", keep_n=20, n_per_iteration=None,
                            navigation_mask=None, signal_mask=None, rechunk=False, dtype=None, *,
                            devices=None, comm=None, verbose=True, compute=None):
        """See `kikuchipy_amd.dictionary_indexing`; `dictionary` is an `EBSD`
        with a 1-D navigation axis and an `xmap` of equal size.  As there, a call that names no
        device shards the dictionary over every visible GPU from this one process."""
        from kikuchipy_amd.indexing._resident_dictionary import ResidentDictionary

        if isinstance(dictionary, ResidentDictionary):
            # prepared once and kept in HBM: only the match runs (metric and signal mask are its own)
            if tuple(dictionary.shape[1:]) != self._signal_shape_rc:
                raise ValueError(
                    f"Experimental {self._signal_shape_rc} and dictionary {tuple(dictionary.shape[1:])} signal "
                    "shapes must be identical"
                )
            return _dictionary_indexing(
                self.data, dictionary, metric, keep_n, n_per_iteration, navigation_mask, signal_mask, rechunk,
                dtype, step_sizes=self.step_sizes, scan_unit=self.scan_unit, device=self._device, comm=comm,
                compute=compute, verbose=verbose,
            )
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
        from kikuchipy_amd.indexing.similarity_metrics import METRICS

        if isinstance(metric, str) and metric in METRICS:
            # on this signal's engine (one context, or a group over several GPUs): its device buffers - and a group's
            # communicator - are reused from call to call
            device, engine = self._engine(devices, comm, dict_size)
            metric = METRICS[metric](device=device, compute=compute, context=engine)
            metric.rechunk = rechunk
        res = _dictionary_indexing(
            self.data, dict_data, metric, keep_n, n_per_iteration, navigation_mask, signal_mask,
            rechunk, dtype, step_sizes=self.step_sizes, dictionary_rotations=dict_xmap.rotations,
            phase_name=getattr(dict_xmap, "phase_name", None), scan_unit=self.scan_unit, device=self._device, comm=comm,
            compute=compute,
            verbose=verbose,
        )
        # the phase list the reference hands to the returned CrystalMap (indexing/_dictionary_indexing.py:165): kept for
        # `to_crystal_map()` when the dictionary's crystal map is orix's
        res.phase_list = getattr(dict_xmap, "phases_in_data", None)
        return res


def _rotations_of(xmap):
    """Quaternions of an indexing result / crystal-map-like object / plain array."""
    rot = getattr(xmap, "rotations", xmap)
    return np.asarray(getattr(rot, "data", rot), dtype=np.float64)


class EBSDMasterPattern:
    """Master pattern in the square Lambert projection: the surface of
    `kikuchipy.signals.EBSDMasterPattern` that `get_patterns` reads.

    Parameters
    ----------
    data
        (npy, npx), (2, npy, npx) [hemisphere], (n_energies, npy, npx) [energy]
        or (2, n_energies, npy, npx) [hemisphere, energy], as HyperSpy orders the
        reference's navigation axes.
    projection
        Must be "lambert" for `get_patterns` (as in the reference).
    hemisphere
        "upper", "lower" or "both"; "both" needs the leading axis of size 2.
    energies
        Energy axis values in kV when `data` has an energy axis.
    has_inversion_symmetry
        Whether the phase's point group contains inversion
        (`phase.point_group.contains_inversion`; orix is not a dependency here).
        `None` = no valid point group.
    """

    def __init__(self, data, projection="lambert", hemisphere=None, energies=None, phase_name="",
                 has_inversion_symmetry=True, device=0):
        self.data = np.asarray(data)
        if self.data.ndim < 2 or self.data.ndim > 4:
            raise ValueError("master pattern data must have 2 signal axes and at most 2 navigation axes")
        # (_utils/vector.py:47-60, _utils/exceptions.py:21-36: case-insensitive, the reference's texts)
        if not isinstance(projection, str) or projection.lower() not in ("stereographic", "lambert"):
            raise ValueError(f"Unknown projection {projection!r}, options are 'stereographic' and 'lambert'")
        self.projection = projection.lower()
        self.energies = None if energies is None else np.asarray(energies, dtype=np.float64)
        n_nav = self.data.ndim - 2
        has_energy = self.energies is not None
        if hemisphere is None:
            hemisphere = "both" if n_nav - int(has_energy) == 1 else "upper"
        if not isinstance(hemisphere, str) or hemisphere.lower() not in ("upper", "lower", "both"):
            raise ValueError(f"Unknown hemisphere {hemisphere!r}, options are 'upper', 'lower', or 'both'")
        hemisphere = hemisphere.lower()
        self.hemisphere = hemisphere
        expected_nav = int(hemisphere == "both") + int(has_energy)
        if expected_nav != n_nav or (hemisphere == "both" and self.data.shape[0] != 2):
            raise ValueError(
                f"data of shape {self.data.shape} does not match hemisphere='{hemisphere}' and "
                f"{'an' if has_energy else 'no'} energy axis"
            )
        if has_energy and self.data.shape[n_nav - 1] != self.energies.size:
            raise ValueError("`energies` must have one value per master pattern along the energy axis")
        self.phase_name = phase_name
        self.has_inversion_symmetry = has_inversion_symmetry
        self._device = device

    @property
    def _has_multiple_energies(self):
        return self.energies is not None

    # signals/ebsd_master_pattern.py:331-377
    def _is_suitable_for_projection(self, raise_if_not=False):
        error = None
        if self.projection != "lambert":
            error = NotImplementedError("Master pattern must be in the square Lambert projection")
        if self.has_inversion_symmetry is None:
            error = AttributeError("Master pattern `phase` attribute must have a valid point group")
        elif self.hemisphere != "both" and not self.has_inversion_symmetry:
            error = AttributeError(
                "For point groups without inversion symmetry, both hemispheres must be present in the "
                "master pattern signal"
            )
        if error is not None and raise_if_not:
            raise error
        return error is None

    # signals/_kikuchi_master_pattern.py:303-345
    def _get_master_pattern_arrays_from_energy(self, energy=None):
        data = self.data
        if self._has_multiple_energies:
            if energy is None:
                energy = self.energies[-1]
            # HyperSpy float indexing (`inav[float]`): the axis value closest to `energy`
            idx = int(np.argmin(np.abs(self.energies - float(energy))))
            data = data[:, idx] if self.hemisphere == "both" else data[idx]
        if self.hemisphere == "both":
            return data[0], data[1]
        return data, data

    def get_patterns(self, rotations, detector, energy=None, dtype_out="float32", compute=False,
                     show_progressbar=None, **kwargs):
        """Patterns projected onto `detector`, one per rotation.

        `rotations`: (..., 4) unit quaternions (a, b, c, d) with at most two
        leading axes (an orix `Rotation`'s `.data` works as is).  Returns an
        `EBSD` whose `data` is a NumPy array (`compute=True`) or a lazy
        `ProjectedDictionary` (`compute=False`, the default, like the
        reference's `LazyEBSD`), with `xmap` holding the rotations.  `chunk_shape`
        in `kwargs` sets the number of patterns per lazy chunk."""
        self._is_suitable_for_projection(raise_if_not=True)
        rot = np.asarray(getattr(rotations, "data", rotations), dtype=np.float64)
        if detector.navigation_size != 1 and rot.shape[:-1] != detector.navigation_shape:
            raise ValueError(
                "`detector.navigation_shape` must be equal to `rotations.shape`, or the"
                " detector must have exactly one projection center"
            )
        if rot.shape[-1] != 4:
            raise ValueError("`rotations` must be an array of quaternions with a last axis of size 4")
        nav_shape = rot.shape[:-1] if rot.ndim > 1 else (1,)
        if len(nav_shape) > 2:
            raise ValueError(
                "`rotations` can only have one or two dimensions, but an instance with "
                f"{len(nav_shape)} dimensions was passed"
            )
        dtype_out = np.dtype(dtype_out)
        # signals/ebsd_master_pattern.py:224-233
        if dtype_out != self.data.dtype:
            rescale = True
            out_min, out_max = DTYPE_RANGE[dtype_out]
        else:
            rescale = False
            out_min, out_max = 1, 2
        master_upper, master_lower = self._get_master_pattern_arrays_from_energy(energy)
        # one PC per rotation (signals/ebsd_master_pattern.py:236-241, :274-283: `nav_shape_det != (1,)`): the lazy
        # dictionary carries the PCs and every chunk is projected with them on the device, inside the indexing loop
        pcs = detector.pc_flattened if detector.navigation_size != 1 else None
        lazy = ProjectedDictionary(np.ascontiguousarray(master_upper), np.ascontiguousarray(master_lower),
                                   rot.reshape(-1, 4), detector, rescale, out_min, out_max, dtype_out,
                                   device=self._device, chunk=kwargs.get("chunk_shape"), pcs=pcs)
        xmap = DictionaryXmap(rot.reshape(-1, 4), self.phase_name)
        if compute:
            data = lazy.compute().reshape(nav_shape + detector.shape)
        elif len(nav_shape) == 1:
            data = lazy
        else:
            raise NotImplementedError("lazy output needs a 1D array of rotations (pass compute=True)")
        out = EBSD(data, xmap=xmap, device=self._device)
        out.detector = detector
        return out
