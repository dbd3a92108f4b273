"""Stand-alone dictionary indexing on the GPU engine.

Counterpart of `EBSD.dictionary_indexing` (signals/ebsd.py:1827-1984 of the
reference) + `_dictionary_indexing` (indexing/_dictionary_indexing.py:36-169)
for plain arrays: same keyword arguments, same validation and messages, same
`scores` / `simulation_indices` outputs.  The difference to running the metric
plugin inside kikuchipy's own loop is that here the running best-k stays on
the GPU across dictionary chunks (the merge of :120-128 is a kernel) and, with
a communicator, the dictionary can be sharded over several GPUs.
"""

import time

import numpy as np

from kikuchipy_amd.indexing.similarity_metrics import METRICS, SimilarityMetric, _HipMetric


class MapData:
    """`get_map_data` of orix's `CrystalMap` (third party, not vendored; used by the reference's tutorials on the maps
    this path returns) for the result holders of this package."""

    def get_map_data(self, item, decimals=None, fill_value=np.nan):
        """A property (by name: "scores", "simulation_indices", "num_evals", ... or an array with one row per point
        in the data) laid out on the map: shape `self.shape` (+ the property's trailing axes), points outside the
        navigation mask filled with `fill_value` (integers: `fill_value` if it is finite, else 0)."""
        if isinstance(item, str):
            if item not in self.prop:
                raise ValueError(f"{item!r} is not among the properties {sorted(self.prop)}")
            values = np.asarray(self.prop[item])
        else:
            values = np.asarray(item)
        in_data = np.asarray(self.is_in_data, dtype=bool).ravel()
        if values.shape[0] != int(in_data.sum()):
            raise ValueError(f"{values.shape[0]} values for {int(in_data.sum())} points in the data")
        if np.issubdtype(values.dtype, np.integer) or values.dtype == bool:
            fill = fill_value if np.isfinite(fill_value) else 0
        else:
            fill = fill_value
        out = np.full((in_data.size,) + values.shape[1:], fill, dtype=values.dtype)
        out[in_data] = values
        if decimals is not None:
            out = np.round(out, decimals=decimals)
        return out.reshape(tuple(self.shape) + values.shape[1:])


class DictionaryIndexingResult(MapData):
    """What the reference stores in the returned `CrystalMap`
    (indexing/_dictionary_indexing.py:141-167): `scores` and
    `simulation_indices` of shape (n_points, keep_n) - squeezed to 1-D when
    `keep_n == 1` and a navigation mask is used, as there - plus `is_in_data`
    and, if dictionary rotations were given, `rotations` (quaternions of the
    matched dictionary entries, identity where masked out)."""

    def __init__(self, scores, simulation_indices, nav_shape, step_sizes, is_in_data, keep_n,
                 rotations=None, phase_name=None, scan_unit=None, patterns_per_second=None,
                 comparisons_per_second=None, float64_certificate=None):
        self.scores = scores
        self.simulation_indices = simulation_indices
        self.shape = tuple(nav_shape)
        self.step_sizes = tuple(step_sizes)
        self.is_in_data = is_in_data
        self.rotations_per_point = keep_n
        self.rotations = rotations
        self.phase_name = phase_name
        self.scan_unit = scan_unit
        self.patterns_per_second = patterns_per_second
        self.comparisons_per_second = comparisons_per_second
        # float64 arithmetic (`dtype=float64`): how the float32 screen was certified - {"mode": "worstcase" (the
        # default) | "statistical" (KPDI_F64_EPS=statistical), "uncertified_patterns": n}; a result is the exact float64
        # best-k for ANY data with mode "worstcase" and 0 uncertified patterns (include/kpdi.h, KPDI_COMPUTE_F64); else None
        self.float64_certificate = float64_certificate
        self.phase_list = None  # (`EBSD.dictionary_indexing`: the dictionary crystal map's `phases_in_data`, if it has one)

    @property
    def size(self):
        """Number of indexed points (`xmap.size` in the reference's tests)."""
        return int(np.count_nonzero(self.is_in_data))

    @property
    def prop(self):
        return {"scores": self.scores, "simulation_indices": self.simulation_indices}

    def to_crystal_map(self, phase_list=None):
        """Build an `orix.crystal_map.CrystalMap` exactly as the reference does
        (needs orix, which is not a dependency of this package)."""
        from orix.crystal_map import CrystalMap, create_coordinate_arrays
        from orix.quaternion import Rotation

        if phase_list is None:
            phase_list = self.phase_list
        kw, _ = create_coordinate_arrays(self.shape, self.step_sizes)
        if self.rotations is None:
            raise ValueError("dictionary_rotations were not given to dictionary_indexing()")
        kw["rotations"] = Rotation(self.rotations)
        kw["prop"] = self.prop
        if not np.all(self.is_in_data):
            kw["is_in_data"] = self.is_in_data
        xmap = CrystalMap(phase_list=phase_list, **kw)
        if self.scan_unit is not None:
            xmap.scan_unit = self.scan_unit
        return xmap


def info_message(metric, n_experimental_all, dictionary_size, phase_name, n_experimental=None):
    """Text of indexing/_dictionary_indexing.py:206-237."""
    info = f"Dictionary indexing information:\n  Phase name: {phase_name}\n"
    if n_experimental is not None and n_experimental != n_experimental_all:
        info += f"  Matching {n_experimental}/{n_experimental_all} experimental pattern(s)"
    else:
        info += f"  Matching {n_experimental_all} experimental pattern(s)"
    info += f" to {dictionary_size} dictionary pattern(s)\n  {metric}"
    return info


def _is_lazy(a):
    return hasattr(a, "compute") and hasattr(a, "chunksize")


# below this many comparisons (experimental x dictionary patterns) a call that did not ask for devices stays on one GPU:
# a group's set-up (one context per device, the in-process communicator) would cost more than the sweep
GROUP_MIN_COMPARISONS = 1 << 26


def pick_devices(device, devices, comm, n_experimental, dictionary_size):
    """Which GPU(s) a `dictionary_indexing` call runs on -> (device, devices) for `make_engine`.
    `comm` (one process per GPU): the rank's own device.  `devices` given: those.  `device` given:
    that one.  Neither: every visible GPU ($KPDI_DEVICES overrides; ONE GPU in a process started by a
    launcher, `_lib.default_devices`) - the reference call uses every core of the host without being
    asked (signals/ebsd.py:1827-1984 on Dask's default scheduler) - unless the job is too small to be
    worth a group."""
    from kikuchipy_amd import _lib

    if comm is not None:
        return (0 if device is None else device), None
    if devices is not None:
        return 0, _lib.resolve_devices(devices)
    if device is not None:
        return device, None
    ids = _lib.default_devices()
    if len(ids) > 1 and n_experimental * dictionary_size < GROUP_MIN_COMPARISONS:
        ids = ids[:1]
    return ids[0], (ids if len(ids) > 1 else None)


def prepare_metric(metric, navigation_mask, signal_mask, dtype, rechunk, n_experimental,
                   n_dictionary_patterns, device=0, compute=None, devices=None):
    """EBSD._prepare_metric (signals/ebsd.py:3049-3088)."""
    if isinstance(metric, str) and metric in METRICS:
        metric = METRICS[metric](device=device, devices=devices, compute=compute)
        metric.rechunk = rechunk
    if not isinstance(metric, SimilarityMetric):
        raise ValueError(
            f"'{metric}' must be either of {METRICS.keys()} or a custom metric class "
            "inheriting from SimilarityMetric. See kikuchipy.indexing.SimilarityMetric"
        )
    metric.n_experimental_patterns = max(n_experimental, 1)
    metric.n_dictionary_patterns = max(n_dictionary_patterns, 1)
    if navigation_mask is not None:
        metric.navigation_mask = navigation_mask
    if signal_mask is not None:
        metric.signal_mask = signal_mask
    if dtype is not None:
        metric.dtype = dtype
    metric.raise_error_if_invalid()
    return metric


def _uncertified(ctx):
    """(pattern, chunk) pairs the float64 mode could not certify so far, over every device of the engine."""
    c = ctx.counters()
    return int(sum(m.get("uncertified_patterns", 0) for m in c.get("members", [c])))


def chunk_bounds(dictionary_size, n_per_iteration):
    """Chunk starts/ends of indexing/_dictionary_indexing.py:100-104."""
    n_iterations = int(np.ceil(dictionary_size / n_per_iteration))
    starts = np.cumsum([0] + [n_per_iteration] * (n_iterations - 1))
    ends = np.cumsum([n_per_iteration] * n_iterations)
    ends[-1] = max(ends[-1], dictionary_size)
    return [(int(s), int(min(e, dictionary_size))) for s, e in zip(starts, ends)]


def dictionary_indexing(
    experimental,
    dictionary,
    metric="ncc",
    keep_n=20,
    n_per_iteration=None,
    navigation_mask=None,
    signal_mask=None,
    rechunk=False,
    dtype=None,
    *,
    step_sizes=None,
    dictionary_rotations=None,
    phase_name="",
    scan_unit=None,
    device=None,
    devices=None,
    comm=None,
    verbose=True,
    compute=None,
    progress=None,
):
    """Index experimental patterns against a dictionary of simulated patterns.

    Parameters (the first nine are those of `EBSD.dictionary_indexing`,
    signals/ebsd.py:1827-1918)
    ----------
    experimental
        Array (..., sy, sx) with 0, 1 or 2 navigation axes.
    dictionary
        Array (N, sy, sx): NumPy, or lazy (Dask-like with `.chunksize` and
        `.compute()`), in which case chunks are computed one at a time inside
        the loop as in the reference (:106-108), or the `ProjectedDictionary`
        of `EBSDMasterPattern.get_patterns()`, whose chunks are simulated
        directly in device memory, or a `ResidentDictionary` (prepared once and
        kept in HBM for a series of maps; it brings its own metric and signal mask).
    metric
        "ncc", "ndp" or an instance of this package's metrics.
    keep_n, n_per_iteration, navigation_mask, signal_mask, rechunk, dtype
        As in the reference.  `n_per_iteration` bounds how many dictionary
        patterns are uploaded and matched per iteration.
    step_sizes, dictionary_rotations, phase_name, scan_unit
        What the reference takes from the signals' axes managers and from
        `dictionary.xmap` (rotations as an (N, 4) quaternion array).
    compute
        Arithmetic of the match kernel: None (default) = "f64" when `dtype` is float64 (float64
        arithmetic like the reference: float32 screening + float64 rescoring), else "f32"; or the
        opt-in "f16x2" / "f16", see `NormalizedCrossCorrelationMetric`; ignored when `metric` is an
        instance.
    device, devices
        Where a metric given by NAME runs.  Neither given (the default): every visible GPU - the
        dictionary is sharded over them inside this one process (`kikuchipy_amd._lib.Group`: one
        host copy of the inputs, every dictionary chunk block-assigned to the devices, the
        per-device best-k lists merged by an in-process RCCL all-gather) - or one GPU when there is
        one, or when the job is tiny; $KPDI_DEVICES ("all" or ids like "0,1") overrides.  `devices="all"`
        / a list of ids / `device=i` say it explicitly.  Results do not depend on the choice.
    progress
        The reference shows a `tqdm` bar over the chunk loop (indexing/_dictionary_indexing.py:105).  None
        (default): the same bar when `verbose`, the dictionary takes more than one iteration and tqdm is
        importable; a callable: called as `progress(chunks_done, n_chunks)` after every pushed chunk; False:
        nothing.  (A push returns when the chunk has been handed over - its sweep may still run.)
    comm
        `kikuchipy_amd.parallel.Communicator` to shard the dictionary over
        ranks (one process per GPU).  Every rank must pass the same arrays; each
        matches its own contiguous block and all ranks return the global result.
    """
    from kikuchipy_amd.indexing._resident_dictionary import ResidentDictionary

    resident = dictionary if isinstance(dictionary, ResidentDictionary) else None
    if resident is not None:
        resident.check_call(metric, signal_mask, comm)
        metric, signal_mask = resident.metric, resident.signal_mask
        metric.navigation_mask = None  # of an earlier map
        if dictionary_rotations is None:
            dictionary_rotations = resident.rotations
        phase_name = phase_name or resident.phase_name
    experimental = experimental if _is_lazy(experimental) else np.asarray(experimental)
    if experimental.ndim < 2 or experimental.ndim > 4:
        raise ValueError("experimental patterns must have 0, 1 or 2 navigation axes and 2 signal axes")
    if dictionary.ndim != 3:
        raise ValueError(
            "Dictionary signal must have a non-empty `EBSD.xmap` attribute of equal size as the "
            "number of dictionary patterns, and both the signal and crystal map must have only one "
            "navigation dimension"
        )
    dict_size = dictionary.shape[0]
    nav_shape_exp = tuple(experimental.shape[:-2])

    # ---- signals/ebsd.py:1925-1964
    if n_per_iteration is None:
        n_per_iteration = dictionary.chunksize[0] if _is_lazy(dictionary) else dict_size
    if navigation_mask is not None:
        if navigation_mask.shape != nav_shape_exp:
            raise ValueError(
                f"The navigation mask shape {navigation_mask.shape} and the "
                f"signal's navigation shape {nav_shape_exp} must be identical"
            )
        elif navigation_mask.all():
            raise ValueError(
                "The navigation mask must allow for indexing of at least one "
                "pattern (at least one value equal to `False`)"
            )
        elif not isinstance(navigation_mask, np.ndarray):
            raise ValueError("The navigation mask must be a NumPy array")
    if signal_mask is not None:
        if not isinstance(signal_mask, np.ndarray):
            raise ValueError("The signal mask must be a NumPy array")
    sig_shape_exp = tuple(experimental.shape[-2:])
    sig_shape_dict = tuple(dictionary.shape[-2:])
    if sig_shape_exp != sig_shape_dict:
        raise ValueError(
            f"Experimental {sig_shape_exp} and dictionary {sig_shape_dict} signal "
            "shapes must be identical"
        )
    if dictionary_rotations is not None:
        # (an orix `Rotation` keeps its quaternions in `.data`)
        dictionary_rotations = np.asarray(getattr(dictionary_rotations, "data", dictionary_rotations))
        if dictionary_rotations.shape != (dict_size, 4):
            raise ValueError(
                "Dictionary signal must have a non-empty `EBSD.xmap` attribute of equal size as "
                "the number of dictionary patterns, and both the signal and crystal map must have "
                "only one navigation dimension"
            )

    n_experimental_all = int(np.prod(nav_shape_exp)) if nav_shape_exp else 1
    own_engine = isinstance(metric, str) and metric in METRICS  # the metric - and its engine - are this call's own
    if own_engine:
        device, devices = pick_devices(device, devices, comm, n_experimental_all, dict_size)
    metric = prepare_metric(metric, navigation_mask, signal_mask, dtype, rechunk, n_experimental_all,
                            dict_size, device=device, compute=compute, devices=devices)
    if not isinstance(metric, _HipMetric):
        raise ValueError("the stand-alone driver runs the GPU metrics of kikuchipy_amd only")
    if own_engine:
        metric._pooled_engine = True  # its engine: the idle one an earlier call left, if any (_lib.acquire_engine)
    try:
        return _run(experimental, dictionary, metric, keep_n, n_per_iteration, resident, dict_size, nav_shape_exp,
                    n_experimental_all, step_sizes, dictionary_rotations, phase_name, scan_unit, verbose, progress, comm,
                    own_engine)
    except BaseException:
        if own_engine and metric._ctx is not None:
            # the engine was made for this call and the call failed half-way: its contexts, host threads and communicator
            # go with it now, not when the garbage collector finds the metric (a healthy one is kept: _lib.release_engine)
            ctx, metric._ctx = metric._ctx, None
            ctx.close()
        raise


def _run(experimental, dictionary, metric, keep_n, n_per_iteration, resident, dict_size, nav_shape_exp, n_experimental_all,
         step_sizes, dictionary_rotations, phase_name, scan_unit, verbose, progress, comm, own_engine):
    """The sweep of `dictionary_indexing` once its arguments are checked and its metric is made."""

    # ---- indexing/_dictionary_indexing.py:66-128
    keep_n = min(keep_n, dict_size)
    prepared = metric.prepare_experimental(experimental)
    n_experimental = prepared.shape[0]
    if verbose:
        print(info_message(metric, n_experimental_all, dict_size, phase_name, n_experimental))

    ctx = metric.context
    ctx.set_keep_n(keep_n)
    if resident is None:
        ctx.set_dictionary_size(dict_size)  # (a group plans which member takes which chunk: kpdi_group_set_dictionary_size)
    rank, world = (comm.rank, comm.world_size) if comm is not None else (0, 1)
    if comm is not None:
        comm.attach(ctx)
    from kikuchipy_amd.parallel import shard_range

    lo, hi = shard_range(dict_size, rank, world)
    bounds = [] if resident is not None else chunk_bounds(dict_size, n_per_iteration)
    bar = None
    if progress is None and verbose and len(bounds) > 1 and rank == 0:
        try:
            from tqdm import tqdm

            bar = tqdm(total=len(bounds))
        except ImportError:
            pass
    time_start = time.time()
    f64 = metric.effective_compute == "f64"
    try:
        if resident is not None:
            ctx.sweep_held()
        for n_done, (start, end) in enumerate(bounds, 1):
            start, end = max(start, lo), min(end, hi)  # this rank's part of the chunk
            if start < end:
                chunk = dictionary[start:end]
                if hasattr(chunk, "push_to_engine"):
                    # simulated on the device from (master pattern, detector, rotations): the chunk
                    # never exists on the host (kikuchipy_amd.simulations.ProjectedDictionary)
                    chunk.push_to_engine(ctx, start)
                else:
                    if _is_lazy(chunk):
                        chunk = chunk.compute()
                    ctx.push_dictionary_chunk(np.asarray(chunk), start)
            # AFTER the chunk has been handed over (or found to belong to another rank); rank 0 only, like the bar
            if bar is not None:
                bar.update(1)
            elif callable(progress) and rank == 0:
                progress(n_done, len(bounds))
        uncertified_before = _uncertified(ctx) if f64 else 0
        scores, simulation_indices = ctx.finalize(keep_n)
    finally:
        if bar is not None:
            bar.close()
    total_time = time.time() - time_start
    certificate = None
    if f64:
        from kikuchipy_amd import _lib

        certificate = {"mode": _lib.F64_CERTIFICATES.get(ctx.counters().get("f64_certificate", 0)),
                       "uncertified_patterns": _uncertified(ctx) - uncertified_before}
    if own_engine and metric._ctx is not None:
        # the engine was made for this call: it is handed back - kept, idle, for the next call on the same devices, or
        # closed with its contexts, host threads and communicator (_lib.release_engine; a caller that wants an engine of
        # its own passes a metric instance, or indexes through an `EBSD`, which keeps its engines)
        from kikuchipy_amd import _lib

        engine, metric._ctx = metric._ctx, None
        _lib.release_engine(engine)
    scores = scores.astype(metric.dtype, copy=False)
    pps = n_experimental / total_time
    cps = n_experimental * dict_size / total_time
    if verbose:
        print(f"  Indexing speed: {pps:.5f} patterns/s, {cps:.5f} comparisons/s")

    # ---- indexing/_dictionary_indexing.py:141-167
    if step_sizes is None:
        step_sizes = (1,) * len(nav_shape_exp)
    rotations = None
    if metric.navigation_mask is not None:
        in_data = ~np.asarray(metric.navigation_mask).ravel()
        scores_all = np.zeros((n_experimental_all, keep_n), dtype=scores.dtype)
        scores_all[in_data] = scores
        indices_all = np.zeros((n_experimental_all, keep_n), dtype=simulation_indices.dtype)
        indices_all[in_data] = simulation_indices
        if dictionary_rotations is not None:
            rotations = np.zeros((n_experimental_all, keep_n, 4), dtype=dictionary_rotations.dtype)
            rotations[..., 0] = 1  # identity quaternion where masked out
            rotations[in_data] = dictionary_rotations[simulation_indices]
        if keep_n == 1:
            scores_all = scores_all.squeeze()
            indices_all = indices_all.squeeze()
            if rotations is not None:
                rotations = rotations.reshape(-1, 4)
        scores, simulation_indices = scores_all, indices_all
    else:
        in_data = np.ones(n_experimental_all, dtype=bool)
        if dictionary_rotations is not None:
            rotations = dictionary_rotations[simulation_indices]
    return DictionaryIndexingResult(
        scores, simulation_indices, nav_shape_exp, step_sizes, in_data, keep_n, rotations=rotations,
        phase_name=phase_name, scan_unit=scan_unit, patterns_per_second=pps, comparisons_per_second=cps,
        float64_certificate=certificate,
    )
