"""A dictionary prepared once and kept in HBM for a series of maps.

The reference's loop prepares every dictionary chunk inside every
`dictionary_indexing()` call (`metric.prepare_dictionary`,
indexing/_dictionary_indexing.py:106-110): the dictionary lives in host memory
(or is simulated lazily) and is streamed through the matcher once per map.  A
lab that indexes map after map of the same phase on the same detector repeats
that upload (6.5 GB of PCIe traffic for 100 000 patterns of 60 x 60 float32, or
the simulation) every time.  On an MI355X the prepared dictionary is 1.45 GB per
100 000 patterns of 288 GB, so it can simply stay: `ResidentDictionary` is
uploaded / simulated and prepared once (`kpdi_hold_*`, include/kpdi.h), and
`dictionary_indexing(resident)` only runs the match + top-k + merge of the held
chunks (`kpdi_sweep_held`).  Results are identical to passing the patterns again.

The metric ("ncc" / "ndp", arithmetic) and the signal mask shape the prepared
layout, so they belong to the resident dictionary, not to the indexing call.
"""

import numpy as np

from kikuchipy_amd.indexing.similarity_metrics import METRICS, _HipMetric


class ResidentDictionary:
    """Parameters
    ----------
    dictionary
        (N, sy, sx) patterns: NumPy, lazy (Dask-like) or the `ProjectedDictionary` of
        `EBSDMasterPattern.get_patterns()` (then simulated in device memory).
    metric
        "ncc", "ndp" or an instance of this package's metrics.
    signal_mask
        Boolean (sy, sx), True = pixel not used.
    n_per_iteration
        Patterns uploaded / simulated per step (bounds the staging memory), default
        the dictionary's chunk size or everything.
    dictionary_rotations, phase_name
        Passed on to the results of `dictionary_indexing`.
    device, devices
        The GPU, or - `devices="all"` / a list of ids - the GPUs of one process the dictionary is
        sharded over (every chunk block-assigned; each device keeps its part prepared).
    comm
        `kikuchipy_amd.parallel.Communicator`: every rank holds its own shard.
    """

    def __init__(self, dictionary, metric="ncc", signal_mask=None, n_per_iteration=None, *,
                 dictionary_rotations=None, phase_name="", device=0, devices=None, compute="f32", comm=None):
        from kikuchipy_amd.indexing._dictionary_indexing import _is_lazy, chunk_bounds
        from kikuchipy_amd.parallel import shard_range

        if dictionary.ndim != 3:
            raise ValueError("the dictionary must have shape (n patterns, detector rows, detector columns)")
        if isinstance(metric, str):
            if metric not in METRICS:
                raise ValueError(f"'{metric}' must be either of {METRICS.keys()}")
            metric = METRICS[metric](device=device, devices=None if comm is not None else devices, compute=compute)
        if not isinstance(metric, _HipMetric):
            raise ValueError("a resident dictionary needs one of the GPU metrics of kikuchipy_amd")
        if metric.compute == "f64":
            raise ValueError("compute='f64' rescoring reads the raw dictionary patterns; a resident dictionary keeps only "
                             "their prepared form - index with the dictionary itself instead")
        if metric.compute is None:
            metric.compute = "f32"  # (dtype=float64 then returns float32 arithmetic as float64, with a warning)
        if signal_mask is not None and not isinstance(signal_mask, np.ndarray):
            raise ValueError("The signal mask must be a NumPy array")
        if dictionary_rotations is not None:
            dictionary_rotations = np.asarray(dictionary_rotations)
            if dictionary_rotations.shape != (dictionary.shape[0], 4):
                raise ValueError("dictionary_rotations must be an (N, 4) quaternion array")
        self.metric = metric
        self.signal_mask = signal_mask
        self.shape = tuple(dictionary.shape)
        self.ndim = 3
        self.rotations = dictionary_rotations
        self.phase_name = phase_name
        self.comm = comm
        metric.signal_mask = signal_mask
        metric._set_problem(self.shape[1:], 1)
        ctx = metric.context
        ctx.release_held()
        ctx.set_dictionary_size(self.shape[0])  # (a group plans which member holds which chunk)
        if n_per_iteration is None:
            n_per_iteration = dictionary.chunksize[0] if _is_lazy(dictionary) else self.shape[0]
        rank, world = (comm.rank, comm.world_size) if comm is not None else (0, 1)
        lo, hi = shard_range(self.shape[0], rank, world)
        for start, end in chunk_bounds(self.shape[0], n_per_iteration):
            start, end = max(start, lo), min(end, hi)
            if start >= end:
                continue
            chunk = dictionary[start:end]
            if hasattr(chunk, "hold_in_engine"):
                chunk.hold_in_engine(ctx, start)
                continue
            if _is_lazy(chunk):
                chunk = chunk.compute()
            ctx.hold_dictionary_chunk(np.asarray(chunk), start)
        ctx.synchronize()

    @classmethod
    def from_signal(cls, dictionary, metric="ncc", signal_mask=None, n_per_iteration=None, **kwargs):
        """From a dictionary `EBSD` signal with an `xmap` (rotations and phase name are taken from it)."""
        xmap = dictionary.xmap
        if xmap is not None:
            kwargs.setdefault("dictionary_rotations", xmap.rotations)
            kwargs.setdefault("phase_name", xmap.phase_name)
        kwargs.setdefault("device", getattr(dictionary, "_device", 0))
        return cls(dictionary.data, metric, signal_mask, n_per_iteration, **kwargs)

    def __len__(self):
        return self.shape[0]

    @property
    def held(self):
        """(patterns held on this rank's GPU, bytes of HBM they occupy)."""
        return self.metric.context.held_size()

    def release(self):
        """Free the HBM; the object cannot be used for indexing afterwards."""
        self.metric.context.release_held()

    def check_call(self, metric, signal_mask, comm):
        """`dictionary_indexing(resident, metric=..., signal_mask=...)`: what is given there must
        be what the dictionary was prepared with."""
        if isinstance(metric, str):
            if METRICS.get(metric) is not type(self.metric):
                raise ValueError(f"the resident dictionary was prepared for {type(self.metric).__name__}, "
                                 f"not '{metric}'")
        elif metric is not self.metric:
            raise ValueError("pass the resident dictionary's own metric (or its name)")
        if signal_mask is not None:
            if self.signal_mask is None or not np.array_equal(signal_mask, self.signal_mask):
                raise ValueError("the signal mask differs from the one the resident dictionary was prepared with")
        if comm is not self.comm:
            raise ValueError("the resident dictionary was sharded with another communicator")
        n, _ = self.held
        if n == 0:
            raise ValueError("the resident dictionary has been released (or another problem was set on its engine)")
