"""Dictionary generation: `EBSDMasterPattern.get_patterns`
(signals/ebsd_master_pattern.py:95-330 of the reference) on the GPU engine.

`ProjectedDictionary` is what `get_patterns(..., compute=False)` hands back in
place of the reference's Dask array: an (N, rows, cols) lazy array that knows
how its patterns are made (master pattern, detector, one rotation per
pattern).  `kikuchipy_amd.dictionary_indexing` recognises it and has the
engine generate every dictionary chunk directly in device memory
(`kpdi_push_rotations_chunk`), so the dictionary never crosses PCIe and is
never materialised on the host; `.compute()` materialises it (also on the
GPU) for any other use.
"""

import numpy as np

from kikuchipy_amd import _lib

# skimage.util.dtype.dtype_range as used at signals/ebsd_master_pattern.py:226-227
DTYPE_RANGE = {
    np.dtype(np.float32): (-1.0, 1.0),
    np.dtype(np.float64): (-1.0, 1.0),
    np.dtype(np.uint8): (0.0, 255.0),
    np.dtype(np.uint16): (0.0, 65535.0),
}


class ProjectedDictionary:
    """Lazy (N, rows, cols) array of simulated patterns; quacks like the Dask
    array the reference returns as far as dictionary indexing reads it
    (`ndim`, `shape`, `dtype`, `chunksize`, slicing along axis 0, `compute()`)."""

    def __init__(self, master_upper, master_lower, rotations, detector, rescale, out_min, out_max,
                 dtype_out=np.float32, device=0, chunk=None, _root=None, pcs=None):
        """`pcs`: None - the detector's one projection centre for every pattern; (N, 3) - one PC per pattern (a detector
        with a PC for every rotation: signals/ebsd_master_pattern.py:236-241, :274-283 of the reference)."""
        self.master_upper = master_upper
        self.master_lower = master_lower
        self.rotations = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        self.pcs = None if pcs is None else np.ascontiguousarray(pcs, dtype=np.float64).reshape(-1, 3)
        if self.pcs is not None and self.pcs.shape[0] != self.rotations.shape[0]:
            raise ValueError(f"{self.rotations.shape[0]} rotations but {self.pcs.shape[0]} projection centres")
        self.detector = detector
        self.rescale = bool(rescale)
        self.out_min, self.out_max = float(out_min), float(out_max)
        self.dtype = np.dtype(dtype_out)
        if self.dtype not in DTYPE_RANGE:
            raise ValueError(f"dtype_out {self.dtype} is not supported (float32, float64, uint8, uint16)")
        self.device = device
        self._chunk = chunk
        self._root = _root if _root is not None else self  # slices share the root's context
        self._ctx = None

    # ---- array protocol
    @property
    def shape(self):
        return (self.rotations.shape[0],) + self.detector.shape

    ndim = 3

    @property
    def chunksize(self):
        n = self.rotations.shape[0]
        if self._chunk is None:
            # 8 GiB of float32 patterns per iteration: a chunk only ever exists in device memory
            # (raw + prepared = 16 of the 288 GiB), and fewer, larger sweeps waste less on launch tails
            per = max(1, (8 << 30) // (4 * self.detector.size))
            return (min(n, per),) + self.detector.shape
        return (min(n, self._chunk),) + self.detector.shape

    def __len__(self):
        return self.rotations.shape[0]

    def __getitem__(self, key):
        if isinstance(key, tuple):
            if len(key) != 1 and any(k != slice(None) for k in key[1:]):
                raise IndexError("a ProjectedDictionary can only be sliced along its first axis")
            key = key[0]
        if isinstance(key, (int, np.integer)):
            return self[key:key + 1 if key != -1 else None].compute()[0]
        if not isinstance(key, slice):
            raise IndexError("a ProjectedDictionary can only be sliced along its first axis")
        return ProjectedDictionary(self.master_upper, self.master_lower, self.rotations[key], self.detector,
                                   self.rescale, self.out_min, self.out_max, self.dtype, self.device,
                                   self._chunk, _root=self._root, pcs=None if self.pcs is None else self.pcs[key])

    # ---- engine
    def configure(self, ctx):
        """Make `ctx` hold this dictionary's master pattern and detector (once)."""
        # `Context.set_master_pattern / set_detector / set_direction_cosines` reset the key, so a
        # refinement on the same context (which loads ITS master pattern) cannot leave a stale match;
        # the detector enters by value: an in-place change of its PC must reach the engine
        det = self.detector
        if self.pcs is not None:  # one PC per pattern: the direction cosines are formed on the device, per pattern
            key = (id(self.master_upper), id(self.master_lower), "one PC per pattern")
            if getattr(ctx, "_projection_key", None) != key:
                ctx.set_master_pattern(self.master_upper, self.master_lower)
                ctx._projection_key = key
                ctx._projection_refs = (self.master_upper, self.master_lower, det)
            return
        key = (id(self.master_upper), id(self.master_lower), tuple(np.ravel(det.gnomonic_bounds)), float(det.pcz),
               det.nrows, det.ncols, tuple(np.ravel(det.detector_to_sample)))
        if getattr(ctx, "_projection_key", None) != key:
            ctx.set_master_pattern(self.master_upper, self.master_lower)
            ctx.set_detector(det.gnomonic_bounds, det.pcz, det.nrows, det.ncols, det.detector_to_sample)
            ctx._projection_key = key
            ctx._projection_refs = (self.master_upper, self.master_lower, det)  # keep the ids alive

    def push_to_engine(self, ctx, global_start):
        """Generate this (slice of the) dictionary in device memory and sweep it."""
        if self.dtype != np.float32:
            # the fused path produces float32 patterns; other dtypes take the reference's
            # route: materialise, then prepare_dictionary casts (cf. .astype(dtype) at
            # similarity_metrics/_normalized_cross_correlation.py:235)
            ctx.push_dictionary_chunk(self.compute(ctx), global_start)
            return
        self.configure(ctx)
        if self.pcs is not None:
            ctx.push_rotations_chunk_varying_pc(self.rotations, self.pcs, global_start, self.detector.detector_to_sample,
                                                self.rescale, self.out_min, self.out_max)
            return
        ctx.push_rotations_chunk(self.rotations, global_start, self.rescale, self.out_min, self.out_max)

    def hold_in_engine(self, ctx, global_start):
        """Generate this (slice of the) dictionary in device memory and keep it prepared there
        (`kikuchipy_amd.ResidentDictionary`)."""
        if self.dtype != np.float32 or self.pcs is not None:
            ctx.hold_dictionary_chunk(self.compute(ctx), global_start)
            return
        self.configure(ctx)
        ctx.hold_rotations_chunk(self.rotations, global_start, self.rescale, self.out_min, self.out_max)

    def compute(self, ctx=None):
        """The patterns as a NumPy array (projected on the GPU)."""
        if ctx is None:
            root = self._root
            if root._ctx is None:
                root._ctx = _lib.Context(self.device)
            ctx = root._ctx
        self.configure(ctx)
        out = np.empty(self.shape, dtype=self.dtype)
        n, step = len(self), self.chunksize[0]
        flat = out.reshape(n, -1)
        for a in range(0, n, step):
            if self.pcs is not None:
                flat[a:a + step] = ctx.project_patterns_varying_pc(self.rotations[a:a + step], self.pcs[a:a + step],
                                                                   self.detector.shape, self.detector.detector_to_sample,
                                                                   self.rescale, self.out_min, self.out_max, self.dtype)
            else:
                flat[a:a + step] = ctx.project_patterns(self.rotations[a:a + step], self.rescale, self.out_min,
                                                        self.out_max, self.dtype)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.compute()
        return a if dtype is None else a.astype(dtype)

    def __repr__(self):
        return (f"ProjectedDictionary(shape={self.shape}, dtype={self.dtype}, rescale={self.rescale}, "
                f"chunksize={self.chunksize})")
