"""`kikuchipy.indexing.orientation_similarity_map` on the GPU engine
(indexing/_orientation_similarity_map.py:30-152 of the reference)."""

import numpy as np

from kikuchipy_amd import _lib


def _footprint_offsets(footprint):
    """Non-zero footprint elements in row-major order, relative to the centre
    `shape // 2` that scipy.ndimage.generic_filter uses."""
    footprint = np.asarray(footprint)
    if footprint.ndim != 2:
        raise ValueError("footprint must be a 2D array")
    cy, cx = footprint.shape[0] // 2, footprint.shape[1] // 2
    return np.array([(i - cy, j - cx) for i in range(footprint.shape[0]) for j in range(footprint.shape[1])
                     if footprint[i, j]], dtype=np.int32).reshape(-1, 2)


def orientation_similarity_map(xmap, n_best=None, simulation_indices_prop="simulation_indices", normalize=False,
                               from_n_best=None, footprint=None, center_index=2, *, shape=None, context=None,
                               device=0):
    """Orientation similarity map (OSM) of a dictionary-indexing result.

    xmap
        A `DictionaryIndexingResult` (or any object with `.prop[...]` and
        `.shape`, like an orix `CrystalMap`), or a `(n_points, keep_n)` integer
        array together with `shape=(ny, nx)`.
    n_best, simulation_indices_prop, normalize, from_n_best, footprint, center_index
        As in the reference.  Returns float32 of shape `(ny, nx)` or
        `(ny, nx, n_best - from_n_best + 1)`.
    context
        A `_lib.Context` to run on.  If its last `finalize()` produced exactly these
        simulation indices (compared element by element), the map is computed from the best-k lists
        still resident in HBM; otherwise the indices are uploaded as usual.
    """
    if hasattr(xmap, "prop"):
        simulation_indices = np.asarray(xmap.prop[simulation_indices_prop])
        data_shape = tuple(xmap.shape)
    else:
        simulation_indices = np.asarray(xmap)
        if shape is None:
            raise ValueError("`shape` is needed with a plain array of simulation indices")
        data_shape = tuple(shape)
    if simulation_indices.ndim != 2:
        raise ValueError("simulation indices must have shape (number of map points, keep_n)")
    nav_size, keep_n = simulation_indices.shape
    if n_best is None:
        n_best = keep_n
    elif n_best > keep_n:
        raise ValueError(f"n_best {n_best} cannot be greater than keep_n {keep_n}")
    if len(data_shape) != 2 or data_shape[0] * data_shape[1] != nav_size:
        raise ValueError(f"map shape {data_shape} does not hold {nav_size} points in two dimensions")
    if from_n_best is None:
        from_n_best = n_best
    if footprint is None:
        footprint = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    offsets = _footprint_offsets(footprint)
    ctx = context if context is not None else _lib.Context(device)
    try:
        osm = None
        if context is not None and ctx.holds_result(simulation_indices):
            try:
                osm = ctx.orientation_similarity_map(None, data_shape, keep_n, n_best, from_n_best, offsets,
                                                     center_index, normalize)
            except _lib.KpdiError:  # the lists were dropped meanwhile (new sweep, reset): upload instead
                osm = None
        if osm is None:
            osm = ctx.orientation_similarity_map(simulation_indices, data_shape, keep_n, n_best, from_n_best, offsets,
                                                 center_index, normalize)
    finally:
        if context is None:
            ctx.close()
    return osm.squeeze()
