"""`kikuchipy.indexing.merge_crystal_maps` for the engine's result objects
(indexing/_merge_crystal_maps.py:28-354 of the reference).

Host-side NumPy: this is bookkeeping over (M, N, K) score arrays (which phase
wins each map point, the merged ranking over all phases), a few MB even for a
200 x 200 map - there is nothing for the GPU to win.  The orix `CrystalMap` /
`PhaseList` handling of the reference is replaced by plain arrays and phase
names; the arithmetic is the reference's.
"""

from math import copysign
import warnings

import numpy as np


class MergedIndexingResult:
    """The merged map: per point the winning phase (`phase_id`, -1 = not indexed
    in any map; `phase_names[id]`), its `rotations`, `scores` and
    `simulation_indices`, and the rankings over all phases `merged_scores`,
    `merged_simulation_indices` (the properties `merged_<name>` of the
    reference's returned `CrystalMap`)."""

    def __init__(self, shape, phase_id, phase_names, rotations, scores, merged_scores, simulation_indices=None,
                 merged_simulation_indices=None, scores_prop="scores", simulation_indices_prop=None,
                 step_sizes=None, scan_unit=None):
        self.shape = tuple(shape)
        self.phase_id = phase_id
        self.phase_names = phase_names
        self.rotations = rotations
        self.scores = scores
        self.merged_scores = merged_scores
        self.simulation_indices = simulation_indices
        self.merged_simulation_indices = merged_simulation_indices
        self._scores_prop = scores_prop
        self._sim_prop = simulation_indices_prop
        self.step_sizes = step_sizes
        self.scan_unit = scan_unit
        self.is_in_data = np.ones(int(np.prod(shape)), dtype=bool)

    @property
    def size(self):
        return int(np.prod(self.shape))

    @property
    def rotations_per_point(self):
        return 1 if self.scores.ndim == 1 else self.scores.shape[1]

    @property
    def prop(self):
        out = {self._scores_prop: self.scores, f"merged_{self._scores_prop}": self.merged_scores}
        if self._sim_prop is not None:
            out[self._sim_prop] = self.simulation_indices
            out[f"merged_{self._sim_prop}"] = self.merged_simulation_indices
        return out


def _prop(xmap, name):
    return np.asarray(xmap.prop[name])


def _phase_ids(xmap, n_points):
    """-1 where a point of the map is marked not indexed."""
    pid = getattr(xmap, "phase_id", None)
    return np.zeros(n_points, dtype=int) if pid is None else np.asarray(pid)


def merge_crystal_maps(crystal_maps, mean_n_best=1, greater_is_better=None, scores_prop="scores",
                       simulation_indices_prop=None, navigation_masks=None):
    """Merge single-phase indexing results of the same map into one multi-phase
    result: per point the phase with the best (mean of the `mean_n_best` best)
    score wins.

    crystal_maps
        `DictionaryIndexingResult`s / `RefinementResult`s (anything with
        `.prop[...]`, `.rotations`, `.shape`, `.is_in_data`, `.phase_name`);
        their property arrays hold the points that are in the data.
    mean_n_best, greater_is_better, scores_prop, simulation_indices_prop, navigation_masks
        As in the reference (masks: True = point NOT in that map).
    """
    n_maps = len(crystal_maps)
    if navigation_masks is None:
        all_in = [np.all(x.is_in_data) for x in crystal_maps]
        if not all(all_in):
            navigation_masks = [~np.asarray(x.is_in_data).reshape(x.shape) for x in crystal_maps]
    if navigation_masks is not None:
        if len(navigation_masks) != n_maps:
            raise ValueError("Number of crystal maps and navigation masks must be equal")
        map_shapes = []
        for i, (mask, xmap) in enumerate(zip(navigation_masks, crystal_maps)):
            if isinstance(mask, np.ndarray):
                mask_is_in_data = np.sum(~mask)
                map_is_in_data = int(np.sum(xmap.is_in_data))
                if mask_is_in_data != map_is_in_data:
                    raise ValueError(
                        f"{i}. navigation mask does not have as many 'False', {mask_is_in_data}, as there are "
                        f"points in the crystal map, {map_is_in_data}"
                    )
                map_shapes.append(mask.shape)
            elif mask is None:
                map_shapes.append(tuple(xmap.shape))
            else:
                raise ValueError(f"{i}. navigation mask must be a NumPy array or 'None'")
    else:
        map_shapes = [tuple(x.shape) for x in crystal_maps]
    if len({len(s) for s in map_shapes}) != 1 or not np.sum(abs(np.diff(map_shapes, axis=0))) == 0:
        raise ValueError("Crystal maps (and/or navigation masks) must have the same navigation shape")
    map_shape = map_shapes[0]
    map_size = int(np.prod(map_shape))
    if navigation_masks is not None:
        masks1d = [np.ones(map_size, dtype=bool) if m is None else ~m.ravel() for m in navigation_masks]
    else:
        masks1d = [None] * n_maps

    def in_data(xmap, arr):
        """Property rows of the points that are in the data (results of this package carry
        full-size arrays with zero rows for masked points)."""
        arr = np.asarray(arr)
        isin = np.asarray(xmap.is_in_data)
        return arr[isin] if arr.shape[0] == isin.size and not isin.all() else arr

    scores_all = [in_data(x, _prop(x, scores_prop)) for x in crystal_maps]
    per_point = [1 if s.ndim == 1 else s.shape[1] for s in scores_all]
    if not all(np.diff(per_point) == 0):
        raise ValueError("Crystal maps must have the same number of rotations and scores per point")
    n_scores_per_point = per_point[0]
    sim_all = None
    if simulation_indices_prop is not None:
        sim_all = [in_data(x, _prop(x, simulation_indices_prop)) for x in crystal_maps]
        n_sim_idx = sim_all[0].shape
        if len(n_sim_idx) > 1 and n_sim_idx[1] > n_scores_per_point:
            raise ValueError("Cannot merge maps with more simulation indices than scores per point")
    if greater_is_better is None:
        sign = copysign(1, mean_n_best)
        mean_n_best = abs(mean_n_best)
    else:
        sign = 1 if greater_is_better else -1

    comb_shape = (map_size,) + ((n_scores_per_point,) if n_scores_per_point > 1 else ()) + (n_maps,)
    scores_dtype = scores_all[0].dtype
    combined_scores = np.full(comb_shape, np.nan, dtype=np.dtype(f"f{scores_dtype.itemsize}"))
    for i, (mask, sc) in enumerate(zip(masks1d, scores_all)):
        if mask is not None:
            combined_scores[mask, ..., i] = sc
        else:
            combined_scores[..., i] = sc
    if n_scores_per_point > 1:
        best_scores = combined_scores[:, :mean_n_best].squeeze()
        if len(best_scores.shape) > 2:
            best_scores = np.nanmean(best_scores, axis=1)
    else:
        best_scores = combined_scores
    phase_id = np.nanargmax(sign * best_scores, axis=1)

    not_indexed = np.zeros((n_maps, map_size), dtype=bool)
    for i, (mask, xmap) in enumerate(zip(masks1d, crystal_maps)):
        pid = _phase_ids(xmap, scores_all[i].shape[0])
        if mask is not None:
            # the reference writes `not_indexed[i, mask][xmap.phase_id == -1] = True` (:167), i.e.
            # into a temporary copy: not-indexed points of a map that comes with a navigation
            # mask are NOT recorded.  Kept as is: results must equal the reference's.
            pass
        else:
            not_indexed[i, pid == -1] = True
    not_indexed = np.logical_and.reduce(not_indexed)
    phase_id[not_indexed] = -1

    new_rotations = np.zeros(comb_shape[:-1] + (4,), dtype="float")
    new_scores = np.zeros(comb_shape[:-1], dtype=scores_dtype)
    new_indices = np.zeros(comb_shape[:-1], dtype="int32") if sim_all is not None else None
    phase_names = []
    for i, (mask, xmap) in enumerate(zip(masks1d, crystal_maps)):
        phase_mask = phase_id == i
        if not phase_mask.any():
            continue
        name = getattr(xmap, "phase_name", "") or ""
        if name in phase_names:
            # same name = same phase here (names are all these result objects know of a phase):
            # not duplicated in the phase list, the points get the first map's ID (:231-237)
            phase_id[phase_mask] = phase_names.index(name)
        else:
            phase_names.append(name)  # PhaseList.add: IDs count up in the order of addition
        rot = in_data(xmap, np.asarray(getattr(xmap.rotations, "data", xmap.rotations)))
        rows = phase_mask[mask] if mask is not None else phase_mask
        new_rotations[phase_mask] = rot[rows]
        new_scores[phase_mask] = scores_all[i][rows]
        if sim_all is not None:
            new_indices[phase_mask] = sim_all[i][rows]

    mergesort_shape = (comb_shape[0], int(np.prod(comb_shape[1:])))
    comb_scores_reshaped = combined_scores.reshape(mergesort_shape)
    best_sorted_idx = np.argsort(sign * -comb_scores_reshaped, kind="mergesort", axis=1)
    merged_best_scores = np.take_along_axis(comb_scores_reshaped, best_sorted_idx, axis=-1)
    merged_sim = None
    if sim_all is not None:
        comb = []
        for mask, si in zip(masks1d, sim_all):
            if mask is not None:
                full = np.full(comb_shape[:-1], np.nan)
                full[mask] = si
                comb.append(full)
            else:
                comb.append(si)
        comb_sim_idx = np.dstack(comb)
        # make the indices unique across the maps so that an orientation similarity map can
        # be computed from the merged lists
        for i in range(1, comb_sim_idx.shape[-1]):
            increment = abs(np.nanmax(comb_sim_idx[..., i - 1]) - np.nanmin(comb_sim_idx[..., i])) + 1
            comb_sim_idx[..., i] += increment
        comb_sim_idx = comb_sim_idx.reshape(mergesort_shape)
        merged_sim = np.take_along_axis(comb_sim_idx, best_sorted_idx, axis=-1)
    first = crystal_maps[0]
    return MergedIndexingResult(map_shape, phase_id, phase_names, new_rotations, new_scores, merged_best_scores,
                                new_indices, merged_sim, scores_prop, simulation_indices_prop,
                                getattr(first, "step_sizes", None), getattr(first, "scan_unit", None))
