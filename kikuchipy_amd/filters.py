"""The two windows of `kikuchipy.filters.Window` that the accelerated path uses
(filters/window.py of the reference): the circular detector mask of the canonical
pipeline, `signal_mask = ~Window("circular", shape).astype(bool)`
(doc/tutorials/pattern_matching.ipynb), and the Gaussian window behind
`remove_dynamic_background`.  Other window types are not part of this path.
"""

import numpy as np


def distance_to_origin(shape, origin=None):
    """filters/window.py:528-555: distance of every pixel to `origin`
    (default: shape // 2 per axis)."""
    shape = tuple(int(s) for s in (shape if np.iterable(shape) else (shape,)))
    if origin is None:
        origin = tuple(s // 2 for s in shape)
    coordinates = np.ogrid[tuple(slice(None, s) for s in shape)]
    if len(shape) == 2:
        (x, y), (ox, oy) = coordinates, origin
        return np.sqrt((x - ox) ** 2 + (y - oy) ** 2)
    return np.abs(coordinates[0] - origin[0])


class Window(np.ndarray):
    """`Window("circular", shape)` / `Window("gaussian", shape, std=...)` as a
    NumPy array subclass with the reference's `name`, `circular`, `origin` and
    `n_neighbours` attributes."""

    def __new__(cls, window=None, shape=None, **kwargs):
        """filters/window.py:117-186: `window` None = "circular"; `shape` None = (3, 3), or `(Nx,)` when the SciPy-style
        keyword `Nx` is given; an array is a "custom" window; any other name is asked of `scipy.signal.get_window`
        (its parameters by keyword, in SciPy's order: `Window("gaussian", (5, 5), std=1)`), per axis, outer product
        in 2-D.  The FFT-filter windows ("lowpass", "highpass", "modified_hann") are not on this package's path."""
        window = "circular" if window is None else window
        if "Nx" in kwargs:
            shape = (kwargs.pop("Nx"),)
        elif shape is None:
            shape = (3, 3)
        else:
            try:
                shape = tuple(shape)
            except TypeError:
                raise TypeError(f"Window shape {shape} must be a sequence of ints.")
            if any(isinstance(v, (float, np.floating)) for v in shape):
                raise TypeError(f"Window shape {shape} must be a sequence of ints.")
            if any(v < 1 for v in shape):
                raise ValueError(f"All window axes {shape} must be > 0.")
        circular = False
        if isinstance(window, np.ndarray) or hasattr(window, "compute"):
            name, data = "custom", np.asarray(window)
        elif isinstance(window, str):
            if window in ("lowpass", "highpass", "modified_hann"):
                raise NotImplementedError(
                    f"kikuchipy_amd.filters.Window does not make the FFT-filter window {window!r} (not used by "
                    "dictionary indexing); pass the array as a custom window"
                )
            if not 1 <= len(shape) <= 2:
                raise ValueError(f"Window shape {shape} must be 1D or 2D")
            from scipy.signal import get_window

            circular = window == "circular"
            name = "rectangular" if circular else window
            fftbins = kwargs.pop("fftbins", False)
            spec = (name,) + tuple(kwargs.values())
            data = get_window(spec, int(shape[0]), fftbins=fftbins)
            if len(shape) == 2:
                data = np.outer(data, get_window(spec, int(shape[1]), fftbins=fftbins))
        else:
            raise ValueError(f"Window {type(window)} must be of type numpy.ndarray, dask.array.Array, or a valid string")
        obj = np.asarray(data).view(cls)
        obj._name = name
        obj._circular = False
        if circular:
            obj.make_circular()
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self._name = getattr(obj, "_name", None)
        self._circular = getattr(obj, "_circular", False)

    @property
    def name(self):
        return self._name

    @property
    def circular(self):
        return self._circular

    @property
    def origin(self):
        return tuple(i // 2 for i in self.shape)

    @property
    def n_neighbours(self):
        return tuple(np.subtract(self.shape, self.origin) - 1)

    @property
    def distance_to_origin(self):
        """Radial distance of every coefficient to the window's origin (filters/window.py:205-208)."""
        return distance_to_origin(self.shape, self.origin)

    @property
    def is_valid(self):
        """filters/window.py:222-230."""
        return isinstance(self.name, str) and self.ndim < 3 and isinstance(self.circular, bool)

    def make_circular(self):
        """filters/window.py:249-269: zero outside the largest centred circle; a rectangular ("boxcar") window is
        then called "circular"; nothing happens to a window with one axis."""
        if self.ndim == 1:
            return
        mask = self.distance_to_origin > max(self.origin)
        self[mask] = 0
        self._circular = True
        if self.name in ("rectangular", "boxcar"):
            self._name = "circular"

    def shape_compatible(self, shape):
        """Whether the window fits into data of `shape` (filters/window.py:271-288)."""
        shape = tuple(shape)
        return len(self.shape) <= len(shape) and not np.any(np.array(self.shape) > np.array(shape))

    def __array_wrap__(self, obj, context=None, return_scalar=False):
        if obj.shape == ():
            return obj[()]
        return np.ndarray.__array_wrap__(self, obj, context, return_scalar)

    def __repr__(self):
        data = np.array_str(np.asarray(self), precision=4, suppress_small=True)
        return f"{self.__class__.__name__} {self.shape} {self.name}\n{data}"
