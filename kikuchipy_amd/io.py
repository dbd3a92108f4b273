"""Reading kikuchipy's h5ebsd files without HyperSpy / h5py
(io/plugins/kikuchipy_h5ebsd/_api.py:64-171, io/plugins/_h5ebsd.py:303-390 of the
reference): `load()` returns this package's `EBSD` holder with the patterns, the
static background, the detector (PCs, tilts, binning) and the step sizes of one
scan.  The file is read by libkpdi through the HDF5 C library."""

import warnings

import numpy as np

from kikuchipy_amd import _lib
from kikuchipy_amd.detectors import EBSDDetector
from kikuchipy_amd.signals import EBSD, DictionaryXmap


def load(filename, lazy=False, *, scan_group_names=None, device=None, devices=None):
    """Load one scan of a kikuchipy h5ebsd file - `kikuchipy.load(filename, lazy=False, **kwargs)`
    (io/_io.py:57-150) with the h5ebsd reader's keyword (io/plugins/kikuchipy_h5ebsd/_api.py:64-78).

    lazy
        Accepted as in the reference.  The patterns are read into host memory at once either way: this
        package's `EBSD` holds an array, the engine takes it from there in pieces.
    scan_group_names
        Name of the scan group ("Scan 1"); the first scan of the file if not
        given (as in the reference).  A list loads several scans and returns a
        list, like `kikuchipy.load`.
    device, devices
        Where the returned signal's engine lives (see `EBSD`).
    """
    if not isinstance(lazy, (bool, np.bool_)):
        raise TypeError(f"`lazy` must be a bool, not {type(lazy).__name__} (the scan is named by the keyword "
                        "`scan_group_names`, as in kikuchipy.load)")
    if isinstance(scan_group_names, (list, tuple)):
        # several scans: one that is not in the file is an error only when it is the only one asked for, else a warning
        # (io/plugins/_h5ebsd.py:285-301)
        out = []
        for name in scan_group_names:
            try:
                out.append(load(filename, lazy, scan_group_names=name, device=device, devices=devices))
            except _lib.KpdiIOError as err:
                if len(scan_group_names) == 1 or "is not among the scans" not in str(err):
                    raise
                warnings.warn(str(err))
        return out
    try:
        info, pats, bg, pc = _lib.h5ebsd_read(str(filename), scan_group_names)
    except _lib.KpdiIOError:
        raise
    except _lib.KpdiError as err:  # (cannot open, no such scan, no patterns: the reference raises OSError / IOError)
        raise _lib.KpdiIOError(str(err)) from None
    ny, nx, sy, sx = info.ny, info.nx, info.sy, info.sx
    # the reference squeezes singleton navigation axes (io/plugins/_h5ebsd.py:366)
    nav_shape = tuple(n for n in (ny, nx) if n > 1)
    data = pats.reshape(nav_shape + (sy, sx))
    if info.n_stored < ny * nx * sy * sx:
        warnings.warn(
            f"Signal shape ({sy}, {sy}) and navigation shape ({ny}, {nx}) larger than file size. "
            "Will attempt to load by zero padding incomplete patterns."
        )
    # detector (io/plugins/kikuchipy_h5ebsd/_api.py:123-169)
    if pc is None:
        pc = np.array([0.5, 0.5, 0.5])
    elif pc.shape[0] == ny * nx and pc.shape[0] > 1:
        pc = pc.reshape((ny, nx, 3))
        pc = pc.reshape(nav_shape + (3,)) if nav_shape else pc.reshape(3)
    elif pc.shape[0] > 1:
        warnings.warn(
            f"Data navigation shape ({(ny, nx)}) differs from the number of projection centers (PCs) "
            f"{pc.shape[0]}; the detector gets the mean PC"
        )
        pc = pc.mean(axis=0)
    else:
        pc = pc.reshape(3)
    detector = EBSDDetector(shape=(sy, sx), px_size=info.detector_pixel_size, binning=info.binning,
                            tilt=info.elevation_angle, azimuthal=info.azimuth_angle, sample_tilt=info.sample_tilt,
                            pc=pc)
    step_sizes = tuple(s for s, n in ((info.step_y, ny), (info.step_x, nx)) if n > 1)
    s = EBSD(data, static_background=bg, xmap=DictionaryXmap.empty(nav_shape or (1,)), step_sizes=step_sizes,
             device=device, devices=devices)
    s.detector = detector
    s.original_metadata = {"scan": info.scan.decode(), "n_rows": ny, "n_columns": nx, "pattern_height": sy,
                           "pattern_width": sx, "binning": info.binning, "step_x": info.step_x, "step_y": info.step_y,
                           "detector_pixel_size": info.detector_pixel_size, "sample_tilt": info.sample_tilt,
                           "azimuth_angle": info.azimuth_angle, "elevation_angle": info.elevation_angle}
    return s
