"""The part of `kikuchipy.detectors.EBSDDetector` that dictionary generation reads
(detectors/_ebsd_detector.py of the reference): detector shape, one projection
centre (PC) and the sample/detector tilts -> gnomonic bounds and the
sample-to-detector orientation matrix that
`_get_direction_cosines_for_fixed_pc` takes
(signals/util/_master_pattern.py:83-124).

One PC, or one PC per map point (`pc` of shape navigation shape + (3,)):
dictionary generation uses a single PC for the whole dictionary (SURVEY.md
8(f1)), refinement takes either (8(f2)).  Plotting, calibration, PC
fitting/extrapolation and file I/O are out of scope.
"""

from collections.abc import Iterable
from numbers import Number

import numpy as np

# detectors/_ebsd_detector.py:71-91
PC_CONVENTIONS_ALIASES = {
    "bruker": ["bruker"],
    "tsl": ["tsl", "edax", "amatek"],
    "oxford": ["oxford", "aztec"],
    "emsoft": ["emsoft", "emsoft4", "emsoft5"],
}


def sample_to_detector_matrix(sample_tilt, tilt, azimuthal, twist):
    """Passive sample -> detector rotation matrix for angles in degrees
    (detectors/_ebsd_detector.py:100-150, :836-845).  Rows are the detector axes
    (X_d, Y_d, Z_d) in sample coordinates: start from (Y_s, Z_s, X_s) and turn
    all three about X_d by -sample_tilt, about X_d by +tilt, about Y_d by
    -azimuthal and about Z_d by -twist (axis-angle formula)."""
    basis = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=np.float64)
    angles = np.deg2rad(np.array([-sample_tilt, tilt, -azimuthal, -twist], dtype=np.float64))
    for axis_row, angle in zip((0, 0, 1, 2), angles):
        u = basis[axis_row] / np.sqrt(np.sum(np.square(basis[axis_row])))
        c, s = np.cos(angle), np.sin(angle)
        for j in range(3):
            v = basis[j].copy()
            basis[j] = v * c + np.cross(u, v) * s + u * np.dot(u, v) * (1.0 - c)
    return basis


class EBSDDetector:
    """EBSD detector with one projection centre, or one per map point.

    Parameters mirror the reference's constructor
    (detectors/_ebsd_detector.py:282-318): `shape` = (rows, columns), `px_size`
    in um, `binning`, detector `tilt`, `azimuthal` and `twist` and `sample_tilt`
    in degrees, `pc` = (PCx, PCy, PCz) in the given `convention` (stored in
    Bruker's convention like the reference does)."""

    def __init__(self, shape=(1, 1), px_size=1.0, binning=1, tilt=0.0, azimuthal=0.0, twist=0.0,
                 sample_tilt=70.0, pc=(0.5, 0.5, 0.5), convention="bruker"):
        self.shape = shape
        self.px_size = px_size
        self.binning = binning
        self.tilt = tilt
        self.azimuthal = azimuthal
        self.twist = twist
        self.sample_tilt = sample_tilt
        pc = np.atleast_2d(np.asarray(pc, dtype=np.float64))
        if pc.shape[-1] != 3 or pc.ndim > 3:
            raise ValueError(
                "`pc` must be (PCx, PCy, PCz) or an array of such triplets with at most two "
                f"navigation axes, got shape {pc.shape}"
            )
        self._pc = self._to_bruker(pc, convention)

    # detectors/_ebsd_detector.py:2207-2248, :2295-2315
    def _to_bruker(self, pc, convention):
        conv = None
        for name, aliases in PC_CONVENTIONS_ALIASES.items():
            if isinstance(convention, str) and convention.lower() in aliases:
                conv = name
        if conv is None:
            options = ", ".join(a for v in PC_CONVENTIONS_ALIASES.values() for a in v)
            raise ValueError(
                f"Invalid projection/pattern center convention {convention!r}. Options are {options}."
            )
        pcx, pcy, pcz = pc[..., 0], pc[..., 1], pc[..., 2]
        if conv == "tsl":
            return np.stack([pcx, 1 - pcy, pcz * min(self.nrows, self.ncols) / self.nrows], axis=-1)
        if conv == "oxford":
            return np.stack([pcx, 1 - pcy * self.aspect_ratio, pcz * self.aspect_ratio], axis=-1)
        if conv == "emsoft":
            version = int(convention[-1]) if convention[-1].isdigit() else 5
            if version < 5:
                pcx = -pcx
            return np.stack([
                0.5 - (pcx / (self.ncols * self._binning)),
                0.5 - (pcy / (self.nrows * self._binning)),
                pcz / (self.nrows * self._binning * self.px_size),
            ], axis=-1)
        return pc.copy()

    # ---- shape (detectors/_ebsd_detector.py:640-668)
    @property
    def nrows(self):
        return self.shape[0]

    @property
    def ncols(self):
        return self.shape[1]

    @property
    def size(self):
        return self.nrows * self.ncols

    @property
    def aspect_ratio(self):
        return self.ncols / self.nrows

    # ---- validated attributes: the reference's checks and texts (detectors/_ebsd_detector.py:2281-2292 and the setters
    # :337-340, :365-368, :392-395, :417-420, :582-585, :691-694)
    @property
    def shape(self):
        return self._shape

    @shape.setter
    def shape(self, value):
        if (not isinstance(value, Iterable) or isinstance(value, (str, bytes)) or len(value) != 2
                or not all(isinstance(v, Number) for v in value) or min(value) < 1):
            raise ValueError(f"Invalid shape {value}. Must be an iterable of two integers.")
        self._shape = tuple(int(v) for v in value)

    @property
    def binning(self):
        return int(self._binning)

    @binning.setter
    def binning(self, value):
        if not isinstance(value, Number):
            raise ValueError(f"Invalid binning {value}. Must be an integer.")
        self._binning = float(value)

    @property
    def px_size(self):
        return self._px_size

    @px_size.setter
    def px_size(self, value):
        if not isinstance(value, Number):
            raise ValueError(f"Invalid pixel size {value}. Must be a number.")
        self._px_size = float(value)

    def _angle(name, text):  # noqa: N805 - a small property factory
        def get(self):
            return getattr(self, "_" + name)

        def set_(self, value):
            if not isinstance(value, Number):
                raise ValueError(f"Invalid {text} {value!r}. Must be a number.")
            setattr(self, "_" + name, float(value))

        return property(get, set_)

    tilt = _angle("tilt", "detector tilt")
    azimuthal = _angle("azimuthal", "azimuthal")
    twist = _angle("twist", "twist")
    sample_tilt = _angle("sample_tilt", "sample tilt")
    del _angle

    @property
    def navigation_shape(self):
        return self._pc.shape[:-1]

    @property
    def navigation_size(self):
        return int(np.prod(self.navigation_shape))

    # ---- PC (Bruker convention); scalars for a single PC, arrays otherwise
    @property
    def pc(self):
        return self._pc

    @pc.setter
    def pc(self, value):
        value = np.atleast_2d(np.asarray(value, dtype=np.float64))
        if value.shape[-1] != 3:
            raise ValueError("`pc` must have a last axis of size 3")
        self._pc = value

    @property
    def pc_flattened(self):
        return self._pc.reshape(-1, 3)

    @property
    def pc_average(self):
        return np.nanmean(self.pc_flattened, axis=0)

    def _component(self, i):
        v = self._pc[..., i]
        return v[0] if v.shape == (1,) else v

    def _set_component(self, i, value):
        v = np.asarray(value, dtype=np.float64)
        if v.size > 1 and v.size == self.navigation_size:
            v = v.reshape(self._pc.shape[:-1])  # one value per projection centre, in any layout
        self._pc[..., i] = v

    @property
    def pcx(self):
        return self._component(0)

    @pcx.setter
    def pcx(self, value):
        self._set_component(0, value)

    @property
    def pcy(self):
        return self._component(1)

    @pcy.setter
    def pcy(self, value):
        self._set_component(1, value)

    @property
    def pcz(self):
        return self._component(2)

    @pcz.setter
    def pcz(self, value):
        self._set_component(2, value)

    # ---- derived sizes (detectors/_ebsd_detector.py:605-727)
    @property
    def navigation_dimension(self):
        return len(self.navigation_shape)

    @property
    def bounds(self):
        """Detector bounds [x0, x1, y0, y1] in pixel coordinates."""
        return np.array([0, self.ncols - 1, 0, self.nrows - 1])

    @property
    def unbinned_shape(self):
        return tuple(int(v) for v in np.array(self.shape, dtype=int) * self._binning)

    @property
    def height(self):
        """Detector height in microns."""
        return self.nrows * self.px_size * self._binning

    @property
    def width(self):
        """Detector width in microns."""
        return self.ncols * self.px_size * self._binning

    @property
    def px_size_binned(self):
        return self.px_size * self._binning

    @property
    def specimen_scintillator_distance(self):
        """PCz x detector height, in microns."""
        return self.pcz * self.height

    # ---- gnomonic coordinates (detectors/_ebsd_detector.py:731-818)
    @property
    def x_min(self):
        return -self.aspect_ratio * (self.pcx / self.pcz)

    @property
    def x_max(self):
        return self.aspect_ratio * (1 - self.pcx) / self.pcz

    @property
    def y_min(self):
        return -(1 - self.pcy) / self.pcz

    @property
    def y_max(self):
        return self.pcy / self.pcz

    @property
    def x_range(self):
        """(x_min, x_max) per projection centre: navigation shape + (2,)  (detectors/_ebsd_detector.py:749-755)."""
        return np.stack([np.atleast_1d(self.x_min), np.atleast_1d(self.x_max)], axis=-1).reshape(self.navigation_shape + (2,))

    @property
    def y_range(self):
        return np.stack([np.atleast_1d(self.y_min), np.atleast_1d(self.y_max)], axis=-1).reshape(self.navigation_shape + (2,))

    @property
    def x_scale(self):
        """Width of a pixel in gnomonic coordinates (detectors/_ebsd_detector.py:785-795)."""
        d = np.diff(self.x_range)
        return (d if self.ncols == 1 else d / (self.ncols - 1)).reshape(self.navigation_shape)

    @property
    def y_scale(self):
        d = np.diff(self.y_range)
        return (d if self.nrows == 1 else d / (self.nrows - 1)).reshape(self.navigation_shape)

    @property
    def r_max(self):
        """Largest distance from the pattern centre to a detector corner in gnomonic coordinates, with the corners the
        reference takes (detectors/_ebsd_detector.py:821-833: its "lower left" repeats the upper left)."""
        x0, x1, y0, y1 = (np.atleast_1d(v) for v in (self.x_min, self.x_max, self.y_min, self.y_max))
        corners = np.stack([x0**2 + y0**2, x1**2 + y0**2, x1**2 + y1**2, x0**2 + y0**2], axis=-1)
        return np.atleast_2d(np.sqrt(corners.max(axis=-1)).reshape(self.navigation_shape))

    @property
    def gnomonic_bounds(self):
        """(x_min, x_max, y_min, y_max): shape (4,) for one PC, else navigation shape + (4,)."""
        return np.stack([self.x_min, self.x_max, self.y_min, self.y_max], axis=-1).astype(np.float64)

    # ---- orientation
    @property
    def sample_to_detector(self):
        """3 x 3 matrix (the reference returns the same rotation as an orix `Rotation`)."""
        return sample_to_detector_matrix(self.sample_tilt, self.tilt, self.azimuthal, self.twist)

    @property
    def detector_to_sample(self):
        """`(~detector.sample_to_detector).to_matrix()`: the transpose."""
        return np.ascontiguousarray(self.sample_to_detector.T)

    def deepcopy(self):
        return EBSDDetector(self.shape, self.px_size, self._binning, self.tilt, self.azimuthal, self.twist,
                            self.sample_tilt, self._pc.copy(), "bruker")

    def __repr__(self):
        """The reference's text (detectors/_ebsd_detector.py:319-331)."""
        pc = tuple(float(v) for v in np.round(self.pc_average, 3))
        deg = "\N{DEGREE SIGN}"
        return ("EBSDDetector\n"
                f"  shape (Ny, Nx):     {self.shape}\n"
                f"  pc (PCx, PCy, PCz): {pc}\n"
                f"  sample_tilt:        {self.sample_tilt}{deg}\n"
                f"  tilt:               {self.tilt}{deg}\n"
                f"  azimuthal:          {self.azimuthal}{deg}\n"
                f"  twist:              {self.twist}{deg}\n"
                f"  binning:            {self.binning}\n"
                f"  px_size:            {self.px_size} um")
