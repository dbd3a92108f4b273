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
        self.shape = tuple(int(v) for v in shape)
        if len(self.shape) != 2 or min(self.shape) < 1:
            raise ValueError("`shape` must be (number of rows, number of columns)")
        self.px_size = float(px_size)
        self._binning = float(binning)
        self.tilt = float(tilt)
        self.azimuthal = float(azimuthal)
        self.twist = float(twist)
        self.sample_tilt = float(sample_tilt)
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

    @property
    def binning(self):
        return int(self._binning)

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

    @property
    def pcx(self):
        return self._component(0)

    @property
    def pcy(self):
        return self._component(1)

    @property
    def pcz(self):
        return self._component(2)

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
        pc = tuple(float(v) for v in np.round(self.pc_average, 3))
        return (f"EBSDDetector(shape={self.shape}, pc={pc}, sample_tilt={self.sample_tilt}, "
                f"tilt={self.tilt}, azimuthal={self.azimuthal}, twist={self.twist}, "
                f"binning={self.binning}, px_size={self.px_size} um)")
