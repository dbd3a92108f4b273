"""Load the reference's hot-path modules, unmodified, in THIS container only.

TEST INFRASTRUCTURE - never imported by the product (kikuchipy_amd/).

`import kikuchipy` is impossible here (no hyperspy/orix/numba/lazy_loader), but
the modules on the dictionary-indexing path only need numpy/dask/scipy.  This
shim (the recipe of SURVEY.md Appendix A) stubs the four things that are
missing and loads each module BY FILE from /root/reference, so the code that
runs is the reference's own:

1. `numba` stub: `njit` returns the undecorated function (== the `.py_func`
   the reference's own tests compare against, tests/test_pattern/test_pattern.py:169).
2. `orix` stub: `CrystalMap`/`Rotation` record their arguments, so
   `_dictionary_indexing()` (indexing/_dictionary_indexing.py:36) runs to the
   end and hands back the `prop` dict with `scores` and `simulation_indices`.
3. namespace-package stubs for `kikuchipy.*` so the packages' `__init__`
   (lazy_loader, hyperspy) never run but intra-package imports resolve.
4. a loader that compiles with postponed annotations (`X | Y` in signatures
   is illegal at def time on Python 3.9).

Must be run with /opt/conda/bin/python3.9 (numpy 1.26.4, dask 2021.10.0,
scipy 1.7.1, h5py 3.3.0): inside the reference's supported range
(pyproject.toml:44).  It reads /root/reference, so it can NOT run on the GPU
box; only its outputs (tests/golden/*.npz) travel.
"""

import __future__

import importlib.machinery
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("KPDI_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF_ROOT, "src", "kikuchipy")


class _PostponedLoader(importlib.machinery.SourceFileLoader):
    def source_to_code(self, data, path, *, _optimize=-1):
        return compile(
            data,
            path,
            "exec",
            flags=__future__.annotations.compiler_flag,
            dont_inherit=True,
        )


def _stub_numba():
    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            f = args[0]
            f.py_func = f
            return f

        def deco(f):
            f.py_func = f
            return f

        return deco

    nb.njit = njit
    nb.jit = njit
    nb.prange = range
    sys.modules["numba"] = nb


class _Recorder:
    """Stands in for orix CrystalMap / Rotation: records what it was given."""

    def __init__(self, *args, **kwargs):
        self.args = args
        self.kw = kwargs

    @classmethod
    def identity(cls, shape):
        return cls(identity=shape)

    def __getitem__(self, item):
        return _Recorder(parent=self, item=item)

    def __setitem__(self, key, value):
        self.kw.setdefault("set", []).append((key, value))

    @property
    def data(self):
        return self

    def flatten(self):
        return self


# ---- minimal stand-ins for the orix objects `merge_crystal_maps` manipulates
# (indexing/_merge_crystal_maps.py): only what that function touches.
class FakeStructure(list):
    class _Lattice:
        def abcABG(self):
            return (1.0, 1.0, 1.0, 90.0, 90.0, 90.0)

    lattice = _Lattice()


class FakePhase:
    def __init__(self, name):
        self.name = name
        self.space_group = None
        self.point_group = None
        self.structure = FakeStructure()

    def deepcopy(self):
        return FakePhase(self.name)


class FakePhaseList:
    """orix.crystal_map.PhaseList: IDs count up from 0 in the order of addition,
    'not_indexed' has ID -1."""

    def __init__(self, phases=None):
        self._dict = {}
        for p in phases or []:
            self.add(p)

    def add(self, phase):
        ids = [i for i in self._dict if i >= 0]
        self._dict[max(ids) + 1 if ids else 0] = phase

    def add_not_indexed(self):
        self._dict[-1] = FakePhase("not_indexed")

    @property
    def names(self):
        return [p.name for p in self._dict.values()]

    @property
    def ids(self):
        return list(self._dict)

    def id_from_name(self, name):
        return [i for i, p in self._dict.items() if p.name == name][0]

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._dict[self.id_from_name(key)]
        return self._dict[key]


class FakeRotations:
    def __init__(self, data):
        import numpy as np

        self.data = np.asarray(data)

    def __getitem__(self, key):
        return FakeRotations(self.data[key])


class FakeCrystalMap:
    """A single-phase map: `prop` arrays hold the points that are in the data."""

    def __init__(self, shape, rotations, prop, phase_name, is_in_data=None, phase_id=None, scan_unit="px"):
        import numpy as np

        self._original_shape = tuple(shape)
        size = int(np.prod(shape))
        self.is_in_data = np.ones(size, dtype=bool) if is_in_data is None else np.asarray(is_in_data)
        n = int(self.is_in_data.sum())
        self.rotations = FakeRotations(rotations)
        self.prop = prop
        self.phase_id = np.zeros(n, dtype=int) if phase_id is None else np.asarray(phase_id)
        self._phase = FakePhase(phase_name)
        self.scan_unit = scan_unit
        self.dx = self.dy = 1.0

    @property
    def shape(self):
        return self._original_shape

    def _data_slices_from_coordinates(self):
        return tuple(slice(None) for _ in self._original_shape)

    @property
    def rotations_per_point(self):
        d = self.rotations.data
        return 1 if d.ndim == 2 else d.shape[1]

    @property
    def phases_in_data(self):
        pl = FakePhaseList()
        if (self.phase_id == -1).any():
            pl.add_not_indexed()
        if (self.phase_id != -1).any():
            pl.add(self._phase)
        return pl


def _stub_orix():
    orix = types.ModuleType("orix")
    cm = types.ModuleType("orix.crystal_map")
    qu = types.ModuleType("orix.quaternion")
    cm.CrystalMap = _Recorder
    cm.Phase = FakePhase
    cm.PhaseList = FakePhaseList
    cm.create_coordinate_arrays = lambda shape, step_sizes=None: ({}, None)
    qu.Rotation = _Recorder
    orix.crystal_map = cm
    orix.quaternion = qu
    sys.modules["orix"] = orix
    sys.modules["orix.crystal_map"] = cm
    sys.modules["orix.quaternion"] = qu


def _stub_tqdm():
    try:
        import tqdm  # noqa: F401
    except ImportError:
        t = types.ModuleType("tqdm")
        t.tqdm = lambda it, **kw: it
        sys.modules["tqdm"] = t


def _stub_skimage():
    # pattern/_pattern.py imports dtype_range from skimage.util.dtype; the conda
    # env has skimage 0.18.3, keep the real one if importable.
    try:
        from skimage.util.dtype import dtype_range  # noqa: F401
    except Exception:
        import numpy as np

        sk = types.ModuleType("skimage")
        sku = types.ModuleType("skimage.util")
        skd = types.ModuleType("skimage.util.dtype")
        skd.dtype_range = {
            bool: (False, True),
            np.bool_: (False, True),
            float: (-1, 1),
            np.float16: (-1, 1),
            np.float32: (-1, 1),
            np.float64: (-1, 1),
            np.uint8: (0, 255),
            np.uint16: (0, 65535),
            np.uint32: (0, 2**32 - 1),
            np.int8: (-128, 127),
            np.int16: (-32768, 32767),
            np.int32: (-(2**31), 2**31 - 1),
        }
        sk.__path__ = []
        sku.__path__ = []
        ske = types.ModuleType("skimage.exposure")
        ske.equalize_adapthist = None  # imported by pattern/_pattern.py, never called here
        sk.util, sk.exposure, sku.dtype = sku, ske, skd
        sys.modules["skimage"] = sk
        sys.modules["skimage.util"] = sku
        sys.modules["skimage.util.dtype"] = skd
        sys.modules["skimage.exposure"] = ske


def _stub_dask_if_missing():
    # the system interpreter (python3.10, scipy 1.15) has no dask; the refinement
    # functions loaded there never touch it, the modules only import it
    try:
        import dask.array  # noqa: F401
    except ImportError:
        d = types.ModuleType("dask")
        da = types.ModuleType("dask.array")
        da.Array = type("Array", (), {})
        d.array = da
        d.delayed = lambda f: f
        d.__path__ = []
        diag = types.ModuleType("dask.diagnostics")
        diag.__path__ = []
        prog = types.ModuleType("dask.diagnostics.progress")
        prog.ProgressBar = type("ProgressBar", (), {})
        diag.progress = prog
        diag.ProgressBar = prog.ProgressBar
        d.diagnostics = diag
        sys.modules["dask"] = d
        sys.modules["dask.array"] = da
        sys.modules["dask.diagnostics"] = diag
        sys.modules["dask.diagnostics.progress"] = prog


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(fullname, relpath):
    path = os.path.join(SRC, relpath)
    loader = _PostponedLoader(fullname, path)
    spec = importlib.util.spec_from_file_location(fullname, path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[fullname] = mod
    loader.exec_module(mod)
    return mod


_loaded = {}


def load_reference():
    """Return a dict of the reference's hot-path modules."""
    if _loaded:
        return _loaded
    _stub_numba()
    _stub_orix()
    _stub_tqdm()
    _stub_skimage()
    _stub_dask_if_missing()
    _ns("kikuchipy", SRC)
    _ns("kikuchipy.indexing", os.path.join(SRC, "indexing"))
    _ns(
        "kikuchipy.indexing.similarity_metrics",
        os.path.join(SRC, "indexing", "similarity_metrics"),
    )
    _ns("kikuchipy.filters", os.path.join(SRC, "filters"))
    _ns("kikuchipy.pattern", os.path.join(SRC, "pattern"))

    sm = "kikuchipy.indexing.similarity_metrics."
    _loaded["similarity_metric"] = _load(
        sm + "_similarity_metric", "indexing/similarity_metrics/_similarity_metric.py"
    )
    _loaded["ncc"] = _load(
        sm + "_normalized_cross_correlation",
        "indexing/similarity_metrics/_normalized_cross_correlation.py",
    )
    _loaded["ndp"] = _load(
        sm + "_normalized_dot_product",
        "indexing/similarity_metrics/_normalized_dot_product.py",
    )
    _loaded["di"] = _load(
        "kikuchipy.indexing._dictionary_indexing", "indexing/_dictionary_indexing.py"
    )
    _loaded["window"] = _load("kikuchipy.filters.window", "filters/window.py")
    # filters/__init__ re-exports Window; pattern/_pattern.py imports
    # `from kikuchipy.filters.window import Window` and fft_barnes helpers.
    sys.modules["kikuchipy.filters"].Window = _loaded["window"].Window
    _loaded["fft_barnes"] = _load("kikuchipy.filters.fft_barnes", "filters/fft_barnes.py")
    _loaded["pattern"] = _load("kikuchipy.pattern._pattern", "pattern/_pattern.py")
    return _loaded


def load_reference_projection():
    """The master-pattern projection modules:
    `_utils/numba.py` (rotate_vector) and `signals/util/_master_pattern.py`
    (direction cosines, Lambert interpolation, pattern projection).  Under the
    numba stub every function is its `.py_func`."""
    ref = load_reference()
    if "master_pattern" in ref:
        return ref
    _ns("kikuchipy._utils", os.path.join(SRC, "_utils"))
    _ns("kikuchipy.signals", os.path.join(SRC, "signals"))
    _ns("kikuchipy.signals.util", os.path.join(SRC, "signals", "util"))
    ref["numba_utils"] = _load("kikuchipy._utils.numba", "_utils/numba.py")
    ref["master_pattern"] = _load(
        "kikuchipy.signals.util._master_pattern", "signals/util/_master_pattern.py"
    )
    return ref


def load_reference_refinement():
    """The refinement objective functions and SciPy solvers:
    `_utils/_gnonomic_bounds.py`, `indexing/_refinement/__init__.py` (the table
    of supported methods), `_objective_functions.py`, `_solvers.py`."""
    ref = load_reference_projection()
    if "solvers" in ref:
        return ref
    ref["gnomonic_bounds"] = _load("kikuchipy._utils._gnonomic_bounds", "_utils/_gnonomic_bounds.py")
    pkg = _ns("kikuchipy.indexing._refinement", os.path.join(SRC, "indexing", "_refinement"))
    init = os.path.join(SRC, "indexing", "_refinement", "__init__.py")
    exec(compile(open(init).read(), init, "exec"), pkg.__dict__)
    ref["objective_functions"] = _load(
        "kikuchipy.indexing._refinement._objective_functions",
        "indexing/_refinement/_objective_functions.py",
    )
    ref["solvers"] = _load(
        "kikuchipy.indexing._refinement._solvers", "indexing/_refinement/_solvers.py"
    )
    return ref


def load_function_source(relpath, name, extra_globals=None):
    """Execute ONE top-level function of a reference module that cannot be
    imported as a whole here (e.g. detectors/_ebsd_detector.py needs orix and
    matplotlib) and return it."""
    import ast

    import numpy as np

    path = os.path.join(SRC, relpath)
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            code = compile(
                mod, path, "exec", flags=__future__.annotations.compiler_flag, dont_inherit=True
            )
            g = {"np": np, "nb": sys.modules.get("numba")}
            g.update(extra_globals or {})
            exec(code, g)
            return g[name]
    raise KeyError(name)


class FakeDictionaryXmap:
    """What `_dictionary_indexing` touches on `dictionary_xmap`
    (indexing/_dictionary_indexing.py:79, :162-166)."""

    class _Phases:
        names = ["ni"]

    phases = _Phases()
    phases_in_data = None
    rotations = _Recorder()
