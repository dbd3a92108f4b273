"""ctypes binding of libkpdi.so (include/kpdi.h) - no PyTorch, no NumPy C-API.

The library is built in-tree by `make -C kikuchipy_amd/csrc` (or
`__graft_entry__.build()`).  There is no CPU fallback anywhere in this
package: if the library is missing, or no gfx950 GPU is visible, calls raise.
"""

import atexit
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPDI_LIB_PATH: another build of the library (kernel experiments, tools/README.md)
LIB_PATH = os.environ.get("KPDI_LIB_PATH") or os.path.join(_HERE, "csrc", "libkpdi.so")

METRIC_NCC, METRIC_NDP = 0, 1
COMPUTE_F32, COMPUTE_F16X2, COMPUTE_F16, COMPUTE_F64 = 0, 1, 2, 3
GATHER_AUTO, GATHER_RCCL, GATHER_P2P, GATHER_NONE = 0, 1, 2, 3
GATHER_NAMES = {GATHER_RCCL: "rccl", GATHER_P2P: "p2p", GATHER_NONE: "none"}
F64_CERTIFICATES = {0: None, 1: "statistical", 2: "worstcase"}
OP_SUBTRACT, OP_DIVIDE = 0, 1
DOMAIN_FREQUENCY, DOMAIN_SPATIAL = 0, 1
UNIQUE_ID_BYTES = 128
REFINE_ORI, REFINE_PC, REFINE_ORI_PC = 0, 1, 2
REFINE_SIZES = {REFINE_ORI: (3, 3), REFINE_PC: (3, 4), REFINE_ORI_PC: (6, 0)}  # (control variables, fixed values)
REFINE_RESULT_STRIDE = 9

DTYPE_CODES = {
    np.dtype(np.uint8): 0,
    np.dtype(np.uint16): 1,
    np.dtype(np.float32): 2,
    np.dtype(np.float64): 3,
    np.dtype(np.int8): 4,
    np.dtype(np.int16): 5,
    np.dtype(np.int32): 6,
    np.dtype(np.uint32): 7,
    np.dtype(np.float16): 8,
}


class KpdiError(RuntimeError):
    """A libkpdi call failed (message from kpdi_last_error())."""


class KpdiIOError(KpdiError, OSError):
    """A file could not be read (`kikuchipy_amd.load`): an `OSError` like the reference's reader raises
    (io/plugins/_h5ebsd.py:210, :292-300), and a `KpdiError` like every failed libkpdi call."""


class Counters(C.Structure):
    _fields_ = [
        ("match_ms", C.c_double),
        ("match_launches", C.c_int64),
        ("match_flops", C.c_double),
        ("prep_ms", C.c_double),
        ("merge_ms", C.c_double),
        ("h2d_bytes", C.c_double),
        ("match_grid", C.c_int32),
        ("match_nsplit", C.c_int32),
        ("kpad", C.c_int32),
        ("k_kept", C.c_int32),
        ("project_ms", C.c_double),
        ("refine_ms", C.c_double),
        ("preproc_ms", C.c_double),
        ("preproc_launches", C.c_int64),
        ("rescore_ms", C.c_double),
        ("rescore_extra_passes", C.c_int64),
        ("uncertified_patterns", C.c_int64),
        ("match_form", C.c_int32),
        ("comm_ranks", C.c_int32),
        ("comm_ms", C.c_double),
        ("fixed_ms", C.c_double),
        ("f64_certificate", C.c_int32),
        ("gather_ranks", C.c_int32),
        ("coalesced_sweeps", C.c_int64),
        ("epi_lists", C.c_int64),
        ("epi_appended", C.c_int64),
        ("epi_overflows", C.c_int64),
        ("epi_direct_first", C.c_int64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class H5ebsdInfo(C.Structure):
    _fields_ = [
        ("scan", C.c_char * 64),
        ("ny", C.c_int32), ("nx", C.c_int32), ("sy", C.c_int32), ("sx", C.c_int32),
        ("dtype", C.c_int32),
        ("has_static_background", C.c_int32),
        ("static_background_dtype", C.c_int32),
        ("binning", C.c_int32),
        ("n_stored", C.c_int64),
        ("n_pc", C.c_int64),
        ("step_y", C.c_double), ("step_x", C.c_double), ("detector_pixel_size", C.c_double),
        ("sample_tilt", C.c_double), ("azimuth_angle", C.c_double), ("elevation_angle", C.c_double),
    ]


class PlanLaunch(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("row_first", "rows", "xcd_rows", "xcd_splits", "rows_grid")]


class Plan(C.Structure):
    """kpdi_plan (include/kpdi.h): what the launch planner decides for a sweep."""
    _fields_ = ([(n, C.c_int32) for n in ("form", "tile", "row_blocks", "n_tiles", "nsplit", "rows_per_launch", "launches")]
                + [("round_rows", C.c_int64)]
                + [(n, C.c_int32) for n in ("n_main", "tail_tiles", "tail_units", "tail_nsplit", "fixed_draws", "tail_first",
                                            "tail_shift", "perm_rounds", "perm_stride", "tail_gemm_rows", "n_launch_desc")]
                + [("launch", PlanLaunch * 64)])


# every symbol include/kpdi.h declares: (restype, argtypes)
_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
SIGNATURES = {
    "kpdi_version": (C.c_char_p, []),
    "kpdi_device_count": (_i, []),
    "kpdi_last_error": (C.c_char_p, []),
    "kpdi_create": (_i, [_i, C.POINTER(_vp)]),
    "kpdi_destroy": (_i, [_vp]),
    "kpdi_synchronize": (_i, [_vp]),
    "kpdi_set_problem": (_i, [_vp, _i, _i, _vp, _i, _i, _i]),
    "kpdi_set_keep_n": (_i, [_vp, _i]),
    "kpdi_set_experimental": (_i, [_vp, _vp, _i, _i64, _vp]),
    "kpdi_set_experimental_dev": (_i, [_vp, _vp, _i, _i64, _vp]),
    "kpdi_n_experimental": (_i64, [_vp]),
    "kpdi_remove_static_background": (_i, [_vp, _vp, _i, _i]),
    "kpdi_remove_dynamic_background": (_i, [_vp, _i, _i, C.c_double, C.c_double]),
    "kpdi_get_experimental": (_i, [_vp, _vp]),
    "kpdi_push_dictionary_chunk": (_i, [_vp, _vp, _i, _i64, _i64]),
    "kpdi_push_dictionary_chunk_dev": (_i, [_vp, _vp, _i, _i64, _i64]),
    "kpdi_set_master_pattern": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "kpdi_set_detector": (_i, [_vp, _vp, C.c_double, _i, _i, _vp]),
    "kpdi_set_direction_cosines": (_i, [_vp, _vp, _i64]),
    "kpdi_get_direction_cosines": (_i, [_vp, _vp]),
    "kpdi_project_patterns": (_i, [_vp, _vp, _i64, _i, C.c_double, C.c_double, _i, _vp]),
    "kpdi_project_patterns_varying_pc": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, C.c_double, C.c_double, _i, _vp]),
    "kpdi_push_rotations_chunk": (_i, [_vp, _vp, _i64, _i64, _i, C.c_double, C.c_double]),
    "kpdi_push_rotations_chunk_varying_pc": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _i, C.c_double, C.c_double]),
    "kpdi_hold_dictionary_chunk": (_i, [_vp, _vp, _i, _i64, _i64]),
    "kpdi_hold_dictionary_chunk_dev": (_i, [_vp, _vp, _i, _i64, _i64]),
    "kpdi_hold_rotations_chunk": (_i, [_vp, _vp, _i64, _i64, _i, C.c_double, C.c_double]),
    "kpdi_sweep_held": (_i, [_vp]),
    "kpdi_release_held": (_i, [_vp]),
    "kpdi_held_size": (_i, [_vp, _vp, _vp]),
    "kpdi_refine_set_patterns": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _i, _vp]),
    "kpdi_refine_get_prepared": (_i, [_vp, _vp, _vp]),
    "kpdi_refine_objective": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp]),
    "kpdi_refine_solve": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, C.c_double, C.c_double, _i, _i, _vp]),
    "kpdi_nelder_mead_selftest": (_i, [_vp, _i, _i, _vp, _vp, _vp, C.c_double, C.c_double, _i, _i, _vp]),
    "kpdi_orientation_similarity_map": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "kpdi_dtype_size": (_sz, [_i]),
    "kpdi_plan_describe": (_i, [_i64, _i64, _i, _i, _i, _i, C.POINTER(Plan)]),
    "kpdi_h5ebsd_info_read": (_i, [C.c_char_p, C.c_char_p, C.POINTER(H5ebsdInfo)]),
    "kpdi_h5ebsd_read_patterns": (_i, [C.c_char_p, C.c_char_p, _vp, _sz]),
    "kpdi_h5ebsd_read_static_background": (_i, [C.c_char_p, C.c_char_p, _vp, _sz]),
    "kpdi_h5ebsd_read_pc": (_i, [C.c_char_p, C.c_char_p, _vp, _i64]),
    "kpdi_set_experimental_h5ebsd": (_i, [_vp, C.c_char_p, C.c_char_p, _vp]),
    "kpdi_reset_topk": (_i, [_vp]),
    "kpdi_finalize": (_i, [_vp, _vp, _vp]),
    "kpdi_finalize_f64": (_i, [_vp, _vp, _vp]),
    "kpdi_finalize_async": (_i, [_vp, C.POINTER(_i)]),
    "kpdi_finalize_wait": (_i, [_vp, _i, _vp, _vp]),
    "kpdi_pending_result_size": (_i, [_vp, _i, C.POINTER(_i64)]),
    "kpdi_result_indices_i32": (_i, [_vp, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(_i64)]),
    "kpdi_comm_unique_id": (_i, [_vp]),
    "kpdi_comm_init": (_i, [_vp, _i, _i, _vp]),
    "kpdi_comm_selftest": (_i, [_vp, _i64, _i]),
    "kpdi_comm_drop": (_i, [_vp]),
    "kpdi_export_lists": (_i, [_vp, _vp, _vp]),
    "kpdi_import_lists": (_i, [_vp, _vp, _vp, _i]),
    "kpdi_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "kpdi_dev_free": (_i, [_vp, _vp]),
    "kpdi_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "kpdi_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "kpdi_set_profiling": (_i, [_vp, _i]),
    "kpdi_get_counters": (_i, [_vp, C.POINTER(Counters)]),
    "kpdi_reset_counters": (_i, [_vp]),
    "kpdi_counters_size": (_sz, []),
    # a group of contexts: several GPUs from one thread of one process
    "kpdi_group_create": (_i, [C.POINTER(_i), _i, _i, C.POINTER(_vp)]),
    "kpdi_group_destroy": (_i, [_vp]),
    "kpdi_group_size": (_i, [_vp]),
    "kpdi_group_gather": (_i, [_vp]),
    "kpdi_group_describe": (C.c_char_p, [_vp]),
    "kpdi_group_member": (_vp, [_vp, _i]),
    "kpdi_group_chunk_share": (_i, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "kpdi_group_assign_chunk": (_i, [_i, _i64, _i64, C.POINTER(_i64), _i64, _i, C.POINTER(_i), C.POINTER(_i64),
                                     C.POINTER(_i64), C.POINTER(_i)]),
    "kpdi_group_set_dictionary_size": (_i, [_vp, _i64]),
    "kpdi_group_push_dictionary_chunk_borrowed": (_i, [_vp, _vp, _i, _i64, _i64, C.POINTER(_i64)]),
    "kpdi_group_chunks_consumed": (_i, [_vp, C.POINTER(_i64)]),
    "kpdi_group_synchronize": (_i, [_vp]),
    "kpdi_group_set_problem": (_i, [_vp, _i, _i, _vp, _i, _i, _i]),
    "kpdi_group_set_keep_n": (_i, [_vp, _i]),
    "kpdi_group_set_experimental": (_i, [_vp, _vp, _i, _i64, _vp]),
    "kpdi_group_set_experimental_dev": (_i, [_vp, C.POINTER(_vp), _i, _i64, _vp]),
    "kpdi_group_n_experimental": (_i64, [_vp]),
    "kpdi_group_remove_static_background": (_i, [_vp, _vp, _i, _i]),
    "kpdi_group_remove_dynamic_background": (_i, [_vp, _i, _i, C.c_double, C.c_double]),
    "kpdi_group_get_experimental": (_i, [_vp, _vp]),
    "kpdi_group_push_dictionary_chunk": (_i, [_vp, _vp, _i, _i64, _i64]),
    "kpdi_group_push_dictionary_chunk_dev": (_i, [_vp, C.POINTER(_vp), _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "kpdi_group_set_master_pattern": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "kpdi_group_set_detector": (_i, [_vp, _vp, C.c_double, _i, _i, _vp]),
    "kpdi_group_push_rotations_chunk": (_i, [_vp, _vp, _i64, _i64, _i, C.c_double, C.c_double]),
    "kpdi_group_hold_dictionary_chunk": (_i, [_vp, _vp, _i, _i64, _i64]),
    "kpdi_group_hold_rotations_chunk": (_i, [_vp, _vp, _i64, _i64, _i, C.c_double, C.c_double]),
    "kpdi_group_sweep_held": (_i, [_vp]),
    "kpdi_group_release_held": (_i, [_vp]),
    "kpdi_group_held_size": (_i, [_vp, _vp, _vp]),
    "kpdi_group_reset_topk": (_i, [_vp]),
    "kpdi_group_finalize": (_i, [_vp, _vp, _vp]),
    "kpdi_group_finalize_f64": (_i, [_vp, _vp, _vp]),
    "kpdi_group_finalize_async": (_i, [_vp, C.POINTER(_i)]),
    "kpdi_group_finalize_wait": (_i, [_vp, _i, _vp, _vp]),
    "kpdi_group_pending_result_size": (_i, [_vp, _i, C.POINTER(_i64)]),
    "kpdi_group_set_profiling": (_i, [_vp, _i]),
    "kpdi_group_reset_counters": (_i, [_vp]),
}

_lib = None


def load():
    """Load libkpdi.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KpdiError(
                f"{LIB_PATH} not found: build it with `make -C kikuchipy_amd/csrc` "
                "(python -c 'import __graft_entry__ as g; g.build()'). kikuchipy_amd has no "
                "CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        # kpdi_counters has grown between versions: a stale struct here would be written past its end
        if lib.kpdi_counters_size() != C.sizeof(Counters):
            raise KpdiError(f"{LIB_PATH} ({lib.kpdi_version().decode()}) was built with a kpdi_counters of "
                            f"{lib.kpdi_counters_size()} bytes, this binding expects {C.sizeof(Counters)}: rebuild "
                            "the library (make -C kikuchipy_amd/csrc)")
        _lib = lib
    return _lib


def last_error():
    return (load().kpdi_last_error() or b"").decode()


def check(rc):
    if rc != 0:
        raise KpdiError(f"libkpdi error {rc}: {last_error()}")


def device_count():
    return int(load().kpdi_device_count())


def version():
    return load().kpdi_version().decode()


def plan_describe(m, n_chunk, k_kept=3600, keep_n=20, n_cu=256, form=-1):
    """The launch plan of a sweep (csrc/plan.h) as a `Plan`; host arithmetic, needs no GPU."""
    p = Plan()
    check(load().kpdi_plan_describe(int(m), int(n_chunk), int(k_kept), int(keep_n), int(n_cu), int(form), C.byref(p)))
    return p


def dtype_code(dtype):
    try:
        return DTYPE_CODES[np.dtype(dtype)]
    except KeyError:
        raise KpdiError(f"pattern dtype {np.dtype(dtype)} is not supported by libkpdi") from None


DTYPE_FROM_CODE = {code: dt for dt, code in DTYPE_CODES.items()}


def _cstr(s):
    return None if s is None else os.fsencode(s)


def h5ebsd_info(path, scan=None):
    info = H5ebsdInfo()
    check(load().kpdi_h5ebsd_info_read(_cstr(path), _cstr(scan), C.byref(info)))
    return info


def h5ebsd_read(path, scan=None):
    """(info, patterns (ny, nx, sy, sx), static background or None, PCs (n_pc, 3) or None)."""
    info = h5ebsd_info(path, scan)
    name = info.scan
    pats = np.empty((info.ny, info.nx, info.sy, info.sx), dtype=DTYPE_FROM_CODE[info.dtype])
    check(load().kpdi_h5ebsd_read_patterns(_cstr(path), name, _ptr(pats), pats.nbytes))
    bg = None
    if info.has_static_background:
        bg = np.empty((info.sy, info.sx), dtype=DTYPE_FROM_CODE[info.static_background_dtype])
        check(load().kpdi_h5ebsd_read_static_background(_cstr(path), name, _ptr(bg), bg.nbytes))
    pc = None
    if info.n_pc > 0:
        pc = np.empty((info.n_pc, 3), dtype=np.float64)
        check(load().kpdi_h5ebsd_read_pc(_cstr(path), name, _ptr(pc), info.n_pc))
    return info, pats, bg, pc


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _mask_bytes(mask):
    if mask is None:
        return None
    return np.ascontiguousarray(np.asarray(mask).ravel().astype(np.uint8))


class _Functions:
    """`f.set_problem` -> `libkpdi.kpdi_set_problem` (prefix "kpdi_") or `kpdi_group_set_problem`
    (prefix "kpdi_group_"): lets `Group` reuse `Context`'s methods for every call that has a group form."""

    def __init__(self, prefix):
        self._prefix = prefix

    def __getattr__(self, name):
        fn = getattr(load(), self._prefix + name)
        setattr(self, name, fn)
        return fn


class Context:
    """One GPU, one stream: thin object wrapper over a `kpdi_ctx*`."""

    _prefix = "kpdi_"

    def __init__(self, device=0, _borrowed=None):
        self._f = _Functions(self._prefix)
        self._owned = _borrowed is None
        if _borrowed is not None:  # a member of a `Group`: the group owns the kpdi_ctx
            self._h = C.c_void_p(_borrowed)
        else:
            self._h = C.c_void_p()
            check(self._f.create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._keep = {}  # host arrays the library may still be reading
        self._keep_n = None       # what kpdi_finalize will write per pattern
        self._compute = COMPUTE_F32
        self._projection_key = None  # simulations.ProjectedDictionary.configure
        self.result_token = 0     # bumped by every finalize()
        self._last_valid = False   # the last finalize() succeeded: its lists may still be resident in HBM
        self._host_gather = None   # a parallel.Communicator whose ranks gather their lists over the control plane

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if self._owned:
                self._f.destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        check(self._f.synchronize(self._h))

    # -- set-up
    def set_problem(self, sy, sx, signal_mask=None, metric=METRIC_NCC, keep_n=20, compute=COMPUTE_F32):
        sm = _mask_bytes(signal_mask)
        if sm is not None and sm.size != sy * sx:
            raise KpdiError(f"signal mask has {sm.size} elements, detector has {sy * sx}")
        check(self._f.set_problem(self._h, int(sy), int(sx), _ptr(sm), int(metric), int(compute),
                                      int(keep_n)))
        self._keep_n = int(keep_n)
        self._compute = int(compute)
        self._detector = (int(sy), int(sx))

    def set_keep_n(self, keep_n):
        check(self._f.set_keep_n(self._h, int(keep_n)))
        self._keep_n = int(keep_n)

    def set_experimental(self, patterns, navigation_mask=None):
        """patterns: (m_all, sy, sx) or (m_all, sy*sx), C-contiguous."""
        p = np.ascontiguousarray(patterns)
        nm = _mask_bytes(navigation_mask)
        m_all = p.shape[0]
        if nm is not None and nm.size != m_all:
            raise KpdiError(f"navigation mask has {nm.size} elements, there are {m_all} patterns")
        check(self._f.set_experimental(self._h, _ptr(p), dtype_code(p.dtype), m_all, _ptr(nm)))
        check(self._f.synchronize(self._h))  # upload done: `p` may be a temporary
        self._exp_shape, self._exp_dtype = p.shape, p.dtype

    def set_experimental_h5ebsd(self, path, scan=None, navigation_mask=None):
        """Patterns of a kikuchipy h5ebsd scan: file -> pinned host buffer -> HBM."""
        info = h5ebsd_info(path, scan)
        nm = _mask_bytes(navigation_mask)
        if nm is not None and nm.size != info.ny * info.nx:
            raise KpdiError(f"navigation mask has {nm.size} elements, the scan has {info.ny * info.nx} patterns")
        check(self._f.set_experimental_h5ebsd(self._h, _cstr(path), info.scan, _ptr(nm)))
        self._exp_shape = (info.ny * info.nx, info.sy, info.sx)
        self._exp_dtype = DTYPE_FROM_CODE[info.dtype]
        return info

    def set_experimental_dev(self, d_ptr, dtype, m_all, navigation_mask=None):
        nm = _mask_bytes(navigation_mask)
        check(self._f.set_experimental_dev(self._h, C.c_void_p(d_ptr), dtype_code(dtype), int(m_all),
                                               _ptr(nm)))
        self._exp_shape, self._exp_dtype = (int(m_all),) + getattr(self, "_detector", (-1,)), np.dtype(dtype)

    @property
    def n_experimental(self):
        return int(self._f.n_experimental(self._h))

    # -- pre-processing
    def remove_static_background(self, static_bg_f32, operation=OP_SUBTRACT, scale_bg=False):
        bg = np.ascontiguousarray(static_bg_f32, dtype=np.float32)
        check(self._f.remove_static_background(self._h, _ptr(bg), int(operation), int(bool(scale_bg))))

    def remove_dynamic_background(self, operation=OP_SUBTRACT, filter_domain=DOMAIN_FREQUENCY, std=0.0,
                                  truncate=4.0):
        check(self._f.remove_dynamic_background(self._h, int(operation), int(filter_domain),
                                                    float(std), float(truncate)))

    def get_experimental(self):
        out = np.empty(self._exp_shape, dtype=self._exp_dtype)
        check(self._f.get_experimental(self._h, _ptr(out)))
        return out

    # -- sweep
    def set_dictionary_size(self, n_total):
        """How many dictionary patterns the coming sweep pushes in total.  One context sweeps them all whatever the
        number; a `Group` plans its chunk assignment with it."""

    def push_dictionary_chunk(self, patterns, global_start):
        p = np.ascontiguousarray(patterns)
        # returns when the upload has consumed `p`; the sweep of the chunk runs on
        check(self._f.push_dictionary_chunk(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0],
                                                int(global_start)))

    def push_dictionary_chunk_dev(self, d_ptr, dtype, n_chunk, global_start):
        check(self._f.push_dictionary_chunk_dev(self._h, C.c_void_p(d_ptr), dtype_code(dtype),
                                                    int(n_chunk), int(global_start)))

    # -- resident dictionary: prepared once, swept against several experimental sets
    def hold_dictionary_chunk(self, patterns, global_start):
        p = np.ascontiguousarray(patterns)
        check(self._f.hold_dictionary_chunk(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0],
                                                int(global_start)))

    def hold_dictionary_chunk_dev(self, d_ptr, dtype, n_chunk, global_start):
        check(self._f.hold_dictionary_chunk_dev(self._h, C.c_void_p(d_ptr), dtype_code(dtype),
                                                    int(n_chunk), int(global_start)))

    def hold_rotations_chunk(self, rotations, global_start, rescale=False, out_min=-1.0, out_max=1.0):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        check(self._f.hold_rotations_chunk(self._h, _ptr(rot), rot.shape[0], int(global_start),
                                               int(bool(rescale)), float(out_min), float(out_max)))

    def sweep_held(self):
        check(self._f.sweep_held(self._h))

    def release_held(self):
        check(self._f.release_held(self._h))

    def held_size(self):
        """(patterns held, bytes of device memory they occupy)."""
        n, b = C.c_int64(0), C.c_int64(0)
        check(self._f.held_size(self._h, C.byref(n), C.byref(b)))
        return n.value, b.value

    # -- dictionary generation on the device
    def set_master_pattern(self, upper, lower=None):
        """upper / lower: (npy, npx) arrays of one dtype (uint8, uint16, float32, float64)."""
        up = np.ascontiguousarray(upper)
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=up.dtype)
        if up.ndim != 2 or (lo is not None and lo.shape != up.shape):
            raise KpdiError("master pattern hemispheres must be 2D arrays of equal shape")
        self._projection_key = None  # whoever cached "my master pattern is loaded" must load it again
        check(self._f.set_master_pattern(self._h, _ptr(up), _ptr(lo), dtype_code(up.dtype),
                                             up.shape[1], up.shape[0]))

    def set_detector(self, gnomonic_bounds, pcz, nrows, ncols, om_detector_to_sample):
        gb = np.ascontiguousarray(gnomonic_bounds, dtype=np.float64).ravel()
        om = np.ascontiguousarray(om_detector_to_sample, dtype=np.float64).ravel()
        if gb.size != 4 or om.size != 9:
            raise KpdiError("gnomonic_bounds must have 4 and om_detector_to_sample 9 elements")
        self._projection_key = None
        check(self._f.set_detector(self._h, _ptr(gb), float(pcz), int(nrows), int(ncols), _ptr(om)))
        self._dc_npix = int(nrows) * int(ncols)

    def set_direction_cosines(self, direction_cosines):
        dc = np.ascontiguousarray(direction_cosines, dtype=np.float64).reshape(-1, 3)
        self._projection_key = None
        check(self._f.set_direction_cosines(self._h, _ptr(dc), dc.shape[0]))
        self._dc_npix = dc.shape[0]

    def get_direction_cosines(self):
        out = np.empty((self._dc_npix, 3), dtype=np.float64)
        check(self._f.get_direction_cosines(self._h, _ptr(out)))
        return out

    def project_patterns(self, rotations, rescale=False, out_min=-1.0, out_max=1.0, dtype_out=np.float32):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        out = np.empty((rot.shape[0], self._dc_npix), dtype=dtype_out)
        check(self._f.project_patterns(self._h, _ptr(rot), rot.shape[0], int(bool(rescale)),
                                           float(out_min), float(out_max), dtype_code(out.dtype), _ptr(out)))
        return out

    def project_patterns_varying_pc(self, rotations, pcs, shape, om_detector_to_sample, rescale=False, out_min=-1.0,
                                    out_max=1.0, dtype_out=np.float32):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        pc = np.ascontiguousarray(pcs, dtype=np.float64).reshape(-1, 3)
        if pc.shape[0] != rot.shape[0]:
            raise KpdiError(f"{rot.shape[0]} rotations but {pc.shape[0]} projection centres")
        om = np.ascontiguousarray(om_detector_to_sample, dtype=np.float64).ravel()
        out = np.empty((rot.shape[0], shape[0] * shape[1]), dtype=dtype_out)
        check(self._f.project_patterns_varying_pc(self._h, _ptr(rot), _ptr(pc), rot.shape[0], int(shape[0]),
                                                      int(shape[1]), _ptr(om), int(bool(rescale)), float(out_min),
                                                      float(out_max), dtype_code(out.dtype), _ptr(out)))
        return out

    def push_rotations_chunk(self, rotations, global_start, rescale=False, out_min=-1.0, out_max=1.0):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        check(self._f.push_rotations_chunk(self._h, _ptr(rot), rot.shape[0], int(global_start),
                                               int(bool(rescale)), float(out_min), float(out_max)))

    def push_rotations_chunk_varying_pc(self, rotations, pcs, global_start, om_detector_to_sample, rescale=False,
                                        out_min=-1.0, out_max=1.0):
        """A dictionary chunk simulated on the device with ONE projection centre per pattern, then swept."""
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        pc = np.ascontiguousarray(pcs, dtype=np.float64).reshape(-1, 3)
        if pc.shape[0] != rot.shape[0]:
            raise KpdiError(f"{rot.shape[0]} rotations but {pc.shape[0]} projection centres")
        om = np.ascontiguousarray(om_detector_to_sample, dtype=np.float64).ravel()
        if om.size != 9:
            raise KpdiError("om_detector_to_sample must have 9 elements")
        check(load().kpdi_push_rotations_chunk_varying_pc(self._h, _ptr(rot), _ptr(pc), rot.shape[0], int(global_start), _ptr(om),
                                                          int(bool(rescale)), float(out_min), float(out_max)))

    # -- refinement
    def refine_set_patterns(self, patterns, signal_mask=None, rescale=False, om_detector_to_sample=None):
        """patterns: (n, nrows, ncols); signal_mask: True = pixel not used."""
        p = np.ascontiguousarray(patterns)
        if p.ndim != 3:
            raise KpdiError("patterns must have shape (n, nrows, ncols)")
        sm = _mask_bytes(signal_mask)
        om = np.ascontiguousarray(om_detector_to_sample, dtype=np.float64).ravel()
        if om.size != 9:
            raise KpdiError("om_detector_to_sample must have 9 elements")
        check(self._f.refine_set_patterns(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0], p.shape[1],
                                              p.shape[2], _ptr(sm), int(bool(rescale)), _ptr(om)))
        self._ref_n = p.shape[0]
        self._ref_k = p.shape[1] * p.shape[2] if sm is None else int(np.count_nonzero(sm == 0))

    def refine_get_prepared(self):
        pat = np.empty((self._ref_n, self._ref_k), dtype=np.float32)
        sqn = np.empty(self._ref_n, dtype=np.float64)
        check(self._f.refine_get_prepared(self._h, _ptr(pat), _ptr(sqn)))
        return pat, sqn

    def refine_objective(self, mode, pattern_index, x, fixed=None):
        idx = np.ascontiguousarray(pattern_index, dtype=np.int32).ravel()
        nvar, nfixed = REFINE_SIZES[mode]
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(idx.size, nvar)
        f = None if nfixed == 0 else np.ascontiguousarray(fixed, dtype=np.float64).reshape(idx.size, nfixed)
        out = np.empty(idx.size, dtype=np.float64)
        check(self._f.refine_objective(self._h, int(mode), idx.size, _ptr(idx), _ptr(x), _ptr(f), _ptr(out)))
        return out

    def refine_solve(self, mode, x0, fixed=None, lower=None, upper=None, xatol=1e-4, fatol=1e-4, maxiter=0,
                     maxfev=0):
        """x0: (n_patterns, n_starts, nvar).  Returns (n_patterns, n_starts, 3 + nvar):
        fun, nfev, nit, x."""
        nvar, nfixed = REFINE_SIZES[mode]
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if x0.ndim != 3 or x0.shape[2] != nvar:
            raise KpdiError(f"x0 must have shape (n_patterns, n_starts, {nvar})")
        n, starts = x0.shape[:2]
        f = None if nfixed == 0 else np.ascontiguousarray(fixed, dtype=np.float64).reshape(n, starts, nfixed)
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64).reshape(x0.shape)
        hi = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64).reshape(x0.shape)
        res = np.empty((n, starts, REFINE_RESULT_STRIDE), dtype=np.float64)
        check(self._f.refine_solve(self._h, int(mode), n, starts, _ptr(x0), _ptr(f), _ptr(lo), _ptr(hi),
                                       float(xatol), float(fatol), int(maxiter or 0), int(maxfev or 0), _ptr(res)))
        return res[:, :, :3 + nvar]

    def nelder_mead_selftest(self, kind, x0, lower=None, upper=None, xatol=1e-4, fatol=1e-4, maxiter=0, maxfev=0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).ravel()
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64).ravel()
        hi = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64).ravel()
        res = np.empty(3 + x0.size, dtype=np.float64)
        check(self._f.nelder_mead_selftest(self._h, int(kind), x0.size, _ptr(x0), _ptr(lo), _ptr(hi),
                                               float(xatol), float(fatol), int(maxiter or 0), int(maxfev or 0),
                                               _ptr(res)))
        return res

    # -- result consumers
    def orientation_similarity_map(self, simulation_indices, shape, keep_n, n_best, from_n_best, offsets,
                                   center_index, normalize):
        """simulation_indices: (ny * nx, keep_n) integers, or None = the lists resident from
        the last finalize().  offsets: (n_fp, 2) (dy, dx).  Returns (ny, nx, layers) float32."""
        ny, nx = shape
        idx = None
        if simulation_indices is not None:
            idx = np.ascontiguousarray(simulation_indices, dtype=np.int64).reshape(ny * nx, keep_n)
        off = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1, 2)
        out = np.empty((ny, nx, n_best - from_n_best + 1), dtype=np.float32)
        check(self._f.orientation_similarity_map(self._h, _ptr(idx), int(ny), int(nx), int(keep_n), int(n_best),
                                                     int(from_n_best), _ptr(off), off.shape[0], int(center_index),
                                                     int(bool(normalize)), _ptr(out)))
        return out

    def reset_topk(self):
        check(self._f.reset_topk(self._h))
        self._last_valid = False

    def holds_result(self, simulation_indices):
        """Whether `simulation_indices` (n, keep_n) are the lists the last finalize() returned
        (and that may therefore still be resident in HBM)."""
        if not self._last_valid:
            return False
        # what kpdi_finalize left in its page-locked staging buffer (no copy was kept: the caller may have masked or
        # remapped the array it got in place, and must then not be told that the device still holds "these" lists)
        p, n = C.POINTER(C.c_int32)(), C.c_int64(0)
        check(self._f.result_indices_i32(self._h, C.byref(p), C.byref(n)))
        idx = np.asarray(simulation_indices)
        if not p or n.value != idx.size:
            return False
        last = np.ctypeslib.as_array(p, shape=(n.value,))
        return bool(np.array_equal(idx.ravel(), last))

    def finalize(self, keep_n=None):
        """(scores (m, keep_n) float32 - float64 with COMPUTE_F64 -, indices (m, keep_n) int64).  `keep_n` must be the value
        last given to `set_problem` / `set_keep_n`: that is what the library writes."""
        if keep_n is None:
            keep_n = self._keep_n
        if self._keep_n is None or int(keep_n) != self._keep_n:
            raise KpdiError(f"finalize(keep_n={keep_n}) but the context keeps {self._keep_n} entries per pattern "
                            "(set_problem / set_keep_n)")
        if self._host_gather is not None:  # no usable RCCL communicator: the ranks' lists travel over the control plane
            self._host_gather.gather_lists(self)
        m = self.n_experimental
        indices = np.empty((m, keep_n), dtype=np.int64)
        if self._compute == COMPUTE_F64:  # float64 arithmetic: the rescored scores (csrc/rescore.hip)
            scores = np.empty((m, keep_n), dtype=np.float64)
            check(self._f.finalize_f64(self._h, _ptr(scores), _ptr(indices)))
        else:
            scores = np.empty((m, keep_n), dtype=np.float32)
            check(self._f.finalize(self._h, _ptr(scores), _ptr(indices)))
        self.result_token += 1
        self._last_valid = False
        self._check_filled(indices, keep_n)
        self._last_valid = self._compute != COMPUTE_F64
        return scores, indices

    @staticmethod
    def _check_filled(indices, keep_n):
        if indices.size and indices[:, -1].max() >= 2**31 - 1:  # unfilled entries rank last
            # unfilled list entries (index INT_MAX, score -inf): fewer than keep_n candidates ranked, which
            # only happens when scores are NaN (NaN / inf in the patterns) - the reference propagates
            # NaN there (out of contract); fail clearly instead of indexing with INT_MAX
            bad = np.flatnonzero((indices >= 2**31 - 1).any(axis=1))
            raise KpdiError(f"{bad.size} experimental pattern(s) (first: {bad[0]}) ranked fewer than {keep_n} "
                            "dictionary patterns: NaN scores (NaN or inf in the patterns?) or a dictionary "
                            "smaller than keep_n")

    def finalize_async(self, keep_n=None):
        """Queue the hand-over of the result (all-gather + merge over the ranks, device-to-host copies) and return a
        ticket at once; `finalize_wait(ticket)` collects it.  In between the NEXT map may already be queued
        (`set_experimental*`, `push_*`): a series of maps then never leaves the GPU idle during a hand-over."""
        if keep_n is None:
            keep_n = self._keep_n
        if self._keep_n is None or int(keep_n) != self._keep_n:
            raise KpdiError(f"finalize_async(keep_n={keep_n}) but the context keeps {self._keep_n} entries per pattern")
        if self._host_gather is not None:  # (the host-staged gather is synchronous; merge and hand-over still overlap)
            self._host_gather.gather_lists(self)
        t = C.c_int(-1)
        check(self._f.finalize_async(self._h, C.byref(t)))
        self.result_token += 1
        self._last_valid = False
        return (t.value, int(keep_n))

    def finalize_wait(self, ticket):
        """(scores (m, keep_n) float32, indices (m, keep_n) int64) of a `finalize_async` ticket."""
        slot, keep_n = ticket
        n = C.c_int64(0)
        check(self._f.pending_result_size(self._h, int(slot), C.byref(n)))
        m = n.value // keep_n
        scores = np.empty((m, keep_n), dtype=np.float32)
        indices = np.empty((m, keep_n), dtype=np.int64)
        check(self._f.finalize_wait(self._h, int(slot), _ptr(scores), _ptr(indices)))
        self._check_filled(indices, keep_n)
        return scores, indices

    # -- multi-GPU
    @staticmethod
    def comm_unique_id():
        buf = np.zeros(UNIQUE_ID_BYTES, dtype=np.uint8)
        check(load().kpdi_comm_unique_id(_ptr(buf)))
        return buf.tobytes()

    def comm_init(self, rank, nranks, unique_id):
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        if buf.size != UNIQUE_ID_BYTES:
            raise KpdiError("unique id must be 128 bytes")
        check(self._f.comm_init(self._h, int(rank), int(nranks), _ptr(buf)))
        self._has_comm = True  # (such an engine is never kept for another call: release_engine)

    def comm_selftest(self, n_bytes=1 << 20, timeout_ms=60000):
        """One all-gather of `n_bytes` per rank, awaited for at most `timeout_ms` (raises KpdiError otherwise)."""
        check(self._f.comm_selftest(self._h, int(n_bytes), int(timeout_ms)))

    def comm_drop(self):
        """Forget the RCCL communicator: `finalize` returns this context's own lists (or imported ones)."""
        check(self._f.comm_drop(self._h))

    def export_lists(self):
        """This context's OWN running best-k lists: (scores (m, keep_n) float32 / float64, indices (m, keep_n) int32)."""
        m = self.n_experimental
        scores = np.empty((m, self._keep_n), dtype=np.float64 if self._compute == COMPUTE_F64 else np.float32)
        indices = np.empty((m, self._keep_n), dtype=np.int32)
        check(self._f.export_lists(self._h, _ptr(scores), _ptr(indices)))
        return scores, indices

    def import_lists(self, scores_all, indices_all):
        """(n_ranks, m, keep_n) lists of all ranks: the next finalize merges THEM (host-staged gather)."""
        s = np.ascontiguousarray(scores_all, dtype=np.float64 if self._compute == COMPUTE_F64 else np.float32)
        i = np.ascontiguousarray(indices_all, dtype=np.int32)
        if s.shape != i.shape or s.ndim != 3 or s.shape[1:] != (self.n_experimental, self._keep_n):
            raise KpdiError(f"import_lists: lists of shape {s.shape} / {i.shape}, expected (n_ranks, {self.n_experimental}, {self._keep_n})")
        check(self._f.import_lists(self._h, _ptr(s), _ptr(i), s.shape[0]))

    # -- device buffers
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self._f.dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, d_ptr):
        check(self._f.dev_free(self._h, C.c_void_p(d_ptr)))

    def h2d(self, d_ptr, array):
        a = np.ascontiguousarray(array)
        check(self._f.h2d(self._h, C.c_void_p(d_ptr), _ptr(a), a.nbytes))

    def d2h(self, array, d_ptr):
        check(self._f.d2h(self._h, _ptr(array), C.c_void_p(d_ptr), array.nbytes))

    # -- measurement
    def set_profiling(self, on=True):
        """False / 0: off; True / 1: every phase of a sweep is bracketed by HIP events; "match" / 2: only the match
        launches (and the all-gather) are - an event record between two kernels idles the GPU for ~6 us."""
        check(self._f.set_profiling(self._h, {"match": 2, "epilogue": 3}.get(on, int(on) if not isinstance(on, str) else -1)))

    def counters(self):
        c = Counters()
        check(self._f.get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        check(self._f.reset_counters(self._h))


def resolve_devices(devices):
    """`devices=` of the host layer -> list of device ids, or None (= not given).
    "all": every visible GPU; an int: that many GPUs (0 .. n-1); a sequence of ids (an id may
    repeat: several members then share that GPU - the rehearsal form for 1-GPU hosts)."""
    if devices is None:
        return None
    if isinstance(devices, str):
        if devices.strip().lower() == "all":
            return list(range(max(device_count(), 1)))
        try:
            return [int(d) for d in devices.replace(",", " ").split()]
        except ValueError:
            raise KpdiError(f"devices={devices!r}: expected 'all' or a list of device ids") from None
    if isinstance(devices, (int, np.integer)):
        if devices < 1:
            raise KpdiError("devices must name at least one GPU")
        return list(range(int(devices)))
    out = [int(d) for d in devices]
    if not out:
        raise KpdiError("devices must name at least one GPU")
    return out


LAUNCHER_VARIABLES = ("WORLD_SIZE", "LOCAL_RANK", "SLURM_PROCID", "SLURM_LOCALID", "OMPI_COMM_WORLD_RANK", "PMI_RANK")
_logged_defaults = set()


def under_a_launcher():
    """A launcher that starts one process per GPU (torchrun, srun, mpirun) has exported its variables: such a process
    must not spread over every GPU by itself - its siblings would all do the same."""
    return any(v in os.environ for v in LAUNCHER_VARIABLES)


def default_devices():
    """What a call that says nothing about devices runs on: $KPDI_DEVICES ("all" or ids like "0,1,2") if set; else, in a
    process started by a launcher (`under_a_launcher`), ONE GPU - the process's local rank ($LOCAL_RANK /
    $SLURM_LOCALID modulo the visible devices); else every visible GPU - the counterpart of the reference using every
    core of the host (its Dask scheduler's default).  What was chosen is logged once per choice
    (`logging.getLogger("kikuchipy_amd")`, level INFO)."""
    env = os.environ.get("KPDI_DEVICES")
    if env:
        ids, why = resolve_devices(env), f"$KPDI_DEVICES={env}"
    elif under_a_launcher():
        local = os.environ.get("LOCAL_RANK") or os.environ.get("SLURM_LOCALID") or "0"
        n = max(device_count(), 1)
        ids = [int(local) % n if local.lstrip("-").isdigit() else 0]
        why = ("a launcher's environment (" + ", ".join(v for v in LAUNCHER_VARIABLES if v in os.environ) +
               "): one process per GPU is assumed; name devices= or set $KPDI_DEVICES to override")
    else:
        ids, why = resolve_devices("all"), "every visible GPU"
    key = (tuple(ids), why)
    if key not in _logged_defaults:
        _logged_defaults.add(key)
        import logging

        logging.getLogger("kikuchipy_amd").info("no device named: running on GPU(s) %s (%s)", ids, why)
    return ids


def make_engine(device=0, devices=None, gather=None):
    """A `Context` on one GPU, or - for more than one device id - a `Group` over them."""
    ids = resolve_devices(devices)
    if ids is None:
        return Context(device)
    if len(ids) == 1 and gather is None:
        return Context(ids[0])
    return Group(ids, gather=gather)


# ---- engines that outlive the call that made them --------------------------------------------------------------------
# `kikuchipy_amd.dictionary_indexing(..., metric="ncc")` makes its own engine.  Creating one (context, streams, the first
# allocation of the prepared matrices and staging buffers) and destroying it (hipFree synchronises) was 6-7 ms of every
# such call - a quarter of a 30 ms call at configs[1].  The engine of a finished call is therefore kept, idle, ONE per set
# of devices, and handed to the next call that names the same devices (its device buffers stay allocated: they are sized
# for the job before).  `clear_engine_cache()` closes them; KPDI_ENGINE_CACHE=0 switches the cache off.
_ENGINE_POOL = {}
_ENGINE_POOL_LOCK = threading.Lock()


def _engine_key(device, devices, gather):
    ids = resolve_devices(devices)
    return (tuple(ids) if ids is not None else (int(device),), gather)


def acquire_engine(device=0, devices=None, gather=None):
    """`make_engine`, or the idle engine an earlier call released for the same devices."""
    key = _engine_key(device, devices, gather)
    engine = None
    if os.environ.get("KPDI_ENGINE_CACHE", "1") != "0":
        with _ENGINE_POOL_LOCK:
            engine = _ENGINE_POOL.pop(key, None)
    if engine is not None and hasattr(engine, "_h") and not (engine._h and engine._h.value):
        engine = None  # (closed behind the pool's back)
    if engine is None:
        engine = make_engine(device, devices, gather)
    engine._pool_key = key
    return engine


def release_engine(engine):
    """The call that acquired `engine` has finished with it, successfully: keep it for the next one (held chunks
    released, profiling off), or close it when the cache is off, holds one already, or the engine was given a
    communicator (that belongs to the job that attached it)."""
    key = getattr(engine, "_pool_key", None)
    keep = key is not None and os.environ.get("KPDI_ENGINE_CACHE", "1") != "0" and getattr(engine, "_host_gather", None) is None \
        and not getattr(engine, "_has_comm", False)
    if keep:
        try:
            if hasattr(engine, "release_held"):
                engine.release_held()
            engine.set_profiling(False)
            engine.reset_counters()  # (the next call's counters are its own)
            getattr(engine, "_keep", {}).clear()
            with _ENGINE_POOL_LOCK:
                if key not in _ENGINE_POOL:
                    _ENGINE_POOL[key] = engine
                    return
        except Exception:  # noqa: BLE001 - an engine that cannot be tidied up is not kept
            pass
    engine.close()


def clear_engine_cache():
    """Close the idle engines `dictionary_indexing` calls left behind (and free their device memory)."""
    with _ENGINE_POOL_LOCK:
        engines = list(_ENGINE_POOL.values())
        _ENGINE_POOL.clear()
    for e in engines:
        try:
            e.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


atexit.register(clear_engine_cache)


class Group(Context):
    """Several GPUs behind the interface of one `Context` (`kpdi_group`, include/kpdi.h): the
    experimental set is replicated, every dictionary chunk is block-assigned to the members,
    `finalize()` returns the one merged result.  One Python thread drives it; the members' host
    work runs on the library's own threads.

    gather: None (automatic: RCCL when the devices are distinct, peer copies otherwise,
    $KPDI_GATHER overrides), "rccl" or "p2p"."""

    _prefix = "kpdi_group_"
    _GATHER = {None: GATHER_AUTO, "auto": GATHER_AUTO, "rccl": GATHER_RCCL, "p2p": GATHER_P2P}

    def __init__(self, devices="all", gather=None):
        ids = resolve_devices(devices)
        if gather not in self._GATHER:
            raise KpdiError(f"gather must be None, 'rccl' or 'p2p', not {gather!r}")
        self._f = _Functions(self._prefix)
        self._owned = True
        self._h = C.c_void_p()
        arr = (C.c_int * len(ids))(*ids)
        check(self._f.create(arr, len(ids), self._GATHER[gather], C.byref(self._h)))
        self.devices = list(ids)
        self.device = ids[0]
        self.members = [Context(d, _borrowed=self._f.member(self._h, i)) for i, d in enumerate(ids)]
        self.root = self.members[0]
        self._keep = {}
        self._borrowed = []  # (ticket, array): host chunks the members' uploads may still be reading
        self._host_gather = None
        self._keep_n = None
        self._compute = COMPUTE_F32
        self._projection_key = None
        self.result_token = 0
        self._last_valid = False

    def __len__(self):
        return len(self.devices)

    @property
    def gather(self):
        """"rccl", "p2p" or "none" (one device): how the members' lists reach member 0."""
        return GATHER_NAMES[int(self._f.gather(self._h))]

    def describe(self):
        return (self._f.describe(self._h) or b"").decode()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for m in self.members:
                m._h = C.c_void_p()
            self._f.destroy(self._h)  # (joins the members' threads: nothing reads a borrowed chunk afterwards)
            self._h = C.c_void_p()
            self._borrowed = []

    @staticmethod
    def chunk_share(n_chunk, i, n_dev):
        """Rows [start, end) of the `i`-th of `n_dev` near-equal contiguous parts of `n_chunk` rows (a member's quota)."""
        a, b = C.c_int64(0), C.c_int64(0)
        check(load().kpdi_group_chunk_share(int(n_chunk), int(i), int(n_dev), C.byref(a), C.byref(b)))
        return a.value, b.value

    @staticmethod
    def assign_chunk(n_dev, n_total, loads, n_chunk, min_piece=1):
        """The library's chunk assignment (csrc/group_assign.h) as a pure function: [(member, first row, rows), ...]
        for a chunk of `n_chunk` patterns; `loads` (list, one entry per member) is updated in place."""
        ld = (C.c_int64 * n_dev)(*[int(v) for v in loads])
        cap = n_dev + 1
        mem, row0, rows, n = (C.c_int * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)(), C.c_int(0)
        check(load().kpdi_group_assign_chunk(int(n_dev), int(n_total), int(min_piece), ld, int(n_chunk), cap, mem, row0,
                                             rows, C.byref(n)))
        loads[:] = list(ld)
        return [(mem[i], row0[i], rows[i]) for i in range(n.value)]

    def set_dictionary_size(self, n_total):
        check(self._f.set_dictionary_size(self._h, int(n_total)))

    def push_dictionary_chunk(self, patterns, global_start):
        """Queue the chunk on the member(s) that take it and return: the array is borrowed by the library (and kept
        alive here) until the members' uploads have read it."""
        p = np.ascontiguousarray(patterns)
        t = C.c_int64(0)
        check(self._f.push_dictionary_chunk_borrowed(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0], int(global_start),
                                                     C.byref(t)))
        self._borrowed.append((t.value, p))
        self._drop_consumed()

    def _drop_consumed(self, everything=False):
        if everything:
            self._borrowed.clear()
        elif self._borrowed:
            t = C.c_int64(0)
            check(self._f.chunks_consumed(self._h, C.byref(t)))
            self._borrowed = [(k, a) for k, a in self._borrowed if k > t.value]

    def synchronize(self):
        super().synchronize()
        self._drop_consumed(True)

    def _after_finalize(self, ok):
        """Borrowed host chunks after a finalize: the C call joins its members' host work before anything is gathered, so
        after a SUCCESSFUL return every chunk has been read and all may go.  After an exception the call may have failed
        before that join (a Python-side check, an allocation): queued uploads may still be reading the arrays - only what
        the library reports as consumed is dropped (use-after-free otherwise)."""
        if ok:
            self._drop_consumed(True)
            return
        try:
            self._drop_consumed()
        except KpdiError:
            pass

    def finalize(self, keep_n=None):
        ok = False
        try:
            res = super().finalize(keep_n)
            ok = True
            return res
        finally:
            self._after_finalize(ok)

    def finalize_async(self, keep_n=None):
        ok = False
        try:
            res = super().finalize_async(keep_n)
            ok = True
            return res
        finally:
            self._after_finalize(ok)

    # -- what differs from a single context
    def set_problem(self, sy, sx, signal_mask=None, metric=METRIC_NCC, keep_n=20, compute=COMPUTE_F32):
        super().set_problem(sy, sx, signal_mask, metric, keep_n, compute)
        for m in self.members:
            m._keep_n, m._compute = self._keep_n, self._compute

    def set_keep_n(self, keep_n):
        super().set_keep_n(keep_n)
        for m in self.members:
            m._keep_n = self._keep_n

    def set_experimental(self, patterns, navigation_mask=None):
        p = np.ascontiguousarray(patterns)
        nm = _mask_bytes(navigation_mask)
        if nm is not None and nm.size != p.shape[0]:
            raise KpdiError(f"navigation mask has {nm.size} elements, there are {p.shape[0]} patterns")
        # (returns when every member's upload has consumed `p`)
        check(self._f.set_experimental(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0], _ptr(nm)))
        self._exp_shape, self._exp_dtype = p.shape, p.dtype

    def set_experimental_h5ebsd(self, path, scan=None, navigation_mask=None):
        info, pats, _, _ = h5ebsd_read(path, scan)
        self.set_experimental(pats.reshape((-1,) + pats.shape[-2:]), navigation_mask)
        return info

    def set_experimental_dev(self, d_ptrs, dtype, m_all, navigation_mask=None):
        """d_ptrs: one device pointer per member (the set in that member's HBM)."""
        if len(d_ptrs) != len(self.members):
            raise KpdiError(f"{len(d_ptrs)} device pointers for {len(self.members)} members")
        nm = _mask_bytes(navigation_mask)
        arr = (C.c_void_p * len(d_ptrs))(*[int(p) for p in d_ptrs])
        check(self._f.set_experimental_dev(self._h, arr, dtype_code(dtype), int(m_all), _ptr(nm)))

    def push_dictionary_chunk_dev(self, d_ptrs, dtype, n_chunk, global_start):
        """One device chunk per member: d_ptrs[i], n_chunk[i] patterns (0 = none), first dictionary
        index global_start[i]."""
        n = len(self.members)
        if not (len(d_ptrs) == len(n_chunk) == len(global_start) == n):
            raise KpdiError(f"push_dictionary_chunk_dev needs one entry per member ({n})")
        ptrs = (C.c_void_p * n)(*[int(p) if p else None for p in d_ptrs])
        cnt = (C.c_int64 * n)(*[int(v) for v in n_chunk])
        st = (C.c_int64 * n)(*[int(v) for v in global_start])
        check(self._f.push_dictionary_chunk_dev(self._h, ptrs, dtype_code(dtype), cnt, st))

    def hold_dictionary_chunk_dev(self, d_ptrs, dtype, n_chunk, global_start):
        for m, p, n, s0 in zip(self.members, d_ptrs, n_chunk, global_start):
            if n > 0:
                m.hold_dictionary_chunk_dev(p, dtype, n, s0)

    def set_direction_cosines(self, direction_cosines):
        for m in self.members:
            m.set_direction_cosines(direction_cosines)
        self._projection_key = None
        self._dc_npix = self.root._dc_npix

    def set_detector(self, gnomonic_bounds, pcz, nrows, ncols, om_detector_to_sample):
        super().set_detector(gnomonic_bounds, pcz, nrows, ncols, om_detector_to_sample)
        self.root._dc_npix = self._dc_npix

    # -- calls without a group form run on member 0 (every member holds the same patterns / master pattern; the merged
    # lists of the last finalize live there)
    def get_direction_cosines(self):
        return self.root.get_direction_cosines()

    def project_patterns(self, *args, **kwargs):
        return self.root.project_patterns(*args, **kwargs)

    def push_rotations_chunk_varying_pc(self, rotations, pcs, global_start, om_detector_to_sample, rescale=False,
                                        out_min=-1.0, out_max=1.0):
        """One contiguous block of the chunk per member (the members' queues are joined first: this path drives the member
        contexts from the caller's thread)."""
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        pc = np.ascontiguousarray(pcs, dtype=np.float64).reshape(-1, 3)
        self.synchronize()
        for i, m in enumerate(self.members):
            a, b = self.chunk_share(rot.shape[0], i, len(self.members))
            if b > a:
                m.push_rotations_chunk_varying_pc(rot[a:b], pc[a:b], global_start + a, om_detector_to_sample, rescale, out_min, out_max)

    def project_patterns_varying_pc(self, *args, **kwargs):
        return self.root.project_patterns_varying_pc(*args, **kwargs)

    def refine_set_patterns(self, *args, **kwargs):
        return self.root.refine_set_patterns(*args, **kwargs)

    def refine_get_prepared(self):
        return self.root.refine_get_prepared()

    def refine_objective(self, *args, **kwargs):
        return self.root.refine_objective(*args, **kwargs)

    def refine_solve(self, *args, **kwargs):
        return self.root.refine_solve(*args, **kwargs)

    def nelder_mead_selftest(self, *args, **kwargs):
        return self.root.nelder_mead_selftest(*args, **kwargs)

    def orientation_similarity_map(self, *args, **kwargs):
        return self.root.orientation_similarity_map(*args, **kwargs)

    def holds_result(self, simulation_indices):
        self.root._last_valid = self._last_valid
        return self.root.holds_result(simulation_indices)

    def comm_init(self, rank, nranks, unique_id):
        raise KpdiError("a Group gathers inside one process; ranks of a multi-process job attach a Context each")

    def dev_alloc(self, nbytes):
        raise KpdiError("device buffers belong to one GPU: use group.members[i].dev_alloc()")

    dev_free = h2d = d2h = dev_alloc

    # -- measurement
    def counters(self):
        """Member 0's counters (its merge / gather figures are the group's) + `members`: every member's."""
        per = [m.counters() for m in self.members]
        out = dict(per[0])
        out["members"] = per
        out["gather"] = self.gather
        return out
