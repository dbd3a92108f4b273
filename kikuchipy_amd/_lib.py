"""ctypes binding of libkpdi.so (include/kpdi.h) - no PyTorch, no NumPy C-API.

The library is built in-tree by `make -C kikuchipy_amd/csrc` (or
`__graft_entry__.build()`).  There is no CPU fallback anywhere in this
package: if the library is missing, or no gfx950 GPU is visible, calls raise.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KPDI_LIB_PATH: another build of the library (kernel experiments, tools/README.md)
LIB_PATH = os.environ.get("KPDI_LIB_PATH") or os.path.join(_HERE, "csrc", "libkpdi.so")

METRIC_NCC, METRIC_NDP = 0, 1
COMPUTE_F32, COMPUTE_F16X2, COMPUTE_F16, COMPUTE_F64 = 0, 1, 2, 3
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
    "kpdi_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "kpdi_dev_free": (_i, [_vp, _vp]),
    "kpdi_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "kpdi_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "kpdi_set_profiling": (_i, [_vp, _i]),
    "kpdi_get_counters": (_i, [_vp, C.POINTER(Counters)]),
    "kpdi_reset_counters": (_i, [_vp]),
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


class Context:
    """One GPU, one stream: thin object wrapper over a `kpdi_ctx*`."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(load().kpdi_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._keep = {}  # host arrays the library may still be reading
        self._keep_n = None       # what kpdi_finalize will write per pattern
        self._compute = COMPUTE_F32
        self._projection_key = None  # simulations.ProjectedDictionary.configure
        self.result_token = 0     # bumped by every finalize()
        self._last_valid = False   # the last finalize() succeeded: its lists may still be resident in HBM

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().kpdi_destroy(self._h)
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
        check(load().kpdi_synchronize(self._h))

    # -- set-up
    def set_problem(self, sy, sx, signal_mask=None, metric=METRIC_NCC, keep_n=20, compute=COMPUTE_F32):
        sm = _mask_bytes(signal_mask)
        if sm is not None and sm.size != sy * sx:
            raise KpdiError(f"signal mask has {sm.size} elements, detector has {sy * sx}")
        check(load().kpdi_set_problem(self._h, int(sy), int(sx), _ptr(sm), int(metric), int(compute),
                                      int(keep_n)))
        self._keep_n = int(keep_n)
        self._compute = int(compute)

    def set_keep_n(self, keep_n):
        check(load().kpdi_set_keep_n(self._h, int(keep_n)))
        self._keep_n = int(keep_n)

    def set_experimental(self, patterns, navigation_mask=None):
        """patterns: (m_all, sy, sx) or (m_all, sy*sx), C-contiguous."""
        p = np.ascontiguousarray(patterns)
        nm = _mask_bytes(navigation_mask)
        m_all = p.shape[0]
        if nm is not None and nm.size != m_all:
            raise KpdiError(f"navigation mask has {nm.size} elements, there are {m_all} patterns")
        check(load().kpdi_set_experimental(self._h, _ptr(p), dtype_code(p.dtype), m_all, _ptr(nm)))
        check(load().kpdi_synchronize(self._h))  # upload done: `p` may be a temporary
        self._exp_shape, self._exp_dtype = p.shape, p.dtype

    def set_experimental_h5ebsd(self, path, scan=None, navigation_mask=None):
        """Patterns of a kikuchipy h5ebsd scan: file -> pinned host buffer -> HBM."""
        info = h5ebsd_info(path, scan)
        nm = _mask_bytes(navigation_mask)
        if nm is not None and nm.size != info.ny * info.nx:
            raise KpdiError(f"navigation mask has {nm.size} elements, the scan has {info.ny * info.nx} patterns")
        check(load().kpdi_set_experimental_h5ebsd(self._h, _cstr(path), info.scan, _ptr(nm)))
        self._exp_shape = (info.ny * info.nx, info.sy, info.sx)
        self._exp_dtype = DTYPE_FROM_CODE[info.dtype]
        return info

    def set_experimental_dev(self, d_ptr, dtype, m_all, navigation_mask=None):
        nm = _mask_bytes(navigation_mask)
        check(load().kpdi_set_experimental_dev(self._h, C.c_void_p(d_ptr), dtype_code(dtype), int(m_all),
                                               _ptr(nm)))

    @property
    def n_experimental(self):
        return int(load().kpdi_n_experimental(self._h))

    # -- pre-processing
    def remove_static_background(self, static_bg_f32, operation=OP_SUBTRACT, scale_bg=False):
        bg = np.ascontiguousarray(static_bg_f32, dtype=np.float32)
        check(load().kpdi_remove_static_background(self._h, _ptr(bg), int(operation), int(bool(scale_bg))))

    def remove_dynamic_background(self, operation=OP_SUBTRACT, filter_domain=DOMAIN_FREQUENCY, std=0.0,
                                  truncate=4.0):
        check(load().kpdi_remove_dynamic_background(self._h, int(operation), int(filter_domain),
                                                    float(std), float(truncate)))

    def get_experimental(self):
        out = np.empty(self._exp_shape, dtype=self._exp_dtype)
        check(load().kpdi_get_experimental(self._h, _ptr(out)))
        return out

    # -- sweep
    def push_dictionary_chunk(self, patterns, global_start):
        p = np.ascontiguousarray(patterns)
        # returns when the upload has consumed `p`; the sweep of the chunk runs on
        check(load().kpdi_push_dictionary_chunk(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0],
                                                int(global_start)))

    def push_dictionary_chunk_dev(self, d_ptr, dtype, n_chunk, global_start):
        check(load().kpdi_push_dictionary_chunk_dev(self._h, C.c_void_p(d_ptr), dtype_code(dtype),
                                                    int(n_chunk), int(global_start)))

    # -- resident dictionary: prepared once, swept against several experimental sets
    def hold_dictionary_chunk(self, patterns, global_start):
        p = np.ascontiguousarray(patterns)
        check(load().kpdi_hold_dictionary_chunk(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0],
                                                int(global_start)))

    def hold_dictionary_chunk_dev(self, d_ptr, dtype, n_chunk, global_start):
        check(load().kpdi_hold_dictionary_chunk_dev(self._h, C.c_void_p(d_ptr), dtype_code(dtype),
                                                    int(n_chunk), int(global_start)))

    def hold_rotations_chunk(self, rotations, global_start, rescale=False, out_min=-1.0, out_max=1.0):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        check(load().kpdi_hold_rotations_chunk(self._h, _ptr(rot), rot.shape[0], int(global_start),
                                               int(bool(rescale)), float(out_min), float(out_max)))

    def sweep_held(self):
        check(load().kpdi_sweep_held(self._h))

    def release_held(self):
        check(load().kpdi_release_held(self._h))

    def held_size(self):
        """(patterns held, bytes of device memory they occupy)."""
        n, b = C.c_int64(0), C.c_int64(0)
        check(load().kpdi_held_size(self._h, C.byref(n), C.byref(b)))
        return n.value, b.value

    # -- dictionary generation on the device
    def set_master_pattern(self, upper, lower=None):
        """upper / lower: (npy, npx) arrays of one dtype (uint8, uint16, float32, float64)."""
        up = np.ascontiguousarray(upper)
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=up.dtype)
        if up.ndim != 2 or (lo is not None and lo.shape != up.shape):
            raise KpdiError("master pattern hemispheres must be 2D arrays of equal shape")
        self._projection_key = None  # whoever cached "my master pattern is loaded" must load it again
        check(load().kpdi_set_master_pattern(self._h, _ptr(up), _ptr(lo), dtype_code(up.dtype),
                                             up.shape[1], up.shape[0]))

    def set_detector(self, gnomonic_bounds, pcz, nrows, ncols, om_detector_to_sample):
        gb = np.ascontiguousarray(gnomonic_bounds, dtype=np.float64).ravel()
        om = np.ascontiguousarray(om_detector_to_sample, dtype=np.float64).ravel()
        if gb.size != 4 or om.size != 9:
            raise KpdiError("gnomonic_bounds must have 4 and om_detector_to_sample 9 elements")
        self._projection_key = None
        check(load().kpdi_set_detector(self._h, _ptr(gb), float(pcz), int(nrows), int(ncols), _ptr(om)))
        self._dc_npix = int(nrows) * int(ncols)

    def set_direction_cosines(self, direction_cosines):
        dc = np.ascontiguousarray(direction_cosines, dtype=np.float64).reshape(-1, 3)
        self._projection_key = None
        check(load().kpdi_set_direction_cosines(self._h, _ptr(dc), dc.shape[0]))
        self._dc_npix = dc.shape[0]

    def get_direction_cosines(self):
        out = np.empty((self._dc_npix, 3), dtype=np.float64)
        check(load().kpdi_get_direction_cosines(self._h, _ptr(out)))
        return out

    def project_patterns(self, rotations, rescale=False, out_min=-1.0, out_max=1.0, dtype_out=np.float32):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        out = np.empty((rot.shape[0], self._dc_npix), dtype=dtype_out)
        check(load().kpdi_project_patterns(self._h, _ptr(rot), rot.shape[0], int(bool(rescale)),
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
        check(load().kpdi_project_patterns_varying_pc(self._h, _ptr(rot), _ptr(pc), rot.shape[0], int(shape[0]),
                                                      int(shape[1]), _ptr(om), int(bool(rescale)), float(out_min),
                                                      float(out_max), dtype_code(out.dtype), _ptr(out)))
        return out

    def push_rotations_chunk(self, rotations, global_start, rescale=False, out_min=-1.0, out_max=1.0):
        rot = np.ascontiguousarray(rotations, dtype=np.float64).reshape(-1, 4)
        check(load().kpdi_push_rotations_chunk(self._h, _ptr(rot), rot.shape[0], int(global_start),
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
        check(load().kpdi_refine_set_patterns(self._h, _ptr(p), dtype_code(p.dtype), p.shape[0], p.shape[1],
                                              p.shape[2], _ptr(sm), int(bool(rescale)), _ptr(om)))
        self._ref_n = p.shape[0]
        self._ref_k = p.shape[1] * p.shape[2] if sm is None else int(np.count_nonzero(sm == 0))

    def refine_get_prepared(self):
        pat = np.empty((self._ref_n, self._ref_k), dtype=np.float32)
        sqn = np.empty(self._ref_n, dtype=np.float64)
        check(load().kpdi_refine_get_prepared(self._h, _ptr(pat), _ptr(sqn)))
        return pat, sqn

    def refine_objective(self, mode, pattern_index, x, fixed=None):
        idx = np.ascontiguousarray(pattern_index, dtype=np.int32).ravel()
        nvar, nfixed = REFINE_SIZES[mode]
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(idx.size, nvar)
        f = None if nfixed == 0 else np.ascontiguousarray(fixed, dtype=np.float64).reshape(idx.size, nfixed)
        out = np.empty(idx.size, dtype=np.float64)
        check(load().kpdi_refine_objective(self._h, int(mode), idx.size, _ptr(idx), _ptr(x), _ptr(f), _ptr(out)))
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
        check(load().kpdi_refine_solve(self._h, int(mode), n, starts, _ptr(x0), _ptr(f), _ptr(lo), _ptr(hi),
                                       float(xatol), float(fatol), int(maxiter or 0), int(maxfev or 0), _ptr(res)))
        return res[:, :, :3 + nvar]

    def nelder_mead_selftest(self, kind, x0, lower=None, upper=None, xatol=1e-4, fatol=1e-4, maxiter=0, maxfev=0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).ravel()
        lo = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64).ravel()
        hi = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64).ravel()
        res = np.empty(3 + x0.size, dtype=np.float64)
        check(load().kpdi_nelder_mead_selftest(self._h, int(kind), x0.size, _ptr(x0), _ptr(lo), _ptr(hi),
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
        check(load().kpdi_orientation_similarity_map(self._h, _ptr(idx), int(ny), int(nx), int(keep_n), int(n_best),
                                                     int(from_n_best), _ptr(off), off.shape[0], int(center_index),
                                                     int(bool(normalize)), _ptr(out)))
        return out

    def reset_topk(self):
        check(load().kpdi_reset_topk(self._h))
        self._last_valid = False

    def holds_result(self, simulation_indices):
        """Whether `simulation_indices` (n, keep_n) are the lists the last finalize() returned
        (and that may therefore still be resident in HBM)."""
        if not self._last_valid:
            return False
        # what kpdi_finalize left in its page-locked staging buffer (no copy was kept: the caller may have masked or
        # remapped the array it got in place, and must then not be told that the device still holds "these" lists)
        p, n = C.POINTER(C.c_int32)(), C.c_int64(0)
        check(load().kpdi_result_indices_i32(self._h, C.byref(p), C.byref(n)))
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
        m = self.n_experimental
        indices = np.empty((m, keep_n), dtype=np.int64)
        if self._compute == COMPUTE_F64:  # float64 arithmetic: the rescored scores (csrc/rescore.hip)
            scores = np.empty((m, keep_n), dtype=np.float64)
            check(load().kpdi_finalize_f64(self._h, _ptr(scores), _ptr(indices)))
        else:
            scores = np.empty((m, keep_n), dtype=np.float32)
            check(load().kpdi_finalize(self._h, _ptr(scores), _ptr(indices)))
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
        t = C.c_int(-1)
        check(load().kpdi_finalize_async(self._h, C.byref(t)))
        self.result_token += 1
        self._last_valid = False
        return (t.value, int(keep_n))

    def finalize_wait(self, ticket):
        """(scores (m, keep_n) float32, indices (m, keep_n) int64) of a `finalize_async` ticket."""
        slot, keep_n = ticket
        n = C.c_int64(0)
        check(load().kpdi_pending_result_size(self._h, int(slot), C.byref(n)))
        m = n.value // keep_n
        scores = np.empty((m, keep_n), dtype=np.float32)
        indices = np.empty((m, keep_n), dtype=np.int64)
        check(load().kpdi_finalize_wait(self._h, int(slot), _ptr(scores), _ptr(indices)))
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
        check(load().kpdi_comm_init(self._h, int(rank), int(nranks), _ptr(buf)))

    # -- device buffers
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(load().kpdi_dev_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, d_ptr):
        check(load().kpdi_dev_free(self._h, C.c_void_p(d_ptr)))

    def h2d(self, d_ptr, array):
        a = np.ascontiguousarray(array)
        check(load().kpdi_h2d(self._h, C.c_void_p(d_ptr), _ptr(a), a.nbytes))

    def d2h(self, array, d_ptr):
        check(load().kpdi_d2h(self._h, _ptr(array), C.c_void_p(d_ptr), array.nbytes))

    # -- measurement
    def set_profiling(self, on=True):
        check(load().kpdi_set_profiling(self._h, int(bool(on))))

    def counters(self):
        c = Counters()
        check(load().kpdi_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        check(load().kpdi_reset_counters(self._h))
