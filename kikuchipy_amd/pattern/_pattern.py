"""Static / dynamic background removal on the GPU.

Array-level counterparts of `EBSD.remove_static_background`
(signals/ebsd.py:442-573) and `EBSD.remove_dynamic_background`
(signals/ebsd.py:575-696): same arguments, defaults, validation messages and
output dtype; the per-pattern kernels of pattern/_pattern.py:392-509 run as
one HIP workgroup per pattern (csrc/preproc.hip).
"""

import numpy as np

from kikuchipy_amd import _lib

_OPS = {"subtract": _lib.OP_SUBTRACT, "divide": _lib.OP_DIVIDE}
_DOMAINS = {"frequency": _lib.DOMAIN_FREQUENCY, "spatial": _lib.DOMAIN_SPATIAL}
_SUPPORTED = (np.uint8, np.uint16, np.int8, np.int16, np.float32, np.float64)


def _context(context, device):
    return context if context is not None else _lib.Context(device)


def _upload(ctx, patterns):
    patterns = np.asarray(patterns)
    if patterns.ndim < 2:
        raise ValueError("patterns need at least the two detector axes")
    if patterns.dtype.type not in _SUPPORTED:
        raise ValueError(f"pattern dtype {patterns.dtype} is not supported by the GPU pre-processing kernels")
    sy, sx = patterns.shape[-2:]
    flat = np.ascontiguousarray(patterns).reshape((-1, sy, sx))
    ctx.set_problem(sy, sx, None, _lib.METRIC_NCC, 1)
    ctx.set_experimental(flat)
    return patterns.shape


def _process(patterns, record, context, device, contexts):
    """Upload -> `record(ctx)` (the recorded step) -> download, on one context or - `contexts`: one per GPU, the members of a
    `_lib.Group` - block-wise: the patterns are independent, every GPU takes a contiguous block of them over its own host
    link from a host thread of its own (the library calls release the GIL); the blocks are concatenated."""
    patterns = np.asarray(patterns)
    if contexts and len(contexts) > 1 and patterns.ndim > 2 and int(np.prod(patterns.shape[:-2])) >= len(contexts):
        from concurrent.futures import ThreadPoolExecutor

        from kikuchipy_amd.parallel import shard_range

        flat = np.ascontiguousarray(patterns).reshape((-1,) + patterns.shape[-2:])
        blocks = [shard_range(len(flat), i, len(contexts)) for i in range(len(contexts))]

        def one(job):
            c, (a, b) = job
            _upload(c, flat[a:b])
            record(c)
            return c.get_experimental()

        with ThreadPoolExecutor(len(contexts)) as pool:
            parts = list(pool.map(one, zip(contexts, blocks)))
        return np.concatenate(parts, axis=0).reshape(patterns.shape)
    ctx = contexts[0] if contexts else _context(context, device)
    try:
        shape = _upload(ctx, patterns)
        record(ctx)
        return ctx.get_experimental().reshape(shape)
    finally:
        if context is None and not contexts:
            ctx.close()


def check_static_background(patterns_dtype, sig_shape, static_bg):
    """Validation of signals/ebsd.py:525-546."""
    if not isinstance(static_bg, np.ndarray):
        if hasattr(static_bg, "compute"):
            static_bg = static_bg.compute()
        else:
            raise ValueError("`EBSD.static_background` is not a valid array")
    dtype_out = np.dtype(patterns_dtype).type
    if dtype_out != static_bg.dtype:
        raise ValueError(
            f"Static background dtype_out {static_bg.dtype} is not the same as "
            f"pattern dtype_out {dtype_out}"
        )
    if static_bg.shape != tuple(sig_shape):
        raise ValueError(
            f"Signal {tuple(sig_shape)} and static background {static_bg.shape} shapes are not "
            "the same"
        )
    return static_bg.astype(np.float32)


def remove_static_background(patterns, static_bg, operation="subtract", scale_bg=False, *,
                             context=None, device=0, contexts=None):
    """Remove the static background from every pattern; returns a new array of
    the input dtype.  `static_bg` must have the patterns' dtype and detector
    shape, as in the reference."""
    if operation not in _OPS:
        raise ValueError(f"operation '{operation}' must be either 'subtract' or 'divide'")
    patterns = np.asarray(patterns)
    bg = check_static_background(patterns.dtype, patterns.shape[-2:], static_bg)
    return _process(patterns, lambda c: c.remove_static_background(bg, _OPS[operation], scale_bg), context, device, contexts)


def remove_dynamic_background(patterns, operation="subtract", filter_domain="frequency", std=None,
                              truncate=4.0, *, context=None, device=0, contexts=None):
    """Remove the dynamic background (Gaussian blur of each pattern, by
    subtraction or division) from every pattern; returns a new array of the
    input dtype.  `std` defaults to an eighth of the pattern width."""
    if filter_domain not in _DOMAINS:
        raise ValueError(f"{filter_domain} must be either of {list(_DOMAINS)}")
    if operation not in _OPS:
        raise ValueError(f"operation '{operation}' must be either 'subtract' or 'divide'")
    patterns = np.asarray(patterns)
    if std is None:
        std = patterns.shape[-1] / 8
    return _process(patterns, lambda c: c.remove_dynamic_background(_OPS[operation], _DOMAINS[filter_domain], std, truncate),
                    context, device, contexts)
