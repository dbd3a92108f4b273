"""ms per step of the two f32 match kernels (forced) and the automatic choice at a few (M, N, K) points:
    KPDI_LIB_PATH=... python tools/probes/form_point.py
(tools/form_probe.py's time_step; the A/B of a rebuilt library against another on one box)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kikuchipy_amd import _lib  # noqa: E402

spec = importlib.util.spec_from_file_location("form_probe", os.path.join(ROOT, "tools", "form_probe.py"))
fp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fp)
rng = np.random.default_rng(5)
points = [(512, 25000, 3600), (4096, 37500, 2819), (4096, 12500, 3600), (4096, 100000, 3600), (10000, 50000, 3600)]
pool = rng.random(100000 * 3600, dtype=np.float32)
exp_pool = rng.integers(0, 256, 40000 * 3600, dtype=np.uint8)
with _lib.Context(0) as ctx:
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    d_dic = ctx.dev_alloc(pool.nbytes)
    ctx.h2d(d_dic, pool)
    d_exp = ctx.dev_alloc(exp_pool.nbytes)
    ctx.h2d(d_exp, exp_pool)
    for m, n, k in points:
        mask = fp.circular_mask(60) if k == 2819 else None
        ms = {}
        for rep in range(2):
            for name, env in (("classic", "0"), ("wide", "1"), ("auto", None)):
                if env is None:
                    os.environ.pop("KPDI_F32_WIDE", None)
                else:
                    os.environ["KPDI_F32_WIDE"] = env
                ms.setdefault(name, []).append(round(fp.time_step(ctx, d_exp, m, d_dic, n, 60, mask, _lib.METRIC_NCC, 8)[0], 4))
        os.environ.pop("KPDI_F32_WIDE", None)
        print(m, n, k, ms, flush=True)
