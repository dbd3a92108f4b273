"""Cost of float64 arithmetic (COMPUTE_F64: f32 screen + float64 rescoring) under its two certification bounds.

    python tools/f64_probe.py [out.txt]

configs[1] (4096 x 100 000 x 60 x 60, ncc, keep_n 20) and an ADVERSARIAL set of the same detector (4096 x 20 000: 40
dictionary patterns are copies of one base pattern that differ by a few 1e-8 relative, every experimental pattern is
that base pattern + noise - the f32 screen cannot rank its best 40): step time, extra screening passes and uncertified
patterns with the worst-case bound (the default since round 5: a proof for any data) and with the statistical one
(KPDI_F64_EPS=statistical, round 4's default), next to the plain f32 sweep."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

lines = []


def say(text):
    print(text, flush=True)
    lines.append(text)


def data(kind):
    rng = np.random.default_rng(2024)
    s = 60
    if kind == "configs[1]":
        return rng.integers(0, 256, (4096, s, s), dtype=np.uint8), rng.random((100000, s, s), dtype=np.float32)
    n = 20000
    dic = rng.random((n, s, s), dtype=np.float32)
    base = rng.random((s, s)).astype(np.float32)
    for j, d in enumerate(rng.permutation(n)[:40]):
        t = base.copy()
        px = rng.integers(0, s * s, 3)
        t.ravel()[px] *= np.float32(1 + (j + 1) * 3e-8)
        dic[d] = t
    exp = np.clip(base * 255 + rng.normal(0, 2, (4096, s, s)), 0, 255).astype(np.uint8)
    return exp, dic


for kind in ("configs[1]", "adversarial near-ties"):
    exp, dic = data(kind)
    m, n = len(exp), len(dic)
    say(f"== {kind}: {m} x {n} x 60 x 60, ncc, keep_n 20, raw dictionary resident")
    base_ms = None
    for name, mode, eps in (("f32", _lib.COMPUTE_F32, None), ("f64 worst-case bound (default)", _lib.COMPUTE_F64, None),
                            ("f64 statistical bound", _lib.COMPUTE_F64, "statistical")):
        if eps:
            os.environ["KPDI_F64_EPS"] = eps
        else:
            os.environ.pop("KPDI_F64_EPS", None)
        with _lib.Context(0) as ctx:
            d = ctx.dev_alloc(dic.nbytes)
            ctx.h2d(d, dic)
            ctx.set_profiling(True)
            ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20, mode)
            ctx.set_experimental(exp)
            best = 1e9
            for rep in range(5):
                ctx.reset_topk()
                ctx.reset_counters()
                ctx.synchronize()
                t0 = time.perf_counter()
                ctx.push_dictionary_chunk_dev(d, np.float32, n, 0)
                ctx.synchronize()
                best = min(best, time.perf_counter() - t0)
                c = ctx.counters()
            if base_ms is None:
                base_ms = best * 1e3
            say(f"  {name:32s} step {best * 1e3:7.2f} ms ({best * 1e3 / base_ms:5.3f} x f32)  match {c['match_ms']:6.2f}  "
                f"rescore {c['rescore_ms']:5.2f}  match launches {c['match_launches']}  extra passes {c['rescore_extra_passes']}  "
                f"uncertified {c['uncertified_patterns']}  certificate {c['f64_certificate']}")
            ctx.dev_free(d)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")
