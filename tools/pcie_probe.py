"""Host-pointer push timing (developer tool): plain H2D copy vs the pipelined sweep."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kikuchipy_amd import _lib  # noqa: E402

rng = np.random.default_rng(0)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((100000, 60, 60), dtype=np.float32)
ctx = _lib.Context(0)
ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
ctx.set_experimental(exp)
d = ctx.dev_alloc(dic.nbytes)
for rep in range(3):
    t0 = time.perf_counter()
    ctx.h2d(d, dic)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"plain H2D {dic.nbytes/1e9:.2f} GB: {dt*1e3:.1f} ms = {dic.nbytes/dt/1e9:.1f} GB/s", flush=True)
for rep in range(3):
    ctx.reset_topk()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.push_dictionary_chunk(dic, 0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"push (host pointer): {dt*1e3:.1f} ms = {4096/dt:.0f} patterns/s", flush=True)
for n in (25000, 50000):
    ctx.reset_topk()
    ctx.synchronize()
    t0 = time.perf_counter()
    for a in range(0, 100000, n):
        ctx.push_dictionary_chunk(dic[a:a + n], a)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"push in chunks of {n}: {dt*1e3:.1f} ms = {4096/dt:.0f} patterns/s", flush=True)
# piece size of the upload/sweep pipeline (KPDI_UPLOAD_TILES=<dictionary tiles per piece>)
for sched in ("", "32", "48", "64", "80", "96", "128", "192"):
    if sched:
        os.environ["KPDI_UPLOAD_TILES"] = sched
    best = 1e9
    for rep in range(4):
        ctx.reset_topk()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.push_dictionary_chunk(dic, 0)
        ctx.finalize(20)
        best = min(best, time.perf_counter() - t0)
    print(f"pieces of {sched or 'default'} tiles: {best*1e3:.1f} ms = {4096/best:.0f} patterns/s", flush=True)
os.environ.pop("KPDI_UPLOAD_TILES")
for m in (1024, 10000):
    ctx.set_experimental(rng.integers(0, 256, (m, 60, 60), dtype=np.uint8))
    for sched in ("", "64", "128", "192", "256"):
        os.environ.pop("KPDI_UPLOAD_TILES", None)
        if sched:
            os.environ["KPDI_UPLOAD_TILES"] = sched
        best = 1e9
        for rep in range(3):
            ctx.reset_topk()
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.push_dictionary_chunk(dic, 0)
            ctx.finalize(20)
            best = min(best, time.perf_counter() - t0)
        print(f"m={m} pieces of {sched or 'default'} tiles: {best*1e3:.1f} ms = {m/best:.0f} patterns/s", flush=True)
# a float16 dictionary (half the bytes over PCIe; exact cast to float32 on the device)
os.environ.pop("KPDI_UPLOAD_TILES", None)
ctx.set_experimental(exp)
dic16 = dic.astype(np.float16)
for rep in range(3):
    ctx.reset_topk()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.push_dictionary_chunk(dic16, 0)
    ctx.finalize(20)
    dt = time.perf_counter() - t0
    print(f"push float16 dictionary: {dt*1e3:.1f} ms = {4096/dt:.0f} patterns/s", flush=True)
