"""One rank's share of configs[1] at N = 8 (4096 x 12 500 x 60 x 60) with the wide f32 kernel, a few steps: what a developer
build of the kernel (tools/build_variant.sh ... -DKPDI16_TIME_PHASES) prints.   KPDI_LIB_PATH=build/variants/libkpdi_<tag>.so python tools/probes/share_step.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["KPDI_F32_WIDE"] = "1"
from kikuchipy_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
rng = np.random.default_rng(3)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((n, 60, 60), dtype=np.float32)
with _lib.Context(0) as ctx:
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    d = ctx.dev_alloc(dic.nbytes)
    ctx.h2d(d, dic)
    for rep in range(3):
        print("--- step", rep, flush=True)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk_dev(d, np.float32, len(dic), 0)
        ctx.finalize(20)
        ctx.synchronize()
