"""One resident-data step of 4096 x 49 152 x 60 x 60 with the wide f32 kernel: what a developer build of the kernel (e.g.
-DKPDI16_TIME_EPI, tools/build_variant.sh) prints."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["KPDI_F32_WIDE"] = "1"
from kikuchipy_amd import _lib
rng = np.random.default_rng(3)
exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
dic = rng.random((49152, 60, 60), dtype=np.float32)
with _lib.Context(0) as ctx:
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    ctx.set_experimental(exp)
    d = ctx.dev_alloc(dic.nbytes)
    ctx.h2d(d, dic)
    ctx.push_dictionary_chunk_dev(d, np.float32, len(dic), 0)
    ctx.finalize(20)
    ctx.synchronize()
