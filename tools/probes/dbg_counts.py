import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kikuchipy_amd import _lib
from oracle import kpdi_oracle as ko
rng = np.random.default_rng(3)
n, m = int(sys.argv[1]), int(sys.argv[2])
dic = rng.random((n, 60, 60), dtype=np.float32)
exp = rng.integers(0, 256, (m, 60, 60), dtype=np.uint8)
for compute in (_lib.COMPUTE_F16, _lib.COMPUTE_F32):
    for chunk in (n, 7000):
        with _lib.Context(0) as c:
            c.set_problem(60, 60, None, _lib.METRIC_NCC, 20, compute)
            c.set_experimental(exp)
            for a in range(0, n, chunk):
                c.push_dictionary_chunk(dic[a:a + chunk], a)
            s, i = c.finalize(20)
            cnt = c.counters()
        fs, fi = ko.dictionary_indexing(exp, dic, keep_n=20)
        err = np.abs(s - fs)
        bad = np.argwhere(err > 1e-3)
        print("compute", compute, "chunk", chunk, "nsplit", cnt["match_nsplit"], "form", cnt["match_form"], "max err", err.max(), "bad entries", len(bad), bad[:5].tolist(), flush=True)
