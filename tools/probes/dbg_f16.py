import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import test_gpu_f16 as T
m, n, sy, sx, k = 257, 900, 6, 37, 64
rng = np.random.default_rng(m * 1000 + n)
exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
dic = rng.random((n, sy, sx)).astype(np.float32)
nav = np.zeros(m, dtype=bool); nav[[2, m - 1]] = True
s, i = T.engine(exp, dic, "ncc", k, None, None, nav)
rs, ri = T.rounded_operand_topk(exp, dic, "ncc", k, None, nav)
bad = np.abs(s - rs) > 3e-5
print("bad entries", bad.sum(), "of", bad.size, "rows with bad", bad.any(1).sum())
print("bad by rank:", bad.sum(0))
r = np.argmax(bad.any(1)); print("row", r); print(np.c_[i[r], ri[r], s[r], rs[r]][20:48])
missing = [len(set(ri[q]) - set(i[q])) for q in range(len(i))]
print("missing candidates per row (max, total):", max(missing), sum(missing))
