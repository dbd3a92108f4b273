"""One member's work of a CHUNKED dictionary_indexing call on a kpdi_group, old rule against new, on ONE GPU.

The reference's canonical call chunks the dictionary (`n_per_iteration` = a tenth of it, 3044 patterns, in
doc/tutorials/pattern_matching.ipynb:582; loop indexing/_dictionary_indexing.py:100-128).  For configs[1]
(4096 x 100 000 x 60 x 60) pushed in chunks of `--per` patterns to a group of `--members` this script times, with the
raw dictionary resident in HBM:

  single   every chunk on one context = what ONE GPU does for the same call (t_1);
  old      the pieces member i got while every chunk was cut n_dev ways (rounds 3-4: kpdi_group_chunk_share per chunk);
  new      the pieces member i gets under csrc/group_assign.h (kpdi_group_assign_chunk with the size announced).

Reported per rule: the slowest member's time and its ratio to the even share t_1 / n_dev (target <= 1.15).

    python tools/group_chunk_probe.py [out.txt] [--per 3044] [--members 8] [--reps 10] [--m 4096] [--n 100000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kikuchipy_amd import _lib  # noqa: E402
from kikuchipy_amd.indexing._dictionary_indexing import chunk_bounds  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?")
ap.add_argument("--per", type=int, default=3044)
ap.add_argument("--members", type=int, default=8)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--n", type=int, default=100000)
a = ap.parse_args()

m, n, sy, sx, keep, n_dev = a.m, a.n, 60, 60, 20, a.members
rng = np.random.default_rng(2024)
exp = rng.integers(0, 256, (m, sy, sx), dtype=np.uint8)
dic = rng.random((n, sy, sx), dtype=np.float32)
chunks = chunk_bounds(n, a.per)

# the pieces (global start, rows) every member gets under each rule
old = [[] for _ in range(n_dev)]
new = [[] for _ in range(n_dev)]
loads = [0] * n_dev
for s, e in chunks:
    for i in range(n_dev):
        lo, hi = _lib.Group.chunk_share(e - s, i, n_dev)
        if hi > lo:
            old[i].append((s + lo, hi - lo))
    for member, row0, rows in _lib.Group.assign_chunk(n_dev, n, loads, e - s):
        new[member].append((s + row0, rows))
single = [[(s, e - s) for s, e in chunks]]

lines = []


def say(*parts):
    text = " ".join(str(p) for p in parts)
    print(text, flush=True)
    lines.append(text)


say(f"configs[1] pushed in chunks of {a.per} patterns ({len(chunks)} chunks) to a group of {n_dev}: one member's work on one "
    f"MI355X, raw dictionary resident, {a.reps} repetitions each (python tools/group_chunk_probe.py --per {a.per} --members {n_dev})")
with _lib.Context(0) as ctx:
    ctx.set_problem(sy, sx, None, _lib.METRIC_NCC, keep, _lib.COMPUTE_F32)
    d_exp = ctx.dev_alloc(exp.nbytes)
    ctx.h2d(d_exp, exp)
    d_dic = ctx.dev_alloc(dic.nbytes)
    ctx.h2d(d_dic, dic)
    row = sy * sx * 4

    def run(pieces):
        """ms per call of one member's pieces (set_experimental_dev + its pushes + finalize), result of the last call"""
        res = None
        for r in range(a.reps + 2):
            if r == 2:
                ctx.synchronize()
                t0 = time.perf_counter()
            ctx.set_experimental_dev(d_exp, exp.dtype, m)
            for start, rows in pieces:
                ctx.push_dictionary_chunk_dev(d_dic + start * row, np.float32, rows, start)
            res = ctx.finalize(keep)
        return (time.perf_counter() - t0) / a.reps * 1e3, res

    t1, (s_all, i_all) = run(single[0])
    ctx.set_experimental_dev(d_exp, exp.dtype, m)
    ctx.push_dictionary_chunk_dev(d_dic, np.float32, n, 0)
    s_one, i_one = ctx.finalize(keep)
    t_pass, _ = run([(0, n)])
    say(f"single   one GPU, all {len(chunks)} chunks: {t1:.3f} ms per call (one pass over the whole dictionary: {t_pass:.3f} ms); "
        f"chunked == single pass bit for bit: {bool(np.array_equal(s_all, s_one) and np.array_equal(i_all, i_one))}")
    even = t1 / n_dev
    say(f"         even share t_1 / {n_dev} = {even:.3f} ms")
    for name, rule in (("old", old), ("new", new)):
        times = []
        lists = []
        for i in range(n_dev):
            if i in (0, n_dev - 1) or len(rule[i]) != len(rule[0]) or sum(r for _, r in rule[i]) != sum(r for _, r in rule[0]):
                t, res = run(rule[i])
                times.append((t, i))
                lists.append(res)
        worst, who = max(times)
        sizes = sorted({r for p in rule for _, r in p})
        say(f"{name:8s} member pieces: {len(rule[0])} of {sizes[0]}..{sizes[-1]} patterns (member 0 takes "
            f"{sum(r for _, r in rule[0])} patterns); slowest measured member ({who}): {worst:.3f} ms = "
            f"{worst / even:.3f} x the even share")
    # the new rule's pieces, merged over all members, are the single sweep (total order of the merge): checked on the GPU
    # by tests/test_gpu_group.py::test_the_tutorial_call_on_eight_members_equals_the_single_sweep_at_full_size
if a.out:
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
