"""Which member of a `kpdi_group` takes which rows of a pushed dictionary chunk (csrc/group_assign.h through the
exported pure function `kpdi_group_assign_chunk`; host arithmetic, no GPU).

The reference's loop feeds the metric `n_per_iteration` patterns per iteration
(indexing/_dictionary_indexing.py:100-128); its tutorial uses a tenth of the dictionary (3044 patterns,
doc/tutorials/pattern_matching.ipynb:582).  The rule must keep such chunks whole (a 3044-pattern chunk cut 8 ways is
3 of 16 workgroup slots per row block for one tile-time), keep the members' totals within one pattern of each other,
and still cut a single-pass call into the contiguous n_dev parts of the multi-process form."""

import numpy as np
import pytest

from kikuchipy_amd import _lib
from kikuchipy_amd.indexing._dictionary_indexing import chunk_bounds
from kikuchipy_amd.parallel import shard_range


def run(n_dev, n_total, chunks, announce=True, min_piece=4096):
    loads = [0] * n_dev
    out, start = [], 0
    for n in chunks:
        pieces = _lib.Group.assign_chunk(n_dev, n_total if announce else 0, loads, n, min_piece)
        # the pieces of a chunk: in row order, disjoint, covering it, one per member at most
        assert pieces[0][1] == 0 and sum(r for _, _, r in pieces) == n
        assert all(a[1] + a[2] == b[1] for a, b in zip(pieces, pieces[1:]))
        assert len({m for m, _, _ in pieces}) == len(pieces) and all(r > 0 for _, _, r in pieces)
        out.append([(m, start + r0, r) for m, r0, r in pieces])
        start += n
    assert sum(loads) == sum(chunks)
    return out, loads


def test_single_pass_call_is_the_contiguous_split():
    for n, n_dev in ((100000, 8), (300000, 8), (12345, 5), (7, 8), (1, 1), (23, 3)):
        (pieces,), loads = run(n_dev, n, [n])
        want = [(i,) + (lambda a, b: (a, b - a))(*shard_range(n, i, n_dev)) for i in range(n_dev)]
        assert pieces == [w for w in want if w[2] > 0]
        assert loads == [b - a for a, b in (shard_range(n, i, n_dev) for i in range(n_dev))]


def test_tutorial_call_keeps_chunks_whole_and_members_level():
    """configs[1] through the tutorial's call: 100 000 patterns, n_per_iteration = 3044, 8 members."""
    n, per, n_dev = 100000, 3044, 8
    sizes = [e - s for s, e in chunk_bounds(n, per)]
    out, loads = run(n_dev, n, sizes)
    whole = [p for p in out if len(p) == 1]
    assert len(whole) == 32 and all(p[0][2] == per for p in whole)           # 32 whole chunks ...
    assert [p[0][0] for p in whole] == [i % n_dev for i in range(32)]        # ... going round the members
    assert len(out[-1]) == n_dev                                             # only the last, shorter chunk is cut
    assert loads == [12500] * n_dev                                          # every member ends on its quota
    # old rule for comparison: every chunk cut 8 ways = 380-pattern pieces
    assert min(r for p in out[:-1] for _, _, r in p) == per


@pytest.mark.parametrize("n,per,n_dev", [(100000, 25000, 8), (100000, 10000, 8), (300000, 3044, 8), (100000, 3044, 3),
                                          (1000, 7, 4), (50, 3, 8), (500000, 50000, 8), (100000, 100000, 2)])
def test_quotas_hold_for_any_chunking(n, per, n_dev):
    sizes = [e - s for s, e in chunk_bounds(n, per)]
    out, loads = run(n_dev, n, sizes)
    assert loads == [b - a for a, b in (shard_range(n, i, n_dev) for i in range(n_dev))]
    quota = n // n_dev
    # consecutive chunks start on different members whenever a chunk fits a quota (overlap of fetch and sweep)
    if per <= quota and n_dev > 1:
        firsts = [p[0][0] for p in out]
        assert all(a != b for a, b in zip(firsts, firsts[1:]))


def test_more_patterns_than_announced_go_to_the_least_loaded():
    out, loads = run(3, 30, [10, 10, 10, 9, 9])
    assert loads[:] == sorted(loads, reverse=True) or max(loads) - min(loads) <= 9
    assert out[3] == [(0, 30, 9)] and out[4] == [(1, 39, 9)]


def test_unannounced_size_cuts_into_pieces_worth_a_launch():
    out, loads = run(4, 0, [40000, 3000, 3000, 3000, 3000, 3000], announce=False, min_piece=4096)
    assert [m for m, _, _ in out[0]] == [0, 1, 2, 3] and all(r == 10000 for _, _, r in out[0])
    assert [p[0][0] for p in out[1:]] == [0, 1, 2, 3, 0] and all(len(p) == 1 for p in out[1:])
    out, _ = run(8, 0, [100000], announce=False, min_piece=8192)   # a single pass nobody announced: still 8 ways
    assert [(m, r) for m, _, r in out[0]] == [(i, 12500) for i in range(8)]
    out, loads = run(8, 0, [25000, 25000], announce=False, min_piece=8192)
    assert [len(p) for p in out] == [3, 3] and {m for p in out for m, _, _ in p} == {0, 1, 2, 3, 4, 5}


def test_bad_arguments_are_refused():
    with pytest.raises(_lib.KpdiError):
        _lib.Group.assign_chunk(0, 10, [], 5)
    with pytest.raises(_lib.KpdiError):
        _lib.Group.assign_chunk(2, 10, [0, 0], 0)
