"""The launch planner of a sweep (csrc/plan.h, through the exported `kpdi_plan_describe`; host arithmetic, no GPU).

What is planned is the reference's `match` + `topk` of one dictionary chunk
(indexing/_dictionary_indexing.py:172-203; similarity_metrics/_normalized_cross_correlation.py:161-183) laid on the 256
compute units of an MI355X: which f32 kernel, how many workgroups share a row block's dictionary tiles, how many row
blocks a launch covers, each launch's XCD grid, the tail of the last round.  Invariants, for shares of 1 ... 300 000
patterns: every row block is covered exactly once, padded grids hold at most 9/8 of the rows, no launch is without
tiles, the tail units cover exactly the tiles the main rounds leave."""

import math

import numpy as np
import pytest

from kikuchipy_amd import _lib

N_CU = 256
SHARES = sorted({1, 2, 31, 32, 33, 127, 128, 129, 255, 256, 257, 324, 380, 1000, 2592, 3044, 4095, 4096, 4097, 6250, 12500, 25000,
                 37500, 50000, 62500, 75000, 100000, 125000, 150000, 250000, 299999, 300000}
                | set(np.random.default_rng(5).integers(1, 300001, 40).tolist()))
MS = (1, 9, 255, 256, 257, 512, 1024, 4096, 7424, 10000, 40000)


def check(p, m, n, form):
    row_blocks = -(-m // 256)
    assert p.row_blocks == row_blocks
    assert p.form in (0, 3) and (form < 0 or p.form == form)
    assert p.tile == (256 if p.form == 3 else 128)
    # tiles cover the chunk exactly
    assert (p.n_tiles - 1) * p.tile < n <= p.n_tiles * p.tile
    # the split never exceeds the tiles there are (no workgroup without a tile) nor the chip
    assert 1 <= p.nsplit <= min(p.n_tiles, N_CU)
    assert 1 <= p.rows_per_launch <= row_blocks and p.rows_per_launch * p.nsplit <= N_CU
    assert p.launches == -(-row_blocks // p.rows_per_launch)
    # launches: contiguous, disjoint, covering every row block exactly once
    n_desc = p.n_launch_desc
    assert n_desc == min(p.launches, 64)
    ls = [p.launch[j] for j in range(n_desc)]
    assert ls[0].row_first == 0 and all(a.row_first + a.rows == b.row_first for a, b in zip(ls, ls[1:]))
    if p.launches <= 64:
        assert ls[-1].row_first + ls[-1].rows == row_blocks
    for l in ls:
        assert l.rows >= 1
        # the XCD grid: 8 XCDs as xcd_rows x xcd_splits, or the plain mapping; a padded grid adds at most an eighth
        assert (l.xcd_rows, l.xcd_splits) == (0, 0) or l.xcd_rows * l.xcd_splits == 8
        if l.xcd_rows:
            assert p.nsplit % l.xcd_splits == 0 and l.rows_grid % l.xcd_rows == 0
        assert l.rows <= l.rows_grid and 8 * (l.rows_grid - l.rows) <= l.rows
        if p.form == 0:
            assert l.rows_grid == l.rows  # match.hip (dynamic draws) never runs a padded grid
        assert l.rows_grid * p.nsplit <= N_CU * 9 // 8
    if p.form == 0:
        # match.hip: main rounds + (single launch, few rounds) a quarter-tile tail launch over the rest
        assert p.n_main + p.tail_tiles == p.n_tiles and p.n_main >= 1
        if p.tail_tiles:
            assert p.launches == 1 and p.n_main % p.nsplit == 0 and p.n_main >= p.nsplit
            assert 4 * p.tail_tiles <= 3 * p.nsplit
            rows_left = min(n, p.n_tiles * 128) - p.n_main * 128
            assert (p.tail_units - 1) * 32 < rows_left <= p.tail_units * 32
            assert 1 <= p.tail_nsplit <= min(p.nsplit, p.tail_units)
        assert p.fixed_draws >= 1
    else:
        # match16.hip f32: whole rounds up to tail_first, then halves / quarters of the remaining tiles
        assert p.tail_first % p.nsplit == 0 and 0 <= p.n_tiles - p.tail_first < p.nsplit
        assert p.tail_shift in (0, 1, 2)
        # the last partial round: inside the kernel (partial units) or as tailgemm.hip's own launch over exactly the rows left
        if p.tail_gemm_rows:
            assert p.tail_shift == 0 and p.n_main == p.tail_first < p.n_tiles
            assert p.tail_gemm_rows == n - p.tail_first * 256 > 0
        else:
            assert p.n_main == p.n_tiles
        if p.tail_first == p.n_tiles:
            assert p.tail_shift == 0
        # the ORDER of a workgroup's whole-tile rounds: a permutation (stride coprime to the rounds) that spreads the first
        # visits over the dictionary; partial units and an incomplete last round stay behind it, in natural order
        whole = p.tail_first if (p.tail_shift or p.tail_gemm_rows) else p.n_tiles
        if p.perm_rounds:
            r, st = p.perm_rounds, p.perm_stride
            assert r == whole // p.nsplit >= 3 and 1 <= st < r and math.gcd(st, r) == 1
            order = [(j * st) % r for j in range(r)]
            assert sorted(order) == list(range(r))
            if r >= 8:  # the running maximum of the walk (= what a dictionary sorted by score shows a workgroup) rises rarely
                records = sum(1 for j in range(r) if order[j] == max(order[:j + 1]))
                assert records <= 1 + 1.5 * math.log2(r), (r, st, records)
                assert order[0] == 0  # (a dictionary sorted by FALLING score shows its best tile first)
        else:
            assert whole // p.nsplit < 3 and p.perm_stride == 1
    assert p.round_rows == max(1, N_CU // row_blocks) * 256


@pytest.mark.parametrize("form", [-1, 0, 3])
def test_invariants_over_shares_and_map_sizes(form):
    for m in MS:
        for n in SHARES:
            check(_lib.plan_describe(m, n, 3600, 20, N_CU, form), m, n, form)


def test_known_plans_of_the_benchmark_configurations():
    p = _lib.plan_describe(4096, 100000)          # configs[1] whole: the wide kernel, 16 x 16 workgroups, XCD grid 2 x 4
    assert (p.form, p.n_tiles, p.nsplit, p.rows_per_launch, p.launches) == (3, 391, 16, 16, 1)
    assert (p.launch[0].xcd_rows, p.launch[0].xcd_splits) == (2, 4)
    # one rank's share at N = 8: the wide kernel's three whole rounds (walked 0, 2, 1) + its last 212 rows on tailgemm.hip
    p = _lib.plan_describe(4096, 12500)
    assert (p.form, p.n_tiles, p.nsplit, p.n_main, p.tail_shift, p.tail_gemm_rows) == (3, 49, 16, 48, 0, 212)
    assert (p.perm_rounds, p.perm_stride) == (3, 2)
    p = _lib.plan_describe(4096, 25000)           # ... at N = 4: six rounds + 424 rows
    assert (p.form, p.n_main, p.tail_gemm_rows) == (3, 96, 424)
    p = _lib.plan_describe(4096, 12500, form=0)   # match.hip, when asked for: 6 rounds + a 7-unit tail launch
    assert (p.form, p.n_tiles, p.nsplit, p.tail_tiles, p.tail_units) == (0, 98, 16, 2, 7)
    p = _lib.plan_describe(40000, 37500)          # configs[3] share: 157 row blocks = 4 x 32 + 29, the last grid padded
    assert (p.form, p.rows_per_launch, p.launches) == (3, 32, 5)
    assert [(p.launch[j].rows, p.launch[j].rows_grid) for j in range(5)] == [(32, 32)] * 4 + [(29, 32)]
    p = _lib.plan_describe(4096, 62500, 14400)    # configs[4] share (f32 form)
    assert p.form == 3 and p.nsplit == 16
    # other chips: the plan scales with the compute units it is given
    for n_cu in (8, 64, 104, 304):
        q = _lib.plan_describe(4096, 100000, 3600, 20, n_cu)
        assert q.rows_per_launch * q.nsplit <= n_cu and q.launches == -(-16 // q.rows_per_launch)


def test_switches_reach_the_planner(monkeypatch):
    monkeypatch.setenv("KPDI_NO_TAIL", "1")
    p = _lib.plan_describe(4096, 12500, form=0)
    assert p.tail_tiles == 0 and p.n_main == p.n_tiles
    p = _lib.plan_describe(4096, 12500, form=3)
    assert p.tail_shift == 0 and p.tail_gemm_rows == 0
    monkeypatch.delenv("KPDI_NO_TAIL")
    monkeypatch.setenv("KPDI_XCD_GRID", "0")
    p = _lib.plan_describe(4096, 100000)
    assert (p.launch[0].xcd_rows, p.launch[0].xcd_splits, p.launch[0].rows_grid) == (0, 0, 16)
    monkeypatch.delenv("KPDI_XCD_GRID")
    assert _lib.plan_describe(4096, 100000).perm_rounds == 24
    monkeypatch.setenv("KPDI_TILE_ORDER", "natural")
    assert _lib.plan_describe(4096, 100000).perm_rounds == 0
    monkeypatch.delenv("KPDI_TILE_ORDER")
    assert _lib.plan_describe(4096, 12500).tail_gemm_rows == 212
    monkeypatch.setenv("KPDI_TAIL_GEMM", "0")
    p = _lib.plan_describe(4096, 12500, form=3)
    assert p.tail_gemm_rows == 0 and p.tail_shift == 2
    monkeypatch.setenv("KPDI_TAIL_GEMM", "1")
    assert _lib.plan_describe(4096, 100000).tail_gemm_rows == 100000 - 384 * 256
    monkeypatch.delenv("KPDI_TAIL_GEMM")
    monkeypatch.setenv("KPDI_XCD_PAD", "0")
    p = _lib.plan_describe(40000, 37500)
    assert p.launch[4].rows_grid == 29


def test_bad_arguments():
    for args in ((0, 10), (10, 0), (10, 10, 0), (10, 10, 5, 0), (10, 10, 5, 5, 0), (10, 10, 5, 5, 256, 2)):
        with pytest.raises(_lib.KpdiError):
            _lib.plan_describe(*args)
