"""BASELINE.json's full sizes on the GPU, checked through size-independent
properties (planted exact matches, invariance to chunking) AND against the C
oracle (oracle/kpdi_oracle_c.c: float64-accumulated dot products, OpenMP over
the host cores) on HUNDREDS of experimental rows over the whole dictionary:
512 rows of configs[1], 320 of configs[2] (tests/test_gpu_config3.py), 256 of a
rank's share of configs[3], 128 of a rank's share of configs[4] (the float32 path and the float16 kernel configs[4] names)."""

import numpy as np
import pytest

from oracle import c_oracle
from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def sweep(ctx, exp, dic, metric, keep_n, chunks=1, signal_mask=None):
    from kikuchipy_amd import _lib

    sy, sx = exp.shape[-2:]
    ctx.set_problem(sy, sx, signal_mask, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric], keep_n)
    ctx.set_experimental(exp)
    bounds = np.linspace(0, len(dic), chunks + 1).astype(int)
    for a, b in zip(bounds[:-1], bounds[1:]):
        ctx.push_dictionary_chunk(dic[a:b], int(a))
    return ctx.finalize(keep_n)


def spot_check(exp, dic, rows, metric, keep_n, scores, idx, signal_mask=None):
    """A few rows against the NumPy oracle (the restatement pinned to the reference's goldens)."""
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric=metric, keep_n=keep_n, n_per_iteration=25000,
                                    signal_mask=signal_mask)
    ko.assert_topk_parity(scores[rows], idx[rows], rs, ri, atol=ATOL)


def rows_check(exp, dic, n_rows, metric, keep_n, scores, idx, signal_mask=None, seed=0, index_offset=0):
    """`n_rows` random rows against the C oracle over the WHOLE dictionary (1e-5, north_star)."""
    rows = np.sort(np.random.default_rng(seed).choice(len(exp), n_rows, replace=False))
    rs, ri = c_oracle.rows_topk_f64(exp, dic, rows, metric, keep_n, signal_mask)
    ko.assert_topk_parity(scores[rows], idx[rows] - index_offset, rs, ri, atol=ATOL)
    return float(np.abs(scores[rows] - rs).max())


@pytest.fixture(scope="module")
def config2():
    """configs[1]: 4096 x 100k x 60x60 (SURVEY.md 8(d) generator), with 32 experimental
    patterns planted into the dictionary as exact (rescaled) copies."""
    rng = np.random.default_rng(2024)
    exp = rng.integers(0, 256, (4096, 60, 60), dtype=np.uint8)
    dic = rng.random((100000, 60, 60), dtype=np.float32)
    planted_rows = rng.choice(4096, 32, replace=False)
    planted_at = rng.choice(100000, 32, replace=False)
    dic[planted_at] = exp[planted_rows].astype(np.float32) / 255.0
    return exp, dic, planted_rows, planted_at


def test_config2_properties(config2):
    from kikuchipy_amd import _lib

    exp, dic, planted_rows, planted_at = config2
    with _lib.Context(0) as ctx:
        s1, i1 = sweep(ctx, exp, dic, "ncc", 20)
        s4, i4 = sweep(ctx, exp, dic, "ncc", 20, chunks=7)
    # chunking invariance: bit-identical
    assert np.array_equal(i1, i4) and np.array_equal(s1, s4)
    # planted copies come out first with score 1 (affine copies under NCC)
    assert np.array_equal(i1[planted_rows, 0], planted_at)
    assert np.allclose(s1[planted_rows, 0], 1, atol=ATOL)
    # order and range
    assert np.all(np.diff(s1, axis=1) <= 0) and s1.max() <= 1 + ATOL and s1.min() >= -1 - ATOL
    assert i1.min() >= 0 and i1.max() < len(dic)
    assert all(len(set(r)) == 20 for r in i1[::97])
    rows = np.concatenate([planted_rows[:2], [0, 1777, 4095]])
    spot_check(exp, dic, rows, "ncc", 20, s1, i1)
    # 512 rows (1/8 of the experimental set) over all 100 000 dictionary patterns
    worst = rows_check(exp, dic, 512, "ncc", 20, s1, i1)
    print(f"configs[1]: 512 rows vs the C oracle, max |dscore| = {worst:.2e}")


def test_config3_mask_properties(config2):
    """configs[2] match stage: circular signal mask (K = 2819)."""
    from kikuchipy_amd import _lib

    exp, dic, planted_rows, planted_at = config2
    mask = ~ko.circular_window((60, 60)).astype(bool)
    with _lib.Context(0) as ctx:
        s, i = sweep(ctx, exp[:1024], dic, "ncc", 20, chunks=3, signal_mask=mask)
    sel = planted_rows < 1024
    assert np.array_equal(i[planted_rows[sel], 0], planted_at[sel])
    spot_check(exp[:1024], dic, np.array([3, 500, 1023]), "ncc", 20, s, i, signal_mask=mask)
    rows_check(exp[:1024], dic, 256, "ncc", 20, s, i, signal_mask=mask)


def test_config4_shard_ndp():
    """configs[3], one rank's share: 200x200 map (40 000 patterns) against a
    37 500-pattern shard (300k / 8), ndp, keep_n=20, indices offset like rank 3's."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(4)
    exp = rng.integers(0, 256, (40000, 60, 60), dtype=np.uint8)
    dic = rng.random((37500, 60, 60), dtype=np.float32)
    start = 3 * 37500
    planted_rows = rng.choice(40000, 16, replace=False)
    planted_at = rng.choice(37500, 16, replace=False)
    dic[planted_at] = exp[planted_rows].astype(np.float32) * 0.5
    with _lib.Context(0) as ctx:
        ctx.set_problem(60, 60, None, _lib.METRIC_NDP, 20)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk(dic, start)
        s, i = ctx.finalize(20)
    assert np.array_equal(i[planted_rows, 0], planted_at + start)
    assert np.allclose(s[planted_rows, 0], 1, atol=ATOL)
    rows = np.array([0, 12345, 39999, planted_rows[0]])
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric="ndp", keep_n=20, n_per_iteration=12500)
    # the NumPy oracle's own float32 `ndp` sums are good to ~1e-5: compare loosely with it,
    # strictly with the float64-accumulated C oracle on 256 rows
    ko.assert_topk_parity(s[rows], i[rows] - start, rs, ri, atol=2e-5)
    worst = rows_check(exp, dic, 256, "ndp", 20, s, i, index_offset=start)
    print(f"configs[3] share: 256 rows vs the C oracle, max |dscore| = {worst:.2e}")


def test_config5_large_detector():
    """configs[4] geometry: 120x120 patterns (K = 14 400), f32 MFMA path."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(5)
    exp = rng.integers(0, 256, (300, 120, 120), dtype=np.uint8)
    dic = rng.random((5000, 120, 120), dtype=np.float32)
    dic[4321] = exp[7].astype(np.float32) + 3.0
    with _lib.Context(0) as ctx:
        s, i = sweep(ctx, exp, dic, "ncc", 20, chunks=2)
    assert i[7, 0] == 4321 and abs(s[7, 0] - 1) < ATOL
    spot_check(exp, dic, np.array([0, 7, 150, 299]), "ncc", 20, s, i)


@pytest.fixture(scope="module")
def config5_share():
    """One rank's share of configs[4]: 4096 patterns of 120 x 120 against 62 500 (= 500k / 8) dictionary patterns
    (3.6 GB raw), 8 experimental patterns planted into the dictionary as affine copies."""
    rng = np.random.default_rng(6)
    exp = rng.integers(0, 256, (4096, 120, 120), dtype=np.uint8)
    dic = rng.random((62500, 120, 120), dtype=np.float32)
    planted_rows = rng.choice(4096, 8, replace=False)
    planted_at = rng.choice(62500, 8, replace=False)
    dic[planted_at] = exp[planted_rows].astype(np.float32) * 2.0 + 1.0
    return exp, dic, planted_rows, planted_at


def test_config5_rank_share_f16(config5_share):
    """The kernel configs[4] NAMES ("fp16 MFMA accumulate-fp32", compute="f16", reduced precision) at the full size of a
    rank's share.  128 rows over the whole shard against (a) the exact products of the float16-rounded prepared
    operands (3e-5: what the mode computes), (b) the float64-accumulated C oracle over the float32 operands (2e-3:
    the mode's documented bound against the reference); planted copies first with score 1."""
    from kikuchipy_amd import _lib

    exp, dic, planted_rows, planted_at = config5_share
    start = 5 * 62500
    with _lib.Context(0) as ctx:
        ctx.set_problem(120, 120, None, _lib.METRIC_NCC, 20, _lib.COMPUTE_F16)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk(dic, start)
        s, i = ctx.finalize(20)
        cnt = ctx.counters()
    assert cnt["match_form"] == 2
    assert np.array_equal(i[planted_rows, 0], planted_at + start)
    assert np.allclose(s[planted_rows, 0], 1, atol=2e-3)
    rows = np.sort(np.random.default_rng(1).choice(4096, 128, replace=False))
    # (a) rounded operands, float64 products, block by block over the shard
    x = np.asarray(ko.prepare_experimental(exp[rows], metric="ncc", dtype=np.float64, n_experimental=len(rows)))
    xh = (x * 4096).astype(np.float32).astype(np.float16).astype(np.float64)
    best_s = np.full((len(rows), 0), -np.inf)
    best_i = np.zeros((len(rows), 0), dtype=np.int64)
    for lo in range(0, len(dic), 5000):
        y = np.asarray(ko.prepare_dictionary(dic[lo:lo + 5000].reshape(-1, 14400), metric="ncc", dtype=np.float64))
        yh = (y * 4096).astype(np.float32).astype(np.float16).astype(np.float64)
        sc = (xh @ yh.T) * 2.0**-24
        cat_s = np.concatenate([best_s, sc], axis=1)
        cat_i = np.concatenate([best_i, np.broadcast_to(np.arange(lo, lo + sc.shape[1]), sc.shape)], axis=1)
        order = np.lexsort((cat_i, -cat_s), axis=1)[:, :20]
        best_s, best_i = np.take_along_axis(cat_s, order, 1), np.take_along_axis(cat_i, order, 1)
    ko.assert_topk_parity(s[rows], i[rows] - start, best_s.astype(np.float32), best_i, atol=3e-5, tie=6e-5)
    # (b) the float32-operand reference arithmetic (float64-accumulated): the ranked scores stay within 2e-3
    rs, ri = c_oracle.rows_topk_f64(exp, dic, rows, "ncc", 20)
    worst = float(np.abs(s[rows] - rs).max())
    assert worst < 2e-3
    agree = float(np.mean(i[rows][:, 0] - start == ri[:, 0]))
    print(f"configs[4] share, float16: 128 rows, max |dscore| vs rounded operands {np.abs(s[rows] - best_s).max():.2e}, "
          f"vs the f32 reference {worst:.2e}, best-match agreement {agree:.3f}")


def test_config5_rank_share(config5_share):
    """configs[4], one rank's share at full size: 4096 patterns of 120 x 120 against 62 500 (= 500k / 8)
    dictionary patterns (3.6 GB raw), f32 MFMA path; 128 rows against the C oracle + planted copies."""
    from kikuchipy_amd import _lib

    exp, dic, planted_rows, planted_at = config5_share
    start = 5 * 62500
    with _lib.Context(0) as ctx:
        ctx.set_problem(120, 120, None, _lib.METRIC_NCC, 20)
        ctx.set_experimental(exp)
        ctx.push_dictionary_chunk(dic, start)
        s, i = ctx.finalize(20)
    assert np.array_equal(i[planted_rows, 0], planted_at + start)
    assert np.allclose(s[planted_rows, 0], 1, atol=ATOL)
    worst = rows_check(exp, dic, 128, "ncc", 20, s, i, index_offset=start)
    print(f"configs[4] share: 128 rows vs the C oracle, max |dscore| = {worst:.2e}")


def smooth_master_pattern(rng, n=401):
    """Low-pass filtered noise: a master pattern with Kikuchi-like smooth structure."""
    f = np.fft.rfft2(rng.standard_normal((n, n)))
    ky, kx = np.meshgrid(np.fft.fftfreq(n), np.fft.rfftfreq(n), indexing="ij")
    return np.fft.irfft2(f * np.exp(-(kx**2 + ky**2) / (2 * 0.03**2)), s=(n, n)).astype(np.float32)


def test_config2_generated_dictionary_round_trip():
    """configs[1] sizes with the dictionary SIMULATED on the device: the
    experimental patterns are noisy projections at 4096 of the 100 000 dictionary rotations,
    so the best match of each must be its own rotation (round trip), whatever the chunking."""
    import kikuchipy_amd as ka

    rng = np.random.default_rng(12)
    mp = ka.EBSDMasterPattern(np.stack([smooth_master_pattern(rng), smooth_master_pattern(rng)]))
    det = ka.EBSDDetector(shape=(60, 60), pc=(0.42, 0.78, 0.5))
    q = rng.standard_normal((100000, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    planted = rng.choice(100000, 4096, replace=False)
    sim = mp.get_patterns(q[planted], det, compute=True).data
    noisy = sim + 0.2 * sim.std() * rng.standard_normal(sim.shape).astype(np.float32)
    exp = ((noisy - noisy.min()) / (noisy.max() - noisy.min()) * 255).astype(np.uint8)
    dictionary = mp.get_patterns(q, det, chunk_shape=30000)
    s = ka.EBSD(exp)
    res = s.dictionary_indexing(dictionary, keep_n=20, verbose=False)
    assert np.array_equal(res.simulation_indices[:, 0], planted)
    assert res.scores[:, 0].min() > 0.9 and np.all(np.diff(res.scores, axis=1) <= 0)
    res2 = s.dictionary_indexing(dictionary, keep_n=20, n_per_iteration=100000, verbose=False)
    assert np.array_equal(res.simulation_indices, res2.simulation_indices)
    assert np.array_equal(res.scores, res2.scores)
    # oracle spot check of three rows against a materialised slice of the dictionary
    rows = np.array([0, 2000, 4095])
    lo = int(planted[rows].min()) // 1000 * 1000
    block = dictionary.data[lo:lo + 1000].compute()
    for r in rows:
        if lo <= planted[r] < lo + 1000:
            rs, ri = ko.dictionary_indexing(exp[r:r + 1], block, keep_n=1)
            assert ri[0, 0] + lo == planted[r] and abs(rs[0, 0] - res.scores[r, 0]) < ATOL


def test_refinement_of_a_whole_map():
    """4096 patterns refined in one launch: starts 1 degree off the truth
    come back to it, every score rises, and a second run is bit-identical."""
    import kikuchipy_amd as ka
    from kikuchipy_amd import _lib
    from kikuchipy_amd.indexing._refinement import rotation_from_euler

    rng = np.random.default_rng(13)
    mp = ka.EBSDMasterPattern(smooth_master_pattern(rng))
    det = ka.EBSDDetector(shape=(60, 60), pc=(0.42, 0.78, 0.5))
    eu = np.column_stack([rng.uniform(0.3, 6, 4096), rng.uniform(0.3, 2.8, 4096), rng.uniform(0.3, 6, 4096)])
    sim = mp.get_patterns(rotation_from_euler(eu), det, compute=True).data
    noisy = sim + 0.3 * sim.std() * rng.standard_normal(sim.shape).astype(np.float32)
    exp = ((noisy - noisy.min()) / (noisy.max() - noisy.min()) * 255).astype(np.uint8)
    eu0 = eu + np.deg2rad(rng.uniform(-1, 1, eu.shape))
    s = ka.EBSD(exp.reshape(64, 64, 60, 60))
    res = s.refine_orientation(rotation_from_euler(eu0).reshape(64, 64, 4), det, mp, verbose=False)
    err0 = np.rad2deg(np.abs(eu0 - eu)).max(axis=1)
    err = np.rad2deg(np.abs(res.euler - eu)).max(axis=1)
    assert np.median(err) < 0.05 and np.median(err) < np.median(err0) / 5
    ctx = s.context
    start = 1 - ctx.refine_objective(_lib.REFINE_ORI, np.arange(4096), eu0, np.tile(det.pc_flattened, (4096, 1)))
    assert np.all(res.scores >= start - 1e-6) and res.scores.mean() > 0.9
    assert res.num_evals.min() >= 4 and res.num_evals.max() <= 600
    again = s.refine_orientation(rotation_from_euler(eu0).reshape(64, 64, 4), det, mp, verbose=False)
    assert np.array_equal(again.scores, res.scores) and np.array_equal(again.euler, res.euler)


def test_experimental_set_beyond_2_31_prepared_elements():
    """A 700 000-pattern map (a large modern scan): 2.52e9 prepared floats on the experimental side."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(0)
    ctx = _lib.Context(0)

    m = 700_000
    exp = rng.integers(0, 256, (m, 60, 60), dtype=np.uint8)
    dic = rng.random((3000, 60, 60), dtype=np.float32)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    ctx.set_experimental(exp)
    ctx.push_dictionary_chunk(dic[:1700], 0)
    ctx.push_dictionary_chunk(dic[1700:], 1700)
    s, i = ctx.finalize(20)
    rows = np.array([0, 1, 255, 256, 596_523, 596_524, 600_000, 699_999])
    rs, ri = ko.dictionary_indexing(exp[rows], dic, keep_n=20)
    ko.assert_topk_parity(s[rows], i[rows], rs, ri, atol=1e-5)
    # every row: descending, valid indices
    assert np.all(np.diff(s, axis=1) <= 0) and i.min() >= 0 and i.max() < 3000
    ctx.close()


def test_dictionary_chunk_beyond_2_31_prepared_elements():
    """One chunk of 620 000 patterns, simulated on the device (8.9 GB raw + 8.9 GB prepared), pushed and held."""
    from kikuchipy_amd import _lib

    rng = np.random.default_rng(1)
    ctx = _lib.Context(0)
    ctx.set_problem(60, 60, None, _lib.METRIC_NCC, 20)
    n = 620_000
    quat = rng.standard_normal((n, 4))
    quat /= np.linalg.norm(quat, axis=1)[:, None]
    f = np.fft.rfft2(rng.standard_normal((2, 401, 401)))
    ky, kx = np.meshgrid(np.fft.fftfreq(401), np.fft.rfftfreq(401), indexing="ij")
    mp = np.fft.irfft2(f * np.exp(-(kx**2 + ky**2) / (2 * 0.03**2)), s=(401, 401)).astype(np.float32)
    ctx.set_master_pattern(mp[0], mp[1])
    pc = (0.421, 0.7794, 0.5049)
    bounds = [-pc[0] / pc[2], (1 - pc[0]) / pc[2], -(1 - pc[1]) / pc[2], pc[1] / pc[2]]
    ct, st = np.cos(np.deg2rad(70.0)), np.sin(np.deg2rad(70.0))
    det_to_sample = np.array([[0, 1, 0], [-st, 0, ct], [ct, 0, st]], dtype=np.float64).T
    ctx.set_detector(bounds, pc[2], 60, 60, det_to_sample)
    picks = np.array([0, 127, 300_000, 596_523, 596_524, 600_001, 619_999])
    planted = ctx.project_patterns(quat[picks], True, -1.0, 1.0, np.float32).reshape(-1, 60, 60)
    noise = planted + 0.05 * rng.standard_normal(planted.shape).astype(np.float32)
    ctx.set_experimental(np.concatenate([planted, noise]))
    ctx.push_rotations_chunk(quat, 5, True, -1.0, 1.0)
    s, i = ctx.finalize(20)
    assert np.array_equal(i[:7, 0], picks + 5) and np.allclose(s[:7, 0], 1, atol=1e-5), (i[:, 0], s[:, 0])
    assert np.array_equal(i[7:, 0], picks + 5), i[7:, 0]
    # the same as a held chunk
    ctx.hold_rotations_chunk(quat, 5, True, -1.0, 1.0)
    ctx.reset_topk()
    ctx.sweep_held()
    s2, i2 = ctx.finalize(20)
    assert np.array_equal(s, s2) and np.array_equal(i, i2)
    ctx.close()
