"""bench_structured.py - a PHYSICALLY STRUCTURED workload of configs[1]'s size, for bench.py's `extra.structured_config2`.

The reference's own benchmark and tutorial index Ni patterns against an orientation-ORDERED dictionary sampled from
a real master pattern (/root/reference/benchmarks/indexing/test_dictionary_indexing.py:30-63: `get_sample_fundamental`
-> `mp.get_patterns` -> `s.dictionary_indexing`; doc/tutorials/pattern_matching.ipynb cell 29, with a circular signal
mask).  bench.py's headline uses i.i.d. uniform-random patterns - the friendliest distribution for a threshold-screened
fused top-k.  This module builds, from what the repo ships and nothing else:

  * the Ni master pattern the reference ships (tests/golden/projection.npz: mp_upper / mp_lower, uint8 401 x 401);
  * a dictionary of 100 000 orientations AS THE REFERENCE'S SAMPLER EMITS THEM: `get_sample_fundamental` (cubochoric
    sampling of the cubic fundamental zone, kikuchipy_amd/sampling.py - the restatement of orix's, pinned by the
    reference's own numbers) at 67 steps per semi-edge (~2 degrees: 100 347 orientations, the first 100 000 of them), in
    its lexicographic order - neighbours in index are neighbours in orientation, long runs of similar scores -
    projected on the device (kpdi::project_kernel);
  * 4096 experimental patterns laid out as a 64 x 64 MAP OF A FEW DOZEN GRAINS (Voronoi cells; neighbouring rows share an
    orientation up to 0.2 degrees of scatter), each a projection of the same master pattern under a smooth detector
    background with noise, quantised to uint8 - then pre-processed by the engine as the tutorial does: static background
    subtract, dynamic background subtract, circular signal mask (K = 2819);
  * two HOSTILE orderings of the same dictionary: sorted ascending and descending by its score against experimental
    pattern 0 (ascending = every tile raises the bound a little: the worst case of an append-then-screen epilogue).

Nothing here reads /root/reference.  The oracle (oracle/) is used as the checker only, by `check`.
"""

import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
F32_MFMA_PEAK_TFLOPS = 157.3

# detector of the reference's Ni test data (tests/golden/projection.npz det60: pc (0.421, 0.7794, 0.5049), tilt 0,
# sample tilt 70)
PC = (0.421, 0.7794, 0.5049)
SEMI_EDGE_STEPS = 67  # cubochoric grid of the dictionary: 100 347 orientations in the cubic fundamental zone (~2 degrees)


def quaternion_multiply(a, b):
    """Hamilton product of (..., 4) quaternions (a, b, c, d)."""
    a0, a1, a2, a3 = np.moveaxis(a, -1, 0)
    b0, b1, b2, b3 = np.moveaxis(b, -1, 0)
    return np.stack([a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                     a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0], axis=-1)


def small_rotations(rng, n, degrees):
    """n rotations by angles ~ N(0, degrees) about random axes, as quaternions."""
    axis = rng.standard_normal((n, 3))
    axis /= np.linalg.norm(axis, axis=1)[:, None]
    half = 0.5 * np.deg2rad(degrees) * rng.standard_normal(n)
    return np.column_stack([np.cos(half), axis * np.sin(half)[:, None]])


def grain_map(ny=64, nx=64, n_grains=40, seed=11):
    """(ny * nx,) grain label of every map point (Voronoi cells of seeded points), row-major like a scan."""
    rng = np.random.default_rng(seed)
    seeds = np.column_stack([rng.uniform(0, ny, n_grains), rng.uniform(0, nx, n_grains)])
    yy, xx = np.mgrid[:ny, :nx]
    d2 = (yy[..., None] - seeds[:, 0]) ** 2 + (xx[..., None] - seeds[:, 1]) ** 2
    return d2.argmin(-1).ravel()


def detector_geometry(sy, sx):
    """(gnomonic bounds, pcz, detector-to-sample matrix) as bench.py's dictionary_generation leg sets them."""
    aspect = sx / sy
    bounds = [-aspect * PC[0] / PC[2], aspect * (1 - PC[0]) / PC[2], -(1 - PC[1]) / PC[2], PC[1] / PC[2]]
    ct, st = np.cos(np.deg2rad(70.0)), np.sin(np.deg2rad(70.0))
    det_to_sample = np.array([[0, 1, 0], [-st, 0, ct], [ct, 0, st]], dtype=np.float64).T
    return bounds, PC[2], det_to_sample


def master_pattern():
    z = np.load(os.path.join(ROOT, "tests", "golden", "projection.npz"))
    return z["mp_upper"].astype(np.float32), z["mp_lower"].astype(np.float32)


def build(ctx, m=4096, sy=60, sx=60, seed=11, n=100000):
    """Inputs of the structured workload, made with the engine's own projection kernel on `ctx`:
    (exp uint8 (m, sy, sx), dictionary float32 (n, sy, sx) in sampler order, static background uint8 (sy, sx))."""
    from kikuchipy_amd.sampling import get_sample_fundamental

    rng = np.random.default_rng(seed)
    rot = get_sample_fundamental(semi_edge_steps=SEMI_EDGE_STEPS, point_group="m-3m")[:n]
    mpu, mpl = master_pattern()
    ctx.set_master_pattern(mpu, mpl)
    bounds, pcz, det_to_sample = detector_geometry(sy, sx)
    ctx.set_detector(bounds, pcz, sy, sx, det_to_sample)
    dic = ctx.project_patterns(rot).reshape(-1, sy, sx)
    side = int(round(np.sqrt(m)))
    assert side * side == m, "the experimental set is a square map"
    labels = grain_map(side, side, 40, seed)
    # a grain = an orientation of the fundamental zone, off the dictionary's grid by ~0.7 degrees; its points scatter by 0.2
    grains = quaternion_multiply(rot[rng.choice(len(rot), 40, replace=False)], small_rotations(rng, 40, 0.7))
    points = quaternion_multiply(grains[labels], small_rotations(rng, m, 0.2))
    sim = ctx.project_patterns(points).reshape(m, sy, sx)
    # raw detector image: a smooth background hump carrying ~10 % Kikuchi contrast, plus noise
    yy, xx = np.mgrid[:sy, :sx]
    hump = 60.0 + 140.0 * np.exp(-((yy - 0.45 * sy) ** 2 + (xx - 0.55 * sx) ** 2) / (2 * (0.47 * sy) ** 2))
    z = (sim - sim.mean(axis=(1, 2), keepdims=True)) / sim.std(axis=(1, 2), keepdims=True)
    raw = hump * (1.0 + 0.10 * z) + 3.0 * rng.standard_normal(sim.shape)
    exp = np.clip(np.rint(raw), 0, 255).astype(np.uint8)
    bg = np.clip(np.rint(hump), 1, 255).astype(np.uint8)
    return exp, np.ascontiguousarray(dic, dtype=np.float32), bg


def hostile_orders(dic, pattern, mask):
    """Permutations of the dictionary sorted by its ncc score against `pattern` (the kept pixels only): (ascending,
    descending).  Host arithmetic; only the ORDER matters."""
    keep = ~mask.ravel()
    x = pattern.ravel()[keep].astype(np.float64)
    x -= x.mean()  # (then y . x = (y - mean(y)) . x)
    s = np.empty(len(dic), dtype=np.float64)
    for a in range(0, len(dic), 10000):
        y = dic[a:a + 10000].reshape(-1, keep.size)[:, keep].astype(np.float64)
        var = (y * y).sum(1) - y.sum(1) ** 2 / y.shape[1]
        s[a:a + 10000] = (y @ x) / np.sqrt(np.maximum(var, 1e-30))
    asc = np.argsort(s, kind="stable")
    return asc, asc[::-1].copy()


def timed_sweeps(_lib, c, d_exp, exp, bg_f32, d_dic, n, keep_n, reps):
    """`reps` pipelined steps (pre-processing -> preparation -> match + top-k -> merge -> hand-over) on resident raw
    inputs, as bench.py's timed region; returns (ms per step, counters, scores, indices)."""
    c.set_profiling("match")
    pending = None
    scores = indices = None
    for r in range(reps + 2):
        if r == 2:
            c.finalize_wait(pending)
            pending = None
            c.reset_counters()
            c.synchronize()
            t0 = time.perf_counter()
        c.set_experimental_dev(d_exp, exp.dtype, len(exp))
        c.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
        c.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
        c.push_dictionary_chunk_dev(d_dic, np.float32, n, 0)
        ticket = c.finalize_async(keep_n)
        if pending is not None:
            scores, indices = c.finalize_wait(pending)
        pending = ticket
    scores, indices = c.finalize_wait(pending)
    c.synchronize()
    dt = (time.perf_counter() - t0) / reps
    cnt = c.counters()
    # untimed: two more steps at profiling level 3, where the epilogues of the match kernel count what they do
    c.set_profiling("epilogue")
    c.reset_counters()
    for _ in range(2):
        c.set_experimental_dev(d_exp, exp.dtype, len(exp))
        c.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
        c.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
        c.push_dictionary_chunk_dev(d_dic, np.float32, n, 0)
        c.finalize(keep_n)
    full = c.counters()
    for key in ("epi_lists", "epi_appended", "epi_overflows", "epi_direct_first"):
        cnt[key] = full.get(key, 0)
    cnt["epi_launches"] = full["match_launches"]
    return dt * 1e3, cnt, scores, indices


def check(exp, bg, mask, pre_engine, dic, scores, indices, keep_n, n_rows):
    """The three-part contract of SURVEY.md 8(a) on a sample of rows: the engine's pre-processed patterns against the
    oracle's (<= 1 grey level on <= 1e-3 of the pixels), then scores / indices of the sweep against the float64 C oracle
    fed the ENGINE's pre-processed patterns (1e-5, ties as sets)."""
    from oracle import c_oracle
    from oracle import kpdi_oracle as ko

    t0 = time.perf_counter()
    rows = np.sort(np.random.default_rng(5).choice(len(exp), n_rows, replace=False))
    want = ko.remove_dynamic_background(ko.remove_static_background(exp[rows], bg))
    got = pre_engine[rows]
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1 and np.mean(d > 0) <= 1e-3, (int(d.max()), float(np.mean(d > 0)))
    rs, ri = c_oracle.rows_topk_f64(got, [(0, dic)], np.arange(n_rows), "ncc", keep_n, mask)
    ko.assert_topk_parity(scores[rows], indices[rows], rs, ri, atol=1e-5)
    return {"rows": int(n_rows), "oracle": "oracle/kpdi_oracle_c.c rows_topk_f64 over the whole dictionary, fed the engine's "
                                           "pre-processed patterns; pre-processing against oracle/kpdi_oracle.py",
            "preprocessed_pixels_off_by_one": float(np.mean(d > 0)),
            "max_abs_score_diff": float(np.abs(scores[rows] - rs).max()),
            "index_agreement": float(np.mean(indices[rows] == ri)),
            "seconds": round(time.perf_counter() - t0, 2)}


def summarize(ms, cnt, m, reps):
    launches = max(cnt["match_launches"], 1)
    match_ms = cnt["match_ms"] / launches
    tf = cnt["match_flops"] / launches / (match_ms * 1e-3) / 1e12
    rec = {"patterns_per_s": round(m / ms * 1e3, 1), "ms_per_step": round(ms, 3), "match_ms": round(match_ms, 4),
           "match_tflops": round(tf, 2), "match_frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
           "match_form": int(cnt.get("match_form", 0))}
    if cnt.get("epi_lists"):
        # (kpdi_counters.epi_*: what the fused top-k's epilogues did in two untimed launches at profiling level 3; a "list" is
        # one lane's share of 32 experimental patterns x the dictionary rows its wave sees: 64 candidates per tile)
        el = max(cnt["epi_launches"], 1)
        rec["candidates_appended_per_lane_list"] = round(cnt["epi_appended"] / cnt["epi_lists"], 3)
        rec["buffer_overflows_per_launch"] = round(cnt["epi_overflows"] / el, 2)       # wave-level events (64 lists each)
        rec["direct_first_tiles_per_launch"] = round(cnt["epi_direct_first"] / el, 2)  # of 4 x the workgroups
    return rec


def leg(_lib, device, reps=8, n_check=64, m=4096, sy=60, sx=60, keep_n=20, hostile=True, baseline_frac=None):
    """bench.py's `extra.structured_config2`."""
    yy, xx = np.ogrid[:sy, :sx]  # `~Window("circular", (sy, sx)).astype(bool)` (filters/window.py:249-269), as configs[2]
    mask = np.sqrt((yy - sy // 2) ** 2 + (xx - sx // 2) ** 2) > max(sy // 2, sx // 2)
    c = _lib.Context(device)
    try:
        t_build = time.perf_counter()
        exp, dic, bg = build(c, m, sy, sx)
        t_build = time.perf_counter() - t_build
        n = len(dic)
        c.set_problem(sy, sx, mask, _lib.METRIC_NCC, keep_n, _lib.COMPUTE_F32)
        d_exp = c.dev_alloc(exp.nbytes)
        c.h2d(d_exp, exp)
        d_dic = c.dev_alloc(dic.nbytes)
        c.h2d(d_dic, dic)
        bg_f32 = bg.astype(np.float32)
        ms, cnt, scores, indices = timed_sweeps(_lib, c, d_exp, exp, bg_f32, d_dic, n, keep_n, reps)
        pre = c.get_experimental().reshape(exp.shape)
        out = {
            "what": f"configs[1]'s size on PHYSICAL structure: {m} patterns = a {int(np.sqrt(m))} x {int(np.sqrt(m))} map of 40 "
                    f"grains (0.2 deg scatter), projections of the Ni master pattern the reference ships under a smooth detector "
                    f"background + noise, uint8; dictionary = the first {n} orientations of get_sample_fundamental (cubochoric, cubic "
                    f"fundamental zone, {SEMI_EDGE_STEPS} steps per semi-edge ~ 2 degrees) in the sampler's order, projected on the device; "
                    "static + dynamic background subtract + circular mask fused pre-kernel, ncc, keep_n=20, raw inputs resident",
            "kept_pixels": int(cnt["k_kept"]),
            "best_score_mean": float(scores[:, 0].mean()),
            "build_seconds": round(t_build, 2),
        }
        out.update(summarize(ms, cnt, m, reps))
        if baseline_frac:  # (configs[2]: the same pipeline on i.i.d. random patterns, this run)
            out["match_frac_on_random_data"] = baseline_frac
        if n_check:
            out["check"] = check(exp, bg, mask, pre, dic, scores, indices, keep_n, n_check)
        if hostile:
            asc, desc = hostile_orders(dic, pre[0], mask)
            for name, perm in (("dictionary_sorted_ascending", asc), ("dictionary_sorted_descending", desc)):
                t0 = time.perf_counter()
                dic_p = np.ascontiguousarray(dic[perm])
                c.h2d(d_dic, dic_p)
                ms_p, cnt_p, s_p, i_p = timed_sweeps(_lib, c, d_exp, exp, bg_f32, d_dic, n, keep_n, max(3, reps // 2))
                rec = summarize(ms_p, cnt_p, m, max(3, reps // 2))
                rec["what"] = ("the same dictionary sorted by its score against experimental pattern 0, "
                               + name.rsplit("_", 1)[1] + " (every pattern of that grain sees rising / falling scores tile after tile)")
                # the permuted sweep must find the same entries (scores bit for bit: same operands, same arithmetic)
                same_s = bool(np.array_equal(s_p, scores))
                rec["scores_identical_to_sampler_order"] = same_s
                if n_check:
                    rec["check"] = check(exp, bg, mask, pre, dic_p, s_p, i_p, keep_n, max(8, n_check // 4))
                rec["seconds"] = round(time.perf_counter() - t0, 2)
                out[name] = rec
                del dic_p
        return out
    finally:
        c.close()


if __name__ == "__main__":  # python bench_structured.py [reps]  -> the leg alone, as JSON
    import json
    import sys

    sys.path.insert(0, ROOT)
    from kikuchipy_amd import _lib

    print(json.dumps(leg(_lib, 0, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 8), indent=1))
