"""bench.py - dictionary-indexing throughput on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3            # spawns its own 8 ranks (one per GPU)
    python bench.py --gpus 8 --single-process                 # ONE process, a kpdi_group over the 8 GPUs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3   # or under a launcher

Workload = BASELINE.json configs[1]: 4096 synthetic experimental patterns
(60x60, uint8) against a 100 000-pattern synthetic dictionary (float32), `ncc`,
keep_n=20.  `--workload config3` adds the circular signal mask and the fused
static + dynamic background pre-kernels (configs[2]).

One step = one full pass of the hot path with the RAW inputs already resident
in HBM: (optional pre-processing) -> cast/mask/normalise experimental patterns
-> cast/mask/normalise this rank's dictionary shard -> f32 MFMA match with fused
top-k -> merge -> (N > 1: RCCL all-gather of the per-shard best-k + merge) ->
best-k scores/indices copied to the host.  With N ranks the SAME job is sharded
over the dictionary axis (strong scaling); value = patterns indexed per second
by the whole job.

Prints ONE JSON line on rank 0.  No PyTorch: rendezvous, barrier and the
max-over-ranks of the timing go over kikuchipy_amd.parallel.SocketGroup (plain
TCP on the loopback interface, reading RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT as a launcher exports them); the data path is libkpdi + RCCL.  For
N > 1 rank 0 checks the MERGED result against the C oracle exactly as for
N = 1, and every rank's result must be bit-identical to rank 0's.

`--single-process`: the same sharded job driven from ONE interpreter - the shape of the reference's own
call (signals/ebsd.py:1827-1984) - through `kikuchipy_amd._lib.Group` (kpdi_group, include/kpdi.h): one
context and one host thread per GPU inside libkpdi, an in-process RCCL communicator (or peer copies,
`--gather p2p`), one merged result.  Same timed region, same check, same JSON line.
"""

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"

WORKLOADS = {
    "config2": dict(m=4096, n=100000, sy=60, sx=60, metric="ncc", keep_n=20, mask=False, preprocess=False,
                    name="configs[1]: 4096 exp (60x60 u8) x 100k dict (f32), ncc, keep_n=20"),
    "config3": dict(m=4096, n=100000, sy=60, sx=60, metric="ncc", keep_n=20, mask=True, preprocess=True,
                    name="configs[2]: configs[1] + circular signal mask + static/dynamic background pre-kernels"),
    # the two 8-GPU configurations also fit ONE MI355X (288 GB): not bench lines, measured for DESIGN.md
    "config4": dict(m=40000, n=300000, sy=60, sx=60, metric="ndp", keep_n=20, mask=False, preprocess=False,
                    name="configs[3]: 200x200 map (40 000 exp, 60x60 u8) x 300k dict (f32), ndp, keep_n=20"),
    "config5": dict(m=4096, n=500000, sy=120, sx=120, metric="ncc", keep_n=20, mask=False, preprocess=False,
                    name="configs[4]: 4096 exp (120x120 u8) x 500k dict (f32), ncc, keep_n=20"),
}
BLOCK = 25000  # dictionary patterns per generated block of the large workloads


def synth(w, seed=2024):
    """SURVEY.md 8(d) config 2 generator."""
    rng = np.random.default_rng(seed)
    exp = rng.integers(0, 256, (w["m"], w["sy"], w["sx"]), dtype=np.uint8)
    dic = rng.random((w["n"], w["sy"], w["sx"]), dtype=np.float32)
    bg = rng.integers(1, 256, (w["sy"], w["sx"]), dtype=np.uint8)
    return exp, dic, bg


def plant_table(w, n_plant=16):
    """(experimental rows, dictionary indices): the dictionary entries that are replaced by exact
    (rescaled) copies of experimental patterns, so that a run can be checked without an oracle sweep:
    row r must come out with dictionary index d first and score 1 (`ncc`) / 1 (`ndp`)."""
    rng = np.random.default_rng(99)
    return rng.choice(w["m"], n_plant, replace=False), np.sort(rng.choice(w["n"], n_plant, replace=False))


def dictionary_block(w, b, exp=None):
    """Block b (patterns [b * BLOCK, (b + 1) * BLOCK)) of a large workload's dictionary: seeded per
    block, so that every rank can generate exactly its own shard.  With `exp`, the planted copies
    of `plant_table` that fall into the block are written in."""
    n = min(BLOCK, w["n"] - b * BLOCK)
    blk = np.random.default_rng([2024, b]).random((n, w["sy"], w["sx"]), dtype=np.float32)
    if exp is not None:
        rows, at = plant_table(w)
        for r, d in zip(rows, at):
            if b * BLOCK <= d < b * BLOCK + n:
                blk[d - b * BLOCK] = exp[r].astype(np.float32) / 255.0
    return blk


def upload_generated_shard(ctx, w, lo, hi, exp, dtype=np.float32):
    """Device buffer with dictionary patterns [lo, hi) of a large workload (never whole on the host), stored as
    `dtype` (float16: a dictionary kept at half the bytes, cast exactly to float32 by the preparation kernels)."""
    row = w["sy"] * w["sx"] * np.dtype(dtype).itemsize
    d = ctx.dev_alloc((hi - lo) * row)
    for b in range(lo // BLOCK, (hi - 1) // BLOCK + 1):
        blk = dictionary_block(w, b, exp)
        a0, a1 = max(lo, b * BLOCK), min(hi, b * BLOCK + len(blk))
        ctx.h2d(d + (a0 - lo) * row, blk[a0 - b * BLOCK:a1 - b * BLOCK].astype(dtype, copy=False))
    return d


def circular_mask(sy, sx):
    """`~Window("circular", (sy, sx)).astype(bool)` (filters/window.py:249-269)."""
    yy, xx = np.ogrid[:sy, :sx]
    return np.sqrt((yy - sy // 2) ** 2 + (xx - sx // 2) ** 2) > max(sy // 2, sx // 2)


def check_result(w, exp, dic, bg, mask, scores, indices, n_rows, large, compute="f32", dict_dtype=np.float32):
    """The result of the timed run against the C oracle (oracle/kpdi_oracle_c.c, float64-accumulated
    dot products, OpenMP over the host cores) on a sample of experimental rows over the WHOLE
    dictionary, plus the planted copies of the large workloads.  The oracle is the checker here,
    never the thing measured.  Raises on a mismatch: a bench line is only printed for a correct run."""
    from oracle import c_oracle
    from oracle import kpdi_oracle as ko

    t0 = time.perf_counter()
    rows = np.sort(np.random.default_rng(5).choice(w["m"], n_rows, replace=False))
    e = exp[rows]
    if w["preprocess"]:
        e = ko.remove_dynamic_background(ko.remove_static_background(e, bg))
    seen = lambda d: d.astype(dict_dtype).astype(np.float32) if dict_dtype != np.float32 else d  # noqa: E731  (what the engine was given)
    if large:
        chunks = ((b * BLOCK, seen(dictionary_block(w, b, exp))) for b in range((w["n"] + BLOCK - 1) // BLOCK))
    else:
        chunks = [(0, seen(dic))]
    rs, ri = c_oracle.rows_topk_f64(e, chunks, np.arange(n_rows), w["metric"], w["keep_n"], mask)
    out = {"rows": int(n_rows), "oracle": "oracle/kpdi_oracle_c.c rows_topk_f64 (float64 accumulation)"}
    if w["preprocess"]:
        # end to end from the raw patterns: a flipped grey level of the pre-processing moves a score by ~1.6e-5
        out["max_abs_score_diff_end_to_end"] = float(np.abs(scores[rows] - rs).max())
        out["best_match_agreement"] = float(np.mean(indices[rows][:, 0] == ri[:, 0]))
        assert out["max_abs_score_diff_end_to_end"] < 1e-4 and out["best_match_agreement"] > 0.98, out
    elif compute == "f16":
        # the opt-in REDUCED-PRECISION arithmetic: its documented bound, not the 1e-5 contract
        out["max_abs_score_diff"] = float(np.abs(scores[rows] - rs).max())
        out["best_match_agreement"] = float(np.mean(indices[rows][:, 0] == ri[:, 0]))
        assert out["max_abs_score_diff"] < 2e-3 and out["best_match_agreement"] > 0.9, out
    else:
        ko.assert_topk_parity(scores[rows], indices[rows], rs, ri, atol=1e-5)
        out["max_abs_score_diff"] = float(np.abs(scores[rows] - rs).max())
        out["index_agreement"] = float(np.mean(indices[rows] == ri))
    if large:
        prow, pat = plant_table(w)
        assert np.array_equal(indices[prow, 0], pat), "planted copies not found first"
        assert np.allclose(scores[prow, 0], 1.0, atol=2e-3 if compute == "f16" else 1e-5), scores[prow, 0]
        out["planted_found"] = int(len(prow))
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


F16_RANDOM_OPERAND_CEILING_TFLOPS = 1740.0  # bare v_mfma_f32_32x32x16_f16 loop on random f16 operands (power-bound at
#                                             ~1.7 GHz): profiles/r03_mfma_power_probe.txt


def rank_share_leg(_lib, make_context, device, key, n_ranks, d_dic, dic_host, reps, n_check, shard_range, per=12500,
                   compute="f32"):
    """One rank's share of an 8-GPU configuration, on this GPU, inside the default run (configs[3] /
    configs[4] are quoted on 8 GPUs; what ONE of them does is measurable here): rank 0's dictionary shard
    (`shard_range(n, 0, n_ranks)`), the whole experimental set, inputs resident, the step pipelined as the timed
    region above, a sample of rows checked against the C oracle over the whole shard.  Informational, never `value`.

    configs[3]: the shard is the first 37 500 patterns of the dictionary already resident (`d_dic`, the same uniform
    float32 generator).  configs[4]: 62 500 patterns of 120 x 120 - one generated block of 12 500 and four copies of it
    with the pixels rotated by a different offset each (distinct, mutually uncorrelated patterns; 3.6 GB on the device,
    0.7 GB generated).  `compute="f16"`: the arithmetic configs[4] NAMES (fp16 MFMA, f32 accumulate; the dictionary
    resident as float16, 1.8 GB): checked against the float64-accumulated oracle over the same float16-stored
    dictionary with the mode's documented bound (2e-3, tests/test_gpu_fullsize.py::test_config5_rank_share_f16)."""
    w = WORKLOADS[key]
    f16 = compute == "f16"
    dict_np = np.float16 if f16 else np.float32
    m, sy, sx, keep = w["m"], w["sy"], w["sx"], w["keep_n"]
    npix = sy * sx
    lo, hi = shard_range(w["n"], 0, n_ranks)
    n_shard = hi - lo
    rng = np.random.default_rng(2024)
    exp = rng.integers(0, 256, (m, sy, sx), dtype=np.uint8)
    c = make_context(device)
    try:
        c.set_problem(sy, sx, None, {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[w["metric"]], keep,
                      _lib.COMPUTE_F16 if f16 else _lib.COMPUTE_F32)
        d_exp = c.dev_alloc(exp.nbytes)
        c.h2d(d_exp, exp)
        if d_dic is not None:  # a prefix of the resident dictionary
            blocks = lambda: [(0, dic_host[:n_shard])]  # noqa: E731
            d_shard = d_dic
        else:
            base = rng.random((per, npix), dtype=np.float32).astype(dict_np)
            shift = lambda b: base if b == 0 else np.roll(base, 2477 * b, axis=1)  # noqa: E731
            blocks = lambda: ((b * per, shift(b)[:min(per, n_shard - b * per)]) for b in range(-(-n_shard // per)))  # noqa: E731
            es = np.dtype(dict_np).itemsize
            d_shard = c.dev_alloc(n_shard * npix * es)
            for at, blk in blocks():
                c.h2d(d_shard + at * npix * es, np.ascontiguousarray(blk))
        c.set_profiling("match")
        pending = None
        scores = indices = None
        for r in range(reps + 1):
            if r == 1:
                c.finalize_wait(pending)
                pending = None
                c.reset_counters()
                c.synchronize()
                t0 = time.perf_counter()
            c.set_experimental_dev(d_exp, exp.dtype, m)
            c.push_dictionary_chunk_dev(d_shard, dict_np, n_shard, lo)
            ticket = c.finalize_async(keep)
            if pending is not None:
                scores, indices = c.finalize_wait(pending)
            pending = ticket
        scores, indices = c.finalize_wait(pending)
        c.synchronize()
        dt = (time.perf_counter() - t0) / reps
        cnt = c.counters()
        c.set_profiling(True)  # (untimed: the preparation of the shard, which is 10 % of the float16 step)
        c.reset_counters()
        c.set_experimental_dev(d_exp, exp.dtype, m)
        c.push_dictionary_chunk_dev(d_shard, dict_np, n_shard, lo)
        c.finalize(keep)
        prep_ms = c.counters()["prep_ms"]
    finally:
        c.close()
    match_ms = cnt["match_ms"] / reps
    tflops = cnt["match_flops"] / reps / (match_ms * 1e-3) / 1e12
    if f16:
        rec = {
            "what": f"{w['name']}: rank 0's share of {n_ranks} ranks on ONE MI355X in the arithmetic configs[4] names - "
                    f"fp16 MFMA, f32 accumulate (compute=f16, REDUCED PRECISION), dictionary resident as float16 "
                    f"({n_shard * npix * 2 / 1e9:.1f} GB), K = {npix}; the whole experimental set x dictionary patterns "
                    f"[{lo}, {hi}), results collected while the next step runs",
            "shard_patterns": int(n_shard), "ms_per_step": round(dt * 1e3, 3),
            "patterns_per_s_before_the_allgather": round(m / dt, 1),
            "match_ms": round(match_ms, 3), "prep_ms": round(prep_ms, 3),
            "match_tflops": round(tflops, 1), "match_frac": round(tflops / 2500.0, 4),
            "match_frac_of_random_operand_ceiling": round(tflops / F16_RANDOM_OPERAND_CEILING_TFLOPS, 4),
            "peaks": {"dense_f16_mfma_tflops": 2500.0, "random_operand_ceiling_tflops": F16_RANDOM_OPERAND_CEILING_TFLOPS,
                      "ceiling_source": "profiles/r03_mfma_power_probe.txt (bare MFMA loop on random f16: power-bound)"},
            "match_form": int(cnt.get("match_form", 0)),
        }
        # the kernel's counters (fabric traffic, MFMA busy, clock) cannot be read inside this run: the committed rocprofv3 passes
        # of the same kernel at the same shape are quoted, labelled as what they are (tools/collect_f16_pmc.sh)
        prof = os.path.join(ROOT, "profiles", "r06_config5_f16_pmc.json")
        if os.path.exists(prof):
            with open(prof) as f:
                pm = json.load(f)
            rec["roofline_profiled"] = {
                "kind": "static: read from the committed profile, NOT measured in this run", "source": os.path.relpath(prof, ROOT),
                "kernel": pm.get("kernel"), "match_ms": pm.get("match_ms"), "frac_of_2500": pm.get("frac_of_2500"),
                "traffic": pm.get("fetch_bytes", 0) + pm.get("write_bytes", 0),
                "algorithmic_operand_bytes": pm.get("algorithmic_operand_bytes"),
                "fetch_over_algorithmic": pm.get("fetch_over_algorithmic"), "fetch_TBps": pm.get("fetch_TBps"),
                "l2_hit_rate": pm.get("l2_hit_rate"), "mfma_busy": pm.get("mfma_busy"), "clock_GHz": pm.get("clock_GHz"),
                "lds_bank_conflicts": pm.get("counters_last_launch", {}).get("SQ_LDS_BANK_CONFLICT"),
                "note": "fabric traffic 8.1 x the prepared operands: 256 x 256 tiles on a 32-CU XCD share at best (4 row blocks + 8 "
                        "splits) / 32 = 5.3 x (DESIGN.md 4.1b); 2.4 TB/s, not the limiter - the clock is (power: 1.63 GHz, MFMA busy 0.67)"}
        if n_check:
            from oracle import c_oracle

            t0 = time.perf_counter()
            rows = np.sort(np.random.default_rng(5).choice(m, n_check, replace=False))
            rs, ri = c_oracle.rows_topk_f64(exp[rows], ((at, blk.astype(np.float32).reshape(-1, sy, sx)) for at, blk in blocks()),
                                            np.arange(n_check), w["metric"], keep, None)
            worst = float(np.abs(scores[rows] - rs).max())
            assert worst < 2e-3, f"float16 share: max |dscore| {worst:.2e} against the float64-accumulated oracle (bound 2e-3)"
            rec["check"] = {"rows": int(n_check), "oracle": "oracle/kpdi_oracle_c.c rows_topk_f64 over the whole (float16-stored) shard",
                            "bound": 2e-3, "max_abs_score_diff": worst,
                            "best_match_agreement": float(np.mean(indices[rows][:, 0] == ri[:, 0])),
                            "index_agreement": float(np.mean(indices[rows] == ri)),
                            "seconds": round(time.perf_counter() - t0, 2)}
        return rec
    rec = {
        "what": f"{w['name']}: rank 0's share of {n_ranks} ranks on ONE MI355X (the whole experimental set x dictionary "
                f"patterns [{lo}, {hi})), inputs resident, results collected while the next step runs",
        "shard_patterns": int(n_shard), "ms_per_step": round(dt * 1e3, 3),
        "patterns_per_s_before_the_allgather": round(m / dt, 1),  # every rank sweeps the whole experimental set
        "match_ms": round(match_ms, 3),
        "match_tflops": round(tflops, 1), "match_frac": round(tflops / F32_MFMA_PEAK_TFLOPS, 4),
        "match_form": int(cnt.get("match_form", 0)),
    }
    if n_check:
        from oracle import c_oracle
        from oracle import kpdi_oracle as ko

        t0 = time.perf_counter()
        rows = np.sort(np.random.default_rng(5).choice(m, n_check, replace=False))
        rs, ri = c_oracle.rows_topk_f64(exp[rows], ((at, blk.reshape(-1, sy, sx)) for at, blk in blocks()),
                                        np.arange(n_check), w["metric"], keep, None)
        ko.assert_topk_parity(scores[rows], indices[rows], rs, ri, atol=1e-5)
        rec["check"] = {"rows": int(n_check), "oracle": "oracle/kpdi_oracle_c.c rows_topk_f64 over the whole shard",
                        "max_abs_score_diff": float(np.abs(scores[rows] - rs).max()),
                        "index_agreement": float(np.mean(indices[rows] == ri)),
                        "seconds": round(time.perf_counter() - t0, 2)}
    return rec


def plugin_seam_leg(exp, dic, keep_n, n_per_iteration, device, ref_scores, ref_indices):
    """What a user of an UNMODIFIED kikuchipy gets: the reference's own loop
    (indexing/_dictionary_indexing.py:94-128 and `_match_chunk`, :172-203 - restated here call for call, NumPy in place
    of Dask's pass-through of NumPy results) driving this package's metric PLUGIN with a host-resident dictionary.
    Two forms, same result: the plugin as it ships - after the first chunk it sweeps the chunks the loop is about to ask
    for ahead of it, on a thread of its own, while the loop runs the reference's host merge
    (similarity_metrics._LookAhead) - and with $KPDI_SEAM_LOOKAHEAD=0: per chunk one upload, one sweep, one synchronous
    hand-over, then the host merge (where the time goes is only separable there).  Best of 3 calls each."""
    import kikuchipy_amd as kpa

    m, n = len(exp), len(dic)

    def one_form(lookahead):
        os.environ["KPDI_SEAM_LOOKAHEAD"] = "1" if lookahead else "0"
        metric = kpa.NormalizedCrossCorrelationMetric(n_experimental_patterns=m, n_dictionary_patterns=n, device=device)
        ctx = metric.context
        t = {"upload": 0.0, "sweep_and_hand_over": 0.0, "host_merge": 0.0}
        push, fin = ctx.push_dictionary_chunk, ctx.finalize

        def timed(fn, key):
            def run(*args, **kw):
                t0 = time.perf_counter()
                try:
                    return fn(*args, **kw)
                finally:
                    t[key] += time.perf_counter() - t0
            return run

        if not lookahead:  # (with the look-ahead most of these calls run on its thread, beside the host merge)
            ctx.push_dictionary_chunk = timed(push, "upload")
            ctx.finalize = timed(fin, "sweep_and_hand_over")
        best = None
        for rep in range(3):
            for k in t:
                t[k] = 0.0
            hits0 = metric.lookahead_hits
            t_start = time.perf_counter()
            experimental = metric.prepare_experimental(exp)                       # :70
            dictionary = dic.reshape((n, -1))                                     # :71
            keep = min(keep_n, n)                                                 # :67
            n_iterations = int(np.ceil(n / n_per_iteration))                      # :68
            indices = np.zeros((m, keep), dtype=np.int32)                         # :97
            scores = np.full((m, keep), -metric.sign, dtype=metric.dtype)         # :98
            starts = np.cumsum([0] + [n_per_iteration] * (n_iterations - 1))      # :100-104
            ends = np.cumsum([n_per_iteration] * n_iterations)
            ends[-1] = max(ends[-1], n)
            for start, end in zip(starts, ends):
                k_i = min(keep, end - start)
                simulated = metric.prepare_dictionary(dictionary[start:end])      # :193
                similarities = metric.match(experimental, simulated)              # :195
                idx_i = similarities.argtopk(k_i, axis=-1).reshape((-1, k_i))     # :197-201
                scores_i = similarities.topk(k_i, axis=-1).reshape((-1, k_i))
                t0 = time.perf_counter()
                idx_i = idx_i + start                                             # :118
                all_scores = np.hstack((scores, scores_i))                        # :120-128
                all_idx = np.hstack((indices, idx_i))
                order = np.argsort(-all_scores, axis=1)[:, :keep]
                scores = np.take_along_axis(all_scores, order, axis=1)
                indices = np.take_along_axis(all_idx, order, axis=1)
                t["host_merge"] += time.perf_counter() - t0
            total = time.perf_counter() - t_start
            if best is None or total < best[0]:
                best = (total, dict(t), metric.lookahead_hits - hits0)
        metric.close()
        total, split, hits = best
        r = {"patterns_per_s": round(m / total, 1), "ms_per_call": round(total * 1e3, 2),
             "ms_reference_host_merge": round(split["host_merge"] * 1e3, 2),
             "max_abs_score_diff_vs_the_timed_result": float(np.abs(scores - ref_scores).max()),
             "index_agreement_with_the_timed_result": float(np.mean(indices == ref_indices))}
        if lookahead:
            r["chunks_served_from_the_lookahead"] = int(hits)
        else:
            r.update({"ms_upload": round(split["upload"] * 1e3, 2),
                      "ms_sweep_and_hand_over": round(split["sweep_and_hand_over"] * 1e3, 2),
                      "ms_other_host": round((total - sum(split.values())) * 1e3, 2)})
        return r, scores, indices

    saved = os.environ.get("KPDI_SEAM_LOOKAHEAD")
    try:
        on, s_on, i_on = one_form(True)
        off, s_off, i_off = one_form(False)
    finally:
        if saved is None:
            os.environ.pop("KPDI_SEAM_LOOKAHEAD", None)
        else:
            os.environ["KPDI_SEAM_LOOKAHEAD"] = saved
    out = {"n_per_iteration": int(n_per_iteration), "iterations": int(np.ceil(n / n_per_iteration))}
    out.update(on)
    out["without_lookahead"] = off
    out["identical_with_and_without_lookahead"] = bool(np.array_equal(s_on, s_off) and np.array_equal(i_on, i_off))
    return out


def cpu_baseline(w, exp, dic, bg, mask, n_sample):
    """The path on this host's CPU cores, on a bounded sample (all M experimental patterns against
    the first `n_sample` dictionary patterns, n_per_iteration=2000; the cost is linear in the
    dictionary size, so patterns/s at the full N = sample rate * n_sample / N).  Two variants:

    numpy_blas  oracle/cpu_port.py: the operations the reference EXECUTES (incl. Dask's lazy
                re-evaluation of the experimental side per chunk and its content hashing), NumPy +
                the BLAS library's threads, spread over processes until all cores are busy.  In the
                build container its wall time is within 10 % of the reference's own
                (tools/cpu_baseline_crosscheck.py).
    c_openmp    oracle/kpdi_oracle_c.c: the same loop in C, OpenMP over all cores, AVX2 + FMA
                register-tiled dot products.
    `value` is the faster of the two."""
    from oracle import c_oracle, cpu_port

    visible = os.cpu_count() or 1
    cores = c_oracle.effective_cpus()  # what the cgroup quota / affinity mask really grant
    blas = min(cpu_port.blas_threads(), cores)
    n_proc = max(1, cores // max(blas, 1))
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=blas, user_api="blas")
    except Exception:
        pass
    t0 = time.perf_counter()
    pp_time = 0.0
    if w["preprocess"]:
        from oracle import kpdi_oracle as ko

        n_pp = min(256, exp.shape[0])  # per-pattern Python loop: time a slice, scale linearly
        tpp = time.perf_counter()
        st = ko.remove_static_background(exp[:n_pp], bg, "subtract")
        ko.remove_dynamic_background(st, "subtract", "frequency")
        pp_time = (time.perf_counter() - tpp) * exp.shape[0] / n_pp / cores  # a map() over all cores
    kw = dict(metric=w["metric"], keep_n=w["keep_n"], n_per_iteration=2000, signal_mask=mask)
    variants = {}
    _, _, t_np = cpu_port.run_parallel(exp, dic[:n_sample], n_proc, **kw)
    variants["numpy_blas"] = {
        "patterns_per_s": w["m"] / (t_np * w["n"] / n_sample + pp_time),
        "sample_seconds": round(t_np, 2),
        "how": f"oracle/cpu_port.py, {n_proc} process(es) x {blas} BLAS threads",
    }
    t1 = time.perf_counter()
    c_oracle.openmp_port(exp, dic[:n_sample], **kw)
    t_c = time.perf_counter() - t1
    variants["c_openmp"] = {
        "patterns_per_s": w["m"] / (t_c * w["n"] / n_sample + pp_time),
        "sample_seconds": round(t_c, 2),
        "how": f"oracle/kpdi_oracle_c.c kpdi_c_match_topk_fast, OpenMP over {cores} threads, AVX2 + FMA",
    }
    best = max(variants, key=lambda k: variants[k]["patterns_per_s"])
    return {
        "value": variants[best]["patterns_per_s"],
        "unit": "patterns/s",
        "cores": int(cores),
        "kind": "port",
        "best_variant": best,
        "variants": variants,
        "sample": (f"{w['m']} exp x " + ("the WHOLE dictionary" if n_sample >= w["n"] else f"first {n_sample} dict patterns")
                   + ", n_per_iteration=2000" + ("" if n_sample >= w["n"] else f", scaled linearly to N={w['n']}")
                   + f"; host: {visible} logical CPUs visible, {cores} usable (cgroup CPU quota / affinity), "
                   f"BLAS pool {blas} threads"
                   + (f"; pre-processing (NumPy oracle) timed on 256 patterns, scaled to all cores "
                      f"({pp_time:.2f} s)" if w["preprocess"] else "")),
        "measured_seconds": round(time.perf_counter() - t0, 2),
    }


def measure_traffic(a, kernel_substring):
    """{"traffic": bytes per launch of the kernels matching `kernel_substring`, ...} from two rocprofv3 counter passes
    of a short sub-run of this command (see main); {"traffic": None, "traffic_note": why} when that is not possible."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"traffic": None, "traffic_note": "not measured: rocprofv3 not found"}
    if any(k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_")) for k in os.environ):
        return {"traffic": None, "traffic_note": "not measured: this process is itself running under a profiler"}
    sub = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", a.workload,
           "--compute", a.compute, "--no-cpu-baseline", "--no-pcie", "--no-generation", "--no-config3", "--no-rank-shares",
           "--no-structured", "--check-rows", "0", "--no-traffic"]
    t0 = time.perf_counter()
    per_counter = {}
    tmp = tempfile.mkdtemp(prefix="kpdi_traffic_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                p = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "t", "--"] + sub, cwd="/tmp",
                                   env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            except subprocess.TimeoutExpired:
                return {"traffic": None, "traffic_note": f"not measured: the {ctr} pass timed out"}
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if kernel_substring in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                            vals.append(float(r["Counter_Value"]))
            if p.returncode != 0 or not vals:
                return {"traffic": None, "traffic_note": f"not measured: the {ctr} pass gave no counters (rc {p.returncode}): "
                                                         + p.stderr.decode(errors="replace")[-200:]}
            # one sweep = one timed region of `launches`; several kernel launches when the experimental set needs them
            per_counter[ctr] = vals
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    n_sweeps = 1 + 2 + max(2, min(2, 5))  # the sub-run's warm-up + 2 timed steps + its untimed breakdown steps (n_break in main)
    fetch = sum(per_counter["FETCH_SIZE"]) / n_sweeps * 1024 * 2   # KiB; gfx950 counts half of a wide coalesced read
    write = sum(per_counter["WRITE_SIZE"]) / n_sweeps * 1024
    return {"traffic": fetch + write,
            "traffic_detail": {"fetch_bytes": fetch, "write_bytes": write, "kernel_launches_per_sweep":
                               len(per_counter["FETCH_SIZE"]) / n_sweeps,
                               "how": "rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE, around a 5-sweep sub-run of this "
                                      "command inside this run; per sweep of the match kernel(s); FETCH_SIZE x 1024 x 2, "
                                      "WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM)",
                               "seconds": round(time.perf_counter() - t0, 1)}}


def spawn_ranks(n_ranks, argv, script=None):
    """`python bench.py --gpus N` as typed (no launcher): one child process per GPU with the
    environment a launcher would export; rank 0's stdout (the JSON line) is relayed, everything
    else goes to stderr.  Returns the exit code (the first failing rank's; the others are stopped)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    job = f"kpdi-bench-{os.getpid()}-{time.time_ns()}"
    script = script or os.path.abspath(__file__)
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KPDI_JOB_ID=job)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    line = []
    reader = threading.Thread(target=lambda: line.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    failed = ""
    live = set(range(n_ranks))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    failed = f"rank {r} exited with code {code}"
                    print(f"bench.py: {failed}", file=sys.stderr)
        time.sleep(0.05)
    for r in live:  # a rank failed: stop exactly the processes started here
        procs[r].terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=10)
    if rc == 0 and line and line[0]:
        sys.stdout.buffer.write(line[0])
        sys.stdout.flush()
        return 0
    if os.environ.get("KPDI_BENCH_NO_FALLBACK") or "--single-process" in argv:
        return rc or 1
    # last step of the fallback chain (see main): the same job as ONE process driving every GPU through a kpdi_group
    why = f"one process per GPU failed ({failed or 'no line from rank 0'})"
    print(f"bench.py: {why} - retrying as ONE process over {n_ranks} GPUs", file=sys.stderr, flush=True)
    env = dict(os.environ, KPDI_BENCH_FALLBACK_REASON=why)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, script] + list(argv) + ["--single-process"], env=env).returncode


_JSON_FD = None  # the real stdout, kept aside for the ONE JSON line (set once per process)


def main(argv=None, context_factory=None, group_factory=None):
    """The bench line - with the last step of the multi-GPU fallback chain around it.  One process per GPU is the form
    the driver launches; its gather falls back from RCCL to the host-staged one by itself (Communicator.attach).  If the
    multi-process run still fails (a rank cannot create its context, the control plane breaks, a rank dies with an
    exception) rank 0 runs the job again as ONE process driving every GPU through a kpdi_group (--single-process: RCCL
    inside the process, else peer copies), the other ranks step aside, and the line says what happened
    (`multi_gpu.gather_fallback_reason`).  $KPDI_BENCH_NO_FALLBACK=1 keeps the failure a failure."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    args = list(sys.argv[1:] if argv is None else argv)
    try:
        return _main(args, context_factory, group_factory)
    except Exception as e:  # noqa: BLE001
        if world <= 1 or os.environ.get("KPDI_BENCH_NO_FALLBACK") or "--single-process" in args:
            raise
        reason = f"one process per GPU failed on rank {rank} ({type(e).__name__}: {e})"
        print("bench.py: " + reason + (" - stepping aside" if rank else f" - retrying as ONE process over {world} GPUs"),
              file=sys.stderr, flush=True)
        if rank != 0:
            return 0  # (exit code 0: a launcher must not tear rank 0 down for it)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        os.environ["KPDI_BENCH_FALLBACK_REASON"] = reason
        if "--gpus" in args:
            args[args.index("--gpus") + 1] = str(world)
        else:
            args += ["--gpus", str(world)]
        return _main(args + ["--single-process"], context_factory, group_factory)


def _main(argv, context_factory=None, group_factory=None):
    """`context_factory(device) -> engine context` / `group_factory(devices, gather) -> engine group` replace
    `kikuchipy_amd._lib.Context` / `_lib.Group` in the CPU rehearsal of the multi-rank / multi-device control flow
    (tests/_bench_worker.py, tests/test_bench_multirank.py); never set otherwise."""
    global _JSON_FD
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="dictionary patterns in the CPU baseline sample (default: sized for ~10 s of CPU work on this "
                         "host's usable cores, at most the whole dictionary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-rows", type=int, default=None,
                    help="experimental rows of the timed result that are checked against the C oracle over the whole "
                         "dictionary before the line is printed (default: 64; 32 for the large workloads; 0 = no check)")
    ap.add_argument("--no-config3", action="store_true", help="skip the configs[2] leg of the default run")
    ap.add_argument("--no-structured", action="store_true",
                    help="skip the informational leg on physically structured data (bench_structured.py)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the informational host-pointer sweep")
    ap.add_argument("--no-rank-shares", action="store_true",
                    help="skip the informational legs that run ONE rank's share of the 8-GPU configurations (configs[3], configs[4])")
    ap.add_argument("--no-generation", action="store_true",
                    help="skip the informational sweep with the dictionary simulated on the device")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="collect every step's result before the next step is queued (kpdi_finalize) instead of while it "
                         "runs (kpdi_finalize_async / kpdi_finalize_wait)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two short rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of a sub-run of this command "
                         "that fill roofline.traffic")
    ap.add_argument("--dict-dtype", default=None, choices=["f32", "f16"],
                    help="dtype the raw dictionary is resident in (default: f32; f16 for --workload config5 --compute f16 - "
                         "BASELINE.json configs[4] is the float16 path: the dictionary at half the bytes on the host, over "
                         "PCIe and in HBM, cast exactly to float32 by the preparation kernels)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1: ONE process drives all N GPUs through a kpdi_group (one host thread per GPU inside libkpdi, "
                         "in-process RCCL communicator) instead of one process per GPU")
    ap.add_argument("--gather", default=None, choices=["rccl", "p2p"],
                    help="--single-process: how the per-GPU best-k lists reach GPU 0 (default: RCCL all-gather when the "
                         "devices are distinct, peer copies otherwise)")
    ap.add_argument("--compute", default="f32", choices=["f32", "f16x2", "f16"],
                    help="arithmetic of the match kernel; f16x2 / f16 are the opt-in float16 modes (never the default; "
                         "f16 is reduced precision)")
    a = ap.parse_args(argv)

    single = a.single_process and a.gpus > 1 and "WORLD_SIZE" not in os.environ
    if "WORLD_SIZE" not in os.environ and a.gpus > 1 and not single:
        return spawn_ranks(a.gpus, argv, script=os.environ.get("KPDI_BENCH_SCRIPT"))

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line
    # version banner through C stdio when a communicator is created, flushed at exit): keep the real
    # stdout aside for the JSON line and point file descriptor 1 - Python's and C's - at stderr.
    sys.stdout.flush()
    if _JSON_FD is None:
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)
    json_fd = _JSON_FD
    fallback_reason = os.environ.get("KPDI_BENCH_FALLBACK_REASON", "")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = a.gpus if single else 1  # devices THIS process drives
    if not single:
        a.gpus = world  # under a launcher the launcher's world size is what runs
    if world > 1 or single:
        # one node: RCCL's bootstrap can always use the loopback interface (the container's
        # hostname may not resolve), and the host driver only supports dmabuf IPC
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from kikuchipy_amd import _lib
    from kikuchipy_amd.parallel import Communicator, shard_range

    comm = Communicator(rank, world)  # world > 1: TCP rendezvous on MASTER_ADDR:MASTER_PORT (no torch)
    make_context = context_factory or _lib.Context
    # KPDI_BENCH_SHARE_GPU=1: ranks beyond the visible GPUs share them (rehearsals on a 1-GPU box)
    device = local_rank
    if os.environ.get("KPDI_BENCH_SHARE_GPU") and context_factory is None:
        device = local_rank % max(_lib.device_count(), 1)

    w = WORKLOADS[a.workload]
    large = a.workload in ("config4", "config5")
    if a.dict_dtype is None:
        a.dict_dtype = "f16" if (a.workload == "config5" and a.compute == "f16") else "f32"
    dict_np = np.float16 if a.dict_dtype == "f16" else np.float32
    if large:
        rng = np.random.default_rng(2024)
        exp = rng.integers(0, 256, (w["m"], w["sy"], w["sx"]), dtype=np.uint8)
        dic, bg = None, np.ones((w["sy"], w["sx"]), dtype=np.uint8)
        a.no_pcie = a.no_generation = True  # the informational legs belong to configs[1]
    else:
        exp, dic, bg = synth(w)
    mask = circular_mask(w["sy"], w["sx"]) if w["mask"] else None
    lo, hi = shard_range(w["n"], rank, world)
    n_local = hi - lo

    if os.environ.get("KPDI_BENCH_FAIL_RANK") == str(rank) and world > 1:  # tests: a rank that dies (fallback chain, step 3)
        sys.exit(3)
    if os.environ.get("KPDI_BENCH_RAISE_RANK") == str(rank) and world > 1:  # tests: ... with an exception, under a launcher
        raise RuntimeError("injected failure (KPDI_BENCH_RAISE_RANK)")
    metric = {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[w["metric"]]
    compute = {"f32": _lib.COMPUTE_F32, "f16x2": _lib.COMPUTE_F16X2, "f16": _lib.COMPUTE_F16}[a.compute]

    def resident(c, a0, a1):
        """raw inputs of one device resident in its HBM before the timed region: (experimental set, dictionary [a0, a1))"""
        de = c.dev_alloc(exp.nbytes)
        c.h2d(de, exp)
        if large:
            return de, upload_generated_shard(c, w, a0, a1, exp, dict_np)
        part = np.ascontiguousarray(dic[a0:a1]).astype(dict_np, copy=False)
        dd = c.dev_alloc(part.nbytes)
        c.h2d(dd, part)
        return de, dd

    if single:
        # ONE process, a group of contexts: member i sweeps dictionary block i (the same contiguous blocks as the ranks of
        # the multi-process form) and the group hands back one merged result
        ids = list(range(n_dev))
        if os.environ.get("KPDI_BENCH_SHARE_GPU") and group_factory is None:
            ids = [i % max(_lib.device_count(), 1) for i in ids]
        ctx = (group_factory or _lib.Group)(ids, a.gather)
        device = ids
        shards = [shard_range(w["n"], i, n_dev) for i in range(n_dev)]
        ctx.set_problem(w["sy"], w["sx"], mask, metric, w["keep_n"], compute)
        res = [resident(mem, a0, a1) for mem, (a0, a1) in zip(ctx.members, shards)]
        # (per-member lists in place of one pointer / size / start: the group forms of the same calls)
        d_exp, d_dic = [r[0] for r in res], [r[1] for r in res]
        n_local, lo = [a1 - a0 for a0, a1 in shards], [a0 for a0, _ in shards]
    else:
        ctx = make_context(device)
        comm.attach(ctx)
        ctx.set_problem(w["sy"], w["sx"], mask, metric, w["keep_n"], compute)
        d_exp, d_dic = resident(ctx, lo, hi)
    bg_f32 = bg.astype(np.float32)

    def queue_step():
        ctx.set_experimental_dev(d_exp, exp.dtype, w["m"])
        if w["preprocess"]:
            ctx.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
            ctx.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
        ctx.push_dictionary_chunk_dev(d_dic, dict_np, n_local, lo)

    def step():
        queue_step()
        return ctx.finalize(w["keep_n"])

    for _ in range(a.warmup):
        step()
    # HIP events around the MATCH launches only (roofline.achieved; + the all-gather for N > 1): an event record between two
    # kernels idles the GPU for ~6 us - bracketing every phase costs 0.05 ms per step (profiles/r04_share_timeline.txt).  The
    # per-phase breakdown of `extra` comes from a few untimed steps with full profiling behind the timed region.
    ctx.set_profiling("match")
    ctx.reset_counters()
    comm.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    if a.no_pipeline:
        for _ in range(a.steps):
            scores, indices = step()
    else:
        # A series of maps: the hand-over of step i's result (synchronisation, device-to-host copies, widening the
        # indices: ~0.1 ms of host time) is collected while step i + 1's kernels are already queued
        # (kpdi_finalize_async / kpdi_finalize_wait).  Every one of the K results reaches host memory inside the timed
        # region; the result of the LAST step is what is checked below.
        pending = None
        for _ in range(a.steps):
            queue_step()
            ticket = ctx.finalize_async(w["keep_n"])
            if pending is not None:
                scores, indices = ctx.finalize_wait(pending)
            pending = ticket
        scores, indices = ctx.finalize_wait(pending)
    ctx.synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    cnt = ctx.counters()
    # ---- untimed: the same step with every phase bracketed, for the breakdown (preparation, merge, bookkeeping) per step
    n_break = max(2, min(a.steps, 5))
    ctx.set_profiling(True)
    ctx.reset_counters()
    for _ in range(n_break):
        step()
    ctx.synchronize()
    brk = ctx.counters()
    ctx.set_profiling(False)

    def with_breakdown(c, b):
        """the timed loop's counters + the per-phase times of the breakdown steps, scaled to the timed loop's step count"""
        c = dict(c)
        for key in ("prep_ms", "merge_ms", "fixed_ms", "preproc_ms"):
            c[key] = b.get(key, 0.0) * a.steps / n_break
        c["preproc_launches"] = b.get("preproc_launches", 0)
        return c

    members_brk = brk.get("members")
    if "members" in cnt and members_brk:
        cnt["members"] = [with_breakdown(c, b) for c, b in zip(cnt["members"], members_brk)]
    cnt = dict(with_breakdown(cnt, brk), **({"members": cnt["members"]} if "members" in cnt else {}))
    per_rank = None
    if single:
        digest = hashlib.sha256(np.ascontiguousarray(scores).tobytes() + np.ascontiguousarray(indices).tobytes()).hexdigest()
        per_rank = [{
            "rank": i, "device": ids[i], "shard": [int(a0), int(a1)], "result_sha256": digest,
            "match_ms": c["match_ms"], "match_launches": int(c["match_launches"]), "match_flops": c["match_flops"],
            "prep_ms": c["prep_ms"], "merge_ms": c["merge_ms"], "comm_ms": c.get("comm_ms", 0.0),
            "fixed_ms": c.get("fixed_ms", 0.0), "comm_ranks": int(c.get("comm_ranks", 0)),
            "match_form": int(c.get("match_form", 0))} for i, (c, (a0, a1)) in enumerate(zip(cnt["members"], shards))]
    if world > 1:
        elapsed = comm.all_reduce_max(elapsed)
        digest = hashlib.sha256(np.ascontiguousarray(scores).tobytes() + np.ascontiguousarray(indices).tobytes()).hexdigest()
        per_rank = comm.all_gather({
            "rank": rank, "device": device, "shard": [int(lo), int(hi)], "result_sha256": digest,
            "match_ms": cnt["match_ms"], "match_launches": int(cnt["match_launches"]), "match_flops": cnt["match_flops"],
            "prep_ms": cnt["prep_ms"], "merge_ms": cnt["merge_ms"], "comm_ms": cnt.get("comm_ms", 0.0),
            "fixed_ms": cnt.get("fixed_ms", 0.0), "comm_ranks": int(cnt.get("comm_ranks", 0)),
            "match_form": int(cnt.get("match_form", 0))})
    comm.close()

    if rank != 0:
        ctx.close()
        return 0

    solo = world == 1 and not single  # one process, one GPU: the informational legs below belong to this form
    n_gpus = n_dev if single else world
    ms_per_step = elapsed / a.steps * 1e3
    value = w["m"] * a.steps / elapsed
    k_kept = cnt["k_kept"]
    launches = max(cnt["match_launches"], 1)
    flops_per_launch = cnt["match_flops"] / launches
    avg_ms = cnt["match_ms"] / launches
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    # the opt-in float16 arithmetics run on the f16 MFMA pipe (dense peak 2.5 PFLOP/s); the split form issues
    # three MFMAs per product term
    peak_tflops = F32_MFMA_PEAK_TFLOPS if a.compute == "f32" else 2500.0
    mfma_per_term = 3 if a.compute == "f16x2" else 1
    out = {
        "metric": ("experimental patterns indexed/sec (whole node), 60x60 px x 100k dict" if not large else
                   f"experimental patterns indexed/sec (whole node), {w['sy']}x{w['sx']} px x {w['n'] // 1000}k dict"),
        "value": round(value, 1),
        "unit": "patterns/s",
        "n_gpus": n_gpus,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": {"f32": "f32", "f16x2": "f16x2 (opt-in: values as two float16, f32 accumulate)",
                  "f16": "f16 (opt-in, REDUCED PRECISION: values as one float16, f32 accumulate)"}[a.compute],
        "data": "synthetic (default_rng(2024): uint8 patterns, uniform float32 dictionary"
                + (" stored as float16" if a.dict_dtype == "f16" else "") + "), raw inputs resident in HBM",
        "config": {
            "workload": w["name"],
            "experimental_patterns": w["m"],
            "dictionary_patterns": w["n"],
            "detector": [w["sy"], w["sx"]],
            "kept_pixels": k_kept,
            "metric": w["metric"],
            "keep_n": w["keep_n"],
            "parallelism": f"dictionary sharded over {n_gpus} GPU(s)" + (
                f", ONE process (kpdi_group), {cnt.get('gather', '?')} gather + merge" if single else
                (", one process per GPU, host-staged gather (RCCL unusable) + merge" if comm.gather == "host" else
                 ", one process per GPU, RCCL all-gather merge") if world > 1 else ""),
            "hand_over": ("every step's result is collected before the next step is queued" if a.no_pipeline else
                          "the result of step i reaches host memory while step i + 1 runs (finalize_async / finalize_wait); "
                          "all K results are collected inside the timed region"),
        },
        "roofline": {
            "kernel": (("kpdi::match16_kernel<20,false,4,true> (exact-f32 MFMA GEMM, 256 x 256 tiles, fused top-k), rank 0"
                        if cnt.get("match_form") == 3 else
                        "kpdi::match_topk_kernel<20,false,0> (f32 MFMA GEMM + fused top-k), rank 0")) if a.compute == "f32"
            else ("kpdi::match16_kernel<20,false,8> (f16 MFMA, f32 accumulate; peak = dense f16 MFMA)" if a.compute == "f16"
                  else "kpdi::match_topk_kernel<20,false,1,4> (split-f16: 3 f16 MFMAs per product term, all counted)"),
            "bound": "mfma",
            "achieved": round(achieved * mfma_per_term, 2),
            "peak": peak_tflops,
            "unit": "TFLOP/s",
            "frac": round(achieved * mfma_per_term / peak_tflops, 4),
            "traffic": None,  # filled below by two short rocprofv3 counter passes of a sub-run (measure_traffic)
            "flops_per_launch": flops_per_launch,
            "avg_launch_ms": round(avg_ms, 4),
            "launches": int(cnt["match_launches"]),
            "grid": int(cnt["match_grid"]),
        },
        "extra": {
            "comparisons_per_s": round(value * w["n"], 1),
            "prep_ms_per_step": round(cnt["prep_ms"] / a.steps, 4),
            "preproc_ms_per_step": round(cnt["preproc_ms"] / a.steps, 4),
            "merge_ms_per_step": round(cnt["merge_ms"] / a.steps, 4),
            "fixed_ms_per_step": round(cnt.get("fixed_ms", 0.0) / a.steps, 4),
            "best_score_mean": float(scores[:, 0].mean()),
        },
    }

    # HBM-side traffic of the match kernel cannot be read from inside the process, so the process asks rocprofv3: two
    # short counter passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, nothing else - the form the pool allows) of a 3-step
    # sub-run of THIS command, per launch of the match kernel, with the guide's corrections (KiB; FETCH_SIZE doubled on
    # gfx950).  A traffic regression then shows in the driver's own line.  Skipped (null + the reason) when rocprofv3 is
    # missing, when this process is itself being profiled, for N > 1, or with --no-traffic.
    if solo and not a.no_traffic and context_factory is None:
        out["roofline"].update(measure_traffic(a, "match"))
    else:
        out["roofline"]["traffic_note"] = "not measured: " + ("N > 1" if not solo else "--no-traffic")
    # The number of the committed rocprofv3 PMC passes of this same command (tools/summarize_pmc.py) is quoted beside it,
    # labelled as what it is: a static, earlier measurement.
    if a.workload == "config2" and solo and a.compute == "f32":
        import glob

        # (r03_pmc.json: the passes of THIS command; not r03_config3_pmc.json / r03_rank_share_pmc.json)
        pmc = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
        if pmc:
            with open(pmc[-1]) as f:
                prof = json.load(f)
            out["roofline"]["traffic_profiled"] = {
                "bytes_per_launch": prof.get("match_traffic_bytes_per_launch"),
                "kind": "static: read from the committed profile, NOT measured in this run",
                "source": os.path.relpath(pmc[-1], ROOT),
            }
        out["roofline"]["algorithmic_operand_bytes"] = float(w["n"] * k_kept * 4 + w["m"] * k_kept * 4)

    # ---- the result of the timed run is checked before anything is printed
    n_check = a.check_rows if a.check_rows is not None else (32 if large else 64)
    if n_check > 0:  # N > 1: the MERGED result (all shards, after the RCCL all-gather) is what rank 0 holds
        out["check"] = check_result(w, exp, dic, bg, mask, scores, indices, n_check, large, a.compute, dict_np)
    if per_rank is not None:
        # every rank must end with the bit-identical global result (total order of the merge)
        same = all(p["result_sha256"] == per_rank[0]["result_sha256"] for p in per_rank)
        assert same, f"ranks disagree on the merged result: {[p['result_sha256'][:12] for p in per_rank]}"
        ranks_in_comm = sorted({p["comm_ranks"] for p in per_rank})
        host_gather = not single and comm.gather == "host"  # the fallback of Communicator.attach (RCCL unusable)
        rccl = (not single and not host_gather) or cnt.get("gather") == "rccl"
        if host_gather:
            assert ranks_in_comm == [0], f"host-staged gather, yet RCCL communicators of sizes {ranks_in_comm} are attached"
            assert cnt.get("gather_ranks") == n_gpus, f"{cnt.get('gather_ranks')} lists merged, expected {n_gpus}"
        if rccl:
            assert ranks_in_comm == [n_gpus], f"RCCL communicator sizes {ranks_in_comm}, expected {n_gpus} on every rank"
        if single:  # member 0 merged one list per member, however they reached it
            assert cnt.get("gather_ranks") == n_gpus, f"{cnt.get('gather_ranks')} lists merged, expected {n_gpus}"
        avg = [p["match_ms"] / max(p["match_launches"], 1) for p in per_rank]
        tf = [p["match_flops"] / max(p["match_launches"], 1) / (m * 1e-3) / 1e12 if m > 0 else 0.0 for p, m in zip(per_rank, avg)]
        slow = int(np.argmax(avg))
        out["roofline"]["max_over_ranks"] = {
            "rank": slow, "avg_launch_ms": round(avg[slow], 4), "achieved": round(tf[slow] * mfma_per_term, 2),
            "frac": round(tf[slow] * mfma_per_term / peak_tflops, 4)}
        out["multi_gpu"] = {
            "rccl_ranks": n_gpus if rccl else 0,  # ncclCommCount of every rank's communicator (asserted above)
            "processes": 1 if single else world,
            "gather": ("rccl" if rccl else
                       "host-staged (kpdi_export_lists -> TCP control plane -> kpdi_import_lists; same merge kernel)" if host_gather
                       else "p2p (hipMemcpyPeerAsync into GPU 0)"),
            "gather_fallback_reason": (comm.gather_reason if host_gather else fallback_reason) or None,
            "lists_merged": int(cnt.get("gather_ranks", 0)) if single else world,
            "identical_result_on_every_rank": same if not single else None,  # (one process: one result)
            "allgather_ms_per_step": round(max(p["comm_ms"] for p in per_rank) / a.steps, 4),
            "allgather_ms_per_step_min_over_ranks": round(min(p["comm_ms"] for p in per_rank) / a.steps, 4),
            "per_rank": [{"rank": p["rank"], "device": p["device"], "shard": p["shard"],
                          "match_ms_per_step": round(p["match_ms"] / a.steps, 4),
                          "prep_ms_per_step": round(p["prep_ms"] / a.steps, 4),
                          "merge_ms_per_step": round(p["merge_ms"] / a.steps, 4),
                          "fixed_ms_per_step": round(p["fixed_ms"] / a.steps, 4),
                          "allgather_ms_per_step": round(p["comm_ms"] / a.steps, 4),
                          "match_form": p["match_form"]} for p in per_rank],
            "control_plane": ("none: one process, one host thread per GPU inside libkpdi (kpdi_group); data path: "
                              + ("ncclAllGather on an ncclCommInitAll communicator" if rccl else "peer copies") + " inside kpdi_group_finalize"
                              if single else
                              "kikuchipy_amd.parallel.SocketGroup (TCP, loopback); data path: "
                              + ("the ranks' lists over the same TCP star, merged by kpdi_finalize" if host_gather
                                 else "ncclAllGather inside kpdi_finalize")),
        }

    # ---- configs[2] inside the default run: circular signal mask (K = 2819) + static and dynamic
    # background removal fused with the preparation of the patterns (ONE pre-kernel), then the match
    if a.workload == "config2" and solo and a.compute == "f32" and not a.no_config3:
        try:
            w3 = WORKLOADS["config3"]
            mask3 = circular_mask(w3["sy"], w3["sx"])
            c3 = _lib.Context(device)
            c3.set_problem(w3["sy"], w3["sx"], mask3, metric, w3["keep_n"], compute)
            c3.set_profiling("match")
            reps = max(3, min(a.steps, 10))

            def step3():
                c3.set_experimental_dev(d_exp, exp.dtype, w3["m"])
                c3.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
                c3.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
                c3.push_dictionary_chunk_dev(d_dic, dict_np, n_local, lo)
                return c3.finalize(w3["keep_n"])

            for r in range(reps + 2):
                if r == 2:
                    c3.reset_counters()
                    c3.synchronize()
                    t0 = time.perf_counter()
                s3, i3 = step3()
            c3.synchronize()
            dt3 = (time.perf_counter() - t0) / reps
            cnt3 = c3.counters()
            c3.set_profiling(True)  # (untimed: the pre-kernel and the preparation, every phase bracketed)
            c3.reset_counters()
            for r in range(reps):
                step3()
            full3 = c3.counters()
            for key in ("preproc_ms", "preproc_launches", "prep_ms"):
                cnt3[key] = full3[key]
            c3.close()
            pre_ms = cnt3["preproc_ms"] / max(cnt3["preproc_launches"], 1)
            # algorithmic bytes of the pre-kernel: the pattern read and written back + its prepared row
            pre_bytes = w3["m"] * (2 * w3["sy"] * w3["sx"] * exp.dtype.itemsize + cnt3["kpad"] * 4)
            match_ms3 = cnt3["match_ms"] / max(cnt3["match_launches"], 1)
            tf3 = cnt3["match_flops"] / max(cnt3["match_launches"], 1) / (match_ms3 * 1e-3) / 1e12
            out["extra"]["config3"] = {
                "what": w3["name"],
                "patterns_per_s": round(w3["m"] / dt3, 1),
                "ms_per_step": round(dt3 * 1e3, 3),
                "kept_pixels": int(cnt3["k_kept"]),
                "match_ms": round(match_ms3, 4),
                "match_tflops": round(tf3, 2),
                "match_frac": round(tf3 / F32_MFMA_PEAK_TFLOPS, 4),
                "prekernel": "kpdi::preproc_fused_kernel<uint8,16> (static + dynamic background + mask gather + normalise)",
                "prekernel_ms": round(pre_ms, 4),
                "prekernel_algorithmic_bytes": int(pre_bytes),
                "prekernel_GBps": round(pre_bytes / (pre_ms * 1e-3) / 1e9, 1) if pre_ms > 0 else None,
                "prekernel_frac_of_8TBps": round(pre_bytes / (pre_ms * 1e-3) / 8e12, 4) if pre_ms > 0 else None,
                "dictionary_prep_ms": round(cnt3["prep_ms"] / reps, 4),
            }
            if n_check > 0:
                out["extra"]["config3"]["check"] = check_result(w3, exp, dic, bg, mask3, s3, i3, n_check, False)
        except AssertionError:
            raise
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["config3_error"] = f"{type(err).__name__}: {err}"

    # ---- configs[1]'s size on PHYSICALLY STRUCTURED data (bench_structured.py): a grain map of projections of the Ni master
    # pattern against an orientation-ordered dictionary (the reference's own benchmark shape,
    # benchmarks/indexing/test_dictionary_indexing.py:30-63), and that dictionary sorted by score both ways
    if a.workload == "config2" and solo and a.compute == "f32" and not a.no_structured and context_factory is None:
        try:
            import bench_structured

            out["extra"]["structured_config2"] = bench_structured.leg(
                _lib, device, reps=max(3, min(a.steps, 8)), n_check=0 if a.check_rows == 0 else 64,
                baseline_frac=out["extra"].get("config3", {}).get("match_frac"))
        except AssertionError:
            raise
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["structured_config2_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_pcie:
        try:
            # informational: the same sweep with the dictionary handed over as a HOST
            # buffer (pageable memory -> PCIe inside the step).  Never `value`.
            for r in range(2):  # the first pass allocates the staging buffers
                ctx.set_experimental_dev(d_exp, exp.dtype, w["m"])
                ctx.synchronize()
                t0 = time.perf_counter()
                ctx.push_dictionary_chunk(dic, 0)
                ctx.finalize(w["keep_n"])
                dt_pcie = time.perf_counter() - t0
                out["extra"]["pcie_inclusive_patterns_per_s"] = round(w["m"] / dt_pcie, 1)
                # its roofline is the host link, not the matrix pipe: the whole dictionary crosses PCIe inside the step
                # (pageable memory; 63 GB/s = the link's rate in MI355X_MICROARCH.md / SURVEY.md 8(f1))
                out["extra"]["pcie_inclusive"] = {
                    "ms_per_step": round(dt_pcie * 1e3, 3), "bound": "host link (PCIe, pageable source)",
                    "bytes_over_link": int(dic.nbytes), "achieved_GBps": round(dic.nbytes / dt_pcie / 1e9, 1),
                    "peak_GBps": 63.0, "frac": round(dic.nbytes / dt_pcie / 63e9, 3)}
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["pcie_inclusive_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_pcie:
        try:
            # informational: a SERIES of maps against one dictionary that was handed over as a host
            # buffer once and stays prepared in HBM (kpdi_hold_dictionary_chunk / kpdi_sweep_held):
            # per map only the experimental set moves.  Never `value` (it skips prepare_dictionary).
            ctx.hold_dictionary_chunk(dic, 0)
            ctx.synchronize()
            reps = 3
            for r in range(reps + 1):
                if r == 1:
                    ctx.synchronize()
                    t0 = time.perf_counter()
                ctx.set_experimental(exp, None)  # host buffer, PCIe inside the step
                if w["preprocess"]:
                    ctx.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
                    ctx.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
                ctx.sweep_held()
                s_held, i_held = ctx.finalize(w["keep_n"])
            dt = (time.perf_counter() - t0) / reps
            out["extra"]["resident_dictionary"] = {
                "what": "maps 2..n of a series: dictionary prepared once and held in HBM, experimental set from host",
                "patterns_per_s": round(w["m"] / dt, 1),
                "held_bytes": ctx.held_size()[1],
                "identical_to_value_run": bool(np.array_equal(s_held, scores) and np.array_equal(i_held, indices)),
            }
            ctx.release_held()
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["resident_dictionary_error"] = f"{type(err).__name__}: {err}"

    for key, mode, mfmas, what in (
            ("split_f16_mode", "COMPUTE_F16X2", 3,
             "opt-in KPDI_COMPUTE_F16X2: same sweep, operands as two float16 terms, 3 f16 MFMAs per product term"),
            ("f16_mode", "COMPUTE_F16", 1,
             "opt-in KPDI_COMPUTE_F16, REDUCED PRECISION: same sweep, operands rounded to one float16, 1 f16 MFMA "
             "per 16 product terms")):
        if not (solo and not a.no_generation and a.compute == "f32"):
            break
        try:
            # informational: the same sweep with the OPT-IN float16 arithmetics of the match kernel.
            # Never `value`: the headline stays the exact-f32 GEMM north_star names.
            c16 = _lib.Context(device)
            c16.set_problem(w["sy"], w["sx"], mask, metric, w["keep_n"], getattr(_lib, mode))
            c16.set_profiling(True)
            for r in range(4):
                if r == 1:
                    c16.reset_counters()
                    c16.synchronize()
                    t0 = time.perf_counter()
                c16.set_experimental_dev(d_exp, exp.dtype, w["m"])
                if w["preprocess"]:
                    c16.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
                    c16.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
                c16.push_dictionary_chunk_dev(d_dic, dict_np, n_local, lo)
                s16, i16 = c16.finalize(w["keep_n"])
            dt16 = (time.perf_counter() - t0) / 3
            cnt16 = c16.counters()
            c16.close()
            tf = mfmas * cnt16["match_flops"] / (cnt16["match_ms"] * 1e-3) / 1e12
            out["extra"][key] = {
                "what": what,
                "patterns_per_s": round(w["m"] / dt16, 1),
                "match_ms": round(cnt16["match_ms"] / 3, 3),
                "prep_ms": round(cnt16["prep_ms"] / 3, 3),
                "f16_mfma_tflops": round(tf, 1),
                "f16_mfma_frac_of_2500": round(tf / 2500.0, 3),
                "max_abs_score_diff_vs_f32": float(np.abs(s16 - scores).max()),
                "mean_abs_score_diff_vs_f32": float(np.abs(s16 - scores).mean()),
                "index_mismatch_fraction_vs_f32": float(np.mean(i16 != indices)),
                "best_match_mismatch_fraction_vs_f32": float(np.mean(i16[:, 0] != indices[:, 0])),
            }
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"][key + "_error"] = f"{type(err).__name__}: {err}"

    # ---- the 8-GPU configurations, as far as one GPU can show them: rank 0's share of configs[3] and of configs[4]
    if a.workload == "config2" and solo and a.compute == "f32" and not a.no_rank_shares and context_factory is None:
        for key, n_ranks, reps, dd, cmp in (("config2", 4, 40, d_dic, "f32"), ("config2", 8, 40, d_dic, "f32"),
                                            ("config4", 8, 3, d_dic, "f32"), ("config5", 8, 3, None, "f32"),
                                            ("config5", 8, 8, None, "f16")):
            name = f"{key}_share_of_{n_ranks}" + ("_f16" if cmp == "f16" else "")
            try:
                rec = rank_share_leg(_lib, _lib.Context, device, key, n_ranks, dd if dict_np == np.float32 else None, dic,
                                     reps, 0 if a.check_rows == 0 else 16, shard_range, compute=cmp)
                if key == "config2":  # strong scaling of the headline job before the all-gather: this step / (t_1 / N)
                    rec["step_over_even_share"] = round(rec["ms_per_step"] / (ms_per_step / n_ranks), 4)
                out["extra"][name] = rec
            except Exception as err:  # an informational leg must not cost the bench line
                out["extra"][name + "_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_generation and a.compute == "f32":
        try:
            # informational: float64 arithmetic (the reference's dtype=float64) - the f32 sweep as the screen,
            # float64 rescoring of keep_n + 12 candidates per pattern from the raw patterns (csrc/rescore.hip).
            c64 = _lib.Context(device)
            c64.set_problem(w["sy"], w["sx"], mask, metric, w["keep_n"], _lib.COMPUTE_F64)
            c64.set_profiling(True)
            for r in range(4):
                if r == 1:
                    c64.reset_counters()
                    c64.synchronize()
                    t0 = time.perf_counter()
                c64.set_experimental_dev(d_exp, exp.dtype, w["m"])
                if w["preprocess"]:
                    c64.remove_static_background(bg_f32, _lib.OP_SUBTRACT, False)
                    c64.remove_dynamic_background(_lib.OP_SUBTRACT, _lib.DOMAIN_FREQUENCY, 0.0, 4.0)
                c64.push_dictionary_chunk_dev(d_dic, dict_np, n_local, lo)
                s64, i64 = c64.finalize(w["keep_n"])
            dt64 = (time.perf_counter() - t0) / 3
            cnt64 = c64.counters()
            c64.close()
            out["extra"]["float64_mode"] = {
                "what": "KPDI_COMPUTE_F64 (dtype=float64): f32 MFMA screen + float64 rescoring of keep_n + 12 candidates "
                        "per pattern and chunk, certified with the WORST-CASE bound of a K-term f32 dot product ((K + 2) 2^-24: a "
                        "proof for any data, the default; KPDI_F64_EPS=statistical = round 4's bound).  Cost of the proof over the "
                        "statistical bound: none at configs[1] nor on an adversarial near-tie set (profiles/r05_f64_bounds.txt)",
                "certificate": {1: "statistical", 2: "worstcase"}.get(int(cnt64.get("f64_certificate", 0))),
                "patterns_per_s": round(w["m"] / dt64, 1),
                "match_ms": round(cnt64["match_ms"] / 3, 3),
                "rescore_ms": round(cnt64["rescore_ms"] / 3, 3),
                "extra_screening_passes": int(cnt64["rescore_extra_passes"]),
                "uncertified_patterns": int(cnt64["uncertified_patterns"]),
                "max_abs_score_diff_vs_f32": float(np.abs(s64 - scores).max()),
                "index_mismatch_fraction_vs_f32": float(np.mean(i64 != indices)),
            }
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["float64_mode_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_rank_shares and a.workload == "config2" and a.compute == "f32" and context_factory is None:
        try:
            # informational: the reference's CHUNKED call shape (n_per_iteration = a tenth of the tutorial's dictionary, 3044
            # patterns, doc/tutorials/pattern_matching.ipynb:582; loop indexing/_dictionary_indexing.py:100-128) on the resident
            # inputs - one GPU taking all 33 chunks, and ONE member's pieces of the same call on a kpdi_group of 8 (the library's own
            # assignment, csrc/group_assign.h); small chunks wait for company and are swept together (csrc/sweep.hip)
            per, n_dev = 3044, 8
            bounds = [(s0, min(s0 + per, w["n"])) for s0 in range(0, w["n"], per)]
            loads = [0] * n_dev
            mine = []
            for s0, e0 in bounds:
                for member, row0, rows in _lib.Group.assign_chunk(n_dev, w["n"], loads, e0 - s0):
                    if member == 0:
                        mine.append((s0 + row0, rows))
            cc = _lib.Context(device)
            cc.set_problem(w["sy"], w["sx"], mask, metric, w["keep_n"], compute)
            row_bytes = w["sy"] * w["sx"] * np.dtype(dict_np).itemsize

            def chunked(pieces, reps=6):
                for r in range(reps + 2):
                    if r == 2:
                        cc.reset_counters()
                        cc.synchronize()
                        t0 = time.perf_counter()
                    cc.set_experimental_dev(d_exp, exp.dtype, w["m"])
                    for s0, rows in pieces:
                        cc.push_dictionary_chunk_dev(d_dic + s0 * row_bytes, dict_np, rows, s0)
                    res = cc.finalize(w["keep_n"])
                return (time.perf_counter() - t0) / reps * 1e3, res, cc.counters()

            t_all, (s_c, i_c), cnt_c = chunked([(s0, e0 - s0) for s0, e0 in bounds])
            t_mem, _, cnt_m = chunked(mine)
            cc.close()
            out["extra"]["chunked_call"] = {
                "what": f"configs[1] pushed as {len(bounds)} chunks of {per} patterns (the reference's n_per_iteration loop), raw inputs "
                        "resident: one GPU taking every chunk, and member 0's pieces of the same call on a kpdi_group of 8",
                "ms_per_call_one_gpu": round(t_all, 3), "patterns_per_s_one_gpu": round(w["m"] / t_all * 1e3, 1),
                "ms_per_step_single_pass": round(ms_per_step, 3),
                "sweeps_per_call": int(cnt_c["match_launches"] // 6), "coalesced_sweeps_per_call": int(cnt_c["coalesced_sweeps"] // 6),
                "identical_to_the_single_pass": bool(np.array_equal(s_c, scores) and np.array_equal(i_c, indices)),
                "group_member_pieces": len(mine), "group_member_patterns": int(sum(r for _, r in mine)),
                "group_member_ms": round(t_mem, 3), "group_member_over_even_share": round(t_mem / (t_all / n_dev), 4),
                "group_member_sweeps": int(cnt_m["match_launches"] // 6),
            }
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["chunked_call_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_pcie and a.workload == "config2" and a.compute == "f32" and context_factory is None:
        # informational: the drop-in seam itself - the reference's loop driving the metric plugin (host-resident dictionary)
        for per in (3044, 25000):  # the tutorial's tenth of its dictionary; a quarter of this one
            try:
                out["extra"].setdefault("plugin_seam", {
                    "what": "an UNMODIFIED kikuchipy's loop (indexing/_dictionary_indexing.py:94-128, :172-203, restated in "
                            "bench.py) driving kikuchipy_amd.NormalizedCrossCorrelationMetric at configs[1], dictionary in host "
                            "memory.  The plugin sweeps the chunks the loop is about to ask for ahead of it, beside the reference's "
                            "host merge (`without_lookahead`: per chunk one upload, one sweep, one synchronous hand-over, then the "
                            "host merge).  Best of 3 calls; the stand-alone driver (`value`) keeps the best-k on the device instead"})
                out["extra"]["plugin_seam"][f"n_per_iteration_{per}"] = plugin_seam_leg(exp, dic, w["keep_n"], per, device, scores, indices)
            except Exception as err:  # an informational leg must not cost the bench line
                out["extra"]["plugin_seam_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_pcie and a.workload == "config2" and a.compute == "f32" and context_factory is None:
        # informational: the stand-alone driver as a USER calls it - kikuchipy_amd.dictionary_indexing(exp, dictionary in host
        # memory, n_per_iteration=...) - engine made and closed by the call; wall time of the whole call, best of 3
        try:
            import kikuchipy_amd as kpa

            leg = {"what": "kikuchipy_amd.dictionary_indexing(4096 patterns, 100 000-pattern dictionary in HOST memory, metric='ncc', "
                           "keep_n=20, n_per_iteration=...) as a user calls it: the call makes its engine - or takes the idle one "
                           "the call before left (kikuchipy_amd._lib: engine pool) - feeds it over the host link and hands it "
                           "back; wall time of the whole call, best of 3 after a warm-up call"}
            for per in (None, 3044):
                best_t, res = None, None
                for rep in range(4):
                    t0 = time.perf_counter()
                    res = kpa.dictionary_indexing(exp, dic, metric=w["metric"], keep_n=w["keep_n"], n_per_iteration=per,
                                                  device=device, verbose=False)
                    dt = time.perf_counter() - t0
                    if rep and (best_t is None or dt < best_t):
                        best_t = dt
                leg["single_pass" if per is None else f"n_per_iteration_{per}"] = {
                    "ms_per_call": round(best_t * 1e3, 2), "patterns_per_s": round(w["m"] / best_t, 1),
                    "identical_to_the_timed_result": bool(np.array_equal(res.scores, scores)
                                                          and np.array_equal(res.simulation_indices, indices))}
            out["extra"]["standalone_call"] = leg
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["standalone_call_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_generation:
        try:
            # informational: the dictionary is SIMULATED on the device inside the step
            # from n rotations (32 B each over PCIe) - the orientations `get_sample_fundamental` emits for the cubic
            # fundamental zone, in its order - and the Ni master pattern the reference ships (2 x 401 x 401,
            # tests/golden/projection.npz), then swept as above.  Never `value`.
            import bench_structured
            from kikuchipy_amd.sampling import get_sample_fundamental

            quat = get_sample_fundamental(semi_edge_steps=bench_structured.SEMI_EDGE_STEPS, point_group="m-3m")
            if len(quat) < w["n"]:  # (a workload larger than the sampler's 100 347: random orientations fill up)
                fill = np.random.default_rng(7).standard_normal((w["n"] - len(quat), 4))
                quat = np.concatenate([quat, fill / np.linalg.norm(fill, axis=1)[:, None]])
            quat = np.ascontiguousarray(quat[:w["n"]])
            ctx.set_master_pattern(*bench_structured.master_pattern())
            pc = (0.421, 0.7794, 0.5049)
            aspect = w["sx"] / w["sy"]
            bounds = [-aspect * pc[0] / pc[2], aspect * (1 - pc[0]) / pc[2], -(1 - pc[1]) / pc[2], pc[1] / pc[2]]
            ct, st = np.cos(np.deg2rad(70.0)), np.sin(np.deg2rad(70.0))
            det_to_sample = np.array([[0, 1, 0], [-st, 0, ct], [ct, 0, st]], dtype=np.float64).T
            ctx.set_detector(bounds, pc[2], w["sy"], w["sx"], det_to_sample)
            ctx.set_profiling(True)
            reps = 3
            for r in range(reps + 1):
                if r == 1:
                    ctx.reset_counters()
                    ctx.synchronize()
                    t0 = time.perf_counter()
                ctx.set_experimental_dev(d_exp, exp.dtype, w["m"])
                ctx.push_rotations_chunk(quat, 0, True, -1.0, 1.0)
                ctx.finalize(w["keep_n"])
            dt = (time.perf_counter() - t0) / reps
            pj = ctx.counters()["project_ms"] / reps
            ctx.set_profiling(False)
            out["extra"]["dictionary_generation"] = {
                "what": "the dictionary's patterns projected from the Ni master pattern (2 x 401 x 401) at the sampler's orientations "
                        "(get_sample_fundamental, cubic fundamental zone, sampler order) inside the step (kpdi::project_kernel)",
                "project_ms_per_step": round(pj, 3),
                "gpixel_per_s": round(w["n"] * w["sy"] * w["sx"] / (pj * 1e-3) / 1e9, 1),
                "patterns_per_s_including_generation": round(w["m"] / dt, 1),
            }
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["dictionary_generation_error"] = f"{type(err).__name__}: {err}"

    if solo and not a.no_generation:
        try:
            # informational: orientation refinement of m patterns simulated from the same
            # master pattern (1 degree off, noise added), SciPy-compatible Nelder-Mead on the device; beside
            # it the oracle (NumPy objective + scipy.optimize.minimize, what the reference runs per pattern)
            # on a few of the patterns.  Never `value`.
            from kikuchipy_amd.indexing._refinement import rotation_from_euler

            rng = np.random.default_rng(7)
            eu = np.column_stack([rng.uniform(0.3, 6, w["m"]), rng.uniform(0.3, 2.8, w["m"]), rng.uniform(0.3, 6, w["m"])])
            mpu = np.fft.irfft2(np.fft.rfft2(rng.standard_normal((401, 401))) * np.exp(
                -(np.add.outer(np.fft.fftfreq(401) ** 2, np.fft.rfftfreq(401) ** 2)) / (2 * 0.03**2)), s=(401, 401))
            mpu = mpu.astype(np.float32)
            ctx.set_master_pattern(mpu)
            sim = ctx.project_patterns(rotation_from_euler(eu))
            noisy = sim + 0.3 * sim.std() * rng.standard_normal(sim.shape).astype(np.float32)
            pats = ((noisy - noisy.min()) / (noisy.max() - noisy.min()) * 255).astype(np.uint8).reshape(-1, w["sy"], w["sx"])
            eu0 = eu + np.deg2rad(rng.uniform(-1, 1, eu.shape))
            pcs = np.tile(pc, (w["m"], 1, 1))
            ctx.refine_set_patterns(pats, mask, False, det_to_sample)
            for r in range(3):
                if r == 1:
                    ctx.reset_counters()
                    t0 = time.perf_counter()
                res = ctx.refine_solve(_lib.REFINE_ORI, eu0[:, None, :], pcs)
            dt = (time.perf_counter() - t0) / 2
            kern = ctx.counters()["refine_ms"] / 2
            evals = float(res[:, 0, 1].sum())
            k_ref = int(w["sy"] * w["sx"] if mask is None else np.count_nonzero(~mask))
            out["extra"]["refinement"] = {
                "what": f"refine_orientation of {w['m']} patterns, Nelder-Mead on the device (kpdi::refine_solve_kernel)",
                "patterns_per_s": round(w["m"] / dt, 1),
                "kernel_ms": round(kern, 3),
                "objective_evaluations_per_s": round(evals / (kern * 1e-3), 1),
                "gpixel_per_s": round(evals * k_ref / (kern * 1e-3) / 1e9, 1),
                "mean_evaluations": round(evals / w["m"], 1),
                "mean_score": round(float(1 - res[:, 0, 0].mean()), 4),
            }
            if not a.no_cpu_baseline:
                from oracle import kpdi_oracle as ko

                keep = None if mask is None else ~mask.ravel()
                dc = ko.direction_cosines_fixed_pc(np.asarray(bounds), pc[2], w["sy"], w["sx"], det_to_sample, keep)
                n_cpu = 8
                t0 = time.perf_counter()
                for i in range(n_cpu):
                    p = pats[i].ravel() if keep is None else pats[i].ravel()[keep]
                    ko.refine_solver(p, "ori", eu0[i], mpu, mpu, False, direction_cosines=dc)
                out["extra"]["refinement"]["cpu_port_patterns_per_s"] = round(n_cpu / (time.perf_counter() - t0), 2)
        except Exception as err:  # an informational leg must not cost the bench line
            out["extra"]["refinement_error"] = f"{type(err).__name__}: {err}"

    if not a.no_cpu_baseline:  # rank 0 only, whatever N (the other ranks have left)
        try:
            # bounded: 50 000 dictionary patterns of configs[1] per 8 usable cores (~10 s of wall time per variant on the
            # build container's 8 vCPUs; on the GPU boxes' 16 usable cores that is the WHOLE 100 000-pattern dictionary
            # in ~3 + ~4.5 s - no extrapolation), whatever the workload; never more than the dictionary
            from oracle import c_oracle

            sample = a.cpu_sample or int(50000 * c_oracle.effective_cpus() / 8)
            n_sample = min(sample, w["n"])
            if large:
                n_sample = max(500, min(BLOCK, int(sample * (4096 * 3600) / (w["m"] * w["sy"] * w["sx"]) / 2.5)))
                dic = dictionary_block(w, 0, exp)
            out["cpu_baseline"] = cpu_baseline(w, exp, dic, bg, mask, n_sample)
        except Exception as err:
            out["cpu_baseline"] = {"value": None, "unit": "patterns/s", "cores": 0, "kind": "port", "sample": "",
                                   "error": f"{type(err).__name__}: {err}"}
    ctx.close()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    return 0


if __name__ == "__main__":
    sys.exit(main())
