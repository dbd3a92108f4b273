"""The driver's command on the GPU: `python bench.py` prints ONE JSON line that carries the contract's keys, `roofline`
and `cpu_baseline`, and none of its informational legs failed (each of them swallows its own exception into
`extra.<leg>_error` so that it cannot cost the line - this test is where such a failure shows)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-sample", "4000"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def timing_failures(out):
    """The TIMING bars of the line (everything else - keys, parity, identities - is asserted strictly below).  A bar missed
    on a first run is measured once more before it fails the suite: these are wall-clock ratios on a shared host."""
    ex, bad = out["extra"], []

    def bar(ok, what):
        if not ok:
            bad.append(what)

    st = ex["structured_config2"]
    # physically structured data (grain map x orientation-ordered Ni dictionary) runs like random data; a dictionary sorted
    # by score (the hostile order for a threshold-screened top-k) stays above 0.8 of the peak
    bar(st["match_frac"] >= 0.87 and st["match_frac"] >= ex["config3"]["match_frac"] - 0.02, ("structured", st["match_frac"]))
    bar(st["dictionary_sorted_ascending"]["match_frac"] >= 0.80, ("ascending", st["dictionary_sorted_ascending"]["match_frac"]))
    bar(st["dictionary_sorted_descending"]["match_frac"] >= 0.87, ("descending", st["dictionary_sorted_descending"]["match_frac"]))
    bar(ex["chunked_call"]["group_member_over_even_share"] < 1.25, ("group member", ex["chunked_call"]["group_member_over_even_share"]))
    sa = ex["standalone_call"]  # the chunked call of the user within 1.2 x the single pass
    bar(sa["n_per_iteration_3044"]["ms_per_call"] < 1.2 * sa["single_pass"]["ms_per_call"],
        ("standalone", sa["n_per_iteration_3044"]["ms_per_call"], sa["single_pass"]["ms_per_call"]))
    # one rank's share of an 8- (4-) rank job stays within 10 (6) % of an even share of the whole step (VERDICT r05 item 1;
    # measured 1.066 - 1.08 and 1.035 - 1.044: profiles/r06_rank_share.json), on the wide kernel + tailgemm.hip
    # (this run times 3 steps after 1 warm-up: its whole step - the denominator - comes out 2-3 % above the steady 21.2 ms,
    # 21.8-21.9 ms, which puts the N = 4 ratio at 1.00-1.01 here; the lower bounds only guard against a nonsensical line)
    share, share4 = ex["config2_share_of_8"], ex["config2_share_of_4"]
    bar(0.95 <= share["step_over_even_share"] < 1.10 and share["match_frac"] >= 0.85,
        ("share of 8", share["step_over_even_share"], share["match_frac"]))
    bar(0.95 <= share4["step_over_even_share"] < 1.06, ("share of 4", share4["step_over_even_share"]))
    return bad


def test_default_bench_line_and_its_legs():
    out = run_bench()
    failed = {k: v for k, v in out["extra"].items() if k.endswith("_error")}
    assert not failed, failed
    bad = timing_failures(out)
    if bad:  # measured once more; the second line is then held to everything
        out = run_bench()
        again = timing_failures(out)
        assert not again, {"first run": bad, "second run": again}
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["unit"] == "patterns/s" and out["dtype"] == "f32"
    assert "configs[1]" in out["config"]["workload"] and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == 157.3 and 0.5 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["avg_launch_ms"] <= out["ms_per_step"]
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] == "port" and cb["cores"] >= 1
    assert out["check"]["index_agreement"] == 1.0
    failed = {k: v for k, v in out["extra"].items() if k.endswith("_error")}
    assert not failed, failed
    for leg in ("config3", "config2_share_of_4", "config2_share_of_8", "config4_share_of_8", "config5_share_of_8",
                "config5_share_of_8_f16", "chunked_call", "plugin_seam", "float64_mode", "f16_mode", "split_f16_mode", "resident_dictionary", "dictionary_generation", "refinement"):
        assert leg in out["extra"], leg
    # physically structured data (grain map x orientation-ordered Ni dictionary) runs like random data; a dictionary sorted
    # by score (the hostile order for a threshold-screened top-k) stays above 0.8 of the peak
    st = out["extra"]["structured_config2"]
    assert st["kept_pixels"] == 2819 and st["check"]["index_agreement"] > 0.99 and st["check"]["max_abs_score_diff"] < 1e-5
    assert st["dictionary_sorted_ascending"]["scores_identical_to_sampler_order"]
    for leg in ("config2_share_of_8", "config4_share_of_8", "config5_share_of_8"):
        assert out["extra"][leg]["check"]["index_agreement"] == 1.0
    # the arithmetic configs[4] NAMES (fp16 MFMA, f32 accumulate, K = 14 400, float16-resident dictionary), in the driver's line
    f16 = out["extra"]["config5_share_of_8_f16"]
    assert f16["match_form"] == 2 and f16["shard_patterns"] == 62500 and "float16" in f16["what"]
    assert f16["check"]["rows"] == 16 and f16["check"]["max_abs_score_diff"] < f16["check"]["bound"] == 2e-3
    assert f16["check"]["best_match_agreement"] >= 0.9
    assert 0.3 < f16["match_frac"] < 1.0 and f16["match_frac"] < f16["match_frac_of_random_operand_ceiling"] < 1.1
    assert out["extra"]["float64_mode"]["certificate"] == "worstcase" and out["extra"]["float64_mode"]["uncertified_patterns"] == 0
    ch = out["extra"]["chunked_call"]  # the reference's chunked call: same result, small chunks swept together, a member near its even share
    assert ch["identical_to_the_single_pass"] and ch["coalesced_sweeps_per_call"] >= 1 and ch["sweeps_per_call"] < 33
    assert ch["group_member_patterns"] == 12500 and ch["group_member_sweeps"] == 1
    for per in (3044, 25000):  # the drop-in seam: the reference's loop around the plugin gives the timed run's result
        seam = out["extra"]["plugin_seam"][f"n_per_iteration_{per}"]
        assert seam["patterns_per_s"] > 0 and seam["max_abs_score_diff_vs_the_timed_result"] < 1e-6, seam
        assert seam["index_agreement_with_the_timed_result"] > 0.999
        assert seam["identical_with_and_without_lookahead"] and seam["chunks_served_from_the_lookahead"] == seam["iterations"] - 1
        assert seam["without_lookahead"]["ms_upload"] > 0
    sa = out["extra"]["standalone_call"]  # the user's call: same result, chunked or not (its time: `timing_failures`)
    assert sa["single_pass"]["identical_to_the_timed_result"] and sa["n_per_iteration_3044"]["identical_to_the_timed_result"]
    share = out["extra"]["config2_share_of_8"]
    assert share["match_form"] == 3
    f16p = f16.get("roofline_profiled")
    assert f16p and f16p["traffic"] > 0 and f16p["fetch_over_algorithmic"] > 1 and 0 < f16p["mfma_busy"] < 1
