"""README.md's and DESIGN.md's measurement blocks, generated from ONE committed bench line (no
hand-typed numbers, no ranges over collections).

    python tools/make_measurements.py [bench.json ...]      (default: BENCH_r06.json if the driver has written it, then
                                                             profiles/r06_bench_n1.json)

Every figure is `python bench.py --gpus 1` (the driver's command) on one MI355X; the first file that exists and parses is the
source of every number, the others are listed beside it.  The driver's own record (BENCH_rNN.json, `parsed` + `tail`) goes
first when present.  The block between `<!-- measurements:begin ... -->` and `<!-- measurements:end -->` of README.md (short)
and DESIGN.md (full) is replaced."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    with open(path) as f:
        text = f.read()
    try:
        d = json.loads(text)
    except json.JSONDecodeError:
        d = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
    if "parsed" in d and "metric" not in d:  # the driver's record: the line itself is in `tail` / `run.stdout_tail`
        for key in ("tail", "stdout"):
            for ln in str(d.get(key, "")).splitlines():
                if ln.startswith("{") and '"metric"' in ln:
                    try:
                        return json.loads(ln), "driver"
                    except json.JSONDecodeError:
                        pass
        return d["parsed"], "driver (parsed keys only)"
    return d, "builder"


def k(v, digits=1):
    return f"{v / 1e3:.{digits}f} k"


def rows(d):
    e = d.get("extra", {})
    r = d["roofline"]
    out = []
    add = lambda what, value, key: out.append((what, value, key))  # noqa: E731
    add("**configs[1]**: 4096 × 100 000 × 60², ncc, keep_n 20, raw inputs resident",
        f"**{k(d['value'])} patterns/s**, {d['ms_per_step']:.2f} ms per step", "`value`, `ms_per_step`")
    add("match kernel (`" + r["kernel"].split(" ")[0] + "`)",
        f"{r['avg_launch_ms']:.2f} ms = {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f} of the f32 MFMA peak** (157.3)", "`roofline`")
    if r.get("traffic"):
        alg = r.get("algorithmic_operand_bytes") or 1.499e9
        add("fabric traffic of that kernel per launch (rocprofv3 counters inside the run)",
            f"{r['traffic'] / 1e9:.2f} GB = {r['traffic'] / alg:.1f} × the operands, {r['traffic'] / r['avg_launch_ms'] / 1e9:.2f} TB/s", "`roofline.traffic`")
    cb = d.get("cpu_baseline") or {}
    if cb.get("value"):
        add(f"CPU beside it ({cb.get('best_variant', cb.get('kind'))}, {cb['cores']} usable cores, {cb.get('sample', '').split(',')[0]})",
            f"{k(cb['value'], 2)} patterns/s → GPU / CPU = {d['value'] / cb['value']:.0f}", "`cpu_baseline`")
    c3 = e.get("config3")
    if c3:
        add("**configs[2]**: + circular mask + static / dynamic background (one fused pre-kernel)",
            f"{k(c3['patterns_per_s'])} patterns/s, match {c3['match_frac']:.3f} of peak, pre-kernel {c3['prekernel_ms'] * 1e3:.0f} µs", "`extra.config3`")
    st = e.get("structured_config2")
    if st:
        add("the same size on **structured data**: 64 × 64 grain map × Ni dictionary of `get_sample_fundamental` (cubochoric, sampler order)",
            f"{k(st['patterns_per_s'])} patterns/s, match **{st['match_frac']:.3f}** of peak; "
            f"{st.get('candidates_appended_per_lane_list', '?')} candidates appended per lane list, "
            f"{st.get('buffer_overflows_per_launch', '?')} buffer overflows per launch", "`extra.structured_config2`")
        for key, name in (("dictionary_sorted_ascending", "… dictionary sorted by RISING score against pattern 0 (hostile)"),
                          ("dictionary_sorted_descending", "… sorted by falling score")):
            if key in st:
                s = st[key]
                add(name, f"match {s['match_frac']:.3f} of peak, {s.get('buffer_overflows_per_launch', '?')} overflows per launch", f"`….{key}`")
    for n in (4, 8):
        s = e.get(f"config2_share_of_{n}")
        if s:
            add(f"one rank's share of configs[1] at **N = {n}** ({s['shard_patterns']} patterns), on one GPU",
                f"{s['ms_per_step']:.3f} ms per step = **{s['step_over_even_share']:.3f} × the even share**; match {s['match_ms']:.3f} ms = "
                f"{s['match_frac']:.3f} of peak (kernel form {s['match_form']})", f"`extra.config2_share_of_{n}`")
    if e.get("config2_share_of_8"):
        add("⇒ whole-node rate at N = 8 before the all-gather (NOT measured on 8 GPUs)",
            f"{k(d['value'] * 8 / e['config2_share_of_8']['step_over_even_share'], 0)} patterns/s of {k(d['value'] * 8, 0)} (linear)", "derived")
    s = e.get("config4_share_of_8")
    if s:
        add("one rank's share of **configs[3]** (40 000 × 37 500, ndp)", f"{s['ms_per_step']:.1f} ms per step, match {s['match_frac']:.3f} of peak",
            "`extra.config4_share_of_8`")
    s = e.get("config5_share_of_8")
    if s:
        add("one rank's share of **configs[4]** in f32 (4096 × 62 500 × 120²)", f"{s['ms_per_step']:.1f} ms per step, match {s['match_frac']:.3f} of peak",
            "`extra.config5_share_of_8`")
    s = e.get("config5_share_of_8_f16")
    if s:
        p = s.get("roofline_profiled") or {}
        extra = (f"; committed counters: {p['traffic'] / 1e9:.1f} GB fabric traffic = {p['fetch_over_algorithmic']:.1f} × the operands, "
                 f"MFMA busy {p['mfma_busy']:.2f} at {p['clock_GHz']:.2f} GHz") if p else ""
        add("… in the arithmetic configs[4] names (f16 MFMA, f32 accumulate; reduced precision)",
            f"{s['ms_per_step']:.2f} ms per step, match {s['match_frac']:.3f} of the 2.5 PFLOP/s f16 peak "
            f"({s['match_frac_of_random_operand_ceiling']:.2f} of the random-operand ceiling){extra}", "`extra.config5_share_of_8_f16`")
    sa = e.get("standalone_call")
    if sa:
        add("`kikuchipy_amd.dictionary_indexing(exp, dictionary in HOST memory)`, whole call",
            f"{sa['single_pass']['ms_per_call']:.1f} ms in one pass; {sa['n_per_iteration_3044']['ms_per_call']:.1f} ms with `n_per_iteration=3044` "
            f"(**{sa['n_per_iteration_3044']['ms_per_call'] / sa['single_pass']['ms_per_call']:.2f} ×**)", "`extra.standalone_call`")
    if e.get("pcie_inclusive"):
        p = e["pcie_inclusive"]
        add("PCIe-inclusive step (dictionary from pageable host memory)", f"{k(e['pcie_inclusive_patterns_per_s'])} patterns/s, "
            f"{p['achieved_GBps']:.0f} GB/s of {p['peak_GBps']:.0f}", "`extra.pcie_inclusive`")
    ps = e.get("plugin_seam")
    if ps:
        a, b = ps.get("n_per_iteration_3044"), ps.get("n_per_iteration_25000")
        if a and b:
            add("the drop-in seam: an unmodified kikuchipy's loop around the metric plugin, host dictionary",
                f"{k(a['patterns_per_s'])} patterns/s at `n_per_iteration=3044` ({k(a['without_lookahead']['patterns_per_s'])} without "
                f"look-ahead), {k(b['patterns_per_s'])} at 25 000", "`extra.plugin_seam`")
    ch = e.get("chunked_call")
    if ch:
        add("configs[1] pushed as 33 resident chunks of 3044", f"{ch['ms_per_call_one_gpu']:.2f} ms (one pass {ch['ms_per_step_single_pass']:.2f}); one member "
            f"of a group of 8: {ch['group_member_over_even_share']:.3f} × its even share", "`extra.chunked_call`")
    for key, name in (("split_f16_mode", "opt-in `compute=\"f16x2\"`"), ("f16_mode", "opt-in `compute=\"f16\"` (reduced precision)")):
        if key in e:
            add(name, f"{k(e[key]['patterns_per_s'], 0)} patterns/s, max score difference to f32 {e[key]['max_abs_score_diff_vs_f32']:.1e}", f"`extra.{key}`")
    if "float64_mode" in e:
        f = e["float64_mode"]
        add("`dtype=float64` (f32 screen + float64 rescoring, certified)", f"{k(f['patterns_per_s'])} patterns/s, {f['uncertified_patterns']} uncertified",
            "`extra.float64_mode`")
    if "dictionary_generation" in e:
        g = e["dictionary_generation"]
        add("dictionary simulated on the device inside the step (f1)", f"{k(g['patterns_per_s_including_generation'])} patterns/s, projection "
            f"{g['project_ms_per_step']:.2f} ms", "`extra.dictionary_generation`")
    if "refinement" in e:
        add("orientation refinement of 4096 patterns (f2)", f"{k(e['refinement']['patterns_per_s'], 0)} patterns/s", "`extra.refinement`")
    return out


def block(d, source, others, short):
    table = rows(d)
    if short:
        keep = ("configs[1]", "match kernel", "fabric", "CPU beside", "configs[2]", "structured", "RISING", "N = 4", "N = 8", "whole-node", "f16 MFMA",
                "HOST memory", "drop-in seam")
        table = [t for t in table if any(s in t[0] for s in keep)]
    lines = [f"Source: `{source[0]}` ({source[1]}; `python bench.py --gpus 1 --steps {d['steps']} --warmup {d['warmup']}` on one MI355X"
             + (f"; also on file: {', '.join('`' + o + '`' for o in others)}" if others else "") + ").", "",
             "| what | measured | key of the JSON line |", "|---|---|---|"]
    lines += [f"| {a} | {b} | {c} |" for a, b, c in table]
    chk = d.get("check", {})
    if chk:
        lines += ["", f"The timed result was checked against the float64 C oracle on {chk.get('rows')} rows before the line was printed "
                      f"(max |Δscore| {chk.get('max_abs_score_diff', float('nan')):.1e}, index agreement {chk.get('index_agreement')})."]
    return "\n".join(lines)


def main():
    cands = sys.argv[1:] or [os.path.join(ROOT, "BENCH_r06.json"), os.path.join(ROOT, "profiles", "r06_bench_n1.json")]
    loaded = []
    for p in cands:
        if os.path.exists(p):
            try:
                d, kind = load(p)
                if "roofline" in d and "extra" in d:
                    loaded.append((os.path.relpath(p, ROOT), kind, d))
            except Exception as err:  # noqa: BLE001
                print("skipping", p, err, file=sys.stderr)
    if not loaded:
        sys.exit("no bench line found among " + ", ".join(cands))
    path, kind, d = loaded[0]
    others = [p for p, _, _ in loaded[1:]]
    for doc, short in (("README.md", True), ("DESIGN.md", False)):
        fn = os.path.join(ROOT, doc)
        text = open(fn).read()
        new = re.sub(r"(<!-- measurements:begin[^>]*-->\n).*?(<!-- measurements:end -->)",
                     lambda m: m.group(1) + block(d, (path, kind), others, short) + "\n" + m.group(2), text, flags=re.S)
        if new == text and "measurements:begin" not in text:
            print(doc, "has no measurement block", file=sys.stderr)
        open(fn, "w").write(new)
        print("wrote", doc, "from", path)


if __name__ == "__main__":
    main()
