"""The last partial round of a sweep as a kernel of its own (csrc/tailgemm.hip: 32 x 128 workgroups, scores to a small
matrix, a select pass against the final shared bound) - against the oracle and BIT FOR BIT against the two other ways the
same rows can be swept (partial units inside match16.hip, KPDI_TAIL_GEMM=0; match.hip, KPDI_F32_WIDE=0).

Reference semantics: `match` + `argtopk` / `topk` of a chunk (indexing/_dictionary_indexing.py:193-203), merged over
chunks (:120-128)."""
import numpy as np
import pytest

from oracle import kpdi_oracle as ko

pytestmark = pytest.mark.gpu


def sweep(monkeypatch, env, exp, dic, metric, keep_n, chunk=None, signal_mask=None):
    from kikuchipy_amd import _lib

    for k in ("KPDI_F32_WIDE", "KPDI_TAIL_GEMM"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    code = {"ncc": _lib.METRIC_NCC, "ndp": _lib.METRIC_NDP}[metric]
    with _lib.Context(0) as c:
        c.set_problem(exp.shape[1], exp.shape[2], signal_mask, code, keep_n)
        c.set_experimental(exp)
        step = chunk or len(dic)
        for a in range(0, len(dic), step):
            c.push_dictionary_chunk(dic[a:a + step], a)
        s, i = c.finalize(keep_n)
        return s, i, c.counters()


FORMS = {"tail kernel": {"KPDI_F32_WIDE": "1", "KPDI_TAIL_GEMM": "1"},
         "partial units": {"KPDI_F32_WIDE": "1", "KPDI_TAIL_GEMM": "0"},
         "match.hip": {"KPDI_F32_WIDE": "0"}}


@pytest.mark.parametrize("m,n,chunk,metric,masked,keep_n", [
    (300, 4700, None, "ncc", False, 20),     # 19 tiles of 256: whole rounds + a few left over
    (300, 4700, 1900, "ndp", False, 8),      # three chunks, each with a tail of its own (the bound persists across them)
    (1000, 2100, None, "ncc", True, 20),     # signal mask: a shorter reduction
    (257, 600, None, "ncc", False, 1),       # keep_n = 1: list length 1
    (4096, 12500, None, "ncc", False, 20),   # one rank's share of configs[1] at N = 8 (24 x 20 pixels here)
    (40, 5000, None, "ndp", False, 32),      # one row block: 20 tiles over up to 256 splits
    (4096, 26000, 9000, "ncc", False, 20),   # 16 row blocks x 16 splits = 64 lists per pattern, three chunks: the merge takes
                                             # the lists' counts from its lanes, beside the running best-k of the chunks before
])
def test_tail_kernel_agrees_with_the_oracle_and_with_the_other_kernels(monkeypatch, m, n, chunk, metric, masked, keep_n):
    rng = np.random.default_rng(m + n)
    sy, sx = 24, 20
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    mask = None
    if masked:
        mask = np.zeros((sy, sx), dtype=bool)
        mask[:4] = True
        mask[:, :3] = True
    out = {name: sweep(monkeypatch, env, exp, dic, metric, keep_n, chunk, mask) for name, env in FORMS.items()}
    assert out["tail kernel"][2]["match_form"] == 3 and out["match.hip"][2]["match_form"] == 0
    for name in ("partial units", "match.hip"):
        assert np.array_equal(out["tail kernel"][0], out[name][0]), name
        assert np.array_equal(out["tail kernel"][1], out[name][1]), name
    rows = np.arange(m) if m <= 300 else np.sort(rng.choice(m, 64, replace=False))
    rs, ri = ko.dictionary_indexing(exp[rows], dic, metric=metric, keep_n=keep_n, n_per_iteration=chunk, signal_mask=mask)
    ko.assert_topk_parity(out["tail kernel"][0][rows], out["tail kernel"][1][rows], rs, ri, atol=1e-5)


def test_the_tail_holds_every_best_match_ties_included(monkeypatch):
    """All the best matches sit in the rows the tail kernel takes - more of them than its candidate arrays hold (the
    exact slow path) - and some are bit-identical copies (ties: lower dictionary index first)."""
    rng = np.random.default_rng(7)
    sy, sx, m, n = 24, 20, 300, 4700
    exp = rng.integers(0, 256, (m, sy, sx)).astype(np.uint8)
    dic = rng.random((n, sy, sx)).astype(np.float32)
    tail0 = 4608                                  # 18 whole tiles of 256; 92 rows behind them
    noise = rng.random((n - tail0, sy, sx)).astype(np.float32)
    dic[tail0:] = exp[0].astype(np.float32) / 255.0 + 0.02 * noise     # 92 near-copies of pattern 0: all of them beat the bound
    dic[tail0 + 5] = dic[tail0 + 40] = dic[tail0 + 77] = dic[tail0 + 3]  # exact ties among them
    dic[100] = dic[tail0 + 3]                                           # ... and one in the main rounds
    dic[tail0 + 60:tail0 + 70] = exp[1].astype(np.float32) / 255.0     # ten identical copies of pattern 1
    for keep_n in (20, 32):
        out = {name: sweep(monkeypatch, env, exp, dic, "ncc", keep_n) for name, env in FORMS.items()}
        for name in ("partial units", "match.hip"):
            assert np.array_equal(out["tail kernel"][0], out[name][0]) and np.array_equal(out["tail kernel"][1], out[name][1]), name
        s, i = out["tail kernel"][:2]
        rs, ri = ko.dictionary_indexing(exp, dic, keep_n=keep_n)
        ko.assert_topk_parity(s, i, rs, ri, atol=1e-5)
        assert (i[0] >= tail0).sum() >= keep_n - 1 and 100 in i[0]
        tied = np.flatnonzero(s[0] == s[0][list(i[0]).index(100)])
        assert list(i[0][tied]) == sorted(i[0][tied]) and 100 == i[0][tied][0]
        assert list(i[1][:10]) == list(range(tail0 + 60, tail0 + 70)) and np.all(s[1][:10] == s[1][0])
