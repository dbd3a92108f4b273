"""A call written for kikuchipy binds the same way here: every mirrored public function / method leads with the
reference's parameters, in the reference's order, with the reference's literal defaults; what this package adds
(`devices`, `comm`, `verbose`, ...) is keyword-only.  The table is data taken from the reference's signatures by
oracle/gen_api_signatures.py (tests/golden/api_signatures.json)."""

import functools
import inspect
import json
import os

import numpy as np
import pytest

import kikuchipy_amd as kpa
from conftest import GOLDEN

TABLE = json.load(open(os.path.join(GOLDEN, "api_signatures.json")))

# Defaults that differ on purpose: the reference's `pseudo_symmetry_checked=False` has to be set to True by hand when the
# deferred refinement looked at pseudo-symmetry operators (indexing/_refinement/_refinement.py:53-120); here the
# deferred result remembers that, None = "as it was run", and a value that contradicts it raises.
LENIENT_DEFAULTS = {("compute_refine_orientation_results", "pseudo_symmetry_checked"): None,
                    ("compute_refine_orientation_projection_center_results", "pseudo_symmetry_checked"): None}


def same(a, b):
    return list(a) == list(b) if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)) else a == b


def resolve(path):
    import kikuchipy_amd.filters  # noqa: F401
    import kikuchipy_amd.indexing  # noqa: F401

    return functools.reduce(getattr, path.split("."), kpa)


@pytest.mark.parametrize("name", sorted(TABLE))
def test_reference_calls_bind_the_same_way(name):
    want = TABLE[name]
    sig = inspect.signature(resolve(want["ours"]))
    params = list(sig.parameters.values())
    positional = [p.name for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    # positional parameters: the reference's, in its order, and no more (an extra one would shift nothing, but a
    # positional call of the reference must never reach a parameter the reference does not have)
    assert positional == want["positional"], (name, positional, want["positional"])
    by_name = {p.name: p for p in params}
    for kw in want["keyword_only"]:
        assert kw in by_name, (name, kw)
    for arg, default in want["defaults"].items():
        default = LENIENT_DEFAULTS.get((name, arg), default)
        assert same(by_name[arg].default, default), (name, arg, by_name[arg].default, default)
    # what this package adds can only be given by keyword, and never has to be
    for p in params:
        if p.name not in want["positional"] and p.name not in want["keyword_only"]:
            if p.kind == p.VAR_KEYWORD and want["var_keyword"]:
                continue
            assert p.kind == p.KEYWORD_ONLY and p.default is not p.empty, (name, p.name)


def test_lazy_output_needs_inplace_false():
    """signals/ebsd.py:518-519, :645-646 - raised before anything touches a device."""
    s = kpa.EBSD(np.zeros((2, 3, 3), np.uint8), static_background=np.ones((3, 3), np.uint8))
    for call in (s.remove_static_background, s.remove_dynamic_background):
        with pytest.raises(ValueError, match="'lazy_output=True' requires 'inplace=False'"):
            call(lazy_output=True)
    # a positional call as the reference reads it: (operation, static_bg, scale_bg, show_progressbar, inplace, lazy_output)
    with pytest.raises(ValueError, match="'lazy_output=True' requires 'inplace=False'"):
        s.remove_static_background("subtract", None, False, False, True, True)
    with pytest.raises(ValueError, match="'lazy_output=True' requires 'inplace=False'"):
        s.remove_dynamic_background("subtract", "frequency", None, 4.0, False, True, True)


def test_load_takes_lazy_second_like_the_reference():
    with pytest.raises(TypeError, match="scan_group_names"):
        kpa.load("nothing.h5", "Scan 1")
