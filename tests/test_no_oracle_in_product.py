"""The product path must not route through the oracle (or any CPU fallback)."""

import os
import re

from conftest import ROOT


def product_files():
    for base, _, files in os.walk(os.path.join(ROOT, "kikuchipy_amd")):
        if "build" in base.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                yield os.path.join(base, f)


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|kpdi_oracle|oracle/", re.M)
    offenders = [p for p in product_files() if pat.search(open(p).read())]
    assert not offenders, offenders


def test_product_never_imports_torch_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+(kikuchipy\b(?!_amd)|numba|dask)", re.M)
    for p in product_files():
        text = open(p).read()
        assert not pat.search(text), p
        # no PyTorch anywhere in the product - the control plane (parallel.py) is plain TCP
        assert not re.search(r"^\s*(from|import)\s+torch\b", text, re.M), p


def test_bench_and_entry_never_import_torch():
    for name in ("bench.py", "__graft_entry__.py"):
        text = open(os.path.join(ROOT, name)).read()
        assert not re.search(r"^\s*(from|import)\s+torch\b", text, re.M), name
