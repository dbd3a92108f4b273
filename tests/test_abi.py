"""The C-ABI library loads and exports every symbol include/kpdi.h declares
(no compute calls: this runs without a GPU)."""

import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "kpdi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kpdi_[a-z0-9_]+)\s*\(", text)))


def test_header_lists_functions():
    names = header_functions()
    assert "kpdi_push_dictionary_chunk" in names and "kpdi_finalize" in names
    assert len(names) >= 25


def test_library_exports_every_declared_symbol():
    from kikuchipy_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), f"{name} declared in include/kpdi.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in kikuchipy_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_version_and_no_cpu_fallback():
    from kikuchipy_amd import _lib

    assert "gfx950" in _lib.version()
    if _lib.device_count() == 0:
        with pytest.raises(_lib.KpdiError, match="no CPU fallback"):
            _lib.Context(0)


def test_counters_struct_matches_header():
    """Field order/types of kpdi_counters in the header == the ctypes Structure."""
    from kikuchipy_amd import _lib

    text = open(os.path.join(ROOT, "include", "kpdi.h")).read()
    body = re.search(r"typedef struct kpdi_counters \{(.*?)\} kpdi_counters;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(double|int64_t|int32_t)\s+(\w+);", body)
    ctype = {"double": ctypes.c_double, "int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32}
    assert [(n, ctype[t]) for t, n in fields] == list(_lib.Counters._fields_)
