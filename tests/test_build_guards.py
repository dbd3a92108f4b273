"""Guards on the GENERATED code of the match kernels (no GPU needed: hipcc cross-compiles).

Their MFMAs are inline asm, invisible to the compiler's hazard recogniser; a spill, reload or copy of an
accumulator placed next to one would read the register before the matrix pipe has written it (this happened:
`match16_kernel<32, true, 8>` once kept one accumulator in scratch and returned wrong scores for keep_n > 32).
tools/check_mfma_loops.py compiles the kernels and fails if any instantiation has scratch traffic between its
first and last MFMA."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("source", ["match.hip", "match16.hip"])
def test_no_accumulator_goes_through_scratch_inside_the_mfma_loop(source):
    import check_mfma_loops

    bad, checked = check_mfma_loops.kernels_with_spills_in_mfma_loop(source)
    assert checked >= 10, f"only {checked} kernels found in {source}: has the check lost track of the assembly?"
    assert not bad, f"scratch traffic inside the MFMA loop of: {sorted(bad)}"
