"""kikuchipy_amd - MI355X-native dictionary indexing for EBSD patterns.

Accelerates ONE path of kikuchipy: `EBSD.dictionary_indexing()` with the
`ncc`/`ndp` similarity metrics, plus the static/dynamic background removal
that feeds it, and the on-device generation of the dictionary from a master
pattern (`EBSDMasterPattern.get_patterns`).  Python host code -> ctypes -> libkpdi.so (hand-written HIP for
gfx950).  No PyTorch, no CPU fallback.
"""

__version__ = "0.2.0"

from kikuchipy_amd.indexing import (  # noqa: E402,F401
    DictionaryIndexingResult,
    NormalizedCrossCorrelationMetric,
    NormalizedDotProductMetric,
    RefinementResult,
    ResidentDictionary,
    SimilarityMetric,
    dictionary_indexing,
    merge_crystal_maps,
    orientation_similarity_map,
)
from kikuchipy_amd.pattern import remove_dynamic_background, remove_static_background  # noqa: E402,F401
from kikuchipy_amd.detectors import EBSDDetector  # noqa: E402,F401
from kikuchipy_amd.signals import EBSD, DictionaryXmap, EBSDMasterPattern  # noqa: E402,F401
from kikuchipy_amd.simulations import ProjectedDictionary  # noqa: E402,F401
from kikuchipy_amd.io import load  # noqa: E402,F401
from kikuchipy_amd import filters  # noqa: E402,F401
from kikuchipy_amd.sampling import get_sample_fundamental  # noqa: E402,F401

from kikuchipy_amd._lib import clear_engine_cache  # noqa: E402,F401

__all__ = [
    "DictionaryIndexingResult",
    "DictionaryXmap",
    "EBSD",
    "EBSDDetector",
    "EBSDMasterPattern",
    "ProjectedDictionary",
    "RefinementResult",
    "ResidentDictionary",
    "NormalizedCrossCorrelationMetric",
    "NormalizedDotProductMetric",
    "SimilarityMetric",
    "clear_engine_cache",
    "dictionary_indexing",
    "filters",
    "get_sample_fundamental",
    "load",
    "merge_crystal_maps",
    "orientation_similarity_map",
    "remove_dynamic_background",
    "remove_static_background",
]
