from kikuchipy_amd.indexing._dictionary_indexing import (  # noqa: F401
    DictionaryIndexingResult,
    dictionary_indexing,
)
from kikuchipy_amd.indexing.similarity_metrics import (  # noqa: F401
    NormalizedCrossCorrelationMetric,
    NormalizedDotProductMetric,
    SimilarityMetric,
)
from kikuchipy_amd.indexing._refinement import (  # noqa: F401
    DeferredRefinement,
    RefinementResult,
    compute_refine_orientation_projection_center_results,
    compute_refine_orientation_results,
    compute_refine_projection_center_results,
    refine,
)
from kikuchipy_amd.indexing._merge_crystal_maps import MergedIndexingResult, merge_crystal_maps  # noqa: F401
from kikuchipy_amd.indexing._orientation_similarity_map import orientation_similarity_map  # noqa: F401
from kikuchipy_amd.indexing._resident_dictionary import ResidentDictionary  # noqa: F401
