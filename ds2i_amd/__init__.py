"""ds2i_amd -- MI355X-native batched query path for ds2i block indexes.

The package is a thin ctypes mirror of the C ABI in include/ds2i_hip.h and
include/ds2i_build.h (libds2i_hip.so, built in-tree by ds2i_amd/build.py).
All query results come from the HIP kernels; there is no CPU fallback.
"""
import os as _os

# see capi.cpp (ds2i_hip_more_hw_queues): must be in the environment before the HIP runtime initialises
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .api import (  # noqa: F401
    CODECS, BLOCK_CODECS, FREQ_INDEX_KINDS, OPS, REFERENCE_ORDER, NO_COUNTERS, Ds2iError, Index, Batch, Pipeline, flatten_queries, lib, library_path,
    encode_block, encode_vbyte, encode_posting_list, build_index, build_wand, gpu_encode_index, synth_build_gpu,
    SynthParams, HybridBuilder, HybridModel, opt_list_directory, synth_list, synth_doc_sizes, set_option, synth_queries, synth_queries_topical, synth_build, synth_build_hybrid,
    and_query, or_query, ranked_and_query, wand_query, maxscore_query, ranked_or_query,
)
