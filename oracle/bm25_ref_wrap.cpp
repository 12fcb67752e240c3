// C wrapper around the REFERENCE's own BM25 scorer, compiled from the source where it lies
// (/root/reference/bm25.hpp -- self-contained apart from the standard headers below, which its includer
// normally provides). Output goes to oracle/_ref/libbm25_ref.so (git-ignored). Test infrastructure only: pins the
// float32 half of the path -- oracle.cpp's bm25, the product's host query_term_weight and the device
// doc_term_weight -- through tests/golden/bm25_reference.json. No reference source is copied into this repo.
#include <stdint.h>

#include <algorithm>
#include REFERENCE_BM25_HEADER

extern "C" float ref_bm25_doc_term_weight(uint64_t freq, float norm_len) { return ds2i::bm25::doc_term_weight(freq, norm_len); }
extern "C" float ref_bm25_query_term_weight(uint64_t freq, uint64_t df, uint64_t num_docs) {
    return ds2i::bm25::query_term_weight(freq, df, num_docs);
}
