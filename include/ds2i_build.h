/* ds2i_build.h -- host-side (CPU) index construction entry points of libds2i_hip.so.
 *
 * These replace the build-side tools the benchmark loop needs because the reference cannot be
 * compiled here (create_freq_index.cpp:45-110, create_wand_data.cpp:8-29,
 * block_freq_index::builder block_freq_index.hpp:18-70). They produce the reference's on-disk
 * images (block_freq_index / wand_data) that ds2i_hip_index_open consumes. Nothing here is on
 * the timed query path and nothing here touches the GPU.
 */
#ifndef DS2I_BUILD_H
#define DS2I_BUILD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ds2i_builder ds2i_builder;
typedef struct ds2i_blob ds2i_blob; /* owned byte buffer returned by the builders */

/* Synthetic Zipf collection; every list is a pure function of (seed, term). */
typedef struct ds2i_synth_params {
    uint64_t seed;
    uint32_t num_docs;
    uint32_t num_terms;
    double zipf_exp;          /* list length(rank r) = max(min_len, top_df_frac*N*r^-zipf_exp) */
    double top_df_frac;
    uint32_t min_len;
    uint32_t clustered_every; /* every k-th list alternates dense/sparse segments; 0 = never */
    /* correlated terms (both 0: independent lists): documents come in runs of 4096 doc-ids of one of `topics` topics, every
     * term has a home topic and is topic_boost times as likely in its documents (list lengths unchanged) */
    uint32_t topics;
    uint32_t topic_boost;
} ds2i_synth_params;

const uint8_t* ds2i_blob_data(const ds2i_blob* b);
size_t ds2i_blob_size(const ds2i_blob* b);
void ds2i_blob_free(ds2i_blob* b);

/* block_freq_index<Codec>::builder (block_freq_index.hpp:18-70) for kinds 0..4, freq_index<...>::builder
 * (freq_index.hpp:19-94) for kinds 5..8; codec = enum ds2i_hip_index_kind */
int ds2i_builder_create(int codec, uint64_t num_docs, ds2i_builder** out);
int ds2i_builder_add_posting_list(ds2i_builder* b, uint64_t n, const uint32_t* docs, const uint32_t* freqs);
int ds2i_builder_freeze(ds2i_builder* b, ds2i_blob** image); /* succinct::mapper::freeze image */
void ds2i_builder_free(ds2i_builder* b);

/* wand_data<bm25> (wand_data.hpp:20-52): sizes = doc lengths; lists are streamed in term order */
typedef struct ds2i_wand_builder ds2i_wand_builder;
int ds2i_wand_create(const uint32_t* doc_sizes, uint64_t num_docs, ds2i_wand_builder** out);
int ds2i_wand_add_list(ds2i_wand_builder* w, uint64_t n, const uint32_t* docs, const uint32_t* freqs);
int ds2i_wand_freeze(ds2i_wand_builder* w, ds2i_blob** image);
void ds2i_wand_free(ds2i_wand_builder* w);

/* single block encoders (test hooks for the codec round trips of test_block_codecs.cpp:9-46).
 * sum_of_values == 0xFFFFFFFF means "unknown" like the reference. Returns a blob. */
int ds2i_encode_block(int codec, const uint32_t* values, uint32_t sum_of_values, uint32_t n, ds2i_blob** out);
int ds2i_encode_vbyte(uint32_t value, ds2i_blob** out);
/* bm25::query_term_weight / doc_term_weight (bm25.hpp:11-24), element-wise, as the host side of the query path
 * computes them (float32) */
int ds2i_bm25_query_term_weight(const uint64_t* qtf, const uint64_t* df, uint64_t num_docs, uint64_t n, float* out);
int ds2i_bm25_doc_term_weight(const uint64_t* freq, const float* norm_len, uint64_t n, float* out);
int ds2i_encode_posting_list(int codec, uint32_t n, const uint32_t* docs, const uint32_t* freqs, ds2i_blob** out);

/* One sequence of the Elias-Fano family written on its own into a fresh bit string (test hook for the reference's
 * layout tests test_compact_elias_fano.cpp:45-80, test_compact_ranked_bitvector.cpp:36-68,
 * test_partitioned_sequence.cpp:13-111, test_uniform_partitioned_sequence.cpp). seq_kind: */
enum ds2i_sequence_kind {
    DS2I_SEQ_ELIAS_FANO = 0,       /* compact_elias_fano */
    DS2I_SEQ_RANKED_BITVECTOR = 1, /* compact_ranked_bitvector (strictly increasing values) */
    DS2I_SEQ_INDEXED = 2,          /* indexed_sequence: 1 type bit + the cheapest of EF / ranked bitvector / all-ones */
    DS2I_SEQ_STRICT = 3,           /* strict_sequence */
    DS2I_SEQ_PARTITIONED_INDEXED = 4, /* partitioned_sequence<indexed_sequence> */
    DS2I_SEQ_PARTITIONED_STRICT = 5,  /* partitioned_sequence<strict_sequence> */
    DS2I_SEQ_UNIFORM_INDEXED = 6,     /* uniform_partitioned_sequence<indexed_sequence> */
    DS2I_SEQ_UNIFORM_STRICT = 7       /* uniform_partitioned_sequence<strict_sequence> */
};
/* params = {ef_log_sampling0, ef_log_sampling1, rb_log_rank1_sampling, rb_log_sampling1, log_partition_size}
 * (global_parameters.hpp:5-31; NULL = defaults). bits = the image as little-endian u64 words, *nbits its length. */
int ds2i_write_sequence(int seq_kind, const uint64_t* values, uint64_t n, uint64_t universe, const uint8_t params[5],
                        ds2i_blob** bits, uint64_t* nbits);

/* The chunk directory ds2i_hip_index_open builds for one list of an opt image (inspection / test hook):
 * cmax = u32[nchunks] last doc-id per chunk, chunks = nchunks x 12 dwords (ds2i_amd/csrc/device_pef.hpp),
 * info = {n, docs_bit0, freqs_bit0, docs bit-vector byte offset in the image, freqs bit-vector byte offset} */
int ds2i_opt_list_directory(const void* opt_image, size_t bytes, uint32_t term, ds2i_blob** cmax, ds2i_blob** chunks,
                            uint64_t info[5]);
/* same for any freq_index kind (DS2I_OPT / DS2I_EF / DS2I_SINGLE / DS2I_UNIFORM) */
int ds2i_freq_list_directory(int kind, const void* image, size_t bytes, uint32_t term, ds2i_blob** cmax,
                             ds2i_blob** chunks, uint64_t info[5]);

/* block_mixed space/time optimiser (reference optimal_hybrid_index.cpp + mixed_block.hpp:119-150).
 * For every block: OptPFor with each usable b, VarInt-G8IU, interpolative -> (bytes, model time x (access + 1));
 * globally: minimise the summed time subject to summed payload bytes <= budget (lower convex hull per block +
 * one Lagrange multiplier). The time model is the MI355X kernels' instruction cost per block (defaults measured with
 * rocprofv3), the access counts come from ds2i_hip_batch_block_profile (2 per block, list by list) or NULL. */
typedef struct ds2i_hybrid ds2i_hybrid;
typedef struct ds2i_hybrid_model {
    float pfor_base, pfor_exc, pfor_exc_many, varint, interp_base, interp_node;
} ds2i_hybrid_model;
void ds2i_hybrid_default_model(ds2i_hybrid_model* m);
int ds2i_hybrid_create(uint64_t num_docs, const ds2i_hybrid_model* model /* NULL = default */, ds2i_hybrid** out);
int ds2i_hybrid_add_posting_list(ds2i_hybrid* h, uint64_t n, const uint32_t* docs, const uint32_t* freqs,
                                 const uint32_t* access /* 2 * ceil(n/128) counters or NULL */);
/* computes every block's candidates; min_space / max_space = payload bytes of the smallest / fastest index */
int ds2i_hybrid_analyse(ds2i_hybrid* h, int threads, uint64_t* min_space, uint64_t* max_space);
/* encodes the block_mixed image for the budget (payload bytes; pass UINT64_MAX for the fastest index).
 * rate = bytes spent per unit of model time saved at the optimum, type_counts = full blocks by
 * {docs pfor, docs varint, docs interpolative, freqs pfor, freqs varint, freqs interpolative} */
int ds2i_hybrid_freeze(ds2i_hybrid* h, uint64_t budget_bytes, int threads, ds2i_blob** image, double* rate,
                       uint64_t* space, double* model_time, uint64_t type_counts[6]);
void ds2i_hybrid_free(ds2i_hybrid* h);

/* synthetic collection */
uint64_t ds2i_synth_list_upper_bound(const ds2i_synth_params* p, uint32_t term);
int ds2i_synth_list(const ds2i_synth_params* p, uint32_t term, uint32_t* docs, uint32_t* freqs, uint64_t capacity,
                    uint64_t* n);
int ds2i_synth_doc_sizes(const ds2i_synth_params* p, uint32_t* sizes);
/* query log: terms gets at most 11*nq entries, offsets nq+1 */
int ds2i_synth_queries(uint64_t seed, uint32_t num_terms, uint32_t nq, uint32_t* terms, uint32_t* offsets);
/* the same log with same_topic_pct percent of the multi-term queries drawn from one topic of a correlated collection */
int ds2i_synth_queries_topical(const ds2i_synth_params* p, uint64_t seed, uint32_t nq, uint32_t same_topic_pct, uint32_t* terms, uint32_t* offsets);
/* generate + encode the whole collection with `threads` host threads: index image + wand image */
int ds2i_synth_build(const ds2i_synth_params* p, int codec, int threads, ds2i_blob** index_image,
                     ds2i_blob** wand_image, uint64_t* total_postings);

/* the synthetic collection encoded by the block_mixed optimiser (lists regenerated for each pass):
 * payload budget = smallest + budget_frac * (fastest - smallest); access = 2 counters per block or NULL (uniform) */
int ds2i_synth_build_hybrid(const ds2i_synth_params* p, int threads, const ds2i_hybrid_model* model, const uint32_t* access,
                            double budget_frac, ds2i_blob** index_image, ds2i_blob** wand_image, uint64_t* total_postings,
                            uint64_t type_counts[6]);

#ifdef __cplusplus
}
#endif
#endif
