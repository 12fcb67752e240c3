/* ds2i_hip.h -- C ABI of libds2i_hip.so: the MI355X-native batched query path for ds2i indexes.
 *
 * ds2i has no FFI; its extension points are two C++ template concepts:
 *   Index concept     size(), num_docs(), operator[](term) -> document_enumerator
 *                     (reference block_freq_index.hpp:72-94, block_posting_list.hpp:84-186)
 *   Query-op concept  uint64_t operator()(Index const&, term_id_vec) [+ topk()]
 *                     (reference queries.hpp:35-591, driven by queries.cpp:13-62)
 * This header is what a binding for those two concepts would call. The index image and the
 * wand image are the reference's own on-disk files, unchanged (block_freq_index.hpp:124-134,
 * wand_data.hpp:71-78). Everything is plain C: no exceptions cross the boundary; every
 * function returns 0 on success or a negative DS2I_E* code, ds2i_hip_last_error() explains.
 * A handle is bound to one device and is not re-entrant; distinct handles may be used
 * concurrently from distinct host threads (index immutable, like profile_queries.cpp:25-37).
 */
#ifndef DS2I_HIP_H
#define DS2I_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* index kinds == reference index_types.hpp:18-39 */
enum ds2i_hip_index_kind {
    DS2I_BLOCK_OPTPFOR = 0,
    DS2I_BLOCK_VARINT = 1,
    DS2I_BLOCK_INTERPOLATIVE = 2,
    DS2I_BLOCK_QMX = 3,
    DS2I_BLOCK_MIXED = 4,
    DS2I_OPT = 5,     /* opt_index: freq_index<partitioned_sequence<>, positive_sequence<partitioned_sequence<strict_sequence>>>
                         (index_types.hpp:29-32) */
    DS2I_EF = 6,      /* ef_index: freq_index<compact_elias_fano, positive_sequence<strict_elias_fano>> (index_types.hpp:18-19) */
    DS2I_SINGLE = 7,  /* single_index: freq_index<indexed_sequence, positive_sequence<>> (index_types.hpp:21-22) */
    DS2I_UNIFORM = 8  /* uniform_index: freq_index<uniform_partitioned_sequence<>,
                         positive_sequence<uniform_partitioned_sequence<strict_sequence>>> (index_types.hpp:24-27) */
};

/* query operators == the strings queries.cpp:104-117 dispatches on (+ ranked_or, queries.hpp:404) */
enum ds2i_hip_op {
    DS2I_OP_AND = 0,        /* and_query<false>   queries.hpp:35-86   */
    DS2I_OP_AND_FREQ = 1,   /* and_query<true>                        */
    DS2I_OP_OR = 2,         /* or_query<false>    queries.hpp:88-131  */
    DS2I_OP_OR_FREQ = 3,    /* or_query<true>                         */
    DS2I_OP_RANKED_AND = 4, /* ranked_and_query   queries.hpp:322-401 */
    DS2I_OP_WAND = 5,       /* wand_query         queries.hpp:200-319 */
    DS2I_OP_MAXSCORE = 6,   /* maxscore_query     queries.hpp:478-591 */
    DS2I_OP_RANKED_OR = 7,  /* ranked_or_query    queries.hpp:404-476 */
    /* OR this flag into any operator to run the reference's
       one-document-per-step traversal on the GPU instead of the block-synchronous kernel (same results; used to
       cross-check and to count the reference traversal's algorithmic bytes on the device). */
    DS2I_OP_REFERENCE_ORDER = 0x100,
    /* OR this flag into the operator of ds2i_hip_query_batch / ds2i_hip_pipeline_submit to run the kernels compiled
       without the statistics counters even though a stats struct is passed (it then carries kernel_ms only). */
    DS2I_OP_NO_COUNTERS = 0x200
};

enum ds2i_hip_error {
    DS2I_OK = 0,
    DS2I_EINVAL = -1,   /* bad argument (null pointer, unknown kind/op, k out of range) */
    DS2I_EFORMAT = -2,  /* index / wand image failed validation */
    DS2I_ETERM = -3,    /* term id >= index size (UB + assert in the reference, block_freq_index.hpp:87) */
    DS2I_EDEVICE = -4,  /* HIP runtime error or no usable device */
    DS2I_ENOWAND = -5,  /* ranked operator without wand data (queries.cpp:108-116 logs "Unsupported") */
    DS2I_ETOOLONG = -6, /* query has more than DS2I_HIP_MAX_TERMS_LONG distinct terms */
    DS2I_ENOMEM = -7,
    DS2I_EBUSY = -8     /* pipeline: every slot is in flight -- collect the oldest ticket first */
};

#define DS2I_HIP_MAX_TERMS 16        /* distinct terms per query held in LDS by one wavefront (the fast kernels) */
#define DS2I_HIP_MAX_TERMS_LONG 1024 /* beyond 16 distinct terms a query runs the one-document-per-step traversal with
                                        its enumerator state in global memory (the reference has no limit, queries.hpp:35-86) */
#define DS2I_HIP_MAX_K 64            /* top-k kept one score per lane (the fast kernels) */
#define DS2I_HIP_MAX_K_LONG 1024     /* beyond 64 the same stream kernels run with 4 (k <= 256) or 16 scores per lane; only a
                                        batch that also holds a query of more than 16 terms, or a natively queried block_mixed
                                        image, falls to the one-document-per-step kernel (the reference's topk_queue has no
                                        limit, queries.hpp:152-197) */

typedef struct ds2i_hip_index ds2i_hip_index;
typedef struct ds2i_hip_batch ds2i_hip_batch;

/* Device-side counters of one batch run; the byte pricing is SURVEY.md §8(d)'s A_skip. */
typedef struct ds2i_hip_stats {
    double kernel_ms;             /* device-side WINDOW of the last run (hipEvents): from the batch's first clear to the
                                     completion of its last kernel. In a pipeline the window also holds whatever other
                                     batches ran on the GPU meanwhile; for wand / maxscore / ranked_or the ranked_and seed
                                     pass is enqueued ahead of the window and may start before it. rocprofv3's per-kernel
                                     durations are the exact figures. */
    uint64_t docs_blocks_decoded; /* block_profiler counter [2b]   (block_posting_list.hpp:316-318) */
    uint64_t freqs_blocks_decoded;/* block_profiler counter [2b+1] (block_posting_list.hpp:328-330) */
    uint64_t block_max_examined;
    uint64_t algorithmic_bytes;
    uint64_t postings_scored;
    uint64_t rounds;              /* block-synchronous rounds (conjunctive kernel) */
} ds2i_hip_stats;

int ds2i_hip_device_count(void);
/* Tuning knobs without the environment: name = one of the twenty documented DS2I_*
 * variables, value = what the variable would hold (NULL = unset). Process-wide like the variables themselves; every
 * ds2i_hip_index_open re-reads them, and they hold for that index and the batches planned until the next upload. Unknown names
 * fail with DS2I_EINVAL. */
int ds2i_hip_set_option(const char* name, const char* value);
const char* ds2i_hip_last_error(void);

/* Copies the index (and optional wand data) to the device's HBM. The images are not referenced
 * after the call returns. Replaces succinct::mapper::map of queries.cpp:76-77,90-95.
 * block_mixed and the freq_index layouts (opt / ef / single / uniform) are decoded once on the device and queried as
 * block_optpfor (same postings, same answers; ds2i_hip_index_info::transcoded_from says so); DS2I_MIXED_NATIVE=1 /
 * DS2I_PEF_NATIVE=1 in the environment (or ds2i_hip_set_option) keep the image and run its own kernels. */
int ds2i_hip_index_open(int device, int index_kind, const void* index_image, size_t index_bytes,
                        const void* wand_image, size_t wand_bytes, ds2i_hip_index** out);
void ds2i_hip_index_close(ds2i_hip_index* idx);

/* Index concept: size() = number of posting lists, num_docs() (block_freq_index.hpp:72-80) */
uint64_t ds2i_hip_index_size(const ds2i_hip_index* idx);
uint64_t ds2i_hip_index_num_docs(const ds2i_hip_index* idx);
uint64_t ds2i_hip_index_device_bytes(const ds2i_hip_index* idx);
/* What the upload put into HBM beside the index image. The pruning tables are accelerators: an upload that cannot
 * afford them (or was told not to build them) still succeeds and the query kernels take their slower table-free paths,
 * so a caller that cares checks has_range_tables here (the library also says so once on stderr). With
 * DS2I_RMW_REQUIRE=1 in the environment such an upload fails with DS2I_ENOMEM instead. */
typedef struct ds2i_hip_index_info {
    uint64_t index_bytes;        /* the posting lists (block indexes) / chunk directory + bit vectors (freq_index layouts) */
    uint64_t skip_table_bytes;   /* interleaved {block_max, end offset} rows (block indexes) */
    uint64_t block_weight_bytes; /* bmw[]: one float per block / chunk */
    uint64_t range_table_bytes;  /* doc-id-range tables, their two coarser levels, membership bit tables, dense bitmaps */
    uint64_t norm_len_bytes;
    uint64_t total_blocks;
    uint64_t total_postings;
    int has_block_weights;
    int has_range_tables;
    int has_bitmaps;
    int has_membership_hints;    /* one more byte per level-1 range-table entry (block_optpfor indexes): see ds2i_hip_list_range_table */
    int range_table_entries_per_posting; /* DS2I_RMW_G in effect */
    uint64_t side_table_bytes;   /* block_optpfor: exception side slots (256 B per block: the OptPFor exceptions of a block as position
                                  * masks + values, so that a decode never parses the Simple16 streams), their overflow area, and the
                                  * lists' partial last blocks in plain form; 0 = not built (DS2I_NO_XSLOTS, other index kinds, no room) */
    int has_side_tables;
    int transcoded_from;         /* the DS2I_* kind of the image the caller handed over when the upload transcoded it to block_optpfor
                                  * (block_mixed, opt / ef / single / uniform by default: DS2I_MIXED_NATIVE / DS2I_PEF_NATIVE), else -1 */
    uint64_t table_budget_bytes; /* DS2I_TABLE_BUDGET in effect at the upload (bytes; 0 = none): the upload built the first of
                                  * {tables at 4 / 2 / 1 entries per posting} x {hints} x {side slots} whose resident bytes fit --
                                  * range_table_entries_per_posting, has_membership_hints, has_side_tables say which */
} ds2i_hip_index_info;
int ds2i_hip_index_get_info(const ds2i_hip_index* idx, ds2i_hip_index_info* out);
/* document_enumerator::size() of index[term] (block_posting_list.hpp:178-181) */
int ds2i_hip_list_size(const ds2i_hip_index* idx, uint32_t term, uint64_t* n);
/* index[term] followed by a full enumeration docid()/freq()/next(): decodes every block of the
 * list on the GPU. docs/freqs hold `capacity` entries; *n receives the list length. */
int ds2i_hip_decode_list(ds2i_hip_index* idx, uint32_t term, uint32_t* docs, uint32_t* freqs, uint64_t capacity,
                         uint64_t* n);

/* One-shot batched query operator: for each query q (terms[query_offsets[q] .. query_offsets[q+1]))
 *   out_count[q]    the operator's return value (matches for and/or, top-k size for ranked ops)
 *   out_topk        nq*k floats, each query's scores descending, padded with -inf. Ranked operators only:
 *                   and / or produce no top-k, ignore k and never touch out_topk (may be NULL)
 *   out_topk_len    nq (may be NULL)
 * Host terms -> HBM, kernels, results -> host. The batch slot (pinned staging, device blocks, events) is cached in
 * the index handle: repeated calls allocate nothing. stats may be NULL; when given, the instrumented kernels run
 * (counters filled) unless DS2I_OP_NO_COUNTERS is set. A query may have any number of distinct terms up to
 * DS2I_HIP_MAX_TERMS_LONG (the reference has no limit); beyond DS2I_HIP_MAX_TERMS it takes the slower long path. */
int ds2i_hip_query_batch(ds2i_hip_index* idx, int op, uint32_t k, const uint32_t* terms,
                         const uint32_t* query_offsets, uint32_t nq, uint64_t* out_count, float* out_topk,
                         uint32_t* out_topk_len, ds2i_hip_stats* stats);

/* Split form: prepare() does query normalisation (query_freqs / remove_duplicate_terms,
 * queries.hpp:29-33,136-150), BM25 query weights, list-length ordering and uploads everything;
 * run() only launches kernels on data already resident in HBM (this is what bench.py times);
 * fetch() copies results back. want_matches != 0 additionally collects the doc-id lists of `and`. */
int ds2i_hip_batch_prepare(ds2i_hip_index* idx, int op, uint32_t k, const uint32_t* terms,
                           const uint32_t* query_offsets, uint32_t nq, int want_matches, ds2i_hip_batch** out);
int ds2i_hip_batch_run(ds2i_hip_batch* b, ds2i_hip_stats* stats);
/* Statistics are an instrumentation option like the reference's block_profiler (`template <bool Profile>`,
 * block_posting_list.hpp:316-318): with on = 0 the conjunctive operators on block_optpfor / opt run kernels compiled
 * without the counters (faster; ds2i_hip_stats then carries kernel_ms only and the per-class counters keep the values
 * of the last instrumented run). Default: on. */
int ds2i_hip_batch_set_instrumented(ds2i_hip_batch* b, int on);

/* Block access profile of a batch on a block index (the GPU-side profile_queries.cpp, reference
 * profile_queries.cpp:21-95): after enable, every instrumented run adds, for each 128-posting block of the index,
 * the number of docs-part and freqs-part decodes. counts receives 2 x total_blocks values, blocks numbered list by
 * list in index order (block b of list t at 2 * (sum_{u<t} ceil(n_u/128) + b)). Pass counts = NULL to query
 * total_blocks only. Input of the block_mixed optimiser (ds2i_hybrid_*, ds2i_build.h). */
int ds2i_hip_batch_enable_block_profile(ds2i_hip_batch* b);
int ds2i_hip_batch_block_profile(ds2i_hip_batch* b, uint32_t* counts, uint64_t capacity, uint64_t* total_blocks);
/* per kernel class of the last run (class 0: <=2 distinct terms, 1: 3..4, 2: 5..8, 3: 9..16 -- four
 * template instantiations with different LDS footprints -- 4: more than 16; launched concurrently on their own streams) */
int ds2i_hip_batch_class_stats(ds2i_hip_batch* b, int cls, ds2i_hip_stats* out, uint32_t* nqueries);
/* A class may run more than one kernel: ranked_and on a block_optpfor index launches the pipelined stream kernel once per
 * exact list count (2, 3, 4 terms) and the class kernel for what is left (one-term queries), back to back on the class
 * stream. Per launch group of class cls in the last run: the hipEvent duration on that stream, the list slots the kernel
 * was launched with, its units (= workgroups) and queries. out may be NULL to ask for *ngroups only. */
typedef struct ds2i_hip_group_stats {
    double kernel_ms;
    uint32_t lists, units, queries;
    int pipelined_stream; /* 1: k_ranked_stream<lists> (ranked_stream.hip); 0: the class kernel */
} ds2i_hip_group_stats;
int ds2i_hip_batch_class_groups(ds2i_hip_batch* b, int cls, ds2i_hip_group_stats* out, uint32_t capacity, uint32_t* ngroups);
/* diagnostic build only (-DDS2I_PHASE_TIMING): per-phase shader-cycle sums {total, docs decode, freqs
 * decode, block search, membership, scoring, top-k} of class cls; zeros otherwise */
int ds2i_hip_batch_phase_cycles(ds2i_hip_batch* b, int cls, uint64_t* out, int n);
int ds2i_hip_batch_fetch(ds2i_hip_batch* b, uint64_t* out_count, float* out_topk, uint32_t* out_topk_len,
                         uint64_t* out_freq_sum);
/* doc-id lists of `and` (want_matches): match_offsets has nq+1 entries; matches has match_offsets[nq] */
int ds2i_hip_batch_match_total(ds2i_hip_batch* b, uint64_t* total);
int ds2i_hip_batch_fetch_matches(ds2i_hip_batch* b, uint64_t* match_offsets, uint32_t* matches);
void ds2i_hip_batch_free(ds2i_hip_batch* b);

/* Pipelined form -- the serving loop. A pipeline owns `depth` reusable batch slots. submit() does the host half of
 * the operator (query normalisation, BM25 query weights, work-unit planning: queries.hpp:29-33,136-150,357-360),
 * then enqueues ONE async H2D copy, the kernels and ONE async D2H copy of the results and returns without waiting
 * for the device: the host plans batch i+1 while the kernels of batch i run, and the kernels of consecutive batches
 * overlap on the device. wait() blocks until the ticket's batch is complete and copies its results out (same
 * meaning as ds2i_hip_query_batch). Tickets must be collected before their slot comes round again (submit() returns
 * DS2I_EBUSY otherwise). This is the loop queries.cpp:25-35 becomes: every query is timed fresh, nothing is
 * pre-staged on the device. */
typedef struct ds2i_hip_pipeline ds2i_hip_pipeline;
int ds2i_hip_pipeline_create(ds2i_hip_index* idx, uint32_t depth, ds2i_hip_pipeline** out);
int ds2i_hip_pipeline_submit(ds2i_hip_pipeline* p, int op, uint32_t k, const uint32_t* terms,
                             const uint32_t* query_offsets, uint32_t nq, uint64_t* ticket);
int ds2i_hip_pipeline_wait(ds2i_hip_pipeline* p, uint64_t ticket, uint64_t* out_count, float* out_topk,
                           uint32_t* out_topk_len, ds2i_hip_stats* stats);
/* like ds2i_hip_batch_class_stats, for the ticket collected last */
int ds2i_hip_pipeline_class_stats(ds2i_hip_pipeline* p, int cls, ds2i_hip_stats* out, uint32_t* nqueries);
int ds2i_hip_pipeline_class_groups(ds2i_hip_pipeline* p, int cls, ds2i_hip_group_stats* out, uint32_t capacity, uint32_t* ngroups);
/* default off: pipelines run the kernels compiled without counters */
int ds2i_hip_pipeline_set_instrumented(ds2i_hip_pipeline* p, int on);
void ds2i_hip_pipeline_destroy(ds2i_hip_pipeline* p);

/* GPU index encoder (build side): block_posting_list::write (block_posting_list.hpp:13-53) with
 * optpfor_block::encode + ds2i's findBestB (block_codecs.hpp:156-208) as HIP kernels, one wavefront per 128-posting
 * block. Input: nlists posting lists in CSR form (list t = docs / freqs [list_offsets[t], list_offsets[t+1])), sorted
 * doc-ids < num_docs, freqs >= 1. Output: the frozen block_freq_index<optpfor_block> image (ds2i_blob_* of
 * ds2i_build.h release it), byte-identical to the host builder's (ds2i_builder_*). index_kind must be
 * DS2I_BLOCK_OPTPFOR. device_ms (may be NULL) receives the hipEvent time of the two kernel passes. */
typedef struct ds2i_blob ds2i_blob;
int ds2i_hip_encode_index(int device, int index_kind, uint64_t num_docs, uint64_t nlists, const uint64_t* list_offsets,
                          const uint32_t* docs, const uint32_t* freqs, ds2i_blob** image, double* device_ms);

/* The synthetic collection of ds2i_build.h generated on `threads` host threads (<= 0: all) and encoded on the GPU;
 * the same two images as ds2i_synth_build(p, DS2I_BLOCK_OPTPFOR, ...), byte for byte. wand_image, total_postings,
 * generate_s (host seconds spent generating the lists) and device_ms may be NULL. */
struct ds2i_synth_params;
int ds2i_hip_synth_encode(int device, const struct ds2i_synth_params* p, int threads, ds2i_blob** index_image,
                          ds2i_blob** wand_image, uint64_t* total_postings, double* generate_s, double* device_ms);

/* Inspection of the upload-time pruning tables of one list (test hooks; both tables exist only with wand data).
 * block weights: bmw[b] = max over block b's postings of bm25::doc_term_weight(freq, norm_len[doc]) (the block-level
 * analogue of wand_data's max_term_weight, wand_data.hpp:40-52); out gets *nblocks floats (capacity in floats).
 * range table, level 1..3: byte e covers the doc-ids [e << *shift, (e + 1) << *shift): 0 = no posting of the list there,
 * else entry * (*list_max / 255) >= the largest doc_term_weight of the range; level l + 1 halves the resolution six
 * times (its entry e is the maximum of entries 64 e .. 64 e + 63 of level l). out gets *entries bytes. A table that
 * was not built (no wand data, DS2I_NO_BMW / DS2I_NO_RMW) reports 0 entries.
 * level 4: the membership hints, one byte per level-1 entry: 0 = the range holds no posting, 255 = two or more, else
 * 1 + (offset of its one posting inside the range) mod 254 (block_optpfor indexes; 0 entries elsewhere). */
int ds2i_hip_list_block_weights(ds2i_hip_index* idx, uint32_t term, float* out, uint64_t capacity, uint64_t* nblocks);
int ds2i_hip_list_range_table(ds2i_hip_index* idx, uint32_t term, uint32_t level, uint8_t* out, uint64_t capacity,
                              uint64_t* entries, uint32_t* shift, float* list_max);

/* profiling aid: streams the whole index arena once with the decoders' load shape (calibrates FETCH_SIZE) */
int ds2i_hip_calibration_read(ds2i_hip_index* idx, uint64_t* bytes_read);
/* GPU unit-test hook: wave64 inclusive prefix sum over rows of 64 values */
int ds2i_hip_selftest_scan(int device, const uint32_t* in, uint32_t* out, uint32_t rows);
/* GPU unit-test hook: the kernels' bm25::doc_term_weight (bm25.hpp:11-15), element-wise */
int ds2i_hip_selftest_bm25(int device, const uint32_t* freqs, const float* norm_lens, float* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
