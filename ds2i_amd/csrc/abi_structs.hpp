// Plain structs shared by the HIP kernels (kernels.hip) and the host C-ABI (capi.cpp).
#pragma once
#include <stdint.h>

namespace ds2i_dev {

// One query term as prepared by the host: byte range of its posting list inside the device
// arena (block indexes) or of its chunk directory (opt index: cmax[] at list_off, 12-dword chunk entries at
// list_end) + BM25 weights (bm25.hpp:17-24 needs logf -> computed on the host).
struct QTerm {
    uint64_t list_off; // byte offset of vbyte(n) in the arena
    uint64_t list_end; // byte offset one past the list
    uint32_t n;        // postings
    float q_weight;
    float max_weight; // q_weight * max_term_weight[term]
    uint32_t term;    // block indexes: term id ; opt index: number of chunks of the list
    uint64_t aux0;    // opt index: absolute bit offset of the list's docs sequence (after its gamma header);
                      // block indexes: number of blocks of all preceding lists (access-profile base)
    uint64_t aux1;    // opt index: absolute bit offset of the list's freqs sequence;
                      // block_optpfor with side tables: dword offset of the list's partial last block in BatchArgs::tails
    uint32_t blk_base; // blocks (chunks) of all preceding lists: index of this list's first entry in bmw[]
    float max_bmw;     // q_weight * (max over the list's blocks of bmw[]): device-computed list bound (ranked_and pruning)
    float suf_bmw;     // sum of max_bmw over the LATER lists of the query (enumerator order)
    float floor1;      // one-term ranked queries: q_weight * (k-th largest bmw of the list) -- k documents reach it
    // doc-id-range max-weight table of the list (rmw, see BatchArgs::rmw): entry (doc >> rmw_shift) of the byte array at
    // rmw + 64 * rmw_off64; rmw_scale * entry bounds q_weight * doc_term_weight of any posting of that doc-id range
    uint32_t rmw_off64;
    uint32_t rmw_shift;
    float rmw_scale;   // q_weight * (list max block weight) / 255
    uint32_t nblocks;  // blocks (block indexes) / chunks (freq_index layouts) of the list
};
static_assert(sizeof(QTerm) == 80, "QTerm is an 80-byte device record");

enum { PH_TOTAL = 0, PH_DOCS, PH_FREQS, PH_FIND, PH_MEMBER, PH_SCORE, PH_TOPK, PH_PROLOG, PH_PROBE, PH_INSERT, PH_STREAM, PH_PREFETCH, PH_FLOOR, PH_UNIT,
       PH_C_VISIT, PH_C_SURV1, PH_C_SURV2, PH_C_BDOCS, PH_C_BFREQS, PH_C_HEAP, PH_C_LIVEROUNDS, PH_C_ALIVE, PH_C_GBLOCKS, /* event counts of the diagnostic build */ PH_COUNT };
struct Stats {
    unsigned long long docs_blocks, freqs_blocks, block_max_examined, algorithmic_bytes, postings_scored, rounds;
    unsigned long long phase_cycles[PH_COUNT]; // summed over waves; only filled with -DDS2I_PHASE_TIMING
};

enum { OP_AND = 0, OP_AND_FREQ = 1, OP_OR = 2, OP_OR_FREQ = 3, OP_RANKED_AND = 4, OP_WAND = 5, OP_MAXSCORE = 6,
       OP_RANKED_OR = 7, OP_REFERENCE_ORDER = 0x100 };

// One schedulable piece of a query: conjunctive queries are split by ranges of blocks of their
// shortest list, the disjunctive operators by doc-id ranges.
struct Unit {
    uint32_t q;         // query id
    uint32_t blk_begin; // first block of list 0 this unit owns
    uint32_t blk_end;   // one past the last block
    uint32_t nparts;    // number of units of query q (1 -> the unit writes the final outputs itself)
};

// k_ranked_stream: everything the start of a unit needs in ONE 32-byte record per ticket of the launch (the unit, its query's
// first term in qterms and its score histogram) -- order[tkt] -> units[uid] -> q_off[q] / q_hist_slot[q] were three dependent
// round trips of a unit that lives for a few dozen blocks
struct UnitRec {
    uint32_t uid, q, blk_begin, blk_end, nparts, qt_off, hist_slot, pad;
};
static_assert(sizeof(UnitRec) == 32, "UnitRec is a 32-byte device record");

struct BatchArgs {
    const uint8_t* arena;     // block indexes: list bytes ; opt index: chunk directory
    const uint8_t* bits0;     // opt index: docs bit vector words
    const uint8_t* bits1;     // opt index: freqs bit vector words
    const float* norm_lens;
    float min_norm_len;       // smallest norm_len of the collection: doc_term_weight(f, min_norm_len) bounds any document's term weight
    const QTerm* qterms;      // terms of all queries, already in enumerator order
    const uint32_t* q_off;    // nq+1 offsets into qterms
    const Unit* units;        // all units, grouped by query
    const uint32_t* order;    // nslice unit ids, scheduling order (costliest first)
    const UnitRec* urec;      // the same order, one record per ticket (ranked conjunctive classes 0 / 1), or null
    uint32_t nslice;
    uint32_t num_docs;
    uint32_t k;
    int codec;
    unsigned long long* unit_clock; // diagnostic (DS2I_UNIT_CLOCK=1, instrumented kernels): {start, end} s_memrealtime of every unit, or null
    unsigned long long* out_count; // nq
    float* out_topk;          // nq*k, descending, padded with -inf
    uint32_t* out_topk_len;   // nq
    unsigned long long* out_freq_sum; // nq (and_freq / or_freq checksum of touched freqs) or null
    uint32_t* out_matches;    // optional doc-id lists (and) or null
    const unsigned long long* match_off; // nq+1 capacity offsets
    // per-unit partial results of split queries (merged by k_merge)
    unsigned long long* unit_count;
    float* unit_topk;         // nunits*k
    uint32_t* unit_topk_len;
    unsigned long long* unit_freq_sum;
    // wand / maxscore: top-k of the ranked_and pass over the same batch. Its k-th score is a valid lower bound of
    // the final k-th score (AND results are a subset of OR results), used as a pruning floor from the first posting
    const float* seed_topk;    // nq*k or null
    const uint32_t* seed_len;  // nq
    // block-synchronous wand / maxscore / ranked_or: the parts of a split query publish their k-th score here (float
    // bits; scores are >= 0 so the bit patterns order like the values) and adopt the maximum as their floor: the final
    // k-th score of the union is >= the k-th score of any part
    unsigned int* q_floor;     // nq or null
    // block-synchronous ranked_and: the parts of a split query share a 256-bucket histogram of the scores that entered
    // their heaps; the lower edge of the highest bucket with >= k documents at or above it is every part's floor
    unsigned int* q_hist;      // (split queries) * 256 or null
    const uint32_t* q_hist_slot; // nq: the query's histogram (rank among the split queries); unused for whole queries
    unsigned int* block_profile; // block indexes: 2 counters per block of the index (docs / freqs decodes) or null
    const void* skip;            // block indexes: interleaved {block_max, block end offset} per block (uint2) or null
    const float* bmw;            // per block / chunk of the index: max doc_term_weight of its postings, or null
    // Doc-id-range max-weight tables (one per list, QTerm::rmw_*), or null: byte e of list t's table covers the doc-ids
    // [e << shift_t, (e + 1) << shift_t); 0 = no posting of the list in that range (=> no document of it can be in an
    // intersection with the list), otherwise an upward-rounded 8-bit quantisation of the largest doc_term_weight in
    // the range relative to the list's maximum. Direct addressing by doc-id: a candidate's bound in every other list
    // costs one byte gather per list and no search (cf. the doc-id-oriented block-max indexes of Dimopoulos, Nepomnyachiy
    // & Suel, WSDM'13). shift_t is chosen per list so that the table holds DS2I_RMW_G..2*DS2I_RMW_G entries per posting.
    // Each table is followed by two coarser levels of itself (level l + 1 entry e = max of level l entries 64 e .. 64 e + 63;
    // every level padded to 64 bytes): the largest entry over the doc-id span of a whole BLOCK of the driving list is found
    // in <= 16 bytes of the level whose entries are wide enough, and bounds every candidate of that block at once.
    const uint8_t* rmw;
    uint32_t rmw_bitmaps;        // the dense lists' exact bitmaps exist behind their tables (RmwLevels::has_bitmap)
    // Membership hints (or null): a second byte per level-1 range-table entry, at the same offsets in a parallel buffer.
    // 0 = the range holds no posting of the list, 255 = it holds two or more, otherwise 1 + (offset of its ONE posting
    // inside the range) mod 254. A candidate whose range holds a single posting at another offset is provably not in the
    // list: for a list with 32 doc-ids per entry that settles 30 of 31 candidates the weight byte lets through, without the
    // block search + block decode a lookup costs (k_ranked_stream; block_optpfor indexes).
    const uint8_t* rmh;
    // Exception side slots (block_optpfor, or null): 64 dwords per block of the index (block b of list t = slot blk_base_t + b)
    // holding the OptPFor exceptions of its docs and freqs parts as position masks + ready-to-OR values (layout:
    // device_codecs.hpp, optpfor_decode_side); xovf = the overflow area of the few blocks whose exceptions do not fit.
    // tails: the partial last block of every list (n % 128 postings; interpolative on disk, serial by construction)
    // expanded to gaps-1 then freqs-1 (+ the byte counts of its two parts), list t's entry at dword aux1_t -- the stream kernels then carry one decoder.
    const uint32_t* xslots;
    const uint32_t* xovf;
    const uint32_t* tails;
    // k_union_topk (wand / maxscore / ranked_or as streams): units belong to VIRTUAL queries = (query, driving list); qterms /
    // q_off then describe the virtual queries, and vq_info holds 3 words per virtual query: the real query, the number
    // of exclusion lists (slots 1 .. nexcl: lists of higher max score -- a document found there is theirs), and the float
    // bits of the real query's score bound (the scale of its shared score histogram). Null for every other kernel.
    const uint32_t* vq_info;
    uint32_t ut_first;    // k_union_topk: optional lists gathered in the first trip (0 = every list in one trip; DS2I_UT_FIRST)
    uint32_t* long_scratch;      // "long" class (> 16 terms): per-unit enumerator state in global memory
    uint32_t long_stride;        // dwords of scratch per unit
    uint32_t dyn_lists;          // union kernels: list slots of decoded blocks in dynamic LDS (>= the longest query of the launch)
    Stats* stats;
};

// upload-time pass computing bmw[] (k_block_max_weights): one wave per item = <=64 consecutive blocks of one list
struct BmwItem {
    uint32_t list, blk_begin;
};
// geometry of one list's range-table levels, derived from the collection size and the list's shift alone
// membership hint of a doc-id inside its level-1 range (BatchArgs::rmh)
__host__ __device__ inline uint32_t rmh_code(uint32_t doc, uint32_t shift) { return 1u + (doc & ((1u << shift) - 1u)) % 254u; }
struct RmwLevels {
    uint32_t e[3];   // entries of level 1, 2, 3
    uint64_t off[3]; // byte offset of each level from the start of the list's table
    __host__ __device__ RmwLevels(uint32_t num_docs, uint32_t shift) {
        e[0] = (num_docs >> shift) + 1u;
        e[1] = (e[0] + 63u) >> 6;
        e[2] = (e[1] + 63u) >> 6;
        off[0] = 0;
        off[1] = ((uint64_t)e[0] + 63u) & ~63ull;
        off[2] = off[1] + (((uint64_t)e[1] + 63u) & ~63ull);
    }
    __host__ __device__ uint64_t bytes() const { return off[2] + (((uint64_t)e[2] + 63u) & ~63ull); }
    // Lists holding at least one document in 64 additionally get an exact BITMAP of their doc-ids behind the levels (at
    // byte offset bytes(); num_docs bits, padded to 64 bytes + one spare line so that a pair of adjacent words can always
    // be read): <= 64 bits per posting. and_query tests membership in such a list with one bit gather and never decodes
    // it; or_query ORs its words into the union instead of decoding its blocks.
    static __host__ __device__ bool has_bitmap(uint32_t n, uint32_t num_docs) { return (uint64_t)n * 64u >= num_docs; }
    static __host__ __device__ uint64_t bitmap_bytes(uint32_t num_docs) { return ((((uint64_t)num_docs + 31u) / 32u * 4u + 63u) & ~63ull) + 64u; }
};
struct BmwArgs {
    const uint8_t* arena;
    const uint8_t* bits0;
    const uint8_t* bits1;
    const float* norm_lens;
    const QTerm* lists; // one per list of the index
    const BmwItem* items;
    uint32_t nitems;
    int codec;
    uint32_t num_docs;
    float* bmw;             // out: one per block
    unsigned int* list_bmw; // out: per list max (float bits; weights are >= 0 so the bit patterns order like the values)
    uint8_t* rmw;           // second pass (k_range_max_weights): the range tables; lists[].max_weight = the list maximum of pass 1
    uint8_t* rmh;           // second pass, level-1 fill: the membership hints (BatchArgs::rmh) or null
    uint32_t bitmaps;       // level-1 pass: also set the bits of the dense lists' bitmaps
    uint32_t rmw_level;     // 0: fill level 1 from the postings; 1 / 2: items = {list, first entry of a 4096-entry run of level
                            // rmw_level + 1}, each entry the maximum of 64 entries of the level below
};

// exception side slot of one block (device_codecs.hpp, optpfor_decode_side): dwords, first add, adds held in-slot, overflow word
// dword 8 / 9: copies of the docs / freqs part's header; dword 10 (XSLOT_FLAG): 0 = the common case -- neither part raw (b < 32),
// both inside the 512 bytes a wave stages from the block's start, every add in the slot -- else bit 31 + (1 + dword offset of
// the block's adds in the overflow area, or 0 when they are in the slot); dwords 11 .. 63: the adds, docs part first
static constexpr uint32_t XSLOT_DW = 64, XSLOT_HDR = 8, XSLOT_FLAG = 10, XSLOT_ADDS = 11, XSLOT_CAP = 53, XSLOT_SLOW = 0x80000000u;

// upload-time pass filling the exception side slots and the tail table (k_build_side_tables): items as in BmwArgs
struct SideArgs {
    const uint8_t* arena;
    const QTerm* lists;
    const BmwItem* items;
    uint32_t nitems;
    uint32_t num_docs;
    const void* skip;          // interleaved skip table
    uint32_t* xslots;          // out: XSLOT_DW dwords per block
    uint32_t* xovf;            // out: overflow area
    unsigned long long xovf_cap; // dwords
    unsigned long long* xovf_cursor; // dwords handed out so far (may run past the capacity: the host then re-runs with more)
    uint32_t* tails;           // out
    unsigned int* bad;         // out: blocks whose header disagrees with their decoded values (corrupt image)
};

// or_query<with_freqs>: the freqs of every posting of every query term, streamed on their own (freq_stream.hip)
struct FreqArgs {
    const uint8_t* arena;
    const void* skip;
    const uint32_t* xslots;
    const uint32_t* xovf;
    const uint32_t* tails;
    const QTerm* qterms;        // the terms of all queries of the batch
    const uint32_t* qterm_q;    // the query each of them belongs to
    unsigned long long* out_freq_sum; // per query: the sum is ADDED (after the union kernels and the merge have written theirs)
};

// and_query / and_query<with_freqs> of queries whose lists are ALL dense enough to carry their exact bitmap (RmwLevels::has_bitmap):
// every list that has to be read is streamed on its own (freq_stream.hip, k_and_stream), membership in the OTHER lists is one bit
// gather per posting and list -- no list is searched, none is decoded for another list's sake
struct StreamTerm {
    uint64_t list_off;   // as QTerm
    uint64_t tail;       // QTerm::aux1: the list's entry of the tail table
    uint32_t n;
    uint32_t blk_base;
    uint32_t q;          // the query the sums belong to
    uint32_t counts;     // 1: this list's matches are the query's result count (its shortest list), 0: freqs only
    uint32_t nother;     // other lists of the query (1..3)
    uint32_t pad;
    uint64_t bm[3];      // byte offsets of their bitmaps from BatchArgs::rmw
};
static_assert(sizeof(StreamTerm) == 64, "StreamTerm is a 64-byte device record");
struct AndStreamArgs {
    const uint8_t* arena;
    const void* skip;
    const uint32_t* xslots;
    const uint32_t* xovf;
    const uint32_t* tails;
    const uint8_t* rmw;
    const StreamTerm* terms;
    unsigned long long* out_count;
    unsigned long long* out_freq_sum; // or null (and_query)
};

struct MergeArgs {
    const uint32_t* split_queries; // ids of queries with nparts > 1
    uint32_t nsplit;
    const uint32_t* q_unit_off;    // nq+1: units of query q are [q_unit_off[q], q_unit_off[q+1])
    uint32_t k;
    int ranked;
    const unsigned long long* unit_count;
    const float* unit_topk;
    const uint32_t* unit_topk_len;
    const unsigned long long* unit_freq_sum;
    unsigned long long* out_count;
    float* out_topk;
    uint32_t* out_topk_len;
    unsigned long long* out_freq_sum;
};

struct DecodeArgs {
    const uint8_t* arena;
    const uint8_t* bits0;
    const uint8_t* bits1;
    QTerm term;
    int codec;
    uint32_t num_docs;
    uint32_t* out_docs;
    uint32_t* out_freqs;
    Stats* stats;
    const void* skip;        // side-slot decode (k_decode_list_side): the interleaved skip table, the slots, their overflow area, the tail table
    const uint32_t* xslots;
    const uint32_t* xovf;
    const uint32_t* tails;
};

} // namespace ds2i_dev
