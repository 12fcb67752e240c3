// Shared by the C-ABI translation units (capi.cpp: index handle, capi_batch.cpp: query batches and the
// pipelined submit/wait form): the index handle, RAII device / pinned buffers, error plumbing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/ds2i_hip.h"
#include "abi_structs.hpp"
#include "capi_error.hpp"

#define HIP_OK(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return ds2i_set_error(DS2I_EDEVICE, m_.c_str());                                       \
        }                                                                                          \
    } while (0)

// Kernel classes by number of distinct query terms: <=2, <=4, <=8, <=16 keep every list's current block in LDS
// (footprint per wave grows with the class); class 4 ("long", > 16 terms -- the reference has no limit,
// queries.hpp:35-86) keeps the per-list state in a global scratch area and runs the one-document-per-step traversal.
static const int NCLS = 5;
static const int CLS_LONG = 4;
static inline int class_of(size_t nterms) { return nterms <= 2 ? 0 : nterms <= 4 ? 1 : nterms <= 8 ? 2 : nterms <= 16 ? 3 : 4; }

// grow-only device allocation: a reused batch slot re-allocates only when a batch needs more than any before it
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() {}
    DevBuf(DevBuf const&) = delete;
    DevBuf& operator=(DevBuf const&) = delete;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        release();
        const size_t c = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, c);
        if (e == hipSuccess) cap = c; else p = nullptr;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DevBuf() { release(); }
    template <class T> T* at(size_t byte_off) const { return (T*)((uint8_t*)p + byte_off); }
};

// grow-only pinned host allocation (async H2D / D2H need page-locked memory to overlap with kernels)
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    PinBuf() {}
    PinBuf(PinBuf const&) = delete;
    PinBuf& operator=(PinBuf const&) = delete;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        release();
        const size_t c = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&p, c, hipHostMallocDefault);
        if (e == hipSuccess) cap = c; else p = nullptr;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    ~PinBuf() { release(); }
    template <class T> T* at(size_t byte_off) const { return (T*)((uint8_t*)p + byte_off); }
};

struct ds2i_hip_batch;

struct ds2i_hip_index {
    int device = 0, kind = 0, num_cus = 256;
    int kind_on_disk = -1;          // the index kind the caller uploaded when it differs from `kind` (block_mixed transcoded at upload), else -1
    uint64_t size = 0, num_docs = 0;
    uint8_t* d_arena = nullptr;
    uint64_t arena_bytes = 0;
    float* d_norm_lens = nullptr;
    float min_norm_len = 0.f;       // smallest norm_len (upper-bounds doc_term_weight by the freq alone)
    bool has_wand = false;
    std::vector<uint64_t> list_off; // arena offsets, size+1 (list i spans [off[i], end[i]))
    std::vector<uint64_t> list_end;
    std::vector<uint32_t> list_n;
    std::vector<uint32_t> list_nb;  // blocks (block indexes) / chunks (opt index) per list
    std::vector<uint64_t> list_aux0, list_aux1; // opt index: docs / freqs sequence bit offsets
    std::vector<uint64_t> list_blk_base;        // blocks / chunks of all preceding lists (access profile, skip table, block-max weights)
    uint64_t total_blocks = 0;
    uint8_t* d_skip = nullptr;                  // block indexes: interleaved {block_max, block end offset} per block
    uint8_t* d_bits0 = nullptr;     // opt index: docs bit vector
    uint8_t* d_bits1 = nullptr;     // opt index: freqs bit vector
    float* d_bmw = nullptr;         // per block / chunk: max doc_term_weight of its postings (ranked_and pruning), or null
    uint8_t* d_rmw = nullptr;       // doc-id-range max-weight tables (abi_structs.hpp, BatchArgs::rmw), or null
    uint64_t rmw_bytes = 0;
    uint8_t* d_rmh = nullptr;       // membership hints: one more byte per level-1 range-table entry, same offsets (BatchArgs::rmh), or null
    int rmw_g = 0;                  // entries per posting the tables were built with (DS2I_RMW_G)
    // block_optpfor: exception side slots (64 dwords per block), their overflow area, and the lists' partial last blocks in
    // plain form (abi_structs.hpp, BatchArgs::xslots / xovf / tails), or null
    uint32_t* d_xslots = nullptr;
    uint32_t* d_xovf = nullptr;
    uint32_t* d_tails = nullptr;
    uint64_t side_bytes = 0;
    // what this upload builds beside the image (capi.cpp: choose_table_plan; DS2I_TABLE_BUDGET or the explicit knobs)
    double plan_g = 4.0;            // range-table entries per posting (0 = no range tables)
    bool plan_hints = true, plan_slots = true;
    uint64_t table_budget = 0;      // bytes the upload was asked to stay under (0 = no budget)
    std::vector<uint64_t> list_tail_off; // postings in the partial last blocks of all preceding lists
    bool d_skip_or_pef() const { return d_skip != nullptr || kind >= DS2I_OPT; } // what the streaming kernels walk the driving list by
    bool has_bitmaps = false;       // dense lists carry an exact bitmap behind their range-table levels
    std::vector<uint32_t> list_rmw_off64, list_rmw_shift;
    std::vector<float> list_bmw;    // per list: max over its blocks of d_bmw (device-computed)
    std::vector<float> list_topbmw; // per list: its DS2I_HIP_MAX_K largest block weights, descending, padded with 0
    uint64_t extra_bytes = 0;
    std::vector<float> max_term_weight;
    // class kernels of consecutive batches queue up on the class streams; uploads and merges / result copies have
    // their own streams so that the next batch's H2D never waits behind the previous batch's merge
    hipStream_t stream[NCLS] = {};
    // Small batches (a rank's share of a batch sharded over several GPUs): the pipeline's odd slots launch on a second set of class
    // streams, so that the class kernels of batch i+1 run beside those of batch i -- a 512-query batch fills a tenth of the wave slots
    // and its two-list kernel's span (its longest unit) was the whole step. Full batches keep one set (two sets: measured -3 % at 4096).
    // Created by the first small batch that wants them (every stream in use is a hardware queue; two ranks sharing one device with twelve
    // streams each ran at half speed).
    hipStream_t stream_alt[NCLS] = {};
    std::mutex stream_alt_mu;
    hipStream_t s_up = nullptr, s_merge = nullptr;
    unsigned int* d_ticket = nullptr; // scratch word(s) for the calibration kernel
    ds2i_hip_batch* oneshot = nullptr; // cached slot of ds2i_hip_query_batch (buffers are reused between calls)
    // Planning reads one 80-byte record per query term instead of eight parallel arrays (a 4096-query batch has ~12 k
    // terms; at configs[1] scale planning, not the kernels, bounds the end-to-end rate): the QTerm as the kernels want
    // it, with the query-independent factors parked in the fields planning overwrites -- q_weight = max_term_weight,
    // max_weight = list max block weight.
    std::vector<ds2i_dev::QTerm> term_proto;
};

void ds2i_batch_destroy(ds2i_hip_batch* b); // capi_batch.cpp

// index[term] as the kernels see it (weights left at 0: they depend on the query)
static inline ds2i_dev::QTerm ds2i_make_qterm(const ds2i_hip_index* idx, uint32_t term) {
    ds2i_dev::QTerm qt;
    const bool freq_layout = idx->kind >= DS2I_OPT;
    qt.list_off = idx->list_off[term];
    qt.list_end = idx->list_end[term];
    qt.n = idx->list_n[term];
    qt.q_weight = 0.f;
    qt.max_weight = 0.f;
    qt.term = freq_layout ? idx->list_nb[term] : term;
    qt.aux0 = freq_layout ? idx->list_aux0[term] : idx->list_blk_base[term];
    qt.aux1 = freq_layout ? idx->list_aux1[term] : (idx->d_tails ? idx->list_tail_off[term] : 0);
    qt.blk_base = (uint32_t)idx->list_blk_base[term];
    qt.max_bmw = 0.f;
    qt.suf_bmw = 0.f;
    qt.floor1 = 0.f;
    qt.rmw_off64 = idx->d_rmw ? idx->list_rmw_off64[term] : 0;
    qt.rmw_shift = idx->d_rmw ? idx->list_rmw_shift[term] : 0;
    qt.rmw_scale = 0.f;
    qt.nblocks = idx->list_nb[term];
    return qt;
}
