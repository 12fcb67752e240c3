// Host implementation of include/ds2i_hip.h: index upload, query-batch preparation
// (the host half of queries.hpp: term normalisation, BM25 query weights, list ordering)
// and kernel launches. The device half lives in kernels.hip. There is NO CPU fallback:
// every query result comes from the HIP kernels or the call fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ds2i_hip.h"
#include "abi_structs.hpp"
#include "capi_error.hpp"
#include "host_index.hpp"
#include "host_pef.hpp"
#include <atomic>
#include <mutex>
#include <thread>

using ds2i_dev::BatchArgs;
using ds2i_dev::DecodeArgs;
using ds2i_dev::MergeArgs;
using ds2i_dev::Unit;
using ds2i_dev::QTerm;
using ds2i_dev::Stats;

extern "C" {
hipError_t ds2i_launch_batch(int op, int tmax_class, const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_decode_list(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_merge(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_copy_seed(const uint32_t* queries, uint32_t n, uint32_t k, const float* seed_topk, const uint32_t* seed_len,
                                 const unsigned long long* seed_count, float* out_topk, uint32_t* out_len,
                                 unsigned long long* out_count, hipStream_t s);
hipError_t ds2i_launch_calib_read(const uint32_t* base, unsigned long long ndw, uint32_t* out, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_selftest(const uint32_t* in, uint32_t* out, unsigned blocks, hipStream_t s);
}

// ------------------------------------------------------------------ errors
namespace {
thread_local std::string g_last_error;
}
int ds2i_set_error(int code, const char* msg) {
    g_last_error = msg ? msg : "";
    return code;
}
const char* ds2i_get_error() { return g_last_error.c_str(); }

#define HIP_OK(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return ds2i_set_error(DS2I_EDEVICE, m_.c_str());                                       \
        }                                                                                          \
    } while (0)

// ------------------------------------------------------------------ handles
// kernel classes by number of distinct query terms: <=2, <=4, <=8, <=16 (LDS per wave grows with it)
static const int NCLS = 4;
static inline int class_of(size_t nterms) { return nterms <= 2 ? 0 : nterms <= 4 ? 1 : nterms <= 8 ? 2 : 3; }
struct ds2i_hip_index {
    int device = 0, kind = 0, num_cus = 256;
    uint64_t size = 0, num_docs = 0;
    uint8_t* d_arena = nullptr;
    uint64_t arena_bytes = 0;
    float* d_norm_lens = nullptr;
    bool has_wand = false;
    std::vector<uint64_t> list_off; // arena offsets, size+1 (list i spans [off[i], end[i]))
    std::vector<uint64_t> list_end;
    std::vector<uint32_t> list_n;
    std::vector<uint32_t> list_nb;  // blocks (block indexes) / chunks (opt index) per list
    std::vector<uint64_t> list_aux0, list_aux1; // opt index: docs / freqs sequence bit offsets
    std::vector<uint64_t> list_blk_base;        // block indexes: blocks of all preceding lists (access profile, skip table)
    uint64_t total_blocks = 0;
    uint8_t* d_skip = nullptr;                  // block indexes: interleaved {block_max, block end offset} per block
    uint8_t* d_bits0 = nullptr;     // opt index: docs bit vector
    uint8_t* d_bits1 = nullptr;     // opt index: freqs bit vector
    uint64_t extra_bytes = 0;
    std::vector<float> max_term_weight;
    hipStream_t stream[NCLS] = {};
    hipEvent_t ev[2 + 2 * NCLS] = {};
    Stats* d_stats = nullptr;     // [NCLS]
    unsigned int* d_ticket = nullptr; // [NCLS]
};

struct ds2i_hip_batch {
    ds2i_hip_batch* seed = nullptr; // wand / maxscore: ranked_and pass over the same queries (pruning floor)
    uint32_t* d_single = nullptr;   // ids of one-term queries answered by the seed pass
    unsigned int* d_qfloor = nullptr; // per-query shared pruning floor of the disjunctive kernel
    bool instrument = true;           // collect ds2i_hip_stats counters (instrumented kernel instantiations)
    unsigned int* d_prof = nullptr;   // block access profile (2 counters per block of the index), optional
    uint32_t nsingle = 0;
    ds2i_hip_index* idx = nullptr;
    int op = 0;
    uint32_t k = 0, nq = 0;
    bool want_matches = false;
    uint32_t ncls[NCLS] = {}; // units per kernel class
    uint32_t nqcls[NCLS] = {}; // queries per kernel class
    uint32_t nunits = 0, nsplit = 0;
    std::vector<Unit> units;
    std::vector<uint32_t> q_unit_off;
    Unit* d_units = nullptr;
    uint32_t* d_q_unit_off = nullptr;
    uint32_t* d_split = nullptr;
    unsigned long long* d_unit_count = nullptr;
    float* d_unit_topk = nullptr;
    uint32_t* d_unit_topk_len = nullptr;
    unsigned long long* d_unit_freq_sum = nullptr;
    QTerm* d_qterms = nullptr;
    uint32_t* d_qoff = nullptr;
    uint32_t* d_order[NCLS] = {};
    unsigned long long* d_count = nullptr;
    float* d_topk = nullptr;
    uint32_t* d_topk_len = nullptr;
    unsigned long long* d_freq_sum = nullptr;
    uint32_t* d_matches = nullptr;
    unsigned long long* d_match_off = nullptr;
    std::vector<unsigned long long> match_off;
    float cls_ms[NCLS] = {};
    Stats cls_stats[NCLS] = {};
};

namespace {

const float kNegInf = -std::numeric_limits<float>::infinity();

uint32_t host_vbyte(const uint8_t* p, size_t avail, uint32_t& val) {
    uint32_t v = 0, shift = 0, i = 0;
    while (i < avail && i < 5) {
        uint8_t c = p[i++];
        v += uint32_t(c & 127) << shift;
        if (c & 128) { val = v; return i; }
        shift += 7;
    }
    return 0;
}

void free_index(ds2i_hip_index* x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->d_arena) (void)hipFree(x->d_arena);
    if (x->d_skip) (void)hipFree(x->d_skip);
    if (x->d_norm_lens) (void)hipFree(x->d_norm_lens);
    if (x->d_bits0) (void)hipFree(x->d_bits0);
    if (x->d_bits1) (void)hipFree(x->d_bits1);
    if (x->d_stats) (void)hipFree(x->d_stats);
    if (x->d_ticket) (void)hipFree(x->d_ticket);
    for (auto& s : x->stream) if (s) (void)hipStreamDestroy(s);
    for (auto& e : x->ev) if (e) (void)hipEventDestroy(e);
    delete x;
}

} // namespace

extern "C" {

const char* ds2i_hip_last_error(void) { return ds2i_get_error(); }

// The four LDS classes of a batch run on four streams; with the HIP default of 4 hardware queues two of them end up
// sharing one (the null stream owns a queue) and serialise. Ask for more before the runtime initialises; an
// explicit setting of the user wins.
__attribute__((constructor)) static void ds2i_hip_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

int ds2i_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ds2i_hip_index_open(int device, int kind, const void* index_image, size_t index_bytes, const void* wand_image,
                        size_t wand_bytes, ds2i_hip_index** out) {
    if (!out || !index_image) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_index_open: null argument");
    if (kind < DS2I_BLOCK_OPTPFOR || kind > DS2I_UNIFORM)
        return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_index_open: unknown index kind");
    int ndev = ds2i_hip_device_count();
    if (device < 0 || device >= ndev) return ds2i_set_error(DS2I_EDEVICE, "ds2i_hip_index_open: no such HIP device");
    std::unique_ptr<ds2i_hip_index, void (*)(ds2i_hip_index*)> x(new ds2i_hip_index, free_index);
    x->device = device;
    x->kind = kind;
    const bool freq_layout = ds2i_host::is_freq_layout(kind); // opt / ef / single / uniform: freq_index images
    ds2i_host::block_index_view view;
    ds2i_host::opt_index_view oview;
    ds2i_host::wand_view wv;
    try {
        oview.layout = kind;
        if (freq_layout) oview.parse(index_image, index_bytes);
        else view.parse(index_image, index_bytes);
        if (wand_image) wv.parse(wand_image, wand_bytes);
    } catch (std::exception const& e) {
        return ds2i_set_error(DS2I_EFORMAT, e.what());
    }
    x->size = freq_layout ? oview.size : view.size;
    x->num_docs = freq_layout ? oview.num_docs : view.num_docs;
    if (wand_image) {
        if (wv.num_docs != x->num_docs || wv.num_terms < x->size)
            return ds2i_set_error(DS2I_EFORMAT, "wand data does not match the index (num_docs / terms)");
        x->has_wand = true;
        x->max_term_weight.resize(wv.num_terms);
        std::memcpy(x->max_term_weight.data(), wv.max_term_weight, 4 * wv.num_terms);
    }
    const uint64_t V = x->size;
    x->list_off.resize(V);
    x->list_end.resize(V);
    x->list_n.resize(V);
    x->list_nb.resize(V);
    std::vector<uint8_t> arena;
    if (freq_layout) {
        // freq_index (opt / ef / single / uniform): the two bit vectors go to HBM unchanged; every list is additionally flattened into a chunk
        // directory (cmax[] + 12-dword entries) so that the device treats <=128-posting chunks like blocks.
        x->list_aux0.resize(V);
        x->list_aux1.resize(V);
        std::vector<ds2i_host::pef_list_dir> dirs(V);
        std::atomic<uint64_t> next(0);
        std::string err;
        std::mutex mu;
        auto worker = [&]() {
            try {
                for (;;) {
                    uint64_t t = next.fetch_add(1);
                    if (t >= V) break;
                    oview.build_dir(t, dirs[t]);
                }
            } catch (std::exception const& e) {
                std::lock_guard<std::mutex> g(mu);
                err = e.what();
            }
        };
        unsigned nth = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < nth; ++i) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
        if (!err.empty()) return ds2i_set_error(DS2I_EFORMAT, err.c_str());
        uint64_t cursor = 0;
        for (uint64_t t = 0; t < V; ++t) {
            const uint64_t nch = dirs[t].chunks.size();
            x->list_off[t] = cursor;                                   // cmax[]
            x->list_end[t] = (cursor + 4 * nch + 15) & ~uint64_t(15);   // chunk entries
            cursor = x->list_end[t] + nch * sizeof(ds2i_host::pef_chunk);
            x->list_n[t] = dirs[t].n;
            x->list_nb[t] = (uint32_t)nch;
            x->list_aux0[t] = dirs[t].docs_bit0;
            x->list_aux1[t] = dirs[t].freqs_bit0;
        }
        x->arena_bytes = ((cursor + 15) & ~uint64_t(15)) + 4096;
        try {
            arena.assign(x->arena_bytes, 0);
        } catch (std::bad_alloc const&) {
            return ds2i_set_error(DS2I_ENOMEM, "out of host memory staging the chunk directory");
        }
        for (uint64_t t = 0; t < V; ++t) {
            std::memcpy(arena.data() + x->list_off[t], dirs[t].cmax.data(), 4 * dirs[t].cmax.size());
            std::memcpy(arena.data() + x->list_end[t], dirs[t].chunks.data(), dirs[t].chunks.size() * sizeof(ds2i_host::pef_chunk));
            ds2i_host::pef_list_dir().chunks.swap(dirs[t].chunks);
        }
    } else {
    // Device arena: every list is copied byte-for-byte, shifted by <= 3 pad bytes so that its
    // block_max / block_endpoint tables (which follow vbyte(n)) are dword aligned in HBM.
    uint64_t cursor = 0;
    for (uint64_t t = 0; t < V; ++t) {
        const uint8_t* lp = view.lists + view.list_offsets[t];
        const uint64_t len = view.list_offsets[t + 1] - view.list_offsets[t];
        uint32_t n = 0;
        uint32_t vl = host_vbyte(lp, len, n);
        if (!vl || !n) return ds2i_set_error(DS2I_EFORMAT, "posting list header is corrupt");
        const uint64_t nb = (uint64_t(n) + 127) / 128;
        if (len < vl + 8 * nb - 4) return ds2i_set_error(DS2I_EFORMAT, "posting list shorter than its block tables");
        uint64_t off = cursor;
        while ((off + vl) & 3) ++off;
        x->list_off[t] = off;
        x->list_end[t] = off + len;
        x->list_n[t] = n;
        x->list_nb[t] = (uint32_t)nb;
        cursor = off + len;
    }
    x->arena_bytes = ((cursor + 3) & ~uint64_t(3)) + 4096; // zero slack: decoders may over-read
    try {
        arena.assign(x->arena_bytes, 0);
    } catch (std::bad_alloc const&) {
        return ds2i_set_error(DS2I_ENOMEM, "out of host memory staging the index");
    }
    for (uint64_t t = 0; t < V; ++t)
        std::memcpy(arena.data() + x->list_off[t], view.lists + view.list_offsets[t],
                    view.list_offsets[t + 1] - view.list_offsets[t]);
    x->list_blk_base.resize(V);
    for (uint64_t t = 0; t < V; ++t) {
        x->list_blk_base[t] = x->total_blocks;
        x->total_blocks += x->list_nb[t];
    }
    }
    HIP_OK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    x->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_OK(hipMalloc((void**)&x->d_arena, x->arena_bytes));
    HIP_OK(hipMemcpy(x->d_arena, arena.data(), x->arena_bytes, hipMemcpyHostToDevice));
    if (!freq_layout && x->total_blocks) {
        // Interleaved skip table (auxiliary, like the list-offset table): entry b of a list = {block_max[b], byte offset
        // where block b ends inside the list's blocks area}. The probe of find_block_info() that locates a block then
        // also delivers its four table words, which saves the dependent table load of a non-sequential decode.
        std::vector<uint32_t> skip;
        try {
            skip.resize(2 * x->total_blocks);
        } catch (std::bad_alloc const&) {
            return ds2i_set_error(DS2I_ENOMEM, "out of host memory staging the skip table");
        }
        for (uint64_t t = 0; t < V; ++t) {
            const uint8_t* lp = view.lists + view.list_offsets[t];
            const uint64_t len = view.list_offsets[t + 1] - view.list_offsets[t];
            uint32_t n = 0;
            const uint32_t vl = host_vbyte(lp, len, n);
            const uint64_t nb = x->list_nb[t];
            const uint8_t* maxs = lp + vl;
            const uint8_t* eps = maxs + 4 * nb;
            const uint64_t data_len = len - vl - (8 * nb - 4);
            uint32_t* out = skip.data() + 2 * x->list_blk_base[t];
            for (uint64_t b = 0; b < nb; ++b) {
                uint32_t mx, ep = (uint32_t)data_len;
                std::memcpy(&mx, maxs + 4 * b, 4);
                if (b + 1 < nb) std::memcpy(&ep, eps + 4 * b, 4);
                out[2 * b] = mx;
                out[2 * b + 1] = ep;
            }
        }
        HIP_OK(hipMalloc((void**)&x->d_skip, 8 * x->total_blocks));
        HIP_OK(hipMemcpy(x->d_skip, skip.data(), 8 * x->total_blocks, hipMemcpyHostToDevice));
        x->extra_bytes += 8 * x->total_blocks;
    }
    if (freq_layout) {
        const uint64_t b0 = oview.docs_bits.nbytes, b1 = oview.freqs_bits.nbytes;
        HIP_OK(hipMalloc((void**)&x->d_bits0, b0 + 4096));
        HIP_OK(hipMalloc((void**)&x->d_bits1, b1 + 4096));
        HIP_OK(hipMemset(x->d_bits0 + b0, 0, 4096));
        HIP_OK(hipMemset(x->d_bits1 + b1, 0, 4096));
        HIP_OK(hipMemcpy(x->d_bits0, oview.docs_bits.bytes, b0, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(x->d_bits1, oview.freqs_bits.bytes, b1, hipMemcpyHostToDevice));
        x->extra_bytes = b0 + b1 + 8192;
    }
    if (x->has_wand) {
        HIP_OK(hipMalloc((void**)&x->d_norm_lens, 4 * (wv.num_docs + 1)));
        HIP_OK(hipMemcpy(x->d_norm_lens, wv.norm_lens, 4 * wv.num_docs, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMalloc((void**)&x->d_stats, NCLS * sizeof(Stats)));
    HIP_OK(hipMalloc((void**)&x->d_ticket, NCLS * sizeof(unsigned int)));
    for (auto& s : x->stream) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& e : x->ev) HIP_OK(hipEventCreate(&e));
    *out = x.release();
    return DS2I_OK;
}

void ds2i_hip_index_close(ds2i_hip_index* idx) { free_index(idx); }
uint64_t ds2i_hip_index_size(const ds2i_hip_index* idx) { return idx ? idx->size : 0; }
uint64_t ds2i_hip_index_num_docs(const ds2i_hip_index* idx) { return idx ? idx->num_docs : 0; }
uint64_t ds2i_hip_index_device_bytes(const ds2i_hip_index* idx) {
    return idx ? idx->arena_bytes + idx->extra_bytes + (idx->has_wand ? 4 * idx->num_docs : 0) : 0;
}

int ds2i_hip_list_size(const ds2i_hip_index* idx, uint32_t term, uint64_t* n) {
    if (!idx || !n) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_size: null argument");
    if (term >= idx->size) return ds2i_set_error(DS2I_ETERM, "term id out of range");
    *n = idx->list_n[term];
    return DS2I_OK;
}

int ds2i_hip_decode_list(ds2i_hip_index* idx, uint32_t term, uint32_t* docs, uint32_t* freqs, uint64_t capacity,
                         uint64_t* n) {
    if (!idx || !docs || !freqs || !n) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_decode_list: null argument");
    if (term >= idx->size) return ds2i_set_error(DS2I_ETERM, "term id out of range");
    const uint64_t len = idx->list_n[term];
    *n = len;
    if (capacity < len) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_decode_list: capacity too small");
    HIP_OK(hipSetDevice(idx->device));
    const uint64_t nb = idx->list_nb[term];
    uint32_t *d_docs = nullptr, *d_freqs = nullptr;
    HIP_OK(hipMalloc((void**)&d_docs, 4 * (len + 128)));
    HIP_OK(hipMalloc((void**)&d_freqs, 4 * (len + 128)));
    DecodeArgs a{};
    a.arena = idx->d_arena;
    a.bits0 = idx->d_bits0;
    a.bits1 = idx->d_bits1;
    a.term.list_off = idx->list_off[term];
    a.term.list_end = idx->list_end[term];
    a.term.n = (uint32_t)len;
    a.term.term = idx->kind >= DS2I_OPT ? idx->list_nb[term] : term;
    a.term.aux0 = idx->kind >= DS2I_OPT ? idx->list_aux0[term] : idx->list_blk_base[term];
    a.term.aux1 = idx->kind >= DS2I_OPT ? idx->list_aux1[term] : 0;
    a.codec = idx->kind >= DS2I_OPT ? (int)DS2I_OPT : idx->kind; // every freq_index layout decodes through the chunk directory
    a.num_docs = (uint32_t)idx->num_docs;
    a.out_docs = d_docs;
    a.out_freqs = d_freqs;
    a.stats = nullptr;
    unsigned grid = (unsigned)std::min<uint64_t>(nb, uint64_t(idx->num_cus) * 16);
    hipError_t e = ds2i_launch_decode_list(&a, grid, idx->stream[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(idx->stream[0]);
    if (e == hipSuccess) e = hipMemcpy(docs, d_docs, 4 * len, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(freqs, d_freqs, 4 * len, hipMemcpyDeviceToHost);
    (void)hipFree(d_docs);
    (void)hipFree(d_freqs);
    if (e != hipSuccess) return ds2i_set_error(DS2I_EDEVICE, hipGetErrorString(e));
    return DS2I_OK;
}

void ds2i_hip_batch_free(ds2i_hip_batch* b) {
    if (!b) return;
    if (b->seed && b->seed->d_prof == b->d_prof) b->seed->d_prof = nullptr; // shared with the owner
    ds2i_hip_batch_free(b->seed);
    (void)hipSetDevice(b->idx->device);
    (void)hipFree(b->d_qterms);
    (void)hipFree(b->d_qfloor);
    (void)hipFree(b->d_prof);
    (void)hipFree(b->d_qoff);
    for (auto& o : b->d_order) (void)hipFree(o);
    (void)hipFree(b->d_count);
    (void)hipFree(b->d_topk);
    (void)hipFree(b->d_topk_len);
    (void)hipFree(b->d_freq_sum);
    (void)hipFree(b->d_matches);
    (void)hipFree(b->d_match_off);
    delete b;
}

int ds2i_hip_batch_prepare(ds2i_hip_index* idx, int op, uint32_t k, const uint32_t* terms,
                           const uint32_t* query_offsets, uint32_t nq, int want_matches, ds2i_hip_batch** out) {
    if (!idx || !out || !query_offsets || (!terms && nq && query_offsets[nq] > 0))
        return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_prepare: null argument");
    const int base_op = op & ~DS2I_OP_REFERENCE_ORDER;
    if (base_op < DS2I_OP_AND || base_op > DS2I_OP_RANKED_OR)
        return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_prepare: unknown query operator");
    const bool conj = base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ || base_op == DS2I_OP_RANKED_AND;
    const bool disj_topk = base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE || base_op == DS2I_OP_RANKED_OR;

    const bool ranked = base_op >= DS2I_OP_RANKED_AND;
    if (ranked && !idx->has_wand) return ds2i_set_error(DS2I_ENOWAND, "ranked operator needs wand data");
    if (ranked && (k == 0 || k > DS2I_HIP_MAX_K)) return ds2i_set_error(DS2I_EINVAL, "k must be in [1,64]");
    if (!ranked && k == 0) k = 1;
    if (k > DS2I_HIP_MAX_K) k = DS2I_HIP_MAX_K;

    std::unique_ptr<ds2i_hip_batch, void (*)(ds2i_hip_batch*)> b(new ds2i_hip_batch, ds2i_hip_batch_free);
    b->idx = idx;
    b->op = op;
    b->k = k;
    b->nq = nq;
    b->want_matches = want_matches && (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ);

    std::vector<QTerm> qterms;
    std::vector<uint32_t> qnbs; // blocks / chunks of each query term's list (parallel to qterms)
    std::vector<uint32_t> qoff(nq + 1, 0);
    std::vector<double> qcost(nq, 0.0);   // estimated block decodes of the whole query
    std::vector<uint32_t> qnb0(nq, 0);    // blocks of the shortest list (conjunctive)
    b->match_off.assign(nq + 1, 0);
    std::vector<uint32_t> t;
    std::vector<std::pair<uint32_t, uint32_t>> tf; // (term, query term frequency)
    const bool split_ok = conj && !(op & DS2I_OP_REFERENCE_ORDER);
    double total_cost[NCLS] = {};
    for (uint32_t q = 0; q < nq; ++q) {
        if (query_offsets[q + 1] < query_offsets[q])
            return ds2i_set_error(DS2I_EINVAL, "query_offsets must be non-decreasing");
        t.assign(terms + query_offsets[q], terms + query_offsets[q + 1]);
        std::sort(t.begin(), t.end()); // queries.hpp:31 / 139
        tf.clear();
        for (size_t i = 0; i < t.size(); ++i) {
            if (t[i] >= idx->size) return ds2i_set_error(DS2I_ETERM, "term id out of range");
            if (i == 0 || t[i] != t[i - 1]) tf.emplace_back(t[i], 1u);
            else tf.back().second += 1;
        }
        if (tf.size() > DS2I_HIP_MAX_TERMS) return ds2i_set_error(DS2I_ETOOLONG, "query has more than 16 distinct terms");
        const size_t begin = qterms.size();
        for (auto const& p : tf) {
            QTerm qt;
            qt.list_off = idx->list_off[p.first];
            qt.list_end = idx->list_end[p.first];
            qt.n = idx->list_n[p.first];
            qt.term = idx->kind >= DS2I_OPT ? idx->list_nb[p.first] : p.first;
            qt.aux0 = idx->kind >= DS2I_OPT ? idx->list_aux0[p.first] : idx->list_blk_base[p.first];
            qt.aux1 = idx->kind >= DS2I_OPT ? idx->list_aux1[p.first] : 0;
            qt.q_weight = 0.f;
            qt.max_weight = 0.f;
            if (ranked) {
                qt.q_weight = ds2i_host::bm25::query_term_weight(p.second, qt.n, idx->num_docs);
                qt.max_weight = qt.q_weight * idx->max_term_weight[p.first];
            }
            qterms.push_back(qt);
            qnbs.push_back(idx->list_nb[p.first]);
        }
        double cost = 0;
        if (conj) { // sort by increasing frequency (queries.hpp:53-56, 357-360)
            std::vector<size_t> perm(qterms.size() - begin);
            for (size_t i = 0; i < perm.size(); ++i) perm[i] = begin + i;
            std::stable_sort(perm.begin(), perm.end(), [&](size_t l, size_t r) { return qterms[l].n < qterms[r].n; });
            std::vector<QTerm> tq;
            std::vector<uint32_t> tn;
            for (size_t i : perm) { tq.push_back(qterms[i]); tn.push_back(qnbs[i]); }
            std::copy(tq.begin(), tq.end(), qterms.begin() + begin);
            std::copy(tn.begin(), tn.end(), qnbs.begin() + begin);
            if (!tf.empty()) {
                const double n0 = qterms[begin].n;
                qnb0[q] = qnbs[begin];
                cost = qnb0[q] * (ranked ? 2.0 : 1.0);
                for (size_t i = begin + 1; i < qterms.size(); ++i) cost += std::min<double>(qnbs[i], n0);
                b->match_off[q + 1] = 128ull * qnb0[q];
            }
        } else {
            for (size_t i = begin; i < qterms.size(); ++i) cost += qnbs[i] * (ranked ? 2.0 : 1.0);
        }
        qoff[q + 1] = (uint32_t)qterms.size();
        qcost[q] = cost;
        total_cost[class_of(tf.size())] += cost;
    }
    for (uint32_t q = 0; q < nq; ++q) b->match_off[q + 1] += b->match_off[q];

    // ---- work units: long conjunctive queries are split by block ranges of their shortest list so
    // that one giant query does not pin a single wavefront (SURVEY.md §7 "Load imbalance")
    std::vector<std::pair<double, uint32_t>> cls[NCLS];
    b->q_unit_off.assign(nq + 1, 0);
    std::vector<uint32_t> split_queries, single_queries;
    // ranked_or takes the seed only in its block-synchronous form: its reference-order traversal stays the unpruned
    // exhaustive OR of queries.hpp:404-476 (the oracle the reference tests wand / maxscore against)
    const bool seeded = nq && (base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE ||
                               (base_op == DS2I_OP_RANKED_OR && !(op & DS2I_OP_REFERENCE_ORDER)));
    double all_cost = 0;
    for (double c : total_cost) all_cost += c;
    const double resident = idx->num_cus * 24.0; // waves the concurrent kernels share
    // units per resident wave (tuning knob, DS2I_UNIT_FACTOR): more = better tail balance, more per-unit overhead
    const char* uf = std::getenv("DS2I_UNIT_FACTOR");
    const double unit_factor = uf && std::atof(uf) > 0 ? std::atof(uf) : 16.0;
    for (uint32_t q = 0; q < nq; ++q) {
        const uint32_t nt = qoff[q + 1] - qoff[q];
        const int c = class_of(nt);
        // multi-list units are latency-bound chains (non-sequential probes): cut them finer so the tail stays parallel
        const double target = std::max(48.0, all_cost / (unit_factor * resident) / (c == 0 ? 1.0 : 4.0));
        if (seeded && nt == 1) { // one list: wand == maxscore == ranked_and, answered by the (block-synchronous) seed pass
            single_queries.push_back(q);
            ++b->nqcls[c];
            b->q_unit_off[q + 1] = (uint32_t)b->units.size();
            continue;
        }
        if (conj) {
            uint32_t parts = 1;
            if (split_ok && nt && qnb0[q] > 1) {
                double want = std::floor(qcost[q] / target);
                parts = (uint32_t)std::min<double>(std::max(1.0, want), qnb0[q]);
            }
            const uint32_t nb0 = std::max(1u, qnb0[q]);
            const uint32_t per = (nb0 + parts - 1) / parts;
            parts = (nb0 + per - 1) / per;
            if (parts > 1) split_queries.push_back(q);
            ++b->nqcls[c];
            for (uint32_t j = 0; j < parts; ++j) {
                Unit u;
                u.q = q;
                u.blk_begin = j * per;
                u.blk_end = std::min(nb0, (j + 1) * per);
                u.nparts = parts;
                cls[c].emplace_back(qcost[q] / parts, (uint32_t)b->units.size());
                b->units.push_back(u);
            }
        } else {
            // or / ranked_or / wand / maxscore: units are equal-width doc-id ranges; every part keeps its own
            // top-k (its own pruning threshold), the merge is exact
            const uint32_t N = (uint32_t)idx->num_docs;
            uint32_t parts = 1;
            // every part re-seeks its lists (the parts of a query share their pruning floor through q_floor), and the
            // many-list classes pay that per list: they want coarser parts than the one/two-list class (measured on
            // the GOV2-scale batch)
            static const double disj_scale[NCLS] = {8.0, 2.0, 1.0, 1.0};
            const double dtarget = std::max(48.0, all_cost / (unit_factor * disj_scale[c] * resident));
            if (nt && N > 1) parts = (uint32_t)std::min<double>(std::max(1.0, std::floor(qcost[q] / dtarget)), std::min<double>(N, 1024.0));
            const uint32_t width = (N + parts - 1) / parts;
            parts = width ? (N + width - 1) / width : 1;
            if (parts > 1) split_queries.push_back(q);
            ++b->nqcls[c];
            for (uint32_t j = 0; j < parts; ++j) {
                Unit u;
                u.q = q;
                u.blk_begin = j * width;
                u.blk_end = (uint32_t)std::min<uint64_t>(N, (uint64_t)(j + 1) * width);
                u.nparts = parts;
                cls[c].emplace_back(qcost[q] / parts, (uint32_t)b->units.size());
                b->units.push_back(u);
            }
        }
        b->q_unit_off[q + 1] = (uint32_t)b->units.size();
    }
    b->nunits = (uint32_t)b->units.size();
    b->nsplit = (uint32_t)split_queries.size();

    HIP_OK(hipSetDevice(idx->device));
    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIP_OK(upload((void**)&b->d_qterms, qterms.data(), qterms.size() * sizeof(QTerm)));
    HIP_OK(upload((void**)&b->d_qoff, qoff.data(), qoff.size() * 4));
    HIP_OK(upload((void**)&b->d_units, b->units.data(), b->units.size() * sizeof(Unit)));
    HIP_OK(upload((void**)&b->d_q_unit_off, b->q_unit_off.data(), b->q_unit_off.size() * 4));
    HIP_OK(upload((void**)&b->d_split, split_queries.data(), split_queries.size() * 4));
    for (int c = 0; c < NCLS; ++c) {
        std::stable_sort(cls[c].begin(), cls[c].end(),
                         [](auto const& l, auto const& r) { return l.first > r.first; }); // costliest first
        std::vector<uint32_t> order(cls[c].size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = cls[c][i].second;
        b->ncls[c] = (uint32_t)order.size();
        HIP_OK(upload((void**)&b->d_order[c], order.data(), order.size() * 4));
    }
    const size_t nu = b->nunits ? b->nunits : 1;
    HIP_OK(hipMalloc((void**)&b->d_unit_count, 8 * nu));
    HIP_OK(hipMalloc((void**)&b->d_unit_topk, 4 * nu * k));
    HIP_OK(hipMalloc((void**)&b->d_unit_topk_len, 4 * nu));
    HIP_OK(hipMalloc((void**)&b->d_unit_freq_sum, 8 * nu));
    HIP_OK(hipMemset(b->d_unit_count, 0, 8 * nu));
    HIP_OK(hipMalloc((void**)&b->d_count, 8 * (size_t)(nq ? nq : 1)));
    HIP_OK(hipMalloc((void**)&b->d_topk, 4 * (size_t)(nq ? nq : 1) * k));
    HIP_OK(hipMalloc((void**)&b->d_topk_len, 4 * (size_t)(nq ? nq : 1)));
    HIP_OK(hipMalloc((void**)&b->d_freq_sum, 8 * (size_t)(nq ? nq : 1)));
    HIP_OK(hipMemset(b->d_topk_len, 0, 4 * (size_t)(nq ? nq : 1)));
    if (b->want_matches) {
        HIP_OK(hipMalloc((void**)&b->d_matches, 4 * (size_t)(b->match_off[nq] ? b->match_off[nq] : 1)));
        HIP_OK(upload((void**)&b->d_match_off, b->match_off.data(), b->match_off.size() * 8));
    }
    if (disj_topk && !(op & DS2I_OP_REFERENCE_ORDER)) HIP_OK(hipMalloc((void**)&b->d_qfloor, 4 * (size_t)(nq ? nq : 1)));
    if (seeded) {
        // The seed is the ranked_and top-k of a SUB-query: any k documents' partial scores bound the final k-th
        // score from below. One- and two-term queries use all their terms (the one-term answer is final); longer
        // queries use their two shortest lists -- the full conjunction of 5+ terms is usually too small to give k
        // documents, while the rarest pair is cheap to intersect and carries the largest term weights.
        const char* sv = std::getenv("DS2I_SEED_TERMS");
        const size_t seed_terms = sv && std::atoi(sv) > 0 ? (size_t)std::atoi(sv) : 2;
        std::vector<uint32_t> sterms, soffs(nq + 1, 0);
        std::vector<uint32_t> dt;
        for (uint32_t q = 0; q < nq; ++q) {
            const uint32_t* qb = terms + query_offsets[q];
            const uint32_t* qe = terms + query_offsets[q + 1];
            dt.assign(qb, qe);
            std::sort(dt.begin(), dt.end());
            dt.erase(std::unique(dt.begin(), dt.end()), dt.end());
            if (dt.size() > seed_terms && dt.size() > 2) {
                std::stable_sort(dt.begin(), dt.end(), [&](uint32_t x, uint32_t y) { return idx->list_n[x] < idx->list_n[y]; });
                dt.resize(std::max<size_t>(2, seed_terms));
                for (const uint32_t* p = qb; p != qe; ++p) // keep multiplicities: the query term weight counts them
                    if (std::find(dt.begin(), dt.end(), *p) != dt.end()) sterms.push_back(*p);
            } else {
                sterms.insert(sterms.end(), qb, qe);
            }
            soffs[q + 1] = (uint32_t)sterms.size();
        }
        int rc = ds2i_hip_batch_prepare(idx, DS2I_OP_RANKED_AND, k, sterms.data(), soffs.data(), nq, 0, &b->seed);
        if (rc) return rc;
        b->nsingle = (uint32_t)single_queries.size();
        HIP_OK(upload((void**)&b->d_single, single_queries.data(), single_queries.size() * 4));
    }
    *out = b.release();
    return DS2I_OK;
}

int ds2i_hip_batch_run(ds2i_hip_batch* b, ds2i_hip_stats* stats) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_run: null batch");
    ds2i_hip_index* idx = b->idx;
    HIP_OK(hipSetDevice(idx->device));
    double seed_ms = 0;
    if (b->seed) { // block-synchronous ranked_and first: its k-th score seeds the pruning floor of every unit
        b->seed->instrument = b->instrument;
        ds2i_hip_stats ss;
        int rc = ds2i_hip_batch_run(b->seed, &ss);
        if (rc) return rc;
        seed_ms = ss.kernel_ms;
    }
    hipStream_t s0 = idx->stream[0];
    HIP_OK(hipMemsetAsync(idx->d_stats, 0, NCLS * sizeof(Stats), s0));
    if (b->d_qfloor) HIP_OK(hipMemsetAsync(b->d_qfloor, 0, 4 * (size_t)(b->nq ? b->nq : 1), s0));
    // ev[0] start (s0); class c kernel on stream c between ev[1+2c], ev[2+2c]; ev[1+2*NCLS] end (s0).
    HIP_OK(hipEventRecord(idx->ev[0], s0));
    // Launch order of the four class kernels (they overlap on separate streams either way; measured on the GOV2-scale
    // batch): the block-synchronous conjunctions run 3 % faster when the issue-bound <=2-list class is enqueued first,
    // the disjunctive operators 2.5 % faster when the many-list classes are.
    const int base_op_run = b->op & ~DS2I_OP_REFERENCE_ORDER;
    const bool small_first = !(b->op & DS2I_OP_REFERENCE_ORDER) &&
                             (base_op_run == DS2I_OP_AND || base_op_run == DS2I_OP_AND_FREQ || base_op_run == DS2I_OP_RANKED_AND);
    for (int ci = NCLS - 1; ci >= 0; --ci) {
        const int c = small_first ? NCLS - 1 - ci : ci;
        hipStream_t s = idx->stream[c];
        if (c) HIP_OK(hipStreamWaitEvent(s, idx->ev[0], 0));
        HIP_OK(hipEventRecord(idx->ev[1 + 2 * c], s));
        if (b->ncls[c]) {
            BatchArgs a{};
            a.arena = idx->d_arena;
            a.bits0 = idx->d_bits0;
            a.bits1 = idx->d_bits1;
            a.norm_lens = idx->d_norm_lens;
            a.qterms = b->d_qterms;
            a.q_off = b->d_qoff;
            a.units = b->d_units;
            a.order = b->d_order[c];
            a.nslice = b->ncls[c];
            a.num_docs = (uint32_t)idx->num_docs;
            a.k = b->k;
            a.codec = idx->kind >= DS2I_OPT ? (int)DS2I_OPT : idx->kind; // every freq_index layout decodes through the chunk directory
            a.ticket = idx->d_ticket + c;
            a.out_count = b->d_count;
            a.out_topk = b->d_topk;
            a.out_topk_len = b->d_topk_len;
            a.out_freq_sum = b->d_freq_sum;
            a.out_matches = b->want_matches ? b->d_matches : nullptr;
            a.match_off = b->d_match_off;
            a.unit_count = b->d_unit_count;
            a.unit_topk = b->d_unit_topk;
            a.unit_topk_len = b->d_unit_topk_len;
            a.unit_freq_sum = b->d_unit_freq_sum;
            a.seed_topk = b->seed ? b->seed->d_topk : nullptr;
            a.seed_len = b->seed ? b->seed->d_topk_len : nullptr;
            a.q_floor = b->d_qfloor;
            a.block_profile = b->instrument ? b->d_prof : nullptr;
            a.skip = std::getenv("DS2I_NO_SKIPTAB") ? nullptr : idx->d_skip;
            a.stats = b->instrument ? idx->d_stats + c : nullptr;
            HIP_OK(ds2i_launch_batch(b->op, c, &a, b->ncls[c], s));
        }
        HIP_OK(hipEventRecord(idx->ev[2 + 2 * c], s));
    }
    for (int c = 1; c < NCLS; ++c) HIP_OK(hipStreamWaitEvent(s0, idx->ev[2 + 2 * c], 0));
    if (b->nsplit) {
        MergeArgs m{};
        m.split_queries = b->d_split;
        m.nsplit = b->nsplit;
        m.q_unit_off = b->d_q_unit_off;
        m.k = b->k;
        m.ranked = (b->op & 0xFF) >= DS2I_OP_RANKED_AND;
        m.unit_count = b->d_unit_count;
        m.unit_topk = b->d_unit_topk;
        m.unit_topk_len = b->d_unit_topk_len;
        m.unit_freq_sum = b->d_unit_freq_sum;
        m.out_count = b->d_count;
        m.out_topk = b->d_topk;
        m.out_topk_len = b->d_topk_len;
        m.out_freq_sum = b->d_freq_sum;
        HIP_OK(ds2i_launch_merge(&m, std::min<unsigned>(b->nsplit, 4096u), s0));
    }
    if (b->seed && b->nsingle)
        HIP_OK(ds2i_launch_copy_seed(b->d_single, b->nsingle, b->k, b->seed->d_topk, b->seed->d_topk_len, b->seed->d_count, b->d_topk,
                                     b->d_topk_len, b->d_count, s0));
    HIP_OK(hipEventRecord(idx->ev[1 + 2 * NCLS], s0));
    HIP_OK(hipStreamSynchronize(s0));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, idx->ev[0], idx->ev[1 + 2 * NCLS]));
    for (int c = 0; c < NCLS; ++c) HIP_OK(hipEventElapsedTime(&b->cls_ms[c], idx->ev[1 + 2 * c], idx->ev[2 + 2 * c]));
    if (b->instrument) HIP_OK(hipMemcpy(b->cls_stats, idx->d_stats, NCLS * sizeof(Stats), hipMemcpyDeviceToHost));
    if (stats) {
        stats->kernel_ms = ms + seed_ms;
        stats->docs_blocks_decoded = stats->freqs_blocks_decoded = stats->block_max_examined = 0;
        stats->algorithmic_bytes = stats->postings_scored = stats->rounds = 0;
        for (int c = 0; c < NCLS; ++c) {
            stats->docs_blocks_decoded += b->cls_stats[c].docs_blocks;
            stats->freqs_blocks_decoded += b->cls_stats[c].freqs_blocks;
            stats->block_max_examined += b->cls_stats[c].block_max_examined;
            stats->algorithmic_bytes += b->cls_stats[c].algorithmic_bytes;
            stats->postings_scored += b->cls_stats[c].postings_scored;
            stats->rounds += b->cls_stats[c].rounds;
        }
    }
    return DS2I_OK;
}

// GPU-side counterpart of profile_queries.cpp: per-block decode counts of the batch (input of the block_mixed
// optimiser, ds2i_hybrid_*). Counting happens in instrumented runs only and accumulates over runs.
int ds2i_hip_batch_enable_block_profile(ds2i_hip_batch* b) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_enable_block_profile: null batch");
    ds2i_hip_index* idx = b->idx;
    if (idx->kind >= DS2I_OPT) return ds2i_set_error(DS2I_EINVAL, "the block access profile exists for block indexes only");
    HIP_OK(hipSetDevice(idx->device));
    const size_t bytes = 8 * (size_t)(idx->total_blocks ? idx->total_blocks : 1);
    if (!b->d_prof) HIP_OK(hipMalloc((void**)&b->d_prof, bytes));
    HIP_OK(hipMemset(b->d_prof, 0, bytes));
    if (b->seed) { // the seed pass decodes blocks too; it shares the buffer (freed by the owner only)
        if (b->seed->d_prof && b->seed->d_prof != b->d_prof) (void)hipFree(b->seed->d_prof);
        b->seed->d_prof = b->d_prof;
    }
    return DS2I_OK;
}
int ds2i_hip_batch_block_profile(ds2i_hip_batch* b, uint32_t* counts, uint64_t capacity, uint64_t* total_blocks) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_block_profile: null batch");
    ds2i_hip_index* idx = b->idx;
    if (total_blocks) *total_blocks = idx->total_blocks;
    if (!counts) return DS2I_OK;
    if (!b->d_prof) return ds2i_set_error(DS2I_EINVAL, "block profile not enabled on this batch");
    if (capacity < 2 * idx->total_blocks) return ds2i_set_error(DS2I_EINVAL, "counts buffer too small (2 per block)");
    HIP_OK(hipSetDevice(idx->device));
    HIP_OK(hipMemcpy(counts, b->d_prof, 8 * (size_t)idx->total_blocks, hipMemcpyDeviceToHost));
    return DS2I_OK;
}

int ds2i_hip_batch_set_instrumented(ds2i_hip_batch* b, int on) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_set_instrumented: null batch");
    b->instrument = on != 0;
    return DS2I_OK;
}

// per kernel-class timing / bytes of the last run (class 0: <=4 distinct terms, class 1: 5..16)
int ds2i_hip_batch_class_stats(ds2i_hip_batch* b, int cls, ds2i_hip_stats* out, uint32_t* nqueries) {
    if (!b || !out || cls < 0 || cls >= NCLS) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_class_stats: bad argument");
    out->kernel_ms = b->cls_ms[cls];
    out->docs_blocks_decoded = b->cls_stats[cls].docs_blocks;
    out->freqs_blocks_decoded = b->cls_stats[cls].freqs_blocks;
    out->block_max_examined = b->cls_stats[cls].block_max_examined;
    out->algorithmic_bytes = b->cls_stats[cls].algorithmic_bytes;
    out->postings_scored = b->cls_stats[cls].postings_scored;
    out->rounds = b->cls_stats[cls].rounds;
    if (nqueries) *nqueries = b->nqcls[cls];
    return DS2I_OK;
}

// diagnostic: phase cycle sums of class `cls` (all zero unless built with -DDS2I_PHASE_TIMING)
int ds2i_hip_batch_phase_cycles(ds2i_hip_batch* b, int cls, uint64_t* out, int n) {
    if (!b || !out || cls < 0 || cls >= NCLS) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_phase_cycles: bad argument");
    for (int i = 0; i < n && i < ds2i_dev::PH_COUNT; ++i) out[i] = b->cls_stats[cls].phase_cycles[i];
    return DS2I_OK;
}

int ds2i_hip_batch_fetch(ds2i_hip_batch* b, uint64_t* out_count, float* out_topk, uint32_t* out_topk_len,
                         uint64_t* out_freq_sum) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_fetch: null batch");
    HIP_OK(hipSetDevice(b->idx->device));
    const size_t nq = b->nq;
    if (!nq) return DS2I_OK;
    if (out_count) HIP_OK(hipMemcpy(out_count, b->d_count, 8 * nq, hipMemcpyDeviceToHost));
    if (out_topk) HIP_OK(hipMemcpy(out_topk, b->d_topk, 4 * nq * b->k, hipMemcpyDeviceToHost));
    if (out_topk_len) HIP_OK(hipMemcpy(out_topk_len, b->d_topk_len, 4 * nq, hipMemcpyDeviceToHost));
    if (out_freq_sum) HIP_OK(hipMemcpy(out_freq_sum, b->d_freq_sum, 8 * nq, hipMemcpyDeviceToHost));
    return DS2I_OK;
}

int ds2i_hip_batch_match_total(ds2i_hip_batch* b, uint64_t* total) {
    if (!b || !total) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_match_total: null argument");
    *total = b->want_matches ? b->match_off[b->nq] : 0; // capacity: 128 per block of each query's shortest list
    return DS2I_OK;
}

// On return matches of query q occupy [match_offsets[q], match_offsets[q] + out_count[q]); the device
// buffer holds one segment per work unit (at 128*blk_begin), compacted here on the host.
int ds2i_hip_batch_fetch_matches(ds2i_hip_batch* b, uint64_t* match_offsets, uint32_t* matches) {
    if (!b || !match_offsets || !matches) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_fetch_matches: null argument");
    if (!b->want_matches) return ds2i_set_error(DS2I_EINVAL, "batch was prepared without want_matches");
    HIP_OK(hipSetDevice(b->idx->device));
    for (size_t i = 0; i <= b->nq; ++i) match_offsets[i] = b->match_off[i];
    const size_t total = (size_t)b->match_off[b->nq];
    if (!total) return DS2I_OK;
    HIP_OK(hipMemcpy(matches, b->d_matches, 4 * total, hipMemcpyDeviceToHost));
    if (b->nsplit) {
        std::vector<unsigned long long> ucount(b->nunits);
        HIP_OK(hipMemcpy(ucount.data(), b->d_unit_count, 8 * (size_t)b->nunits, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < b->nq; ++q) {
            const uint32_t u0 = b->q_unit_off[q], u1 = b->q_unit_off[q + 1];
            if (u1 - u0 < 2) continue;
            uint32_t* base = matches + b->match_off[q];
            size_t w = 0;
            for (uint32_t u = u0; u < u1; ++u) {
                const uint32_t* seg = base + 128ull * b->units[u].blk_begin;
                std::memmove(base + w, seg, 4 * (size_t)ucount[u]);
                w += (size_t)ucount[u];
            }
        }
    }
    return DS2I_OK;
}

int ds2i_hip_query_batch(ds2i_hip_index* idx, int op, uint32_t k, const uint32_t* terms,
                         const uint32_t* query_offsets, uint32_t nq, uint64_t* out_count, float* out_topk,
                         uint32_t* out_topk_len, ds2i_hip_stats* stats) {
    ds2i_hip_batch* b = nullptr;
    int rc = ds2i_hip_batch_prepare(idx, op, k, terms, query_offsets, nq, 0, &b);
    if (rc) return rc;
    rc = ds2i_hip_batch_run(b, stats);
    if (!rc) rc = ds2i_hip_batch_fetch(b, out_count, out_topk, out_topk_len, nullptr);
    ds2i_hip_batch_free(b);
    return rc;
}

// profiling aid: one pass over the whole index arena with the decoders' load shape (4 B per lane); the
// bytes read are returned so that rocprofv3's FETCH_SIZE can be calibrated against a known count
int ds2i_hip_calibration_read(ds2i_hip_index* idx, uint64_t* bytes_read) {
    if (!idx || !bytes_read) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_calibration_read: null argument");
    HIP_OK(hipSetDevice(idx->device));
    const unsigned long long ndw = idx->arena_bytes / 4;
    HIP_OK(ds2i_launch_calib_read((const uint32_t*)idx->d_arena, ndw, (uint32_t*)idx->d_ticket, idx->num_cus * 32, idx->stream[0]));
    HIP_OK(hipStreamSynchronize(idx->stream[0]));
    *bytes_read = ndw * 4;
    return DS2I_OK;
}

int ds2i_hip_selftest_scan(int device, const uint32_t* in, uint32_t* out, uint32_t rows) {
    if (!in || !out || !rows) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_selftest_scan: bad argument");
    if (device < 0 || device >= ds2i_hip_device_count()) return ds2i_set_error(DS2I_EDEVICE, "no such HIP device");
    HIP_OK(hipSetDevice(device));
    uint32_t *di = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc((void**)&di, 256 * rows));
    HIP_OK(hipMalloc((void**)&dout, 256 * rows));
    hipError_t e = hipMemcpy(di, in, 256 * rows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ds2i_launch_selftest(di, dout, rows, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, 256 * rows, hipMemcpyDeviceToHost);
    (void)hipFree(di);
    (void)hipFree(dout);
    if (e != hipSuccess) return ds2i_set_error(DS2I_EDEVICE, hipGetErrorString(e));
    return DS2I_OK;
}

} // extern "C"
