// Host implementation of include/ds2i_hip.h, index half: validation and upload of the index / wand images, the
// upload-time auxiliary tables (list offsets, interleaved skip table, chunk directory, per-block max BM25 weights),
// list decode. Query batches live in capi_batch.cpp, the device code in kernels.hip. There is NO CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "capi_internal.hpp"
#include "knobs.hpp"
#include "../../include/ds2i_build.h"
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
hipError_t ds2i_launch_decode_list(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_decode_list_side(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_block_max_weights(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_list_top_bmw(const float* bmw, const void* lists, uint32_t nlists, float* out, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_build_side_tables(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_calib_read(const uint32_t* base, unsigned long long ndw, uint32_t* out, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_selftest(const uint32_t* in, uint32_t* out, unsigned blocks, hipStream_t s);
hipError_t ds2i_launch_selftest_bm25(const uint32_t* freqs, const float* norm_lens, float* out, uint32_t n, hipStream_t s);
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
    if (x->oneshot) ds2i_batch_destroy(x->oneshot);
    if (x->d_bmw) (void)hipFree(x->d_bmw);
    if (x->d_rmw) (void)hipFree(x->d_rmw);
    if (x->d_rmh) (void)hipFree(x->d_rmh);
    if (x->d_xslots) (void)hipFree(x->d_xslots);
    if (x->d_xovf) (void)hipFree(x->d_xovf);
    if (x->d_tails) (void)hipFree(x->d_tails);
    if (x->d_ticket) (void)hipFree(x->d_ticket);
    for (auto& s : x->stream) if (s) (void)hipStreamDestroy(s);
    for (auto& s : x->stream_alt) if (s) (void)hipStreamDestroy(s);
    if (x->s_up) (void)hipStreamDestroy(x->s_up);
    if (x->s_merge) (void)hipStreamDestroy(x->s_merge);
    delete x;
}

// device temporaries of one function: freed on every path out of it
struct DevTemps {
    std::vector<void*> p;
    ~DevTemps() { for (void* x : p) if (x) (void)hipFree(x); }
    template <class T> hipError_t alloc(T** out, size_t bytes) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, bytes ? bytes : 4);
        if (e == hipSuccess) p.push_back(q);
        *out = (T*)q;
        return e;
    }
};

// Per-block (per-chunk) maximum of bm25::doc_term_weight over the block's postings -- the block-level analogue of
// wand_data's max_term_weight (wand_data.hpp:40-52), computed ON THE DEVICE with the kernels' own float32 arithmetic:
// one pass decodes every block of the index (k_block_max_weights, <=64 blocks of one list per wave). ranked_and uses
// it as an exact upper bound to skip blocks and windows that cannot enter the heap (kernels.hip, k_conjunctive).
// A second pass of the same shape fills the doc-id-range tables (BatchArgs::rmw): it needs every list's maximum, which
// the first pass delivers.
int build_block_max_weights(ds2i_hip_index* x) {
    const uint64_t V = x->size;
    std::vector<QTerm> lists(V);
    std::vector<ds2i_dev::BmwItem> items;
    items.reserve(x->total_blocks / 64 + V);
    for (uint64_t t = 0; t < V; ++t) {
        lists[t] = ds2i_make_qterm(x, (uint32_t)t);
        for (uint32_t b = 0; b < x->list_nb[t]; b += 64) items.push_back(ds2i_dev::BmwItem{(uint32_t)t, b});
    }
    DevTemps tmp;
    QTerm* d_lists = nullptr;
    ds2i_dev::BmwItem* d_items = nullptr;
    unsigned int* d_lmax = nullptr;
    float* d_top = nullptr;
    HIP_OK(hipMalloc((void**)&x->d_bmw, 4 * x->total_blocks)); // (owned by the handle: free_index releases it on failure)
    HIP_OK(tmp.alloc(&d_lists, sizeof(QTerm) * V));
    HIP_OK(tmp.alloc(&d_items, sizeof(ds2i_dev::BmwItem) * items.size()));
    HIP_OK(tmp.alloc(&d_lmax, 4 * V));
    HIP_OK(tmp.alloc(&d_top, 4 * (size_t)DS2I_HIP_MAX_K * V));
    HIP_OK(hipMemcpy(d_lists, lists.data(), sizeof(QTerm) * V, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_items, items.data(), sizeof(ds2i_dev::BmwItem) * items.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_lmax, 0, 4 * V));
    ds2i_dev::BmwArgs a{};
    a.arena = x->d_arena;
    a.bits0 = x->d_bits0;
    a.bits1 = x->d_bits1;
    a.norm_lens = x->d_norm_lens;
    a.lists = d_lists;
    a.items = d_items;
    a.nitems = (uint32_t)items.size();
    a.codec = x->kind >= DS2I_OPT ? (int)DS2I_OPT : x->kind;
    a.num_docs = (uint32_t)x->num_docs;
    a.bmw = x->d_bmw;
    a.list_bmw = d_lmax;
    a.rmw = nullptr;
    const unsigned grid = (unsigned)std::min<size_t>(items.size(), (size_t)x->num_cus * 64);
    HIP_OK(ds2i_launch_block_max_weights(&a, grid, x->stream[0]));
    HIP_OK(hipStreamSynchronize(x->stream[0]));
    x->list_bmw.assign(V, 0.f);
    HIP_OK(hipMemcpy(x->list_bmw.data(), d_lmax, 4 * V, hipMemcpyDeviceToHost));
    // the DS2I_HIP_MAX_K largest block weights of every list: a one-term ranked query knows k documents reaching
    // q_weight * (k-th largest) before it decodes anything
    HIP_OK(ds2i_launch_list_top_bmw(x->d_bmw, d_lists, (uint32_t)V, d_top, (unsigned)std::min<uint64_t>(V, (uint64_t)x->num_cus * 64), x->stream[0]));
    HIP_OK(hipStreamSynchronize(x->stream[0]));
    x->list_topbmw.assign((size_t)DS2I_HIP_MAX_K * V, 0.f);
    HIP_OK(hipMemcpy(x->list_topbmw.data(), d_top, 4 * (size_t)DS2I_HIP_MAX_K * V, hipMemcpyDeviceToHost));
    x->extra_bytes += 4 * x->total_blocks;

    // ---- doc-id-range tables. Granularity per list: the largest power of two of doc-ids per entry that still gives
    // the list at least G entries per posting (G = DS2I_RMW_G, default 4; 0 = no tables): a long list gets fine ranges
    // (a 6 M-posting list of a 25 M-doc collection: one doc-id per byte), a short one coarse ranges of about the same
    // number of bytes per posting -- 1..2 G bytes per posting altogether, every table padded to 64 bytes. Measured on
    // the GOV2-scale ranked_and batch (queries/s): G = 1: 335 k, 2: 460 k, 4: 503 k, 8: 474 k.
    const double G = x->plan_g; // (choose_table_plan: DS2I_RMW_G, default 4; DS2I_TABLE_BUDGET may have lowered it)
    if (!(G > 0)) return DS2I_OK; // asked not to build them
    x->rmw_g = (int)G;
    // The tables are an accelerator, never a reason to fail an upload -- but running without them changes which kernel
    // family answers wand / maxscore / ranked_or and costs ranked_and and `and` most of their speed, so the decision is
    // never silent: it is reported through ds2i_hip_index_get_info(), said once on stderr, and DS2I_RMW_REQUIRE=1 turns
    // it into DS2I_ENOMEM.
    auto without_tables = [&](const char* why, uint64_t want) -> int {
        x->list_rmw_off64.clear();
        x->list_rmw_shift.clear();
        if (ds2i_knobs().rmw_require) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "doc-id-range tables (%.2f GB) cannot be built: %s (DS2I_RMW_REQUIRE is set)", want / 1e9, why);
            return ds2i_set_error(DS2I_ENOMEM, msg);
        }
        std::fprintf(stderr, "ds2i_hip: index uploaded WITHOUT doc-id-range tables (%.2f GB wanted: %s); ranked_and / and / wand / "
                             "maxscore / or take their slower table-free kernels\n", want / 1e9, why);
        return DS2I_OK;
    };
    x->list_rmw_off64.assign(V, 0);
    x->list_rmw_shift.assign(V, 0);
    uint64_t cursor = 0; // in units of 64 bytes
    for (uint64_t t = 0; t < V; ++t) {
        uint32_t sh = 0;
        while (sh < 31 && (double)(x->num_docs >> (sh + 1)) >= G * (double)x->list_n[t]) ++sh;
        x->list_rmw_shift[t] = sh;
        x->list_rmw_off64[t] = (uint32_t)cursor;
        cursor += ds2i_dev::RmwLevels((uint32_t)x->num_docs, sh).bytes() / 64; // level 1 + its two coarser levels
        if (ds2i_dev::RmwLevels::has_bitmap(x->list_n[t], (uint32_t)x->num_docs) && !ds2i_knobs().no_bitmaps)
            cursor += ds2i_dev::RmwLevels::bitmap_bytes((uint32_t)x->num_docs) / 64; // dense list: + its exact bitmap
        if (cursor >= (1ull << 32)) return without_tables("more than 256 GB of tables", cursor * 64);
    }
    const uint64_t bytes = cursor * 64 + 64;
    {
        // several replicas may be uploaded to one device from parallel host threads (gpu_index_set): the free-memory test
        // and the allocation it guards are one critical section, and an allocation that fails all the same is not an error
        static std::mutex table_alloc_mu;
        std::lock_guard<std::mutex> g(table_alloc_mu);
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        if (bytes > free_b / 2) return without_tables("less than twice their size is free on the device", bytes);
        if (hipMalloc((void**)&x->d_rmw, bytes) != hipSuccess) {
            (void)hipGetLastError();
            x->d_rmw = nullptr;
            return without_tables("hipMalloc failed", bytes);
        }
    }
    x->rmw_bytes = bytes;
    // membership hints (abi_structs.hpp, BatchArgs::rmh): a parallel buffer with the tables' offsets (only the level-1 regions are written). Optional like the tables.
    if (x->plan_hints) { // (every index kind: the ranked / and / wand kernels of all of them consult the hints)
        static std::mutex hint_alloc_mu;
        std::lock_guard<std::mutex> g(hint_alloc_mu);
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        if (bytes <= free_b / 2 && hipMalloc((void**)&x->d_rmh, bytes) == hipSuccess) {
            HIP_OK(hipMemsetAsync(x->d_rmh, 0, bytes, x->stream[0]));
        } else {
            (void)hipGetLastError();
            x->d_rmh = nullptr;
            std::fprintf(stderr, "ds2i_hip: index uploaded without membership hints (%.2f GB wanted); ranked_and looks more candidates up\n", bytes / 1e9);
        }
    }
    HIP_OK(hipMemsetAsync(x->d_rmw, 0, bytes, x->stream[0]));
    for (uint64_t t = 0; t < V; ++t) {
        lists[t].rmw_off64 = x->list_rmw_off64[t];
        lists[t].rmw_shift = x->list_rmw_shift[t];
        lists[t].max_weight = x->list_bmw[t];
    }
    HIP_OK(hipMemcpyAsync(d_lists, lists.data(), sizeof(QTerm) * V, hipMemcpyHostToDevice, x->stream[0]));
    a.rmw = x->d_rmw;
    a.rmh = x->d_rmh;
    a.rmw_level = 0;
    a.bitmaps = ds2i_knobs().no_bitmaps ? 0u : 1u;
    x->has_bitmaps = a.bitmaps != 0;
    HIP_OK(ds2i_launch_block_max_weights(&a, grid, x->stream[0]));
    for (uint32_t lvl = 1; lvl <= 2; ++lvl) { // level lvl + 1 = maxima of 64 entries of level lvl, 4096 entries per item
        items.clear();
        for (uint64_t t = 0; t < V; ++t) {
            const ds2i_dev::RmwLevels g((uint32_t)x->num_docs, x->list_rmw_shift[t]);
            for (uint32_t e = 0; e < g.e[lvl]; e += 4096) items.push_back(ds2i_dev::BmwItem{(uint32_t)t, e});
        }
        ds2i_dev::BmwItem* d_it = nullptr;
        HIP_OK(tmp.alloc(&d_it, sizeof(ds2i_dev::BmwItem) * items.size()));
        HIP_OK(hipMemcpyAsync(d_it, items.data(), sizeof(ds2i_dev::BmwItem) * items.size(), hipMemcpyHostToDevice, x->stream[0]));
        HIP_OK(hipStreamSynchronize(x->stream[0])); // (`items` is reused by the next level)
        a.items = d_it;
        a.nitems = (uint32_t)items.size();
        a.rmw_level = lvl;
        HIP_OK(ds2i_launch_block_max_weights(&a, (unsigned)std::min<size_t>(items.size(), (size_t)x->num_cus * 64), x->stream[0]));
    }
    HIP_OK(hipStreamSynchronize(x->stream[0]));
    x->extra_bytes += bytes + (x->d_rmh ? bytes : 0);
    return DS2I_OK;
}

// Bytes of the doc-id-range tables (levels + dense lists' bitmaps) at G entries per posting; the hints are a buffer of the same size.
static uint64_t range_table_bytes_at(const ds2i_hip_index* x, double G) {
    uint64_t cursor = 0;
    const bool bitmaps = !ds2i_knobs().no_bitmaps;
    for (uint64_t t = 0; t < x->size; ++t) {
        uint32_t sh = 0;
        while (sh < 31 && (double)(x->num_docs >> (sh + 1)) >= G * (double)x->list_n[t]) ++sh;
        cursor += ds2i_dev::RmwLevels((uint32_t)x->num_docs, sh).bytes() / 64;
        if (bitmaps && ds2i_dev::RmwLevels::has_bitmap(x->list_n[t], (uint32_t)x->num_docs)) cursor += ds2i_dev::RmwLevels::bitmap_bytes((uint32_t)x->num_docs) / 64;
    }
    return cursor * 64 + 64;
}

// What an upload builds beside the image. Without a budget: range tables at DS2I_RMW_G (4) entries per posting, membership
// hints, exception side slots (block_optpfor) -- 29.9 GB resident for the 2.8 GB GOV2-scale index. DS2I_TABLE_BUDGET=<bytes>
// (or <factor>x, e.g. "5x" = five times the image) makes the upload pick, in the order of the rates measured at GOV2 scale
// (profiles/r05_table_budget.txt), the first configuration whose resident bytes stay under it: whole structures are dropped or
// the tables' granularity halved -- there is no per-list choice (a list either has every structure the index has or none does).
// A knob set explicitly (DS2I_RMW_G, DS2I_NO_RMH, DS2I_NO_XSLOTS) is not overridden. Reported by ds2i_hip_index_get_info.
// DS2I_TABLE_BUDGET in bytes ("<bytes>" or "<factor>x" of the CALLER's image -- for a transcoded upload the image handed to
// ds2i_hip_index_open, not its re-encoded form), 0 = none
static uint64_t table_budget_bytes(size_t image_bytes) {
    const Ds2iKnobs kn = ds2i_knobs();
    const char* eb = kn.table_budget;
    if (!*eb) return 0;
    char* end = nullptr;
    const double v = std::strtod(eb, &end);
    if (!(v > 0)) return 0;
    return (end && (*end == 'x' || *end == 'X')) ? (uint64_t)(v * (double)image_bytes) : (uint64_t)v;
}
static void choose_table_plan(ds2i_hip_index* x, size_t image_bytes) {
    const Ds2iKnobs kn = ds2i_knobs();
    x->plan_g = kn.rmw_g;
    if (!(x->plan_g > 0) || kn.no_rmw) x->plan_g = 0;
    x->plan_hints = !kn.no_rmh;
    x->plan_slots = !kn.no_xslots && x->kind == DS2I_BLOCK_OPTPFOR;
    x->table_budget = table_budget_bytes(image_bytes);
    if (!x->table_budget) return;
    const bool g_pinned = kn.rmw_g_set || kn.no_rmw; // (an explicit knob pins the granularity: DS2I_NO_RMW = none)
    // resident whatever is chosen: the image and its skip table (counted in extra_bytes by now), block weights (4 B per block), norm_lens
    const uint64_t base = x->arena_bytes + x->extra_bytes + 4ull * x->total_blocks + (x->has_wand ? 4 * x->num_docs : 0);
    // side tables = a slot per block + the lists' partial last blocks in plain form + the overflow area at its first-attempt size
    // (build_side_tables: the same arithmetic; the 256 KB floor of the overflow area is what a small index notices)
    uint64_t slots = 0;
    if (x->kind == DS2I_BLOCK_OPTPFOR) {
        uint64_t tail_dw = 0;
        for (uint64_t t = 0; t < x->size; ++t)
            if (x->list_n[t] & 127u) tail_dw += 2 * (x->list_n[t] & 127u) + 2;
        // (the overflow area is sized by the build pass itself -- blocks with more exceptions than a slot holds, raw and oversize
        // parts; 38 bytes per block on the GOV2-scale index: the plan reserves 40, and never less than the pass's first attempt)
        const uint64_t ovf = std::max<uint64_t>(4 * (x->total_blocks / 8 + 65536), 40 * x->total_blocks);
        slots = 4ull * ds2i_dev::XSLOT_DW * x->total_blocks + (4 * tail_dw + 1024) + ovf;
    }
    struct Plan { double g; bool hints, slots; };
    // (measured, GOV2 scale, ranked_and, queries/s at GB resident: {4, hints} 958 k at 29.9; {2, hints} 969 k at 19.6; {4, none} 886 k at 18.5;
    // {2, none} 767 k at 13.3; {1, none} without side slots 314 k at 6.7; side slots alone are worth 958 k against 510 k for 4 GB)
    static const Plan order[] = {{4, true, true}, {2, true, true}, {4, false, true}, {2, false, true}, {1, true, true}, {1, false, true},
                                 {2, false, false}, {1, false, false}, {0, false, false}};
    for (const Plan& c : order) {
        if (g_pinned && c.g != x->plan_g) continue;
        if (kn.no_rmh && c.hints) continue;
        if ((kn.no_xslots || !slots) && c.slots) continue;
        const uint64_t tables = c.g > 0 ? range_table_bytes_at(x, c.g) : 0;
        const uint64_t total = base + tables * (c.hints ? 2 : 1) + (c.slots ? slots : 0);
        if (total <= x->table_budget || c.g == 0) {
            x->plan_g = c.g;
            x->plan_hints = c.hints && c.g > 0;
            x->plan_slots = c.slots;
            std::fprintf(stderr, "ds2i_hip: DS2I_TABLE_BUDGET %.2f GB: range tables at %g entries per posting, %s hints, %s side slots (%.2f GB resident)\n",
                         x->table_budget / 1e9, c.g, x->plan_hints ? "with" : "no", x->plan_slots ? "with" : "no", total / 1e9);
            return;
        }
    }
    // no candidate matches the pinned knobs (e.g. DS2I_RMW_G outside {4, 2, 1}): the knobs win, the budget is not enforced -- and says so
    std::fprintf(stderr, "ds2i_hip: DS2I_TABLE_BUDGET %.2f GB is not enforced: no table plan matches the pinned knobs (DS2I_RMW_G = %g)\n", x->table_budget / 1e9, x->plan_g);
    x->table_budget = 0;
}

// block_optpfor: the exception side slots, their overflow area and the tail table (abi_structs.hpp, BatchArgs::xslots).
// One more pass over the index with the general decoders (k_build_side_tables). Like the range tables they are an
// accelerator: an upload that cannot afford them (or DS2I_NO_XSLOTS) runs the kernels that parse the Simple16 streams.
int build_side_tables(ds2i_hip_index* x) {
    if (!x->plan_slots || !x->d_skip || !x->total_blocks || x->total_blocks >= (1ull << 32)) return DS2I_OK;
    const uint64_t V = x->size;
    x->list_tail_off.assign(V, 0);
    uint64_t tail = 0;
    for (uint64_t t = 0; t < V; ++t) {
        x->list_tail_off[t] = tail; // dwords: an entry = sz gaps-1, sz freqs-1, bytes of the docs part, bytes of the freqs part
        if (x->list_n[t] & 127u) tail += 2 * (x->list_n[t] & 127u) + 2;
    }
    const uint64_t slot_bytes = 4ull * ds2i_dev::XSLOT_DW * x->total_blocks, tail_bytes = 4 * tail + 1024;
    auto give_up = [&](const char* why) -> int {
        if (x->d_xslots) (void)hipFree(x->d_xslots);
        if (x->d_xovf) (void)hipFree(x->d_xovf);
        if (x->d_tails) (void)hipFree(x->d_tails);
        x->d_xslots = x->d_xovf = x->d_tails = nullptr;
        (void)hipGetLastError();
        std::fprintf(stderr, "ds2i_hip: index uploaded WITHOUT exception side slots (%.2f GB wanted: %s); block decodes parse the Simple16 streams\n",
                     (slot_bytes + tail_bytes) / 1e9, why);
        return DS2I_OK;
    };
    {
        static std::mutex side_alloc_mu; // (several replicas may be uploaded to one device at once)
        std::lock_guard<std::mutex> g(side_alloc_mu);
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        if (slot_bytes + tail_bytes > free_b / 2) return give_up("less than twice their size is free on the device");
        if (hipMalloc((void**)&x->d_xslots, slot_bytes) != hipSuccess || hipMalloc((void**)&x->d_tails, tail_bytes) != hipSuccess) return give_up("hipMalloc failed");
    }
    std::vector<QTerm> lists(V);
    std::vector<ds2i_dev::BmwItem> items;
    items.reserve(x->total_blocks / 64 + V);
    for (uint64_t t = 0; t < V; ++t) {
        lists[t] = ds2i_make_qterm(x, (uint32_t)t);
        lists[t].aux1 = x->list_tail_off[t]; // (d_tails is set only once the tables are complete)
        for (uint32_t b = 0; b < x->list_nb[t]; b += 64) items.push_back(ds2i_dev::BmwItem{(uint32_t)t, b});
    }
    DevTemps tmp;
    QTerm* d_lists = nullptr;
    ds2i_dev::BmwItem* d_items = nullptr;
    unsigned long long* d_cursor = nullptr;
    HIP_OK(tmp.alloc(&d_lists, sizeof(QTerm) * V));
    HIP_OK(tmp.alloc(&d_items, sizeof(ds2i_dev::BmwItem) * items.size()));
    HIP_OK(tmp.alloc(&d_cursor, 16));
    HIP_OK(hipMemcpy(d_lists, lists.data(), sizeof(QTerm) * V, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_items, items.data(), sizeof(ds2i_dev::BmwItem) * items.size(), hipMemcpyHostToDevice));
    uint64_t ovf_cap = x->total_blocks / 8 + 65536; // dwords; the pass says how much it wanted and is re-run once if that was more
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (x->d_xovf) (void)hipFree(x->d_xovf);
        x->d_xovf = nullptr;
        if (hipMalloc((void**)&x->d_xovf, 4 * ovf_cap) != hipSuccess) return give_up("hipMalloc failed (overflow area)");
        HIP_OK(hipMemsetAsync(x->d_xslots, 0, slot_bytes, x->stream[0]));
        HIP_OK(hipMemsetAsync(d_cursor, 0, 16, x->stream[0]));
        ds2i_dev::SideArgs a{};
        a.arena = x->d_arena;
        a.lists = d_lists;
        a.items = d_items;
        a.nitems = (uint32_t)items.size();
        a.num_docs = (uint32_t)x->num_docs;
        a.skip = x->d_skip;
        a.xslots = x->d_xslots;
        a.xovf = x->d_xovf;
        a.xovf_cap = ovf_cap;
        a.xovf_cursor = d_cursor;
        a.tails = x->d_tails;
        a.bad = (unsigned int*)(d_cursor + 1);
        HIP_OK(ds2i_launch_build_side_tables(&a, (unsigned)std::min<size_t>(items.size(), (size_t)x->num_cus * 64), x->stream[0]));
        HIP_OK(hipStreamSynchronize(x->stream[0]));
        unsigned long long res[2] = {0, 0};
        HIP_OK(hipMemcpy(res, d_cursor, 16, hipMemcpyDeviceToHost));
        if ((unsigned int)res[1]) return give_up("block headers disagree with the decoded values (corrupt image?)");
        if (res[0] >= 0x7FFFFFFFull) return give_up("overflow area beyond 8 GB"); // (a slot stores XSLOT_SLOW | (offset + 1): 31 bits of offset)
        if (res[0] <= ovf_cap) {
            x->side_bytes = slot_bytes + tail_bytes + 4 * ovf_cap;
            x->extra_bytes += x->side_bytes;
            return DS2I_OK;
        }
        ovf_cap = res[0] + 64;
    }
    return give_up("overflow area did not converge");
}

} // namespace

extern "C" {

const char* ds2i_hip_last_error(void) { return ds2i_get_error(); }

// The kernel classes of a batch run on one stream each, uploads and merges on two more; with the HIP default of 4
// hardware queues several of them end up sharing one (the null stream owns a queue) and serialise. Ask for more
// before the runtime initialises; an explicit setting of the user wins.
__attribute__((constructor)) static void ds2i_hip_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// The knobs (knobs.hpp) are read from the environment by every ds2i_hip_index_open -- the one place the library reads it -- and hold
// for that index and for every batch planned until the next upload; ds2i_hip_set_option sets one without the environment.
namespace {
const char* const kKnobs[] = {"DS2I_RMW_G", "DS2I_NO_RMW", "DS2I_NO_RMH", "DS2I_NO_BITMAPS", "DS2I_NO_BMW", "DS2I_NO_XSLOTS", "DS2I_RMW_REQUIRE", "DS2I_MIXED_NATIVE",
                              "DS2I_PEF_NATIVE", "DS2I_TABLE_BUDGET", "DS2I_PLAN_THREADS", "DS2I_UNIT_FACTOR", "DS2I_UNIT_CAP", "DS2I_UT_BLOCKS", "DS2I_STREAM_NT_MAX",
                              "DS2I_NO_RANKED_STREAM", "DS2I_NO_UNION_RSTREAM", "DS2I_NO_LIST_STREAMS", "DS2I_DECODE_GENERAL", "DS2I_UNIT_CLOCK"};
Ds2iKnobs g_knobs{};
std::mutex g_knobs_mu;
bool g_knobs_loaded = false;
void load_knobs_locked() {
    auto env = [](const char* n) -> const char* { return std::getenv(n); }; // (the one place the library reads its knobs)
    auto on = [&](const char* n) { return env(n) != nullptr; };
    auto num = [&](const char* n, double dflt) { const char* e = env(n); return e && std::atof(e) > 0 ? std::atof(e) : dflt; };
    Ds2iKnobs v{};
    v.rmw_g_set = on("DS2I_RMW_G");
    v.rmw_g = v.rmw_g_set ? std::atof(env("DS2I_RMW_G")) : 4.0;
    v.no_rmw = on("DS2I_NO_RMW");
    v.no_rmh = on("DS2I_NO_RMH");
    v.no_bitmaps = on("DS2I_NO_BITMAPS");
    v.no_bmw = on("DS2I_NO_BMW");
    v.no_xslots = on("DS2I_NO_XSLOTS");
    v.rmw_require = on("DS2I_RMW_REQUIRE");
    v.mixed_native = on("DS2I_MIXED_NATIVE");
    v.pef_native = on("DS2I_PEF_NATIVE");
    if (env("DS2I_TABLE_BUDGET")) std::snprintf(v.table_budget, sizeof v.table_budget, "%s", env("DS2I_TABLE_BUDGET"));
    v.plan_threads = (unsigned)num("DS2I_PLAN_THREADS", 0);
    v.unit_factor = num("DS2I_UNIT_FACTOR", 0);
    v.unit_cap = (uint32_t)num("DS2I_UNIT_CAP", 0);
    v.ut_blocks = (uint32_t)num("DS2I_UT_BLOCKS", 320);
    v.stream_nt_max = (uint32_t)std::min(16.0, std::max(2.0, num("DS2I_STREAM_NT_MAX", 16)));
    v.no_ranked_stream = on("DS2I_NO_RANKED_STREAM");
    v.no_union_rstream = on("DS2I_NO_UNION_RSTREAM");
    v.no_list_streams = on("DS2I_NO_LIST_STREAMS");
    v.decode_general = on("DS2I_DECODE_GENERAL");
    v.unit_clock = on("DS2I_UNIT_CLOCK");
    g_knobs = v;
    g_knobs_loaded = true;
}
}

extern "C++" Ds2iKnobs ds2i_knobs() {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    if (!g_knobs_loaded) load_knobs_locked();
    return g_knobs;
}
static void ds2i_reload_knobs() {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    load_knobs_locked();
}

int ds2i_hip_set_option(const char* name, const char* value) {
    if (!name || std::strncmp(name, "DS2I_", 5) != 0 || std::strlen(name) > 48) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_set_option: not a DS2I_* knob");
    for (const char* c = name; *c; ++c)
        if (!((*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '_')) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_set_option: not a DS2I_* knob");
    bool known = false;
    for (const char* k : kKnobs) known = known || std::strcmp(k, name) == 0;
    if (!known) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_set_option: unknown knob (DESIGN.md 7 lists them)");
    if (value) setenv(name, value, 1); else unsetenv(name);
    return DS2I_OK;
}

int ds2i_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int index_open_impl(int device, int kind, const void* index_image, size_t index_bytes, const void* wand_image, size_t wand_bytes, bool bare,
                           ds2i_hip_index** out, size_t budget_base_bytes = 0);

// block_mixed, the default upload: the image is TRANSCODED to the block codec this device decodes fastest. The mixed
// image is uploaded bare (no tables), every list decoded by the mixed-block kernels (mixed_block.hpp:198-217: OptPFor /
// VarInt-G8IU / interpolative by type byte), the postings re-encoded by the GPU index encoder as block_optpfor
// (encode_kernels.hip, byte-identical to the host builder), and THAT image is what the query kernels see -- with its skip
// table, block weights, range tables and exception side slots. The answers are those of the mixed index (same postings); what
// changes is that a query never waits for the lane-0 interpolative walk (12 % of the fixed policy's blocks at 11 x the cost
// of an OptPFor block) or the descriptor-driven VarInt-G8IU gather. DS2I_MIXED_NATIVE=1 keeps the image as it is and runs the
// mixed-codec kernels (k_ranked_stream_mixed, the CODEC_MIXED instantiations).
// The freq_index layouts (opt / ef / single / uniform) take the same way: partitioned Elias-Fano is the space-optimal form for
// disk and for a CPU's caches; on a device with 288 GB behind 8 TB/s the form to query is the one whose block is one wave's
// worth of work with its skip entry, block weight, range-table bytes and side slot beside it. Their chunk directory is built
// (it is what the decode pass walks), every list decoded by the partitioned-sequence kernels (partitioned_sequence.hpp:198-326,
// compact_elias_fano.hpp:184-214, compact_ranked_bitvector.hpp, positive / strict sequences), and the postings re-encoded as
// above. DS2I_PEF_NATIVE=1 keeps the bit vectors and runs the CODEC_PEF kernels over the chunk directory.
static int index_open_transcoded(int device, int kind, const void* index_image, size_t index_bytes, const void* wand_image, size_t wand_bytes, ds2i_hip_index** out) {
    ds2i_hip_index* raw = nullptr;
    int rc = index_open_impl(device, kind, index_image, index_bytes, nullptr, 0, true, &raw);
    if (rc) return rc;
    std::unique_ptr<ds2i_hip_index, void (*)(ds2i_hip_index*)> guard(raw, free_index);
    const uint64_t V = raw->size;
    std::vector<uint64_t> offs(V + 1, 0);
    uint32_t longest = 1;
    for (uint64_t t = 0; t < V; ++t) {
        offs[t + 1] = offs[t] + raw->list_n[t];
        longest = std::max(longest, raw->list_n[t]);
    }
    // Transcoding is a choice, not an obligation: what a transcoded index needs at the very least -- ~2 bytes per posting of
    // block_optpfor image, a side slot per block, its skip table and block weights -- is several times a partitioned Elias-Fano image.
    // Under a DS2I_TABLE_BUDGET it does not fit, or when the host cannot hold the decoded postings (8 bytes each), the image is
    // uploaded AS IT IS and queried by its own kernels (DS2I_PEF_NATIVE / DS2I_MIXED_NATIVE behaviour), with the tables the budget allows.
    auto native = [&](const char* why) {
        std::fprintf(stderr, "ds2i_hip: %s: the %s image is uploaded as it is (native kernels)\n", why, kind == DS2I_BLOCK_MIXED ? "block_mixed" : "freq_index");
        guard.reset();
        return index_open_impl(device, kind, index_image, index_bytes, wand_image, wand_bytes, false, out);
    };
    const uint64_t budget = table_budget_bytes(index_bytes);
    const uint64_t least = 2 * offs[V] + (4ull * ds2i_dev::XSLOT_DW + 8 + 4) * ((offs[V] + 127) / 128) + 4 * raw->num_docs;
    if (budget && least > budget) return native("DS2I_TABLE_BUDGET is below what a transcoded index needs");
    std::vector<uint32_t> docs, freqs;
    try {
        docs.resize(offs[V]);
        freqs.resize(offs[V]);
    } catch (std::bad_alloc const&) {
        std::vector<uint32_t>().swap(docs);
        return native("not enough host memory to hold the decoded postings");
    }
    {
        // lists are decoded a CHUNK at a time: every list of the chunk by a launch of its own into its place in two device buffers
        // (no host synchronisation in between), then one copy per buffer -- not a launch, two copies and a synchronisation per list
        // (a collection has millions of lists, most of them a few postings long)
        DevTemps tmp;
        uint32_t *d_docs = nullptr, *d_freqs = nullptr;
        const uint64_t chunk_cap = std::max<uint64_t>((uint64_t)longest + 128, 64ull << 20); // postings per chunk (256 MB per buffer)
        if (tmp.alloc(&d_docs, 4 * (size_t)chunk_cap) != hipSuccess || tmp.alloc(&d_freqs, 4 * (size_t)chunk_cap) != hipSuccess) {
            (void)hipGetLastError();
            return native("not enough device memory for the decode buffers");
        }
        for (uint64_t t0 = 0; t0 < V;) {
            uint64_t t1 = t0, fill = 0;
            while (t1 < V && fill + raw->list_n[t1] + 128 <= chunk_cap) { // (+128: the decode kernels write whole blocks)
                DecodeArgs a{};
                a.arena = raw->d_arena;
                a.bits0 = raw->d_bits0;
                a.bits1 = raw->d_bits1;
                a.skip = raw->d_skip;
                a.term = ds2i_make_qterm(raw, (uint32_t)t1);
                a.codec = kind >= DS2I_OPT ? (int)DS2I_OPT : kind; // (every freq_index layout decodes through the chunk directory)
                a.num_docs = (uint32_t)raw->num_docs;
                a.out_docs = d_docs + fill;
                a.out_freqs = d_freqs + fill;
                HIP_OK(ds2i_launch_decode_list(&a, (unsigned)std::min<uint64_t>(raw->list_nb[t1], uint64_t(raw->num_cus) * 16), raw->stream[0]));
                fill += raw->list_n[t1];
                ++t1;
            }
            HIP_OK(hipMemcpyAsync(docs.data() + offs[t0], d_docs, 4 * (size_t)fill, hipMemcpyDeviceToHost, raw->stream[0]));
            HIP_OK(hipMemcpyAsync(freqs.data() + offs[t0], d_freqs, 4 * (size_t)fill, hipMemcpyDeviceToHost, raw->stream[0]));
            HIP_OK(hipStreamSynchronize(raw->stream[0])); // (the two device buffers are reused by the next chunk)
            t0 = t1;
        }
    }
    const uint64_t num_docs = raw->num_docs;
    guard.reset(); // the mixed image leaves the device before the transcoded one arrives
    ds2i_blob* img = nullptr;
    rc = ds2i_hip_encode_index(device, DS2I_BLOCK_OPTPFOR, num_docs, V, offs.data(), docs.data(), freqs.data(), &img, nullptr);
    if (rc) return rc;
    std::vector<uint32_t>().swap(docs);
    std::vector<uint32_t>().swap(freqs);
    rc = index_open_impl(device, DS2I_BLOCK_OPTPFOR, ds2i_blob_data(img), ds2i_blob_size(img), wand_image, wand_bytes, false, out, index_bytes);
    ds2i_blob_free(img);
    if (rc == DS2I_OK) (*out)->kind_on_disk = kind;
    return rc;
}

int ds2i_hip_index_open(int device, int kind, const void* index_image, size_t index_bytes, const void* wand_image,
                        size_t wand_bytes, ds2i_hip_index** out) {
    if (!out || !index_image) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_index_open: null argument");
    if (kind < DS2I_BLOCK_OPTPFOR || kind > DS2I_UNIFORM) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_index_open: unknown index kind");
    if (device < 0 || device >= ds2i_hip_device_count()) return ds2i_set_error(DS2I_EDEVICE, "ds2i_hip_index_open: no such HIP device");
    ds2i_reload_knobs(); // (every upload re-reads the knobs: they hold for this index and the batches planned until the next upload)
    const bool transcode = (kind == DS2I_BLOCK_MIXED && !ds2i_knobs().mixed_native) ||
                           (kind >= DS2I_OPT && kind <= DS2I_UNIFORM && !ds2i_knobs().pef_native);
    if (transcode) {
        return index_open_transcoded(device, kind, index_image, index_bytes, wand_image, wand_bytes, out);
    }
    return index_open_impl(device, kind, index_image, index_bytes, wand_image, wand_bytes, false, out);
}

static int index_open_impl(int device, int kind, const void* index_image, size_t index_bytes, const void* wand_image, size_t wand_bytes, bool bare,
                           ds2i_hip_index** out, size_t budget_base_bytes) {
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
    }
    x->list_blk_base.resize(V);
    for (uint64_t t = 0; t < V; ++t) { // blocks (chunks) are numbered list by list in index order
        x->list_blk_base[t] = x->total_blocks;
        x->total_blocks += x->list_nb[t];
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
        float mn = std::numeric_limits<float>::infinity();
        for (uint64_t i = 0; i < wv.num_docs; ++i) {
            float v;
            std::memcpy(&v, (const uint8_t*)wv.norm_lens + 4 * i, 4);
            mn = v < mn ? v : mn;
        }
        x->min_norm_len = wv.num_docs && mn >= 0.f ? mn : 0.f; // 0 is a valid lower bound for any collection
    }
    HIP_OK(hipMalloc((void**)&x->d_ticket, 64 * sizeof(unsigned int)));
    {
        // One stream per class, all of ONE priority. Rounds 2-4 gave the many-list classes a higher queue priority than the <=2-list
        // flood; with the stream kernels what the priorities do turned out to depend on the history of the process (an index
        // uploaded after another one had been closed -- every transcoding upload -- ran 18 % slower): equal priorities since round 5.
        int lo_pri = 0, hi_pri = 0; // numerically lower = higher priority
        HIP_OK(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        for (int c = 0; c < NCLS; ++c) HIP_OK(hipStreamCreateWithPriority(&x->stream[c], hipStreamNonBlocking, (lo_pri + hi_pri) / 2));
    }
    HIP_OK(hipStreamCreateWithFlags(&x->s_up, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&x->s_merge, hipStreamNonBlocking));
    if (!bare) choose_table_plan(x.get(), budget_base_bytes ? budget_base_bytes : index_bytes);
    else x->plan_g = 0, x->plan_hints = x->plan_slots = false;
    if (!bare && x->has_wand && x->total_blocks && x->total_blocks < (1ull << 32) && !ds2i_knobs().no_bmw) {
        int rc = build_block_max_weights(x.get());
        if (rc) return rc;
    }
    if (!bare && kind == DS2I_BLOCK_OPTPFOR) {
        int rc = build_side_tables(x.get());
        if (rc) return rc;
    }
    x->term_proto.resize(V);
    for (uint64_t t = 0; t < V; ++t) {
        QTerm qt = ds2i_make_qterm(x.get(), (uint32_t)t);
        qt.q_weight = x->has_wand ? x->max_term_weight[t] : 0.f;
        qt.max_weight = x->d_bmw ? x->list_bmw[t] : 0.f;
        x->term_proto[t] = qt;
    }
    *out = x.release();
    return DS2I_OK;
}

void ds2i_hip_index_close(ds2i_hip_index* idx) { free_index(idx); }
uint64_t ds2i_hip_index_size(const ds2i_hip_index* idx) { return idx ? idx->size : 0; }
uint64_t ds2i_hip_index_num_docs(const ds2i_hip_index* idx) { return idx ? idx->num_docs : 0; }
uint64_t ds2i_hip_index_device_bytes(const ds2i_hip_index* idx) {
    return idx ? idx->arena_bytes + idx->extra_bytes + (idx->has_wand ? 4 * idx->num_docs : 0) : 0;
}
int ds2i_hip_index_get_info(const ds2i_hip_index* idx, ds2i_hip_index_info* out) {
    if (!idx || !out) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_index_get_info: null argument");
    std::memset(out, 0, sizeof *out);
    const bool freq_layout = idx->kind >= DS2I_OPT;
    out->block_weight_bytes = idx->d_bmw ? 4 * idx->total_blocks : 0;
    out->range_table_bytes = idx->d_rmw ? idx->rmw_bytes + (idx->d_rmh ? idx->rmw_bytes : 0) : 0;
    out->has_membership_hints = idx->d_rmh != nullptr;
    out->skip_table_bytes = idx->d_skip ? 8 * idx->total_blocks : 0;
    out->norm_len_bytes = idx->has_wand ? 4 * idx->num_docs : 0;
    out->side_table_bytes = idx->d_xslots ? idx->side_bytes : 0;
    out->has_side_tables = idx->d_xslots != nullptr;
    out->index_bytes = idx->arena_bytes + idx->extra_bytes - out->block_weight_bytes - out->range_table_bytes - out->skip_table_bytes - out->side_table_bytes;
    (void)freq_layout;
    out->total_blocks = idx->total_blocks;
    for (uint32_t n : idx->list_n) out->total_postings += n;
    out->has_block_weights = idx->d_bmw != nullptr;
    out->has_range_tables = idx->d_rmw != nullptr;
    out->has_bitmaps = idx->d_rmw != nullptr && idx->has_bitmaps;
    out->range_table_entries_per_posting = idx->d_rmw ? idx->rmw_g : 0;
    out->transcoded_from = idx->kind_on_disk;
    out->table_budget_bytes = idx->table_budget;
    return DS2I_OK;
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
    a.term = ds2i_make_qterm(idx, term);
    a.codec = idx->kind >= DS2I_OPT ? (int)DS2I_OPT : idx->kind; // every freq_index layout decodes through the chunk directory
    a.num_docs = (uint32_t)idx->num_docs;
    a.out_docs = d_docs;
    a.out_freqs = d_freqs;
    a.stats = nullptr;
    a.skip = idx->d_skip;
    a.xslots = idx->d_xslots;
    a.xovf = idx->d_xovf;
    a.tails = idx->d_tails;
    unsigned grid = (unsigned)std::min<uint64_t>(nb, uint64_t(idx->num_cus) * 16);
    // block_optpfor with side tables: through the stream kernels' decoder (DS2I_DECODE_GENERAL=1: the general decoders)
    const bool side = idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots && !ds2i_knobs().decode_general;
    hipError_t e = side ? ds2i_launch_decode_list_side(&a, grid, idx->stream[0]) : ds2i_launch_decode_list(&a, grid, idx->stream[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(idx->stream[0]);
    if (e == hipSuccess) e = hipMemcpy(docs, d_docs, 4 * len, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(freqs, d_freqs, 4 * len, hipMemcpyDeviceToHost);
    (void)hipFree(d_docs);
    (void)hipFree(d_freqs);
    if (e != hipSuccess) return ds2i_set_error(DS2I_EDEVICE, hipGetErrorString(e));
    return DS2I_OK;
}

int ds2i_hip_list_block_weights(ds2i_hip_index* idx, uint32_t term, float* out, uint64_t capacity, uint64_t* nblocks) {
    if (!idx || !nblocks) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_block_weights: null argument");
    if (term >= idx->size) return ds2i_set_error(DS2I_ETERM, "term id out of range");
    *nblocks = idx->d_bmw ? idx->list_nb[term] : 0;
    if (!*nblocks) return DS2I_OK;
    if (!out || capacity < *nblocks) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_block_weights: capacity too small");
    HIP_OK(hipSetDevice(idx->device));
    HIP_OK(hipMemcpy(out, idx->d_bmw + idx->list_blk_base[term], 4 * *nblocks, hipMemcpyDeviceToHost));
    return DS2I_OK;
}

int ds2i_hip_list_range_table(ds2i_hip_index* idx, uint32_t term, uint32_t level, uint8_t* out, uint64_t capacity,
                              uint64_t* entries, uint32_t* shift, float* list_max) {
    if (!idx || !entries || !shift || !list_max) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_range_table: null argument");
    if (term >= idx->size) return ds2i_set_error(DS2I_ETERM, "term id out of range");
    *entries = 0;
    *shift = 0;
    *list_max = 0.f;
    if (!idx->d_rmw) return DS2I_OK;
    if (level > 4) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_range_table: level must be 0 (the bitmap), 1, 2, 3 or 4 (the membership hints)");
    if (level == 4) { // membership hints: one byte per level-1 entry (0 entries = this index has none)
        *shift = idx->list_rmw_shift[term];
        *list_max = idx->list_bmw[term];
        if (!idx->d_rmh) return DS2I_OK;
        const ds2i_dev::RmwLevels gh((uint32_t)idx->num_docs, idx->list_rmw_shift[term]);
        *entries = gh.e[0];
        if (!out || capacity < *entries) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_range_table: capacity too small");
        HIP_OK(hipSetDevice(idx->device));
        HIP_OK(hipMemcpy(out, idx->d_rmh + 64ull * idx->list_rmw_off64[term], *entries, hipMemcpyDeviceToHost));
        return DS2I_OK;
    }
    const ds2i_dev::RmwLevels g((uint32_t)idx->num_docs, idx->list_rmw_shift[term]);
    if (level == 0) { // the exact bitmap of a dense list: (num_docs + 7) / 8 bytes, bit d = doc-id d; 0 entries = the list has none
        *list_max = idx->list_bmw[term];
        if (!idx->has_bitmaps || !ds2i_dev::RmwLevels::has_bitmap(idx->list_n[term], (uint32_t)idx->num_docs)) return DS2I_OK;
        *entries = (idx->num_docs + 7) / 8;
        if (!out || capacity < *entries) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_range_table: capacity too small");
        HIP_OK(hipSetDevice(idx->device));
        HIP_OK(hipMemcpy(out, idx->d_rmw + 64ull * idx->list_rmw_off64[term] + g.bytes(), *entries, hipMemcpyDeviceToHost));
        return DS2I_OK;
    }
    *shift = idx->list_rmw_shift[term] + 6 * (level - 1);
    *entries = g.e[level - 1];
    *list_max = idx->list_bmw[term];
    if (!out || capacity < *entries) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_list_range_table: capacity too small");
    HIP_OK(hipSetDevice(idx->device));
    HIP_OK(hipMemcpy(out, idx->d_rmw + 64ull * idx->list_rmw_off64[term] + g.off[level - 1], *entries, hipMemcpyDeviceToHost));
    return DS2I_OK;
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

int ds2i_hip_selftest_bm25(int device, const uint32_t* freqs, const float* norm_lens, float* out, uint32_t n) {
    if (!freqs || !norm_lens || !out || !n) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_selftest_bm25: bad argument");
    if (device < 0 || device >= ds2i_hip_device_count()) return ds2i_set_error(DS2I_EDEVICE, "no such HIP device");
    HIP_OK(hipSetDevice(device));
    DevTemps tmp;
    uint32_t* df = nullptr;
    float *dn = nullptr, *dout = nullptr;
    HIP_OK(tmp.alloc(&df, 4 * (size_t)n));
    HIP_OK(tmp.alloc(&dn, 4 * (size_t)n));
    HIP_OK(tmp.alloc(&dout, 4 * (size_t)n));
    HIP_OK(hipMemcpy(df, freqs, 4 * (size_t)n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dn, norm_lens, 4 * (size_t)n, hipMemcpyHostToDevice));
    HIP_OK(ds2i_launch_selftest_bm25(df, dn, dout, n, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, dout, 4 * (size_t)n, hipMemcpyDeviceToHost));
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
