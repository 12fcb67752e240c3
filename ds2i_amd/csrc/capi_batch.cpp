// Host implementation of include/ds2i_hip.h, query half: the host part of queries.hpp (term normalisation, BM25
// query weights, list ordering: queries.hpp:29-33, 53-56, 136-150, 357-360), work-unit planning, and the kernel
// launches. A batch is a reusable SLOT: its pinned staging block, its device blocks and its events are allocated
// once and only grow, so a serving loop (ds2i_hip_pipeline_*) pays no allocation per batch; everything a batch
// uploads travels in ONE async H2D copy, everything it returns in ONE async D2H copy, and nothing in submit()
// blocks on the device -- the host plans batch i+1 while the kernels of batch i run.
// There is NO CPU fallback: every query result comes from the HIP kernels or the call fails.
#include <algorithm>
#include <atomic>
#include <exception>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <limits>
#include <memory>
#include <new>
#include <pthread.h>

#include "capi_internal.hpp"
#include "knobs.hpp"
#include "host_index.hpp"

using ds2i_dev::BatchArgs;
using ds2i_dev::MergeArgs;
using ds2i_dev::QTerm;
using ds2i_dev::Stats;
using ds2i_dev::Unit;

extern "C" {
hipError_t ds2i_launch_batch(int op, int tmax_class, const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_ranked_stream(int nt, const void* args, unsigned grid, hipStream_t s); // ranked_stream.hip
hipError_t ds2i_launch_ranked_stream_bigk(int cap, const void* args, unsigned grid, hipStream_t s); // ranked_stream.hip compiled with -DDS2I_RS_BIGK_TU (k > 64)
hipError_t ds2i_launch_freq_stream(const void* args, unsigned longest, unsigned nqterms, hipStream_t s); // freq_stream.hip
hipError_t ds2i_launch_and_stream(const void* args, int with_freqs, unsigned longest, unsigned nterms, hipStream_t s); // freq_stream.hip
hipError_t ds2i_launch_and_rstream(int cap, int with_freqs, const void* args, unsigned grid, hipStream_t s);       // ranked_stream.hip (AND = true)
hipError_t ds2i_launch_union_stream(int nt, const void* args, unsigned grid, hipStream_t s);     // union_stream.hip (wand / maxscore / ranked_or)
hipError_t ds2i_launch_union_stream_bigk(int cap, const void* args, unsigned grid, hipStream_t s); // union_stream.hip compiled with -DDS2I_US_BIGK_TU (k > 64)
hipError_t ds2i_launch_ranked_stream_mixed(int nt, const void* args, unsigned grid, hipStream_t s); // ranked_stream_mixed.hip
hipError_t ds2i_launch_merge(const void* args, unsigned grid, hipStream_t s);
hipError_t ds2i_launch_copy_seed(const uint32_t* queries, uint32_t n, uint32_t k, const float* seed_topk, const uint32_t* seed_len,
                                 const unsigned long long* seed_count, float* out_topk, uint32_t* out_len,
                                 unsigned long long* out_count, hipStream_t s);
uint32_t ds2i_meta_words(void); // kernels.hip: dwords of enumerator state per list slot (M_WORDS)
}

// one contiguous range of a batch's queries, planned by one thread (plan_batch)
struct PlanChunk {
    uint32_t q0 = 0, q1 = 0;
    std::vector<QTerm> qterms;
    std::vector<uint32_t> qnbs;
    double total_cost[NCLS] = {};
    uint32_t long_terms = 0;
    bool over16 = false; // some query of the range has more than DS2I_HIP_MAX_TERMS distinct terms
    int rc = 0;
    const char* err = nullptr;
};

// fn(0) .. fn(n-1), fn(0) on the calling thread and the others on a small pool of planning threads that lives as long as the
// library (created on first use; a batch is planned ~200 times a second, so the threads are kept, not spawned per batch)
namespace {
struct PlanPool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    const std::function<void(unsigned)>* fn = nullptr;
    unsigned next = 0, total = 0, done = 0;
    uint64_t epoch = 0;
    std::exception_ptr failed; // the first exception a piece threw (any thread); rethrown by run() once every piece is done
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return epoch != seen && next < total; });
            while (next < total) {
                const unsigned i = next++;
                const std::function<void(unsigned)>* f = fn;
                lk.unlock();
                std::exception_ptr ex;
                try { (*f)(i); } catch (...) { ex = std::current_exception(); } // (a throw on a detached thread would be std::terminate)
                lk.lock();
                if (ex && !failed) failed = ex;
                if (++done == total) cv_done.notify_all();
            }
            seen = epoch;
        }
    }
    void run(unsigned n, const std::function<void(unsigned)>& f) {
        if (n <= 1) { if (n) f(0); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            while (threads.size() + 1 < n) { threads.emplace_back([this] { worker(); }); threads.back().detach(); }
            fn = &f;
            next = 1; // (index 0 is the caller's)
            total = n;
            done = 1;
            failed = nullptr;
            ++epoch;
        }
        cv_work.notify_all();
        std::exception_ptr ex;
        try { f(0); } catch (...) { ex = std::current_exception(); }
        // whatever happened to piece 0, the workers hold a pointer to f and to the caller's chunks: they are waited for
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done == total; });
        total = 0;
        fn = nullptr;
        if (!ex) ex = failed;
        failed = nullptr;
        lk.unlock();
        if (ex) std::rethrow_exception(ex);
    }
};
std::mutex g_plan_pool_user; // one batch at a time uses the pool (several pipelines / replicas may plan concurrently)
std::atomic<PlanPool*> g_plan_pool{nullptr};
// fork(): the child has the parent's PlanPool bookkeeping (threads.size() > 0, perhaps a batch half planned, perhaps the user
// mutex held by a thread that does not exist there) and none of its threads -- it starts over with an empty pool and a fresh mutex
void plan_pool_after_fork_in_child() {
    g_plan_pool.store(new PlanPool, std::memory_order_release); // (the old one is leaked: its mutex may be held)
    new (&g_plan_pool_user) std::mutex;
}
void ds2i_plan_pool_run(unsigned n, const std::function<void(unsigned)>& f) {
    static const bool once = [] {
        g_plan_pool.store(new PlanPool, std::memory_order_release); // (leaked on purpose: its detached threads may outlive static destruction)
        pthread_atfork(nullptr, nullptr, plan_pool_after_fork_in_child);
        return true;
    }();
    (void)once;
    PlanPool* const pool = g_plan_pool.load(std::memory_order_acquire);
    if (n > 1 && g_plan_pool_user.try_lock()) {
        std::lock_guard<std::mutex> user(g_plan_pool_user, std::adopt_lock); // (released when run() throws, too)
        pool->run(n, f);
    } else {
        for (unsigned i = 0; i < n; ++i) f(i); // the pool is busy with another batch: plan this one on the caller's thread
    }
}
} // namespace

// k_ranked_stream is compiled for the list capacities 2 | 4 | 6 | 8; the planner hands it the queries of 2 .. DS2I_STREAM_NT_MAX lists
// (default 8; 4 = the 5..8-term class keeps k_conjunctive<.., 8>). Round 5, interleaved on one box: 1 047-1 060 k queries/s with 8
// against 985-992 k with 4.
static uint32_t rs_stream_nt_max() { return ds2i_knobs().stream_nt_max; }
static int rs_stream_classes() { return rs_stream_nt_max() > 8 ? 4 : rs_stream_nt_max() > 4 ? 3 : 2; } // classes that get unit records (BatchArgs::urec)

struct ds2i_hip_batch {
    ds2i_hip_index* idx = nullptr;
    int op = 0;
    uint32_t k = 0, nq = 0;
    bool want_matches = false;
    bool instrument = true;           // collect ds2i_hip_stats counters (instrumented kernel instantiations)
    uint32_t pool_batches = 1;        // batches the caller keeps in flight beside this one (pipeline depth): unit sizing
    bool alt_streams = false;         // launch on the index's second set of class streams (pipeline: odd slots of small batches)
    bool use_seed = false;            // wand / maxscore / ranked_or: `seed` holds this batch's ranked_and pass
    ds2i_hip_batch* seed = nullptr;   // ranked_and pass over the same queries (pruning floor); the slot is kept for reuse
    bool profile_on = false;          // block access profile requested (d_prof)
    unsigned int* prof_ptr = nullptr; // where instrumented runs count block decodes (the owner's d_prof; a seed borrows it)
    // ---- host plan (vectors keep their capacity between uses of the slot)
    std::vector<QTerm> qterms;
    std::vector<uint32_t> qnbs, qoff, qnb0, scratch_u32;
    std::vector<double> qcost;
    std::vector<Unit> units;
    std::vector<uint32_t> q_unit_off, split_queries, single_queries, hist_slot, order[NCLS];
    std::vector<float> unit_cost;
    std::vector<PlanChunk> plan_chunks;
    std::vector<unsigned long long> match_off;
    std::vector<uint32_t> seed_terms, seed_offs;
    // wand / maxscore / ranked_or as streams (k_union_topk): the virtual queries (query, driving list) their units belong to
    std::vector<QTerm> vterms;
    std::vector<uint32_t> voff, vinfo; // vinfo: {real query, exclusion lists, float bits of the query's score bound} per virtual query
    // prepared batches (ds2i_hip_batch_prepare) keep their query arrays: enabling the block profile re-plans the batch without the list
    // streams (k_and_stream / k_freq_stream count no block decodes -- ADVICE r5)
    std::vector<uint32_t> keep_terms, keep_offs;
    int keep_want_matches = 0;
    bool no_list_streams = false;
    bool union_stream = false;
    bool union_rstream = false;   // ... and some class of it runs k_union_stream (union_stream.hip): unit records + the floor words
    // or_freq on a block_optpfor index with the side tables: the union's size by the `or` kernels, the freqs -- which do not
    // depend on the union -- by a stream of their own after the merge (freq_stream.hip)
    bool freq_stream = false;
    // and / and_freq: the queries whose lists all carry their exact bitmap are answered by list streams (freq_stream.hip,
    // k_and_stream) and get no work units; sterms = one record per list that has to be read
    std::vector<ds2i_dev::StreamTerm> sterms;
    uint32_t sterm_longest = 0;
    uint32_t ncls[NCLS] = {};  // units per kernel class
    uint32_t nqcls[NCLS] = {}; // queries per kernel class
    uint32_t nunits = 0, nsplit = 0, nsingle = 0, long_terms = 0;
    // ---- one upload block (pinned mirror h_up -> d_up), byte offsets
    size_t o_vinfo = 0;
    size_t o_qterms = 0, o_qoff = 0, o_units = 0, o_q_unit_off = 0, o_split = 0, o_single = 0, o_hslot = 0, o_order[NCLS] = {}, o_urec[4] = {}, o_qterm_q = 0, o_sterms = 0,
           o_match_off = 0, up_bytes = 0;
    // ---- one result block (d_out -> pinned mirror h_out)
    size_t o_count = 0, o_topk = 0, o_topk_len = 0, o_freq_sum = 0, out_bytes = 0;
    // ---- device-only scratch: per-unit partial results of split queries + the shared floors
    size_t o_unit_count = 0, o_unit_topk = 0, o_unit_topk_len = 0, o_unit_freq_sum = 0, o_qfloor = 0, o_qfloorw = 0, scr_bytes = 0;
    DevBuf d_up, d_out, d_scr, d_matches, d_prof, d_stats, d_long, d_clk;
    // union kernels (k_disjunctive) keep their decoded blocks in dynamic LDS sized per launch: a class's units are grouped
    // by the list count of their query and every group is launched with just that many list slots
    struct SubLaunch { uint32_t begin, end, lists; bool stream = false; /* ranked_and: k_ranked_stream<lists> (ranked_stream.hip) */ };
    std::vector<SubLaunch> sub[NCLS];
    PinBuf h_up, h_out;
    hipEvent_t ev_up = nullptr, ev_clear = nullptr, ev_done = nullptr, ev_c0[NCLS] = {}, ev_c1[NCLS] = {};
    // a class may run several kernels back to back (ranked_and: one per exact list count): hipEvents around each launch group
    std::vector<hipEvent_t> ev_g[NCLS];
    std::vector<float> grp_ms[NCLS];
    bool uploaded = false, launched = false;
    float cls_ms[NCLS] = {};
    Stats cls_stats[NCLS] = {};
    double total_ms = 0;
};

namespace {

inline size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

int ensure_events(ds2i_hip_batch* b) {
    if (b->ev_up) return DS2I_OK;
    HIP_OK(hipEventCreate(&b->ev_up));
    HIP_OK(hipEventCreate(&b->ev_clear));
    HIP_OK(hipEventCreate(&b->ev_done));
    for (int c = 0; c < NCLS; ++c) {
        HIP_OK(hipEventCreate(&b->ev_c0[c]));
        HIP_OK(hipEventCreate(&b->ev_c1[c]));
    }
    return DS2I_OK;
}

// units in decreasing cost order. The order only steers the dispatcher (big units first, small ones fill the tail), so
// a counting sort on the float's exponent and top mantissa bits replaces the comparison sort: O(n), stable.
void order_by_cost(const std::vector<float>& cost, const std::vector<uint32_t>& ids, std::vector<uint32_t>& out,
                   std::vector<uint32_t>& tmp) {
    const int BITS = 14; // 8 exponent bits + 6 mantissa bits of a positive float
    const size_t NB = size_t(1) << BITS;
    tmp.assign(NB + 1, 0);
    auto key = [&](uint32_t id) -> uint32_t {
        uint32_t bits;
        const float c = cost[id] > 0.f ? cost[id] : 0.f;
        std::memcpy(&bits, &c, 4);
        return (uint32_t)(NB - 1) - (bits >> (31 - BITS)); // descending
    };
    for (uint32_t id : ids) ++tmp[key(id) + 1];
    for (size_t i = 0; i < NB; ++i) tmp[i + 1] += tmp[i];
    out.resize(ids.size());
    for (uint32_t id : ids) out[tmp[key(id)]++] = id;
}

// ---------------------------------------------------------------- plan: host half of the query operators
static int plan_batch_impl(ds2i_hip_batch* b, int op, uint32_t k, const uint32_t* terms, const uint32_t* query_offsets, uint32_t nq, int want_matches);
// (the plan's vectors grow with the batch: an allocation failure -- on the caller's thread or on a pool thread, see PlanPool -- is
// an error code at the C boundary, not an exception crossing it)
int plan_batch(ds2i_hip_batch* b, int op, uint32_t k, const uint32_t* terms, const uint32_t* query_offsets, uint32_t nq,
               int want_matches) {
    try {
        return plan_batch_impl(b, op, k, terms, query_offsets, nq, want_matches);
    } catch (std::bad_alloc const&) {
        return ds2i_set_error(DS2I_ENOMEM, "out of host memory planning the batch");
    } catch (std::exception const& e) {
        return ds2i_set_error(DS2I_EINVAL, e.what());
    }
}
static int plan_batch_impl(ds2i_hip_batch* b, int op, uint32_t k, const uint32_t* terms, const uint32_t* query_offsets, uint32_t nq,
                           int want_matches) {
    ds2i_hip_index* idx = b->idx;
    const Ds2iKnobs kn = ds2i_knobs(); // (as the last upload read them: knobs.hpp)
    const auto plan_t0 = std::chrono::steady_clock::now();
    if (!query_offsets || (!terms && nq && query_offsets[nq] > 0))
        return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_prepare: null argument");
    const int base_op = op & 0xFF;
    if (base_op < DS2I_OP_AND || base_op > DS2I_OP_RANKED_OR || (op & ~(0xFF | DS2I_OP_REFERENCE_ORDER | DS2I_OP_NO_COUNTERS)))
        return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_prepare: unknown query operator");
    const bool conj = base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ || base_op == DS2I_OP_RANKED_AND;
    const bool ranked = base_op >= DS2I_OP_RANKED_AND;
    if (ranked && !idx->has_wand) return ds2i_set_error(DS2I_ENOWAND, "ranked operator needs wand data");
    if (ranked && (k == 0 || k > DS2I_HIP_MAX_K_LONG)) return ds2i_set_error(DS2I_EINVAL, "k must be in [1, DS2I_HIP_MAX_K_LONG]");
    // k > 64 (one score per lane no longer suffices): every query takes the one-document-per-step kernel with a 16-scores-
    // per-lane heap and its enumerator state in global scratch -- slower, same results (the reference has no limit on k)
    const bool bigk = ranked && k > DS2I_HIP_MAX_K;
    // ... except ranked_and on block_optpfor with every table: k_ranked_stream is compiled with 4 / 16 scores per lane too, and those
    // instantiations also take the one-term queries -- every query of 1 .. DS2I_STREAM_NT_MAX terms stays on the pruned stream path
    const bool bigk_stream = bigk && base_op == DS2I_OP_RANKED_AND && !(op & DS2I_OP_REFERENCE_ORDER) && idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots &&
                             idx->d_tails && idx->d_skip && idx->d_bmw && idx->d_rmw && !kn.no_ranked_stream && kn.stream_nt_max >= 4;
    // ... and wand / maxscore / ranked_or there (k_union_stream with the same heaps; their one-term queries are answered by the ranked_and seed pass)
    bool bigk_union = bigk && (base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE || base_op == DS2I_OP_RANKED_OR) && !(op & DS2I_OP_REFERENCE_ORDER) &&
                            idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots && idx->d_tails && idx->d_skip && idx->d_bmw && idx->d_rmw && !kn.no_union_rstream &&
                            !kn.no_ranked_stream && kn.stream_nt_max >= 4;
    auto goes_long = [&](size_t nterms) {
        if (!bigk) return nterms > DS2I_HIP_MAX_TERMS;
        if (bigk_stream) return !(nterms >= 1 && nterms <= kn.stream_nt_max);
        if (bigk_union) return nterms > DS2I_HIP_MAX_TERMS; // (an empty query: the empty virtual query's unit, as for k <= 64)
        return true;
    };
    if (!ranked) k = 1; // and / or return counts only: k is ignored, no top-k is produced or copied

    b->op = op;
    b->k = k;
    b->nq = nq;
    b->want_matches = want_matches && (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ);
    b->uploaded = b->launched = false;

    auto& qterms = b->qterms;
    auto& qnbs = b->qnbs;
    auto& qoff = b->qoff;
    auto& qcost = b->qcost;
    auto& qnb0 = b->qnb0;
    qterms.clear();
    qnbs.clear();
    qoff.assign(nq + 1, 0);
    qcost.assign(nq, 0.0);
    qnb0.assign(nq, 0);
    b->match_off.assign(nq + 1, 0);
    b->long_terms = 0;
    const bool split_ok = conj && !(op & DS2I_OP_REFERENCE_ORDER);
    double total_cost[NCLS] = {};
    // The per-query half of planning (normalisation, BM25 query weights, list order, costs, bounds) is independent from query
    // to query: the batch is cut into contiguous ranges, one per planning thread (DS2I_PLAN_THREADS, default 4, the caller's
    // thread takes the first range), each range fills vectors of its own, and the ranges are joined by one copy. At
    // configs[1] scale the host's 0.9 ms of planning per 1.3 ms of kernels is what bounds the end-to-end rate, and the
    // same holds for a batch cut over 8 GPUs; the unit list and the class orders below stay sequential.
    typedef PlanChunk Chunk;
    auto plan_range = [&](Chunk& ch) {
        std::vector<QTerm>& qterms = ch.qterms;
        std::vector<uint32_t>& qnbs = ch.qnbs;
        std::vector<uint32_t> t;
        std::vector<std::pair<uint32_t, uint32_t>> tf; // (term, query term frequency)
        qterms.clear();
        qnbs.clear();
        for (uint32_t q = ch.q0; q < ch.q1; ++q) {
        if (query_offsets[q + 1] < query_offsets[q])
            { ch.rc = DS2I_EINVAL; ch.err = "query_offsets must be non-decreasing"; return; }
        t.assign(terms + query_offsets[q], terms + query_offsets[q + 1]);
        std::sort(t.begin(), t.end()); // queries.hpp:31 / 139
        tf.clear();
        for (size_t i = 0; i < t.size(); ++i) {
            if (t[i] >= idx->size) { ch.rc = DS2I_ETERM; ch.err = "term id out of range"; return; }
            if (i == 0 || t[i] != t[i - 1]) tf.emplace_back(t[i], 1u);
            else tf.back().second += 1;
        }
        if (tf.size() > DS2I_HIP_MAX_TERMS_LONG)
            { ch.rc = DS2I_ETOOLONG; ch.err = "query has more than DS2I_HIP_MAX_TERMS_LONG distinct terms"; return; }
        const size_t begin = qterms.size();
        for (auto const& p : tf) {
            QTerm qt = idx->term_proto[p.first]; // one cache line: see capi_internal.hpp
            const float mtw = qt.q_weight, lbmw = qt.max_weight;
            const uint32_t nb = qt.nblocks;
            qt.q_weight = qt.max_weight = 0.f;
            if (ranked) {
                qt.q_weight = ds2i_host::bm25::query_term_weight(p.second, qt.n, idx->num_docs);
                qt.max_weight = qt.q_weight * mtw;
                qt.max_bmw = idx->d_bmw ? qt.q_weight * lbmw : std::numeric_limits<float>::infinity();
                qt.rmw_scale = qt.q_weight * (lbmw * (1.0f / 255.0f)); // range-table byte -> bound of the term score
            }
            qterms.push_back(qt);
            qnbs.push_back(nb);
        }
        double cost = 0;
        if (conj) { // sort by increasing frequency (queries.hpp:53-56, 357-360); stable insertion sort of <= a few lists
            for (size_t i = begin + 1; i < qterms.size(); ++i) {
                const QTerm v = qterms[i];
                const uint32_t vn = qnbs[i];
                size_t j = i;
                while (j > begin && v.n < qterms[j - 1].n) {
                    qterms[j] = qterms[j - 1];
                    qnbs[j] = qnbs[j - 1];
                    --j;
                }
                qterms[j] = v;
                qnbs[j] = vn;
            }
            if (!tf.empty()) {
                const double n0 = qterms[begin].n;
                qnb0[q] = qnbs[begin];
                cost = qnb0[q] * (ranked ? 2.0 : 1.0);
                for (size_t i = begin + 1; i < qterms.size(); ++i) cost += std::min<double>(qnbs[i], n0);
                // With range tables (BatchArgs::rmw) a ranked conjunction is a walk over the blocks of its shortest list --
                // most of them are left after one decode and one gather per other list -- so a unit's time goes with its
                // number of list-0 blocks, a little more per block the more lists there are (measured wave time per round:
                // 6 / 10 / 15 us for the <=2- / <=4- / <=8-list classes)
                if (ranked && idx->d_rmw && tf.size() > 1) {
                    cost = qnb0[q] * (3.0 + (double)tf.size());
                    // ... unless the bounds have little to work with. A block of the driving list is cheap when the heap threshold
                    // kills its candidates at the range tables; how many instead get as far as a lookup in the other lists goes with
                    // (a) how many sit in an occupied range of every other list (~1/4 per list: the tables hold 4-8 entries per
                    // posting) and (b) how high the threshold can be -- the k-th best of an expected M = n0 * prod(n_j / N) matches:
                    // with M in the hundreds a few candidates of every block pass, with M in the 100 000s none. A lookup costs a
                    // block search + a decode of another list's block, about 4x the block's own cost (measured: 9 us per block for
                    // two 80 k-posting lists against 2 us for the typical unit; left unsplit such queries were the kernel's tail).
                    double M = (double)qterms[begin].n, inrange = 128.0;
                    for (size_t i = begin + 1; i < qterms.size(); ++i) {
                        M *= (double)qterms[i].n / (double)idx->num_docs;
                        inrange *= 0.25;
                    }
                    const double span_entries = 128.0 * (double)idx->num_docs / std::max(1.0, (double)qterms[begin].n) /
                                                (double)(1u << std::min(31u, qterms[begin + 1].rmw_shift));
                    const double lookups = inrange * std::min(1.0, 3.0 * (double)k / std::max(1.0, M)) * std::min(1.0, span_entries / 128.0);
                    cost *= 1.0 + 3.5 * std::min(1.0, lookups);
                }
                // a one-term ranked query scans its block weights (64 per probe) and decodes about k blocks
                if (ranked && idx->d_bmw && tf.size() == 1) cost = qnb0[q] / 16.0 + 4.0 * k;
                b->match_off[q + 1] = 128ull * qnb0[q];
            }
        } else {
            for (size_t i = begin; i < qterms.size(); ++i) cost += qnbs[i] * (ranked ? 2.0 : 1.0);
        }
        if (ranked && idx->d_bmw && qterms.size() > begin) {
            // ranked_and pruning bounds (kernels.hip): suffix sums of the list bounds in enumerator order; a one-term
            // query additionally starts from the k-th largest block weight of its list
            float suf = 0.f;
            for (size_t i = qterms.size(); i-- > begin;) {
                qterms[i].suf_bmw = suf;
                suf += qterms[i].max_bmw;
            }
            // (list_topbmw holds DS2I_HIP_MAX_K entries per list; the k > 64 path does not use static floors)
            if (!bigk && qterms.size() - begin == 1)
                qterms[begin].floor1 = qterms[begin].q_weight * idx->list_topbmw[(size_t)tf[0].first * DS2I_HIP_MAX_K + (k - 1)];
            if (!bigk && !conj) { // top-k of the UNION: any one term's k-th best block weight is a floor of the k-th score
                float f = 0.f;
                for (size_t i = 0; i < tf.size(); ++i)
                    f = std::max(f, qterms[begin + i].q_weight * idx->list_topbmw[(size_t)tf[i].first * DS2I_HIP_MAX_K + (k - 1)]);
                qterms[begin].floor1 = f;
            }
        }
        qoff[q + 1] = (uint32_t)(qterms.size() - begin); // (terms of this query: turned into offsets once the ranges are joined)
        qcost[q] = cost;
        ch.total_cost[goes_long(tf.size()) ? CLS_LONG : class_of(tf.size())] += cost;
        if (tf.size() > DS2I_HIP_MAX_TERMS) ch.over16 = true;
        if (goes_long(tf.size())) ch.long_terms = std::max<uint32_t>(ch.long_terms, (uint32_t)std::max<size_t>(1, tf.size()));
        }
    };
    // default: 4, but never more than this process's share of the CPUs it may use -- one rank per GPU under torchrun
    // (LOCAL_WORLD_SIZE) on a host whose cgroup grants 16 CPUs leaves each of 8 ranks one planning thread
    static const unsigned default_threads = [] {
        double cpus = (double)std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
            char q[32] = {0};
            double per = 0;
            if (std::fscanf(f, "%31s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) cpus = std::min(cpus, std::atof(q) / per);
            std::fclose(f);
        }
        const char* lw = std::getenv("LOCAL_WORLD_SIZE");
        const double ranks = lw && std::atoi(lw) > 0 ? (double)std::atoi(lw) : 1.0;
        const double mine = cpus / ranks - 1.0; // (one for the thread that drives the pipeline)
        return (unsigned)std::max(1.0, std::min(4.0, mine));
    }();
    const unsigned want_threads = kn.plan_threads ? kn.plan_threads : default_threads;
    const unsigned nchunks = nq >= 1024 ? std::max(1u, std::min(want_threads, 16u)) : 1u;
    std::vector<Chunk>& chunks = b->plan_chunks;
    if (chunks.size() < nchunks) chunks.resize(nchunks);
    for (unsigned c = 0; c < nchunks; ++c) {
        chunks[c].q0 = (uint32_t)((uint64_t)nq * c / nchunks);
        chunks[c].q1 = (uint32_t)((uint64_t)nq * (c + 1) / nchunks);
        chunks[c].rc = 0;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (unsigned c = 0; c < nchunks; ++c) {
            chunks[c].long_terms = 0;
            chunks[c].over16 = false;
            for (double& v : chunks[c].total_cost) v = 0;
        }
        ds2i_plan_pool_run(nchunks, [&](unsigned c) { plan_range(chunks[c]); });
        for (unsigned c = 0; c < nchunks; ++c)
            if (chunks[c].rc) return ds2i_set_error(chunks[c].rc, chunks[c].err);
        // k > 64 through the union streams needs the whole batch on them (the kernels see virtual queries; k_daat_long does not): a
        // query beyond 16 terms sends the batch back to the one-document-per-step kernels -- planned again, once
        bool over = false;
        for (unsigned c = 0; c < nchunks; ++c) over = over || chunks[c].over16;
        if (!(bigk_union && over)) break;
        bigk_union = false;
    }
    {
        size_t total = 0;
        for (unsigned c = 0; c < nchunks; ++c) total += chunks[c].qterms.size();
        qterms.resize(total);
        qnbs.resize(total);
        size_t at = 0;
        for (unsigned c = 0; c < nchunks; ++c) {
            if (!chunks[c].qterms.empty()) {
                std::memcpy(qterms.data() + at, chunks[c].qterms.data(), chunks[c].qterms.size() * sizeof(QTerm));
                std::memcpy(qnbs.data() + at, chunks[c].qnbs.data(), chunks[c].qnbs.size() * 4);
            }
            at += chunks[c].qterms.size();
            for (int k2 = 0; k2 < NCLS; ++k2) total_cost[k2] += chunks[c].total_cost[k2];
            b->long_terms = std::max(b->long_terms, chunks[c].long_terms);
        }
        for (uint32_t q = 0; q < nq; ++q) qoff[q + 1] += qoff[q];
    }
    for (uint32_t q = 0; q < nq; ++q) b->match_off[q + 1] += b->match_off[q];

    // ---- work units: long conjunctive queries are split by block ranges of their shortest list so
    // that one giant query does not pin a single wavefront (SURVEY.md §7 "Load imbalance")
    b->units.clear();
    b->unit_cost.clear();
    b->q_unit_off.assign(nq + 1, 0);
    b->split_queries.clear();
    b->single_queries.clear();
    std::vector<uint32_t> cls_ids[NCLS];
    for (int c = 0; c < NCLS; ++c) b->nqcls[c] = 0;
    // ranked_or takes the seed only in its block-synchronous form: its reference-order traversal stays the unpruned
    // exhaustive OR of queries.hpp:404-476 (the oracle the reference tests wand / maxscore against)
    const bool seeded = nq && (!bigk || bigk_union) && (base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE ||
                                        (base_op == DS2I_OP_RANKED_OR && !(op & DS2I_OP_REFERENCE_ORDER)));
    double all_cost = 0;
    for (double c : total_cost) all_cost += c;
    // A pipeline keeps several batches in flight: a small batch shares the wave slots with its neighbours, so its units are sized as if
    // two or three of them were one batch. Sized against its own cost alone a 512-query batch was cut into 20 k units -- three and a half
    // full rounds of the wave slots, each unit paying its window fill and its heap's warm-up. Measured at GOV2 scale (queries/s with the
    // multiplier 1 | 1.5 | 2 | 3 | depth): 256 queries 305 k | 349 k | 376 k | 414 k | 359 k; 512: 438 k | 474 k | 522 k | 552 k | 343 k;
    // 1024: 695 k | 784 k | 844 k | 687 k | 591 k; 2048 (two in 4096): 780 k | 944 k | 990 k.
    all_cost *= std::min<double>(std::min<double>(nq <= 512 ? 3.0 : 2.0, std::max<uint32_t>(1u, b->pool_batches)), std::max(1.0, 4096.0 / std::max(1u, nq)));
    const double resident = idx->num_cus * 24.0; // waves the concurrent kernels share
    // units per resident wave (tuning knob, DS2I_UNIT_FACTOR): more = better tail balance, more per-unit overhead
    // With range tables a ranked conjunction is cheap per block and the parts of a split query each pay for warming up
    // their own heap: coarser units win (measured on the GOV2-scale batch, queries/s: factor 16: 355 k, 8: 344 k,
    // 4 with the many-list classes cut 4x finer: 430-457 k, 2: 251 k)
    const bool rmw_units = ranked && conj && idx->d_rmw; // (and / and_freq verify every candidate the tables let through: their cost
                                                         // stays with the blocks of all lists, and they are throughput-, not tail-bound: measured)
    // (a small batch -- the per-GPU share of a batch sharded over several GPUs -- is cut coarser still: at 512 queries factor 2
    // measured 391 k queries/s against 321-328 k with 4; at 4096 it is the other way round)
    const double unit_factor = kn.unit_factor > 0 ? kn.unit_factor : rmw_units ? (nq < 1536 ? 2.0 : 4.0) : 16.0;
    // wand / maxscore / ranked_or: the streaming form (kernels.hip, k_union_topk) needs the range tables and the block weights;
    // queries beyond 16 terms and k > 64 keep the one-document-per-step kernel, and the whole batch keeps the windowed
    // kernel when any query does (one operator = one kernel family per batch)
    const bool disj_topk_op = base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE || base_op == DS2I_OP_RANKED_OR;
    b->union_stream = disj_topk_op && !(op & DS2I_OP_REFERENCE_ORDER) && (!bigk || bigk_union) && idx->d_rmw && idx->d_bmw && idx->d_skip_or_pef() &&
                      !b->long_terms;
    // (list_stream: what a stream over ONE list needs -- its blocks through the side slots; and_stream: the other lists' bitmaps as well)
    const bool list_stream = (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ) && !(op & DS2I_OP_REFERENCE_ORDER) && !b->want_matches &&
                             idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots && idx->d_tails && idx->d_skip && !kn.no_list_streams && !b->no_list_streams;
    const bool and_stream = list_stream && idx->d_rmw && idx->has_bitmaps;
    const uint32_t and_unit_blocks = 96u; // (measured, `and`: 48: 965 k, 96: 1 068 k, 192: 1 190 k against 1 360 k of the same build, whole queries: 802 k; and_freq: 24 | 48 | 96 all 334-337 k)
    const bool and_rs_units = (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ) && !(op & DS2I_OP_REFERENCE_ORDER) && idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots && idx->d_skip &&
                              idx->d_bmw && idx->d_rmw && !kn.no_ranked_stream;
    b->sterms.clear();
    b->sterm_longest = 0;
    b->union_rstream = false;
    b->freq_stream = base_op == DS2I_OP_OR_FREQ && !(op & DS2I_OP_REFERENCE_ORDER) && idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots && idx->d_tails &&
                     idx->d_skip && !kn.no_list_streams && !b->no_list_streams;
    b->vterms.clear();
    b->voff.assign(1, 0);
    b->vinfo.clear();
    const uint32_t ut_blocks = kn.ut_blocks; // DS2I_UT_BLOCKS, default 320: blocks of the driving list per unit (k_union_topk, round 4: 96: 335 k, 128-256: 345-352 k, 384: 335 k queries/s; k_union_stream, round 6: 64: 240 k, 160: 378 k, 320: 401 k)
    auto add_unit = [&](int c, uint32_t q, uint32_t lo, uint32_t hi, uint32_t parts, double cost) {
        Unit u;
        u.q = q;
        u.blk_begin = lo;
        u.blk_end = hi;
        u.nparts = parts;
        cls_ids[c].push_back((uint32_t)b->units.size());
        b->units.push_back(u);
        b->unit_cost.push_back((float)cost);
    };
    for (uint32_t q = 0; q < nq; ++q) {
        const uint32_t nt = qoff[q + 1] - qoff[q];
        const int c = goes_long(nt) ? CLS_LONG : class_of(nt);
        // multi-list units are latency-bound chains (non-sequential probes): cut 4x finer so the tail stays parallel; the 9-16-term
        // class (one or two waves per SIMD; its few, long units were the last thing every batch waited for) 4x finer still
        const double unit_div = 4.0, unit_div_rmw = 4.0, unit_div_many = 4.0;
        const bool rmw_cost = rmw_units; // (cost already counts the class's time per block)
        // a unit pays for itself (a dozen dependent round trips before its first block, its own heap to warm up): below a few
        // dozen blocks that is most of its time. The floor only binds for small batches -- the per-GPU share of a batch sharded over 8 GPUs
        const double floor_cost = (rmw_cost && c <= 2) ? 256.0 : 48.0;
        const double target = std::max(floor_cost, all_cost / (unit_factor * resident) / (c == 0 ? 1.0 : rmw_cost ? unit_div_rmw : unit_div) /
                                                       (c == 3 && rmw_cost ? unit_div_many : 1.0));
        ++b->nqcls[c];
        if ((and_stream && nt >= 2 && nt <= 4) || (list_stream && nt == 1)) {
            // every list carries its exact bitmap: the query is a sum over the postings of its shortest list (and, with the freqs,
            // over every list's own postings) of one bit test per other list -- list streams, no units (k_and_stream)
            // (and_query reads only its shortest list: that one needs no bitmap of its own)
            // One-term queries (and_query walks the list and counts it, queries.hpp:58-84): the same stream with no other list -- beside
            // the class kernels instead of in front of the two-term group on class 0's stream (424 of 4096 queries, 5 of that class's 9 ms)
            bool dense = true;
            for (uint32_t i = qoff[q] + (base_op == DS2I_OP_AND ? 1u : 0u); nt > 1 && i < qoff[q + 1]; ++i)
                dense = dense && ds2i_dev::RmwLevels::has_bitmap(qterms[i].n, (uint32_t)idx->num_docs);
            if (dense) {
                const uint32_t lists = base_op == DS2I_OP_AND_FREQ ? nt : 1u;
                for (uint32_t i = 0; i < lists; ++i) {
                    const QTerm& t = qterms[qoff[q] + i];
                    ds2i_dev::StreamTerm st{};
                    st.list_off = t.list_off;
                    st.tail = t.aux1;
                    st.n = t.n;
                    st.blk_base = t.blk_base;
                    st.q = q;
                    st.counts = i == 0 ? 1u : 0u;
                    st.nother = nt - 1;
                    uint32_t k2 = 0;
                    for (uint32_t j = 0; j < nt; ++j) {
                        if (j == i) continue;
                        const QTerm& o = qterms[qoff[q] + j];
                        st.bm[k2++] = 64ull * o.rmw_off64 + ds2i_dev::RmwLevels((uint32_t)idx->num_docs, o.rmw_shift).bytes();
                    }
                    b->sterms.push_back(st);
                    b->sterm_longest = std::max(b->sterm_longest, t.nblocks);
                }
                b->q_unit_off[q + 1] = (uint32_t)b->units.size();
                continue;
            }
        }
        if (seeded && nt == 1) { // one list: wand == maxscore == ranked_and, answered by the (block-synchronous) seed pass
            b->single_queries.push_back(q);
            b->q_unit_off[q + 1] = (uint32_t)b->units.size();
            continue;
        }
        if (c == CLS_LONG) { // > 16 terms: one unit, reference-order traversal over global scratch
            add_unit(c, q, 0, conj ? std::max(1u, qnb0[q]) : (uint32_t)idx->num_docs, 1, qcost[q]);
        } else if (conj) {
            uint32_t parts = 1;
            if (split_ok && nt && qnb0[q] > 1) {
                double want = std::floor(qcost[q] / target);
                parts = (uint32_t)std::min<double>(std::max(1.0, want), qnb0[q]);
            }
            const uint32_t nb0 = std::max(1u, qnb0[q]);
            if (rmw_units && split_ok && nt > 1) {
                // The cost model prices a block of the driving list at its average, but a query whose conjunction holds fewer than k
                // documents never gets a threshold: every one of its blocks goes all the way through the other lists (10 us per
                // block with 2 lists, 25 us with 4, against 2-4 us). Left whole, a 600-block query of that kind ran for 6 ms and WAS
                // the kernel's span (DS2I_UNIT_CLOCK: 1 700 of 6 144 wave slots busy on average). The cost model above now prices
                // such queries; DS2I_UNIT_CAP (blocks; half of it beyond 2 lists) additionally bounds every unit -- off by
                // default: cutting EVERY query that fine cost 20 % (each part warms up its own heap). The tests use it to split everything.
                const uint32_t cap_env = kn.unit_cap;
                if (cap_env) parts = std::max(parts, (nb0 + (c == 0 ? cap_env : std::max(8u, cap_env / 2)) - 1) / (c == 0 ? cap_env : std::max(8u, cap_env / 2)));
            }
            // `and` through the stream pipeline has no heap to warm up -- a part costs its blocks and nothing else -- and a two-list query
            // left whole was a single wave for up to 13 ms (DS2I_UNIT_CLOCK: class 0 = 487 units, median 4.9 ms, 229 waves busy on
            // average, the span of the whole batch): at most 96 blocks of the shortest list per unit
            if (and_rs_units && split_ok && nt > 1 && nt <= rs_stream_nt_max()) parts = std::max(parts, (nb0 + and_unit_blocks - 1) / and_unit_blocks);
            const uint32_t per = (nb0 + parts - 1) / parts;
            parts = (nb0 + per - 1) / per;
            if (parts > 1) b->split_queries.push_back(q);
            for (uint32_t j = 0; j < parts; ++j) add_unit(c, q, j * per, std::min(nb0, (j + 1) * per), parts, qcost[q] / parts);
        } else if (b->union_stream && !nt) { // empty query: one unit of an empty virtual query writes the empty answer
            b->voff.push_back((uint32_t)b->vterms.size());
            b->vinfo.push_back(q);
            b->vinfo.push_back(0);
            b->vinfo.push_back(0);
            add_unit(c, (uint32_t)b->voff.size() - 2, 0, 0, 1, 0.0);
        } else if (b->union_stream) {
            // lists by decreasing max score (device-computed list maxima x query weight); a document belongs to the first
            // list of that order that holds it, so what list e owns scores at most S_e = the maxima from e down. A list
            // whose S_e is below the static floor (some term's k-th best block weight) gets no units at all -- MaxScore's
            // non-essential lists, decided at plan time; the kernel re-checks against the live threshold.
            const size_t begin = qoff[q];
            uint32_t ord[DS2I_HIP_MAX_TERMS];
            for (uint32_t i = 0; i < nt; ++i) ord[i] = i;
            for (uint32_t i = 1; i < nt; ++i) { // stable insertion sort, descending max score
                const uint32_t v = ord[i];
                uint32_t j = i;
                while (j > 0 && qterms[begin + v].max_bmw > qterms[begin + ord[j - 1]].max_bmw) { ord[j] = ord[j - 1]; --j; }
                ord[j] = v;
            }
            float suffix[DS2I_HIP_MAX_TERMS + 1];
            suffix[nt] = 0.f;
            for (uint32_t e = nt; e-- > 0;) suffix[e] = suffix[e + 1] + qterms[begin + ord[e]].max_bmw;
            const float f1 = qterms[begin].floor1;
            const size_t first_unit = b->units.size();
            for (uint32_t e = 0; e < nt; ++e) {
                if (e && suffix[e] * (1.0f + 1.0f / 65536.0f) < f1 * (1.0f - 1.0e-5f)) break; // this list and all after it: non-essential
                const uint32_t vq = (uint32_t)b->voff.size() - 1;
                QTerm drv = qterms[begin + ord[e]];
                drv.suf_bmw = suffix[e + 1];
                drv.floor1 = f1;
                drv.max_weight = suffix[0]; // (k_union_stream: the query's score bound = the scale of its shared histogram)
                b->vterms.push_back(drv);
                for (uint32_t j = 0; j < e; ++j) { // exclusion lists
                    QTerm t = qterms[begin + ord[j]];
                    t.suf_bmw = suffix[e + 1];
                    b->vterms.push_back(t);
                }
                for (uint32_t j = e + 1; j < nt; ++j) { // optional lists
                    QTerm t = qterms[begin + ord[j]];
                    t.suf_bmw = suffix[j + 1];
                    b->vterms.push_back(t);
                }
                b->voff.push_back((uint32_t)b->vterms.size());
                uint32_t sbits;
                std::memcpy(&sbits, &suffix[0], 4);
                b->vinfo.push_back(q);
                b->vinfo.push_back(e);
                b->vinfo.push_back(sbits);
                const uint32_t nbe = std::max(1u, qnbs[begin + ord[e]]);
                // (the 9-16-term class runs two waves per SIMD: its units are cut 4 times finer, for more of them at once)
                const uint32_t utb_c = c == 3 ? std::max(4u, ut_blocks / 4u) : ut_blocks;
                const uint32_t lo0 = 0;
                const uint32_t nrest = nbe - lo0;
                const uint32_t parts_e = (nrest + utb_c - 1) / utb_c, per = (nrest + parts_e - 1) / parts_e;
                for (uint32_t lo = lo0; lo < nbe; lo += per) // (the driving lists of higher max score first: they raise the threshold)
                    add_unit(c, vq, lo, std::min(nbe, lo + per), 0, (double)(nt - e) * 1.0e7 + (double)(std::min(nbe, lo + per) - lo));
            }
            const uint32_t total = (uint32_t)(b->units.size() - first_unit);
            for (size_t ui = first_unit; ui < b->units.size(); ++ui) b->units[ui].nparts = total;
            if (total > 1) b->split_queries.push_back(q);
        } else {
            // or / ranked_or / wand / maxscore: units are equal-width doc-id ranges; every part keeps its own
            // top-k (its own pruning threshold), the merge is exact
            const uint32_t N = (uint32_t)idx->num_docs;
            uint32_t parts = 1;
            // every part re-seeks its lists; how fine the parts should be cut per class was measured on the GOV2-scale
            // batch (wand, queries/s): {8,2,1,1} 50.6 k, {8,2,2,2} 54.5 k, {8,4,4,4} 55.4 k, {4,2,4,4} 55.7 k -- with the
            // shared score histogram a part no longer has to warm up its own threshold, so the latency-bound many-list
            // classes gain from finer parts
            static const double disj_scale[NCLS] = {4.0, 2.0, 4.0, 4.0, 1.0};
            // or / or_freq run as a stream (k_union): a part costs a positioning of every list plus its blocks, once each
            const bool stream_or = !ranked && !(op & DS2I_OP_REFERENCE_ORDER);
            const double dtarget = std::max(48.0, all_cost / (unit_factor * (stream_or ? 1.0 : disj_scale[c]) * resident));
            if (nt && N > 1)
                parts = (uint32_t)std::min<double>(std::max(1.0, std::floor(qcost[q] / dtarget)), std::min<double>(N, 1024.0));
            const uint32_t width = (N + parts - 1) / parts;
            parts = width ? (N + width - 1) / width : 1;
            if (parts > 1) b->split_queries.push_back(q);
            for (uint32_t j = 0; j < parts; ++j)
                add_unit(c, q, j * width, (uint32_t)std::min<uint64_t>(N, (uint64_t)(j + 1) * width), parts, qcost[q] / parts);
        }
        b->q_unit_off[q + 1] = (uint32_t)b->units.size();
    }
    b->nunits = (uint32_t)b->units.size();
    b->nsplit = (uint32_t)b->split_queries.size();
    b->nsingle = (uint32_t)b->single_queries.size();
    for (int c = 0; c < NCLS; ++c) {
        order_by_cost(b->unit_cost, cls_ids[c], b->order[c], b->scratch_u32); // costliest first
        b->ncls[c] = (uint32_t)b->order[c].size();
        b->sub[c].clear();
        if (!b->ncls[c]) continue;
        // Residency hides the union kernels' dependent round trips and LDS per wave caps it. The <=2- and <=4-list kernels
        // are capped by registers first (6 / 5 waves per SIMD), so only the many-list classes are cut into groups: the
        // longest queries first, costliest first inside a group (stable). The groups of a class run back to back on its
        // stream, so finer is not better: granularity in lists, measured on the GOV2-scale wand batch (round 2: k_disjunctive)
        // 0 (off) 60.6 k, 1: 60.0 k, 2: 62.4 k, 4: 61.0 k queries/s.
        const uint32_t dyn_group = 2u;
        const uint32_t cls_lists = c == 0 ? 2u : c == 1 ? 4u : c == 2 ? 8u : 16u;
        const bool union_kernel = !conj && !(op & DS2I_OP_REFERENCE_ORDER) && c != CLS_LONG;
        const int dyn_mincls = 2;
        if (b->union_stream) {
            // block_optpfor with every upload-time table: the virtual queries of 2 .. 8 lists run the pipelined stream kernel compiled for
            // exactly their list count (union_stream.hip), one launch group per count, back to back on the class stream -- as ranked_and
            // does (below); everything else (other codecs, 9-16 lists, the empty query's unit): k_union_topk, static LDS, one launch per class
            const bool us_ok = !kn.no_union_rstream && c <= 3 && idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots && idx->d_tails && idx->d_skip && idx->d_bmw && idx->d_rmw;
            b->union_rstream = b->union_rstream || us_ok;
            if (!us_ok) {
                b->sub[c].push_back({0u, b->ncls[c], cls_lists});
                continue;
            }
            // (list CAPACITIES 2 | 4 | 6 | 8 | 16: a launch group holds the virtual queries of cap - 1 and cap (9 .. 16) lists -- five groups
            // and five tails per batch; inside a group the units stay in cost order)
            auto cap_of = [&](uint32_t uid) { const uint32_t vq = b->units[uid].q, n = b->voff[vq + 1] - b->voff[vq]; return n < 2 ? 0u : n > DS2I_HIP_MAX_TERMS ? DS2I_HIP_MAX_TERMS + 1u : n > 8 ? (uint32_t)DS2I_HIP_MAX_TERMS : (n + 1u) & ~1u; };
            {   // stable partition by capacity, largest first
                uint32_t cnt[DS2I_HIP_MAX_TERMS + 2] = {};
                for (uint32_t uid : b->order[c]) ++cnt[cap_of(uid)];
                uint32_t start[DS2I_HIP_MAX_TERMS + 2], acc = 0;
                for (int n = DS2I_HIP_MAX_TERMS + 1; n >= 0; --n) { start[n] = acc; acc += cnt[n]; }
                std::vector<uint32_t>& tmp = b->scratch_u32;
                tmp.resize(b->order[c].size());
                for (uint32_t uid : b->order[c]) tmp[start[cap_of(uid)]++] = uid;
                b->order[c].swap(tmp);
            }
            for (uint32_t i = 0; i < b->ncls[c];) {
                uint32_t j = i;
                const uint32_t l = cap_of(b->order[c][i]);
                while (j < b->ncls[c] && cap_of(b->order[c][j]) == l) ++j;
                ds2i_hip_batch::SubLaunch sl{i, j, l >= 2 && l <= DS2I_HIP_MAX_TERMS ? l : cls_lists};
                sl.stream = l >= 2 && l <= DS2I_HIP_MAX_TERMS;
                b->sub[c].push_back(sl);
                i = j;
            }
            continue;
        }
        // ranked_and on block_optpfor with every upload-time table: the 2- .. 8-term queries (block_mixed native: 2 .. 4) run the pipelined stream
        // kernel compiled for exactly their list count (ranked_stream.hip), one launch group per count, back to back on the
        // class stream; one-term queries and everything else keep the class kernel
        // (the 5..8-term class takes the stream kernel too, up to DS2I_STREAM_NT_MAX lists -- block_optpfor only)
        const bool no_rs = kn.no_ranked_stream;
        const uint32_t rs_nt = idx->kind == DS2I_BLOCK_OPTPFOR ? rs_stream_nt_max() : 4u;
        // `and` batches that do not ask for the doc-id lists take the same pipeline with AND = true -- a candidate whose hints settle
        // its membership in every other list is counted without any of them being searched or decoded (k_conjunctive<false, ...>
        // verifies every survivor of its filters by a probe).
        const bool rs_and = (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ) && idx->kind == DS2I_BLOCK_OPTPFOR; // (with or without the doc-id lists)
        const bool rs_ok = (base_op == DS2I_OP_RANKED_AND || rs_and) && !(op & DS2I_OP_REFERENCE_ORDER) && (!bigk || bigk_stream) && c <= (rs_nt > 8 ? 3 : rs_nt > 4 ? 2 : 1) && !no_rs &&
                           ((idx->kind == DS2I_BLOCK_OPTPFOR && idx->d_xslots) || idx->kind == DS2I_BLOCK_MIXED) && idx->d_skip && idx->d_bmw && idx->d_rmw;
        if (rs_ok) {
            // launch groups by list CAPACITY 2 | 4 | 6 | 8 (block_optpfor: a group holds the queries of cap - 1 and cap lists, UnitRec::pad says
            // which; block_mixed native: the exact count, 2 .. 4): four groups and four tails per batch instead of seven; queries beyond
            // DS2I_STREAM_NT_MAX lists and one-term queries form the class kernel's groups. Inside a group the units stay in cost order.
            const bool exact = idx->kind != DS2I_BLOCK_OPTPFOR;
            auto nt_of = [&](uint32_t uid) { const uint32_t q = b->units[uid].q; return qoff[q + 1] - qoff[q]; };
            auto cap_of = [&](uint32_t uid) {
                const uint32_t n = nt_of(uid);
                if (n == 1 && bigk_stream) return 4u; // (k > 64: the one-term queries ride in the capacity-4 launch)
                return n < 2 ? n : n > rs_nt ? DS2I_HIP_MAX_TERMS + 1u : exact ? n : n > 8 ? (uint32_t)DS2I_HIP_MAX_TERMS : (n + 1u) & ~1u;
            };
            {   // stable partition by capacity, largest first
                uint32_t cnt[DS2I_HIP_MAX_TERMS + 2] = {};
                for (uint32_t uid : b->order[c]) ++cnt[cap_of(uid)];
                uint32_t start[DS2I_HIP_MAX_TERMS + 2], acc = 0;
                for (int n = DS2I_HIP_MAX_TERMS + 1; n >= 0; --n) { start[n] = acc; acc += cnt[n]; }
                std::vector<uint32_t>& tmp = b->scratch_u32;
                tmp.resize(b->order[c].size());
                for (uint32_t uid : b->order[c]) tmp[start[cap_of(uid)]++] = uid;
                b->order[c].swap(tmp);
            }
            for (uint32_t i = 0; i < b->ncls[c];) {
                uint32_t j = i;
                const uint32_t l = cap_of(b->order[c][i]);
                while (j < b->ncls[c] && cap_of(b->order[c][j]) == l) ++j;
                ds2i_hip_batch::SubLaunch sl{i, j, l >= 2 && l <= DS2I_HIP_MAX_TERMS ? l : cls_lists};
                sl.stream = l >= 2 && l <= DS2I_HIP_MAX_TERMS;
                b->sub[c].push_back(sl);
                i = j;
            }
            continue;
        }
        if (union_kernel && !ranked) { // or / or_freq: the streaming kernel serves every list count (lists = ~0 says so)
            b->sub[c].push_back({0u, b->ncls[c], 0xFFFFFFFFu});
            continue;
        }
        if (!union_kernel || c < dyn_mincls || dyn_group == 0) {
            b->sub[c].push_back({0u, b->ncls[c], cls_lists});
            continue;
        }
        auto lists_of = [&](uint32_t uid) {
            const uint32_t q = b->units[uid].q, n = qoff[q + 1] - qoff[q];
            const uint32_t g = c == 1 ? 1u : dyn_group; // (3- and 4-list queries: exact)
            return std::min(cls_lists, (n + g - 1) / g * g);
        };
        std::stable_sort(b->order[c].begin(), b->order[c].end(), [&](uint32_t x, uint32_t y) { return lists_of(x) > lists_of(y); });
        for (uint32_t i = 0; i < b->ncls[c];) {
            uint32_t j = i;
            const uint32_t l = lists_of(b->order[c][i]);
            while (j < b->ncls[c] && lists_of(b->order[c][j]) == l) ++j;
            // a group narrower than one of its queries would make the kernel answer that query with an empty result
            for (uint32_t t = i; t < j; ++t) {
                const uint32_t q = b->units[b->order[c][t]].q;
                if (qoff[q + 1] - qoff[q] > l) return ds2i_set_error(DS2I_EINVAL, "internal: launch group has fewer list slots than a query of it");
            }
            b->sub[c].push_back({i, j, l});
            i = j;
        }
    }

    if (kn.unit_clock) // (diagnostic)
        std::fprintf(stderr, "ds2i plan: op %d nq %u units %u (per class %u %u %u %u %u) split queries %u cost %.0f, %.0f us\n", op, nq, b->nunits,
                     b->ncls[0], b->ncls[1], b->ncls[2], b->ncls[3], b->ncls[4], b->nsplit, all_cost,
                     1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - plan_t0).count());

    // ---- layouts
    size_t o = 0;
    auto place = [&](size_t bytes) { size_t at = o; o = align16(o + bytes); return at; };
    if (b->union_stream) { // the kernels see the virtual queries; the per-query arrays (units by query, histogram slots) stay real
        qterms.swap(b->vterms);
        qoff.swap(b->voff);
    }
    b->o_qterms = place(qterms.size() * sizeof(QTerm));
    b->o_qoff = place(qoff.size() * 4);
    b->o_vinfo = place(b->union_stream ? b->vinfo.size() * 4 : 0);
    b->o_units = place(b->units.size() * sizeof(Unit));
    b->o_q_unit_off = place(b->q_unit_off.size() * 4);
    b->o_split = place(b->split_queries.size() * 4);
    b->o_single = place(b->single_queries.size() * 4);
    b->hist_slot.assign(nq ? nq : 1, 0xFFFFFFFFu); // a split query's score histogram = its rank among the split queries
    for (uint32_t i = 0; i < b->nsplit; ++i) b->hist_slot[b->split_queries[i]] = i;
    b->o_hslot = place(b->hist_slot.size() * 4);
    for (int c = 0; c < NCLS; ++c) b->o_order[c] = place(b->order[c].size() * 4);
    for (int c = 0; c < 4; ++c) b->o_urec[c] = place((b->union_rstream || c < rs_stream_classes()) ? b->order[c].size() * sizeof(ds2i_dev::UnitRec) : 0); // (classes of k_ranked_stream / k_union_stream)
    b->o_qterm_q = place(b->freq_stream ? qterms.size() * 4 : 0);
    b->o_sterms = place(b->sterms.size() * sizeof(ds2i_dev::StreamTerm));
    b->o_match_off = place(b->want_matches ? b->match_off.size() * 8 : 0);
    b->up_bytes = o + 16;
    const size_t nq1 = nq ? nq : 1, nu1 = b->nunits ? b->nunits : 1;
    o = 0;
    b->o_count = place(8 * nq1);
    b->o_topk = place(4 * nq1 * k);
    b->o_topk_len = place(4 * nq1);
    b->o_freq_sum = place(8 * nq1);
    b->out_bytes = o;
    o = 0;
    b->o_unit_count = place(8 * nu1);
    b->o_unit_topk = place(4 * nu1 * k);
    b->o_unit_topk_len = place(4 * nu1);
    b->o_unit_freq_sum = place(8 * nu1);
    // ranked_and / wand / maxscore / ranked_or: a 256-bucket score histogram per split query (kernels.hip, ScoreHist)
    const bool disj_ranked = base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE || base_op == DS2I_OP_RANKED_OR;
    const bool hist = !(op & DS2I_OP_REFERENCE_ORDER) && b->nsplit && ((base_op == DS2I_OP_RANKED_AND && idx->d_bmw) || disj_ranked);
    b->o_qfloor = place(hist ? 1024 * (size_t)b->nsplit : 16);
    b->o_qfloorw = place(hist ? 4 * (size_t)b->nsplit : 16); // k_ranked_stream: the floor the histogram implies, one word per split query
    b->scr_bytes = o;

    b->use_seed = seeded;
    if (seeded) {
        // The seed is the ranked_and top-k of a SUB-query: any k documents' partial scores bound the final k-th
        // score from below. One- and two-term queries use all their terms (the one-term answer is final); longer
        // queries use their two shortest lists -- the full conjunction of 5+ terms is usually too small to give k
        // documents, while the rarest pair is cheap to intersect and carries the largest term weights.
        const size_t seed_terms = 2;
        // The streams (k_union_topk) start from the static floor and share a score histogram per query: measured on the
        // GOV2-scale wand batch the sub-query pass costs 8.3 ms to save 17 % of the block decodes (209 k queries/s with it,
        // 278 k without), so there only the one-term queries keep it -- it is what answers them.
        const bool seed_single_only = b->union_stream;
        auto& sterms = b->seed_terms;
        auto& soffs = b->seed_offs;
        sterms.clear();
        soffs.assign(nq + 1, 0);
        std::vector<uint32_t> dt;
        for (uint32_t q = 0; q < nq; ++q) {
            const uint32_t* qb = terms + query_offsets[q];
            const uint32_t* qe = terms + query_offsets[q + 1];
            dt.assign(qb, qe);
            std::sort(dt.begin(), dt.end());
            dt.erase(std::unique(dt.begin(), dt.end()), dt.end());
            if (seed_single_only && dt.size() > 1) {
                // no sub-query: the stream kernels find their floor themselves (static floor + shared histogram)
            } else if (dt.size() > seed_terms && dt.size() > 2) {
                std::stable_sort(dt.begin(), dt.end(), [&](uint32_t x, uint32_t y) { return idx->list_n[x] < idx->list_n[y]; });
                dt.resize(std::max<size_t>(2, seed_terms));
                for (const uint32_t* p = qb; p != qe; ++p) // keep multiplicities: the query term weight counts them
                    if (std::find(dt.begin(), dt.end(), *p) != dt.end()) sterms.push_back(*p);
            } else {
                sterms.insert(sterms.end(), qb, qe);
            }
            soffs[q + 1] = (uint32_t)sterms.size();
        }
        if (!b->seed) {
            b->seed = new ds2i_hip_batch;
            b->seed->idx = idx;
        }
        b->seed->pool_batches = b->pool_batches;
        int rc = plan_batch(b->seed, DS2I_OP_RANKED_AND, k, sterms.data(), soffs.data(), nq, 0);
        if (rc) return rc;
    }
    return DS2I_OK;
}

// ---------------------------------------------------------------- upload: one async H2D copy of the packed plan
int upload_batch(ds2i_hip_batch* b) {
    ds2i_hip_index* idx = b->idx;
    int rc = ensure_events(b);
    if (rc) return rc;
    if (b->use_seed) {
        rc = upload_batch(b->seed);
        if (rc) return rc;
    }
    HIP_OK(b->h_up.reserve(b->up_bytes));
    HIP_OK(b->d_up.reserve(b->up_bytes));
    HIP_OK(b->d_out.reserve(b->out_bytes));
    HIP_OK(b->h_out.reserve(b->out_bytes));
    HIP_OK(b->d_scr.reserve(b->scr_bytes));
    HIP_OK(b->d_stats.reserve(NCLS * sizeof(Stats)));
    if (b->want_matches) HIP_OK(b->d_matches.reserve(4 * (size_t)(b->match_off[b->nq] ? b->match_off[b->nq] : 1)));
    if (b->long_terms) {
        const size_t stride = (size_t)b->long_terms * (256 + ds2i_meta_words() + 2) + 16;
        HIP_OK(b->d_long.reserve(4 * stride * std::max<uint32_t>(1, b->ncls[CLS_LONG])));
    }
    uint8_t* h = (uint8_t*)b->h_up.p;
    auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) std::memcpy(h + off, src, bytes); };
    put(b->o_qterms, b->qterms.data(), b->qterms.size() * sizeof(QTerm));
    put(b->o_qoff, b->qoff.data(), b->qoff.size() * 4);
    if (b->union_stream) put(b->o_vinfo, b->vinfo.data(), b->vinfo.size() * 4);
    put(b->o_units, b->units.data(), b->units.size() * sizeof(Unit));
    put(b->o_q_unit_off, b->q_unit_off.data(), b->q_unit_off.size() * 4);
    put(b->o_split, b->split_queries.data(), b->split_queries.size() * 4);
    put(b->o_single, b->single_queries.data(), b->single_queries.size() * 4);
    put(b->o_hslot, b->hist_slot.data(), b->hist_slot.size() * 4);
    for (int c = 0; c < NCLS; ++c) put(b->o_order[c], b->order[c].data(), b->order[c].size() * 4);
    put(b->o_sterms, b->sterms.data(), b->sterms.size() * sizeof(ds2i_dev::StreamTerm));
    if (b->freq_stream) { // or_freq: the query of every term (k_freq_stream adds a term's freqs to its query's checksum)
        uint32_t* qq = (uint32_t*)(h + b->o_qterm_q);
        for (uint32_t q = 0; q < b->nq; ++q)
            for (uint32_t i = b->qoff[q]; i < b->qoff[q + 1]; ++i) qq[i] = q;
    }
    for (int c = 0; c < 4 && b->union_rstream; ++c) { // k_union_stream: the unit, its virtual query's terms, its REAL query (results, histogram), its exclusion lists
        ds2i_dev::UnitRec* r = (ds2i_dev::UnitRec*)(h + b->o_urec[c]);
        for (size_t i = 0; i < b->order[c].size(); ++i) {
            const uint32_t uid = b->order[c][i];
            const Unit& u = b->units[uid];
            const uint32_t rq = b->vinfo[3 * (size_t)u.q];
            r[i] = ds2i_dev::UnitRec{uid, rq, u.blk_begin, u.blk_end, u.nparts, b->qoff[u.q], b->hist_slot[rq], b->vinfo[3 * (size_t)u.q + 1] | ((b->qoff[u.q + 1] - b->qoff[u.q]) << 8)};
        }
    }
    for (int c = 0; c < rs_stream_classes() && !b->union_stream; ++c) { // one record per ticket: what k_ranked_stream reads where a unit starts (conjunctive batches)
        ds2i_dev::UnitRec* r = (ds2i_dev::UnitRec*)(h + b->o_urec[c]);
        for (size_t i = 0; i < b->order[c].size(); ++i) {
            const uint32_t uid = b->order[c][i];
            const Unit& u = b->units[uid];
            r[i] = ds2i_dev::UnitRec{uid, u.q, u.blk_begin, u.blk_end, u.nparts, b->qoff[u.q], b->hist_slot[u.q], b->qoff[u.q + 1] - b->qoff[u.q]};
        }
    }
    if (b->want_matches) put(b->o_match_off, b->match_off.data(), b->match_off.size() * 8);
    HIP_OK(hipMemcpyAsync(b->d_up.p, b->h_up.p, b->up_bytes, hipMemcpyHostToDevice, idx->s_up));
    HIP_OK(hipEventRecord(b->ev_up, idx->s_up));
    b->uploaded = true;
    return DS2I_OK;
}

// ---------------------------------------------------------------- launch: kernels + merge + one async D2H; no host sync
int launch_batch(ds2i_hip_batch* b) {
    ds2i_hip_index* idx = b->idx;
    if (!b->uploaded) return ds2i_set_error(DS2I_EINVAL, "batch has not been prepared");
    if (b->use_seed) { // block-synchronous ranked_and first: its k-th score seeds the pruning floor of every unit
        b->seed->instrument = b->instrument;
        b->seed->profile_on = b->profile_on;
        b->seed->prof_ptr = b->prof_ptr;
        b->seed->alt_streams = b->alt_streams;
        int rc = launch_batch(b->seed);
        if (rc) return rc;
    }
    // per-unit partials, shared floors, result block and counters start from zero. The clears go to the upload
    // stream: they depend on nothing but the slot being free, so the class kernels of this batch can start while the
    // previous batch is still being merged
    hipStream_t sm = idx->s_merge, su = idx->s_up;
    HIP_OK(hipMemsetAsync(b->d_scr.p, 0, b->scr_bytes, su));
    HIP_OK(hipMemsetAsync(b->d_out.p, 0, b->out_bytes, su));
    if (b->instrument) HIP_OK(hipMemsetAsync(b->d_stats.p, 0, NCLS * sizeof(Stats), su));
    HIP_OK(hipEventRecord(b->ev_clear, su)); // also orders this launch after the batch's upload (same stream)
    // Launch order of the class kernels (they overlap on separate streams either way; measured on the GOV2-scale
    // batch): the block-synchronous conjunctions run 3 % faster when the issue-bound <=2-list class is enqueued first,
    // the disjunctive operators 2.5 % faster when the many-list classes are.
    const int base_op = b->op & 0xFF;
    const bool small_first = !(b->op & DS2I_OP_REFERENCE_ORDER) &&
                       (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ || base_op == DS2I_OP_RANKED_AND);
    const bool unit_clock = ds2i_knobs().unit_clock; // diagnostic: per-unit start / end times
    if (unit_clock && b->instrument) {
        HIP_OK(b->d_clk.reserve(16 * (size_t)(b->nunits ? b->nunits : 1)));
        HIP_OK(hipMemsetAsync(b->d_clk.p, 0, 16 * (size_t)(b->nunits ? b->nunits : 1), idx->s_up));
        HIP_OK(hipStreamSynchronize(idx->s_up));
    }
    if (b->alt_streams) { // (capi_internal.hpp: the second set exists from the first small batch on)
        std::lock_guard<std::mutex> lk(idx->stream_alt_mu);
        if (!idx->stream_alt[0]) {
            int lo_pri = 0, hi_pri = 0;
            HIP_OK(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
            for (int c = NCLS - 1; c >= 0; --c) // (slot 0 last: it is the "set exists" mark)
                if (!idx->stream_alt[c]) HIP_OK(hipStreamCreateWithPriority(&idx->stream_alt[c], hipStreamNonBlocking, (lo_pri + hi_pri) / 2));
        }
    }
    hipStream_t* const cstreams = b->alt_streams ? idx->stream_alt : idx->stream;
    auto cls_stream = [&](int c) { return cstreams[c]; };
    // every class stream first waits for the upload + cleared buffers, and for the seed pass when its floors feed the kernels. (In the
    // union decomposition the seed pass only ANSWERS the one-term queries -- copied into the result block on the merge stream below --
    // and the kernels of the longer queries start beside it: waiting cost the wand batch the one-term kernel's 1.2 ms in series.)
    const bool seed_feeds = b->use_seed && !b->union_stream;
    for (int c = 0; c < NCLS; ++c) {
        if (!b->ncls[c]) continue;
        HIP_OK(hipStreamWaitEvent(cls_stream(c), b->ev_clear, 0));
        if (seed_feeds) HIP_OK(hipStreamWaitEvent(cls_stream(c), b->seed->ev_done, 0));
    }
    HIP_OK(hipStreamWaitEvent(sm, b->ev_clear, 0));
    if (b->use_seed) HIP_OK(hipStreamWaitEvent(sm, b->seed->ev_done, 0));
    if (!b->sterms.empty()) { // and / and_freq of the all-dense queries: list streams beside the class kernels (nobody else writes these queries' results)
        ds2i_dev::AndStreamArgs g{};
        g.arena = idx->d_arena;
        g.skip = idx->d_skip;
        g.xslots = idx->d_xslots;
        g.xovf = idx->d_xovf;
        g.tails = idx->d_tails;
        g.rmw = idx->d_rmw;
        g.out_count = b->d_out.at<unsigned long long>(b->o_count);
        g.out_freq_sum = base_op == DS2I_OP_AND_FREQ ? b->d_out.at<unsigned long long>(b->o_freq_sum) : nullptr;
        const bool own = b->ncls[CLS_LONG] == 0;
        hipStream_t sf = own ? cstreams[CLS_LONG] : sm;
        if (own) HIP_OK(hipStreamWaitEvent(sf, b->ev_clear, 0));
        for (size_t t0 = 0; t0 < b->sterms.size(); t0 += 32768) { // (grid.y is limited to 65535)
            g.terms = b->d_up.at<ds2i_dev::StreamTerm>(b->o_sterms) + t0;
            HIP_OK(ds2i_launch_and_stream(&g, base_op == DS2I_OP_AND_FREQ ? 1 : 0, b->sterm_longest, (unsigned)std::min<size_t>(32768, b->sterms.size() - t0), sf));
        }
        if (own) {
            HIP_OK(hipEventRecord(b->ev_c1[CLS_LONG], sf));
            HIP_OK(hipStreamWaitEvent(sm, b->ev_c1[CLS_LONG], 0));
        }
    }
    if (b->freq_stream && !b->qterms.empty()) {
        // beside the union kernels, on the stream of the >16-term class when the batch has no such query (else on the merge
        // stream, ahead of the merge): nobody else writes the checksums (the union kernels and k_merge get no pointer to them)
        const bool own = b->ncls[CLS_LONG] == 0;
        hipStream_t sf = own ? cstreams[CLS_LONG] : sm;
        if (own) HIP_OK(hipStreamWaitEvent(sf, b->ev_clear, 0));
        ds2i_dev::FreqArgs f{};
        f.arena = idx->d_arena;
        f.skip = idx->d_skip;
        f.xslots = idx->d_xslots;
        f.xovf = idx->d_xovf;
        f.tails = idx->d_tails;
        f.qterms = b->d_up.at<QTerm>(b->o_qterms);
        f.qterm_q = b->d_up.at<uint32_t>(b->o_qterm_q);
        f.out_freq_sum = b->d_out.at<unsigned long long>(b->o_freq_sum);
        uint32_t longest = 1; // (the grid has one row per term, as wide as the longest list needs)
        for (const QTerm& t : b->qterms) longest = std::max(longest, t.nblocks);
        for (size_t t0 = 0; t0 < b->qterms.size(); t0 += 32768) { // (grid.y is limited to 65535)
            ds2i_dev::FreqArgs g = f;
            g.qterms += t0;
            g.qterm_q += t0;
            HIP_OK(ds2i_launch_freq_stream(&g, longest, (unsigned)std::min<size_t>(32768, b->qterms.size() - t0), sf));
        }
        if (own) {
            HIP_OK(hipEventRecord(b->ev_c1[CLS_LONG], sf));
            HIP_OK(hipStreamWaitEvent(sm, b->ev_c1[CLS_LONG], 0));
        }
    }
    // Two launch groups leave their class stream for the stream of a class this batch has no queries in (no further hardware queue is
    // opened; spreading EVERY second group that way was measured and lost): ranked_and's one-term queries (the class kernel's group of class 0: 0.9 ms behind the two-term stream
    // kernel's 2.5 ms on that class's stream) go to a spare stream -- class 0 is one of three co-critical class streams of the step
    const bool side_group0 = base_op == DS2I_OP_RANKED_AND && !(b->op & DS2I_OP_REFERENCE_ORDER) && b->ncls[0] && b->sub[0].size() > 1 && b->sub[0].front().stream;
    // ... and the second stream group of the 5-8-list class (capacity 6 behind capacity 8; wand: 3.6 ms behind 7.5 ms, and: 1.3 behind 1.8)
    hipStream_t spare[NCLS];
    int nspare = 0, next_spare = 0;
    const bool side_group2 = b->ncls[2] && b->sub[2].size() > 1 && b->sub[2][0].stream && b->sub[2][1].stream;
    if (side_group0 || side_group2)
        for (int c = NCLS - 1; c >= 0; --c)
            if (!b->ncls[c]) {
                spare[nspare] = cstreams[c];
                HIP_OK(hipStreamWaitEvent(spare[nspare], b->ev_clear, 0));
                if (seed_feeds) HIP_OK(hipStreamWaitEvent(spare[nspare], b->seed->ev_done, 0));
                ++nspare;
            }
    for (int ci = NCLS - 1; ci >= 0; --ci) {
        const int c = small_first ? NCLS - 1 - ci : ci;
        if (!b->ncls[c]) continue;
        hipStream_t s = cls_stream(c);
        HIP_OK(hipEventRecord(b->ev_c0[c], s));
        BatchArgs a{};
        a.arena = idx->d_arena;
        a.bits0 = idx->d_bits0;
        a.bits1 = idx->d_bits1;
        a.norm_lens = idx->d_norm_lens;
        a.min_norm_len = idx->min_norm_len;
        a.qterms = b->d_up.at<QTerm>(b->o_qterms);
        a.q_off = b->d_up.at<uint32_t>(b->o_qoff);
        a.vq_info = b->union_stream ? b->d_up.at<uint32_t>(b->o_vinfo) : nullptr;
        a.ut_first = 1u;
        a.units = b->d_up.at<Unit>(b->o_units);
        a.order = b->d_up.at<uint32_t>(b->o_order[c]);
        a.urec = (c < 4 && (b->union_rstream || c < rs_stream_classes())) ? b->d_up.at<ds2i_dev::UnitRec>(b->o_urec[c]) : nullptr;
        a.nslice = b->ncls[c];
        a.dyn_lists = 0;
        a.num_docs = (uint32_t)idx->num_docs;
        a.k = b->k;
        a.codec = idx->kind >= DS2I_OPT ? (int)DS2I_OPT : idx->kind; // every freq_index layout decodes through the chunk directory
        a.unit_clock = (unit_clock && b->instrument) ? (unsigned long long*)b->d_clk.p : nullptr;
        a.out_count = b->d_out.at<unsigned long long>(b->o_count);
        a.out_topk = b->d_out.at<float>(b->o_topk);
        a.out_topk_len = b->d_out.at<uint32_t>(b->o_topk_len);
        a.out_freq_sum = b->freq_stream ? nullptr : b->d_out.at<unsigned long long>(b->o_freq_sum);
        a.out_matches = b->want_matches ? (uint32_t*)b->d_matches.p : nullptr;
        a.match_off = b->want_matches ? b->d_up.at<unsigned long long>(b->o_match_off) : nullptr;
        a.unit_count = b->d_scr.at<unsigned long long>(b->o_unit_count);
        a.unit_topk = b->d_scr.at<float>(b->o_unit_topk);
        a.unit_topk_len = b->d_scr.at<uint32_t>(b->o_unit_topk_len);
        a.unit_freq_sum = b->d_scr.at<unsigned long long>(b->o_unit_freq_sum);
        a.seed_topk = seed_feeds ? b->seed->d_out.at<float>(b->seed->o_topk) : nullptr;
        a.seed_len = seed_feeds ? b->seed->d_out.at<uint32_t>(b->seed->o_topk_len) : nullptr;
        const bool disj_topk = base_op == DS2I_OP_WAND || base_op == DS2I_OP_MAXSCORE || base_op == DS2I_OP_RANKED_OR;
        a.q_floor = (base_op == DS2I_OP_RANKED_AND || b->union_rstream) ? b->d_scr.at<unsigned int>(b->o_qfloorw) : nullptr;
        a.q_hist = (!(b->op & DS2I_OP_REFERENCE_ORDER) && b->nsplit && ((base_op == DS2I_OP_RANKED_AND && idx->d_bmw) || disj_topk))
                       ? b->d_scr.at<unsigned int>(b->o_qfloor) : nullptr;
        a.q_hist_slot = b->d_up.at<uint32_t>(b->o_hslot);
        a.block_profile = (b->instrument && b->profile_on) ? b->prof_ptr : nullptr;
        a.skip = idx->d_skip;
        a.bmw = idx->d_bmw;
        a.rmw = (base_op == DS2I_OP_RANKED_AND && !a.bmw) ? nullptr : idx->d_rmw;
        a.rmw_bitmaps = (a.rmw && idx->has_bitmaps) ? 1u : 0u;
        a.rmh = a.rmw ? idx->d_rmh : nullptr;
        a.xslots = idx->d_xslots;
        a.xovf = idx->d_xovf;
        a.tails = idx->d_tails;
        a.long_scratch = (uint32_t*)b->d_long.p;
        a.long_stride = (uint32_t)((size_t)b->long_terms * (256 + ds2i_meta_words() + 2) + 16);
        a.stats = b->instrument ? b->d_stats.at<Stats>(0) + c : nullptr;
        const uint32_t* order_base = a.order;
        const ds2i_dev::UnitRec* urec_base = a.urec;
        for (const auto& sl : b->sub[c]) { // one launch per group of the class (a single group for everything but the union kernels)
            a.order = order_base + sl.begin;
            a.urec = urec_base ? urec_base + sl.begin : nullptr;
            a.nslice = sl.end - sl.begin;
            a.dyn_lists = sl.lists;
            // (the groups of a class run back to back on its stream: launching them beside each other on further streams
            // was tried -- with that many streams the hardware queues are oversubscribed and steps of 0.5-0.9 s appear)
            const size_t gi = (size_t)(&sl - &b->sub[c].front());
            while (b->ev_g[c].size() < 2 * (gi + 1)) {
                hipEvent_t e = nullptr;
                HIP_OK(hipEventCreate(&e));
                b->ev_g[c].push_back(e);
            }
            hipStream_t sg = (gi > 0 && nspare && ((c == 0 && side_group0 && !sl.stream) || (c == 2 && side_group2 && gi == 1))) ? spare[next_spare++ % nspare] : s;
            HIP_OK(hipEventRecord(b->ev_g[c][2 * gi], sg));
            if (sl.stream && !a.block_profile && a.skip && a.bmw && a.rmw)
                HIP_OK(b->union_stream ? (b->k > DS2I_HIP_MAX_K ? ds2i_launch_union_stream_bigk((int)sl.lists, &a, a.nslice, sg) : ds2i_launch_union_stream((int)sl.lists, &a, a.nslice, sg))
                       : (base_op == DS2I_OP_AND || base_op == DS2I_OP_AND_FREQ) ? ds2i_launch_and_rstream((int)sl.lists, base_op == DS2I_OP_AND_FREQ ? 1 : 0, &a, a.nslice, sg)
                       : idx->kind == DS2I_BLOCK_MIXED ? ds2i_launch_ranked_stream_mixed((int)sl.lists, &a, a.nslice, sg)
                       : b->k > DS2I_HIP_MAX_K ? ds2i_launch_ranked_stream_bigk((int)sl.lists, &a, a.nslice, sg) : ds2i_launch_ranked_stream((int)sl.lists, &a, a.nslice, sg));
            else HIP_OK(ds2i_launch_batch(b->freq_stream ? (int)DS2I_OP_OR : (b->op & (0xFF | DS2I_OP_REFERENCE_ORDER)), c, &a, a.nslice, sg));
            HIP_OK(hipEventRecord(b->ev_g[c][2 * gi + 1], sg));
            if (sg != s) HIP_OK(hipStreamWaitEvent(sm, b->ev_g[c][2 * gi + 1], 0));
        }
        HIP_OK(hipEventRecord(b->ev_c1[c], s));
        HIP_OK(hipStreamWaitEvent(sm, b->ev_c1[c], 0));
    }
    if (b->nsplit) {
        MergeArgs m{};
        m.split_queries = b->d_up.at<uint32_t>(b->o_split);
        m.nsplit = b->nsplit;
        m.q_unit_off = b->d_up.at<uint32_t>(b->o_q_unit_off);
        m.k = b->k;
        m.ranked = (b->op & 0xFF) >= DS2I_OP_RANKED_AND;
        m.unit_count = b->d_scr.at<unsigned long long>(b->o_unit_count);
        m.unit_topk = b->d_scr.at<float>(b->o_unit_topk);
        m.unit_topk_len = b->d_scr.at<uint32_t>(b->o_unit_topk_len);
        m.unit_freq_sum = b->d_scr.at<unsigned long long>(b->o_unit_freq_sum);
        m.out_count = b->d_out.at<unsigned long long>(b->o_count);
        m.out_topk = b->d_out.at<float>(b->o_topk);
        m.out_topk_len = b->d_out.at<uint32_t>(b->o_topk_len);
        m.out_freq_sum = b->freq_stream ? nullptr : b->d_out.at<unsigned long long>(b->o_freq_sum);
        HIP_OK(ds2i_launch_merge(&m, std::min<unsigned>(b->nsplit, 4096u), sm));
    }
    if (b->use_seed && b->nsingle)
        HIP_OK(ds2i_launch_copy_seed(b->d_up.at<uint32_t>(b->o_single), b->nsingle, b->k, b->seed->d_out.at<float>(b->seed->o_topk),
                                     b->seed->d_out.at<uint32_t>(b->seed->o_topk_len),
                                     b->seed->d_out.at<unsigned long long>(b->seed->o_count), b->d_out.at<float>(b->o_topk),
                                     b->d_out.at<uint32_t>(b->o_topk_len), b->d_out.at<unsigned long long>(b->o_count), sm));
    HIP_OK(hipMemcpyAsync(b->h_out.p, b->d_out.p, b->out_bytes, hipMemcpyDeviceToHost, sm));
    HIP_OK(hipEventRecord(b->ev_done, sm));
    b->launched = true;
    return DS2I_OK;
}

// ---------------------------------------------------------------- finish: wait for the batch, collect timings / counters
int finish_batch(ds2i_hip_batch* b, ds2i_hip_stats* stats) {
    if (!b->launched) return ds2i_set_error(DS2I_EINVAL, "batch has not been launched");
    HIP_OK(hipEventSynchronize(b->ev_done));
    if (b->use_seed) { // (its kernels ran inside this batch's [ev_clear, ev_done] window: not added to kernel_ms again)
        ds2i_hip_stats ss;
        int rc = finish_batch(b->seed, &ss);
        if (rc) return rc;
    }
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, b->ev_clear, b->ev_done));
    b->total_ms = ms;
    for (int c = 0; c < NCLS; ++c) {
        b->cls_ms[c] = 0.f;
        if (b->ncls[c]) HIP_OK(hipEventElapsedTime(&b->cls_ms[c], b->ev_c0[c], b->ev_c1[c]));
        b->grp_ms[c].assign(b->ncls[c] ? b->sub[c].size() : 0, 0.f);
        for (size_t g = 0; g < b->grp_ms[c].size() && 2 * g + 1 < b->ev_g[c].size(); ++g)
            HIP_OK(hipEventElapsedTime(&b->grp_ms[c][g], b->ev_g[c][2 * g], b->ev_g[c][2 * g + 1]));
    }
    if (b->instrument) HIP_OK(hipMemcpy(b->cls_stats, b->d_stats.p, NCLS * sizeof(Stats), hipMemcpyDeviceToHost));
    else std::memset(b->cls_stats, 0, sizeof(b->cls_stats));
    if (b->instrument && b->d_clk.p && ds2i_knobs().unit_clock) { // diagnostic: where each class kernel's time goes
        std::vector<unsigned long long> clk(2 * (size_t)b->nunits);
        HIP_OK(hipMemcpy(clk.data(), b->d_clk.p, 16 * (size_t)b->nunits, hipMemcpyDeviceToHost));
        for (int c = 0; c < NCLS; ++c) {
            if (!b->ncls[c]) continue;
            unsigned long long t0 = ~0ull, t1 = 0;
            double busy = 0;
            std::vector<std::pair<unsigned long long, uint32_t>> durs;
            for (uint32_t uid : b->order[c]) {
                const unsigned long long s0 = clk[2 * (size_t)uid], e0 = clk[2 * (size_t)uid + 1];
                if (!e0) continue;
                t0 = std::min(t0, s0);
                t1 = std::max(t1, e0);
                busy += (double)(e0 - s0);
                durs.push_back({e0 - s0, uid});
            }
            if (durs.empty()) continue;
            std::sort(durs.begin(), durs.end());
            const double tick_us = 0.01; // s_memrealtime: 100 MHz
            std::fprintf(stderr, "ds2i unit clock: class %d: %zu units, span %.0f us, sum of unit times %.0f us (= %.0f waves busy on average), median unit %.1f us, p99 %.1f us\n",
                         c, durs.size(), (t1 - t0) * tick_us, busy * tick_us, busy / (double)(t1 - t0), durs[durs.size() / 2].first * tick_us,
                         durs[durs.size() * 99 / 100].first * tick_us);
            if (b->union_stream && !b->vinfo.empty()) { // wand / maxscore / ranked_or: where the time goes by driving list (e = its exclusion lists)
                double by_e[DS2I_HIP_MAX_TERMS + 1] = {}, blocks_e[DS2I_HIP_MAX_TERMS + 1] = {};
                size_t n_e[DS2I_HIP_MAX_TERMS + 1] = {};
                for (const auto& d : durs) {
                    const Unit& u = b->units[d.second];
                    const uint32_t e = std::min<uint32_t>(b->vinfo[3 * (size_t)u.q + 1], DS2I_HIP_MAX_TERMS);
                    by_e[e] += (double)d.first;
                    blocks_e[e] += (double)(u.blk_end - u.blk_begin);
                    ++n_e[e];
                }
                for (uint32_t e = 0; e <= DS2I_HIP_MAX_TERMS; ++e)
                    if (n_e[e]) std::fprintf(stderr, "    driving list %u: %zu units of %.0f blocks, %.0f us of unit time (%.1f us per owned block)\n", e, n_e[e], blocks_e[e], by_e[e] * tick_us,
                                             by_e[e] * tick_us / std::max(1.0, blocks_e[e]));
            }
            for (size_t i = 0; i < 8 && i < durs.size(); ++i) {
                const auto& d = durs[durs.size() - 1 - i];
                const Unit& u = b->units[d.second];
                std::fprintf(stderr, "    unit %u: query %u part [%u, %u) of %u parts, %.0f us, started at +%.0f us, ended at +%.0f us\n", d.second, u.q, u.blk_begin,
                             u.blk_end, u.nparts, d.first * tick_us, (clk[2 * (size_t)d.second] - t0) * tick_us, (clk[2 * (size_t)d.second + 1] - t0) * tick_us);
            }
        }
    }
    if (stats) {
        stats->kernel_ms = ms;
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

// results of the last finished run, from the pinned mirror
void copy_results(const ds2i_hip_batch* b, uint64_t* out_count, float* out_topk, uint32_t* out_topk_len, uint64_t* out_freq_sum) {
    const size_t nq = b->nq;
    if (!nq) return;
    const uint8_t* h = (const uint8_t*)b->h_out.p;
    const bool ranked = (b->op & 0xFF) >= DS2I_OP_RANKED_AND;
    if (out_count) std::memcpy(out_count, h + b->o_count, 8 * nq);
    if (out_topk && ranked) std::memcpy(out_topk, h + b->o_topk, 4 * nq * b->k); // and / or have no top-k (k is ignored)
    if (out_topk_len) std::memcpy(out_topk_len, h + b->o_topk_len, 4 * nq);
    if (out_freq_sum) std::memcpy(out_freq_sum, h + b->o_freq_sum, 8 * nq);
}

} // namespace

void ds2i_batch_destroy(ds2i_hip_batch* b) {
    if (!b) return;
    if (b->seed) ds2i_batch_destroy(b->seed);
    (void)hipSetDevice(b->idx->device);
    if (b->launched) (void)hipEventSynchronize(b->ev_done); // never free buffers under a running kernel
    if (b->ev_up) {
        (void)hipEventDestroy(b->ev_up);
        (void)hipEventDestroy(b->ev_clear);
        (void)hipEventDestroy(b->ev_done);
        for (int c = 0; c < NCLS; ++c) {
            (void)hipEventDestroy(b->ev_c0[c]);
            (void)hipEventDestroy(b->ev_c1[c]);
        }
    }
    for (auto& v : b->ev_g) for (hipEvent_t e : v) (void)hipEventDestroy(e);

    delete b; // DevBuf / PinBuf members release their memory
}

struct ds2i_hip_pipeline {
    ds2i_hip_index* idx = nullptr;
    std::vector<ds2i_hip_batch*> slots;
    std::vector<uint64_t> slot_ticket;
    std::vector<char> busy;
    uint64_t next_ticket = 0;
    ds2i_hip_batch* last_waited = nullptr;
    // (the host half of a batch -- normalisation, BM25 query weights, work-unit planning -- runs on the caller's thread, spread over
    // DS2I_PLAN_THREADS planning threads; a planner thread OF the pipeline was built in round 4, measured slower, and removed in round 6)
};

namespace {
// plan + upload + launch of one slot
int pipeline_launch(ds2i_hip_pipeline* p, size_t slot, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq) {
    HIP_OK(hipSetDevice(p->idx->device));
    ds2i_hip_batch* b = p->slots[slot];
    // (capi_internal.hpp, stream_alt; with the pool-sized units above, GOV2 scale: 512 queries 536 k against 367 k queries/s on one set,
    // 1024: 833 k against 601 k, wand 512: 265 k against 181 k)
    // (three sets, slot % 3: 256 queries 367 k against 403 k on two, 512: 564 k / 532 k, 1024: 831 k / 821 k, wand 512: 232 k / 261 k -- two)
    b->alt_streams = (slot & 1) != 0 && nq < 2048;
    b->pool_batches = (uint32_t)p->slots.size();
    int rc = plan_batch(b, op, k, terms, offs, nq, 0);
    if (rc) return rc; // (nothing enqueued yet)
    rc = upload_batch(b);
    if (!rc) rc = launch_batch(b);
    if (rc) {
        const std::string keep = ds2i_get_error();
        (void)hipDeviceSynchronize(); // nothing of the slot is still running when the error is reported
        return ds2i_set_error(rc, keep.c_str());
    }
    return DS2I_OK;
}
} // namespace

extern "C" {

void ds2i_hip_batch_free(ds2i_hip_batch* b) { ds2i_batch_destroy(b); }

int ds2i_hip_batch_prepare(ds2i_hip_index* idx, int op, uint32_t k, const uint32_t* terms,
                           const uint32_t* query_offsets, uint32_t nq, int want_matches, ds2i_hip_batch** out) {
    if (!idx || !out) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_prepare: null argument");
    HIP_OK(hipSetDevice(idx->device));
    std::unique_ptr<ds2i_hip_batch, void (*)(ds2i_hip_batch*)> b(new ds2i_hip_batch, ds2i_batch_destroy);
    b->idx = idx;
    int rc = plan_batch(b.get(), op, k, terms, query_offsets, nq, want_matches);
    if (!rc) rc = upload_batch(b.get());
    if (rc) return rc;
    HIP_OK(hipEventSynchronize(b->ev_up));
    try {
        b->keep_offs.assign(query_offsets, query_offsets + nq + 1);
        b->keep_terms.assign(terms, terms + (nq ? query_offsets[nq] : 0));
        b->keep_want_matches = want_matches;
    } catch (std::bad_alloc const&) {
        return ds2i_set_error(DS2I_ENOMEM, "out of host memory preparing the batch");
    }
    *out = b.release();
    return DS2I_OK;
}

// A launch that failed part-way may have kernels of this slot in flight (the seed pass, some class kernels) with no
// completion event recorded: wait for the device before the caller can reuse or free the slot's buffers. The error
// being reported is kept (the drain's own status is secondary).
static int drain_after_failure(ds2i_hip_batch* b, int rc) {
    const std::string keep = ds2i_get_error();
    (void)hipSetDevice(b->idx->device);
    (void)hipDeviceSynchronize();
    return ds2i_set_error(rc, keep.c_str());
}

int ds2i_hip_batch_run(ds2i_hip_batch* b, ds2i_hip_stats* stats) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_run: null batch");
    HIP_OK(hipSetDevice(b->idx->device));
    int rc = launch_batch(b);
    if (rc) return drain_after_failure(b, rc);
    return finish_batch(b, stats);
}

// GPU-side counterpart of profile_queries.cpp: per-block decode counts of the batch (input of the block_mixed
// optimiser, ds2i_hybrid_*). Counting happens in instrumented runs only and accumulates over runs.
int ds2i_hip_batch_enable_block_profile(ds2i_hip_batch* b) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_enable_block_profile: null batch");
    ds2i_hip_index* idx = b->idx;
    if (idx->kind >= DS2I_OPT) return ds2i_set_error(DS2I_EINVAL, "the block access profile exists for block indexes only");
    HIP_OK(hipSetDevice(idx->device));
    const size_t bytes = 8 * (size_t)(idx->total_blocks ? idx->total_blocks : 1);
    HIP_OK(b->d_prof.reserve(bytes));
    HIP_OK(hipMemset(b->d_prof.p, 0, bytes));
    if ((!b->sterms.empty() || b->freq_stream) && !b->keep_offs.empty()) {
        // queries answered by list streams have no work units and their kernels count nothing: plan the batch again without them, so
        // that every block decode of the batch shows in the profile (the input of the block_mixed optimiser)
        b->no_list_streams = true;
        HIP_OK(hipDeviceSynchronize());
        int rc = plan_batch(b, b->op, b->k, b->keep_terms.empty() ? nullptr : b->keep_terms.data(), b->keep_offs.data(), b->nq, b->keep_want_matches);
        if (!rc) rc = upload_batch(b);
        if (rc) return rc;
        HIP_OK(hipEventSynchronize(b->ev_up));
    }
    b->profile_on = true;
    b->prof_ptr = (unsigned int*)b->d_prof.p; // the seed pass decodes blocks too: launch_batch hands it the same buffer
    return DS2I_OK;
}
int ds2i_hip_batch_block_profile(ds2i_hip_batch* b, uint32_t* counts, uint64_t capacity, uint64_t* total_blocks) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_block_profile: null batch");
    ds2i_hip_index* idx = b->idx;
    if (total_blocks) *total_blocks = idx->total_blocks;
    if (!counts) return DS2I_OK;
    if (!b->profile_on) return ds2i_set_error(DS2I_EINVAL, "block profile not enabled on this batch");
    if (capacity < 2 * idx->total_blocks) return ds2i_set_error(DS2I_EINVAL, "counts buffer too small (2 per block)");
    HIP_OK(hipSetDevice(idx->device));
    HIP_OK(hipMemcpy(counts, b->d_prof.p, 8 * (size_t)idx->total_blocks, hipMemcpyDeviceToHost));
    return DS2I_OK;
}

int ds2i_hip_batch_set_instrumented(ds2i_hip_batch* b, int on) {
    if (!b) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_set_instrumented: null batch");
    b->instrument = on != 0;
    return DS2I_OK;
}

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

// the launch groups of class `cls` in the last run: hipEvent duration, list slots the group's kernel was launched with, units
// (= workgroups), queries, and whether it ran the pipelined ranked_and kernel (k_ranked_stream<lists>, ranked_stream.hip)
int ds2i_hip_batch_class_groups(ds2i_hip_batch* b, int cls, ds2i_hip_group_stats* out, uint32_t capacity, uint32_t* ngroups) {
    if (!b || !ngroups || cls < 0 || cls >= NCLS) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_class_groups: bad argument");
    const uint32_t n = b->ncls[cls] ? (uint32_t)b->sub[cls].size() : 0u;
    *ngroups = n;
    if (!out) return DS2I_OK;
    for (uint32_t g = 0; g < n && g < capacity; ++g) {
        const auto& sl = b->sub[cls][g];
        out[g].kernel_ms = g < b->grp_ms[cls].size() ? b->grp_ms[cls][g] : 0.0;
        out[g].lists = sl.lists;
        out[g].units = sl.end - sl.begin;
        out[g].pipelined_stream = sl.stream ? 1 : 0;
        uint32_t nq = 0, last = 0xFFFFFFFFu; // distinct queries of the group (its units are grouped by query only loosely: count by marking)
        std::vector<char> seen(b->nq ? b->nq : 1, 0);
        for (uint32_t i = sl.begin; i < sl.end; ++i) {
            const uint32_t q = b->union_stream ? 0u : b->units[b->order[cls][i]].q;
            if (q < seen.size() && !seen[q]) { seen[q] = 1; ++nq; }
        }
        (void)last;
        out[g].queries = nq;
    }
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
    if (!b->launched) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_fetch: the batch has not been run");
    HIP_OK(hipSetDevice(b->idx->device));
    HIP_OK(hipEventSynchronize(b->ev_done));
    copy_results(b, out_count, out_topk, out_topk_len, out_freq_sum);
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
    if (!b->launched) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_batch_fetch_matches: the batch has not been run");
    HIP_OK(hipSetDevice(b->idx->device));
    HIP_OK(hipEventSynchronize(b->ev_done));
    for (size_t i = 0; i <= b->nq; ++i) match_offsets[i] = b->match_off[i];
    const size_t total = (size_t)b->match_off[b->nq];
    if (!total) return DS2I_OK;
    HIP_OK(hipMemcpy(matches, b->d_matches.p, 4 * total, hipMemcpyDeviceToHost));
    if (b->nsplit) {
        std::vector<unsigned long long> ucount(b->nunits);
        HIP_OK(hipMemcpy(ucount.data(), b->d_scr.at<uint8_t>(b->o_unit_count), 8 * (size_t)b->nunits, hipMemcpyDeviceToHost));
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

// One-shot form: the slot (pinned staging, device blocks, events) is cached in the index handle, so repeated calls
// -- the per-query latency loop of the `queries` driver, the Python op(index, terms) form -- allocate nothing.
int ds2i_hip_query_batch(ds2i_hip_index* idx, int op, uint32_t k, const uint32_t* terms,
                         const uint32_t* query_offsets, uint32_t nq, uint64_t* out_count, float* out_topk,
                         uint32_t* out_topk_len, ds2i_hip_stats* stats) {
    if (!idx) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_query_batch: null index");
    HIP_OK(hipSetDevice(idx->device));
    if (!idx->oneshot) {
        idx->oneshot = new ds2i_hip_batch;
        idx->oneshot->idx = idx;
    }
    ds2i_hip_batch* b = idx->oneshot;
    b->profile_on = false;
    // the counters cost throughput: collected only when the caller asks for them (stats given, no DS2I_OP_NO_COUNTERS)
    b->instrument = stats != nullptr && !(op & DS2I_OP_NO_COUNTERS);
    int rc = plan_batch(b, op, k, terms, query_offsets, nq, 0);
    if (!rc) rc = upload_batch(b);
    if (!rc) rc = launch_batch(b);
    if (!rc) rc = finish_batch(b, stats);
    if (!rc) copy_results(b, out_count, out_topk, out_topk_len, nullptr);
    return rc;
}

// ---------------------------------------------------------------- pipelined form
int ds2i_hip_pipeline_create(ds2i_hip_index* idx, uint32_t depth, ds2i_hip_pipeline** out) {
    if (!idx || !out || depth == 0 || depth > 16) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_create: bad argument (depth in [1,16])");
    ds2i_hip_pipeline* p = new ds2i_hip_pipeline;
    p->idx = idx;
    for (uint32_t i = 0; i < depth; ++i) {
        ds2i_hip_batch* b = new ds2i_hip_batch;
        b->idx = idx;
        b->instrument = false;
        p->slots.push_back(b);
    }
    p->slot_ticket.assign(depth, 0);
    p->busy.assign(depth, 0);
    *out = p;
    return DS2I_OK;
}

void ds2i_hip_pipeline_destroy(ds2i_hip_pipeline* p) {
    if (!p) return;
    for (auto* b : p->slots) ds2i_batch_destroy(b);
    delete p;
}

int ds2i_hip_pipeline_submit(ds2i_hip_pipeline* p, int op, uint32_t k, const uint32_t* terms, const uint32_t* query_offsets,
                             uint32_t nq, uint64_t* ticket) {
    if (!p || !ticket) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_submit: null argument");
    const size_t slot = (size_t)(p->next_ticket % p->slots.size());
    if (p->busy[slot]) return ds2i_set_error(DS2I_EBUSY, "ds2i_hip_pipeline_submit: all slots in flight; wait for the oldest ticket first");
    {
        int rc = pipeline_launch(p, slot, op, k, terms, query_offsets, nq);
        if (rc) return rc; // the slot stays free
    }
    p->busy[slot] = 1;
    p->slot_ticket[slot] = p->next_ticket;
    *ticket = p->next_ticket++;
    return DS2I_OK;
}

int ds2i_hip_pipeline_wait(ds2i_hip_pipeline* p, uint64_t ticket, uint64_t* out_count, float* out_topk, uint32_t* out_topk_len,
                           ds2i_hip_stats* stats) {
    if (!p) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_wait: null pipeline");
    const size_t slot = (size_t)(ticket % p->slots.size());
    if (!p->busy[slot] || p->slot_ticket[slot] != ticket)
        return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_wait: unknown or already collected ticket");
    HIP_OK(hipSetDevice(p->idx->device));
    ds2i_hip_batch* b = p->slots[slot];
    int rc = finish_batch(b, stats);
    p->busy[slot] = 0;
    if (rc) return rc;
    p->last_waited = b;
    copy_results(b, out_count, out_topk, out_topk_len, nullptr);
    return DS2I_OK;
}

// per kernel class of the ticket collected last (hipEvent duration of the class kernel, counters when instrumented)
int ds2i_hip_pipeline_class_stats(ds2i_hip_pipeline* p, int cls, ds2i_hip_stats* out, uint32_t* nqueries) {
    if (!p || !p->last_waited) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_class_stats: no collected ticket");
    return ds2i_hip_batch_class_stats(p->last_waited, cls, out, nqueries);
}

int ds2i_hip_pipeline_class_groups(ds2i_hip_pipeline* p, int cls, ds2i_hip_group_stats* out, uint32_t capacity, uint32_t* ngroups) {
    if (!p || !p->last_waited) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_class_groups: no collected ticket");
    return ds2i_hip_batch_class_groups(p->last_waited, cls, out, capacity, ngroups);
}

int ds2i_hip_pipeline_set_instrumented(ds2i_hip_pipeline* p, int on) {
    if (!p) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_pipeline_set_instrumented: null pipeline");
    for (auto* b : p->slots) b->instrument = on != 0;
    return DS2I_OK;
}

} // extern "C"
