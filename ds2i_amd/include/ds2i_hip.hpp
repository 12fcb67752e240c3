// ds2i_hip.hpp -- header-only C++ adaptor over the C ABI (include/ds2i_hip.h) that presents the two
// template concepts ds2i's drivers are written against, so queries.cpp-style code can switch to the
// MI355X path by changing a typedef:
//
//   Index concept (block_freq_index.hpp:72-94)          -> ds2i_hip::gpu_index
//       size(), num_docs(), operator[](term) -> document_enumerator, warmup(term)
//       document_enumerator (block_posting_list.hpp:84-186): docid(), freq(), next(), next_geq(),
//       move(), reset(), size(), position(); exhausted <=> docid() == num_docs()
//   Query-operator concept (queries.hpp:35-591, queries.cpp:13-28) -> ds2i_hip::gpu_query_op<OP>
//       uint64_t operator()(Index const&, term_id_vec) ; topk() for ranked operators
//       + the batched form this framework adds: operator()(Index const&, vector<term_id_vec> const&)
//
// Errors: the C ABI's negative codes become std::runtime_error (builders in the reference throw
// std::invalid_argument / std::runtime_error too; its query path only asserts).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <deque>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/ds2i_hip.h"

namespace ds2i_hip {

typedef uint32_t term_id_type;
typedef std::vector<term_id_type> term_id_vec;

inline void check(int rc, const char* what) {
    if (rc != DS2I_OK) throw std::runtime_error(std::string(what) + ": " + ds2i_hip_last_error());
}

class gpu_index {
public:
    // The whole list is decoded on the GPU once (k_decode_list) and iterated on the host: this keeps
    // the reference's per-posting enumerator API usable (tools, verification) -- the query operators
    // below never go through it.
    class document_enumerator {
    public:
        document_enumerator() {}
        document_enumerator(std::vector<uint32_t>&& docs, std::vector<uint32_t>&& freqs, uint64_t universe)
            : m_docs(std::move(docs)), m_freqs(std::move(freqs)), m_universe(universe) {}
        void reset() { m_pos = 0; }
        void next() { ++m_pos; }
        void next_geq(uint64_t lower_bound) {
            while (m_pos < m_docs.size() && m_docs[m_pos] < lower_bound) ++m_pos;
        }
        void move(uint64_t pos) { m_pos = pos; }
        uint64_t docid() const { return m_pos < m_docs.size() ? m_docs[m_pos] : m_universe; }
        uint64_t freq() const { return m_freqs[m_pos]; }
        uint64_t position() const { return m_pos; }
        uint64_t size() const { return m_docs.size(); }

    private:
        std::vector<uint32_t> m_docs, m_freqs;
        uint64_t m_universe = 0;
        uint64_t m_pos = 0;
    };

    gpu_index() {}
    // index_kind = enum ds2i_hip_index_kind; the images are the reference's on-disk files (mmap them)
    gpu_index(int index_kind, const void* image, size_t bytes, const void* wand = nullptr, size_t wand_bytes = 0,
              int device = 0) {
        check(ds2i_hip_index_open(device, index_kind, image, bytes, wand, wand_bytes, &m_h), "ds2i_hip_index_open");
    }
    gpu_index(gpu_index const&) = delete;
    gpu_index& operator=(gpu_index const&) = delete;
    gpu_index(gpu_index&& o) : m_h(o.m_h) { o.m_h = nullptr; }
    ~gpu_index() { ds2i_hip_index_close(m_h); }

    size_t size() const { return (size_t)ds2i_hip_index_size(m_h); }
    uint64_t num_docs() const { return ds2i_hip_index_num_docs(m_h); }
    void warmup(size_t) const {} // resident in HBM; nothing to touch (block_freq_index.hpp:96-114)
    document_enumerator operator[](size_t term) const {
        uint64_t n = 0;
        check(ds2i_hip_list_size(m_h, (uint32_t)term, &n), "ds2i_hip_list_size");
        std::vector<uint32_t> d(n), f(n);
        check(ds2i_hip_decode_list(m_h, (uint32_t)term, d.data(), f.data(), n, &n), "ds2i_hip_decode_list");
        return document_enumerator(std::move(d), std::move(f), num_docs());
    }
    ds2i_hip_index* handle() const { return m_h; }

private:
    ds2i_hip_index* m_h = nullptr;
};

template <int OP>
class gpu_query_op {
public:
    explicit gpu_query_op(uint64_t k = 10) : m_k((uint32_t)k) {}
    // reference signature: ranked operators are constructed (wand_data const&, k); the wand data already
    // lives next to the index in HBM, so it is accepted and ignored
    template <class WandData> gpu_query_op(WandData const&, uint64_t k) : m_k((uint32_t)k) {}
    void prefer_latency(bool) {} // (set_query<OP>'s ticket policy; one index has no tickets)

    // queries.cpp:26-28 -- one query (a batch of one)
    uint64_t operator()(gpu_index const& index, term_id_vec const& terms) {
        std::vector<term_id_vec> one(1, terms);
        return (*this)(index, one)[0];
    }
    // the batch boundary: all queries cross the C ABI at once
    std::vector<uint64_t> const& operator()(gpu_index const& index, std::vector<term_id_vec> const& queries) {
        const uint32_t nq = (uint32_t)queries.size();
        std::vector<uint32_t> terms, offs(nq + 1, 0);
        for (uint32_t q = 0; q < nq; ++q) {
            terms.insert(terms.end(), queries[q].begin(), queries[q].end());
            offs[q + 1] = (uint32_t)terms.size();
        }
        if (terms.empty()) terms.push_back(0);
        m_counts.assign(nq, 0);
        const uint32_t k = ranked() ? m_k : 1;
        std::vector<float> topk((size_t)nq * k, -std::numeric_limits<float>::infinity());
        std::vector<uint32_t> len(nq, 0);
        // the device counters are an instrumentation option (the reference's block_profiler is a template flag too):
        // off unless asked for, so timing loops run the uninstrumented kernels; kernel_ms is always filled
        check(ds2i_hip_query_batch(index.handle(), OP | (m_counters ? 0 : DS2I_OP_NO_COUNTERS), k, terms.data(), offs.data(), nq,
                                   m_counts.data(), topk.data(), len.data(), &m_stats),
              "ds2i_hip_query_batch");
        m_topk.assign(nq, std::vector<float>());
        if (ranked())
            for (uint32_t q = 0; q < nq; ++q) m_topk[q].assign(topk.begin() + (size_t)q * k, topk.begin() + (size_t)q * k + len[q]);
        return m_counts;
    }
    std::vector<float> const& topk() const {                                  // last query (reference shape)
        static const std::vector<float> none;
        return m_topk.empty() ? none : m_topk.back();
    }
    std::vector<std::vector<float>> const& topk_batch() const { return m_topk; }
    ds2i_hip_stats const& stats() const { return m_stats; }
    void collect_counters(bool on) { m_counters = on; }
    static constexpr bool ranked() { return OP >= DS2I_OP_RANKED_AND; }

private:
    uint32_t m_k;
    bool m_counters = false;
    std::vector<uint64_t> m_counts;
    std::vector<std::vector<float>> m_topk;
    ds2i_hip_stats m_stats{};
};

// Pipelined serving loop over the C ABI's submit / wait (ds2i_hip_pipeline_*): `depth` batches in flight, the host
// plans batch i+1 while the kernels of batch i run. Results arrive in submission order.
class gpu_pipeline {
public:
    struct result {
        std::vector<uint64_t> counts;
        std::vector<float> topk; // nq * k, descending per query, padded with -inf (ranked operators)
        std::vector<uint32_t> topk_len;
        ds2i_hip_stats stats{};
    };
    gpu_pipeline(gpu_index const& index, uint32_t depth = 3) : m_depth(depth) {
        check(ds2i_hip_pipeline_create(index.handle(), depth, &m_h), "ds2i_hip_pipeline_create");
    }
    gpu_pipeline(gpu_pipeline const&) = delete;
    gpu_pipeline& operator=(gpu_pipeline const&) = delete;
    ~gpu_pipeline() { ds2i_hip_pipeline_destroy(m_h); }
    uint32_t depth() const { return m_depth; }
    uint64_t submit(int op, uint32_t k, std::vector<term_id_vec> const& queries) { return submit(op, k, queries.begin(), queries.end()); }
    // a run of queries of a larger batch, without copying it
    template <class It> uint64_t submit(int op, uint32_t k, It first, It last) {
        const size_t nq = (size_t)(last - first);
        std::vector<uint32_t> terms, offs(nq + 1, 0);
        size_t q = 0;
        for (It it = first; it != last; ++it, ++q) {
            terms.insert(terms.end(), it->begin(), it->end());
            offs[q + 1] = (uint32_t)terms.size();
        }
        if (terms.empty()) terms.push_back(0);
        uint64_t ticket = 0;
        check(ds2i_hip_pipeline_submit(m_h, op, k, terms.data(), offs.data(), (uint32_t)nq, &ticket), "ds2i_hip_pipeline_submit");
        if (m_meta.size() <= ticket % m_depth) m_meta.resize(m_depth);
        m_meta[ticket % m_depth] = {(uint32_t)nq, ((op & 0xFF) >= DS2I_OP_RANKED_AND) ? k : 1u};
        return ticket;
    }
    void set_instrumented(bool on) { check(ds2i_hip_pipeline_set_instrumented(m_h, on ? 1 : 0), "ds2i_hip_pipeline_set_instrumented"); }
    result wait(uint64_t ticket) {
        const auto meta = m_meta.at(ticket % m_depth);
        result r;
        r.counts.assign(meta.first, 0);
        r.topk.assign((size_t)meta.first * meta.second, -std::numeric_limits<float>::infinity());
        r.topk_len.assign(meta.first, 0);
        check(ds2i_hip_pipeline_wait(m_h, ticket, r.counts.data(), r.topk.data(), r.topk_len.data(), &r.stats), "ds2i_hip_pipeline_wait");
        return r;
    }

private:
    ds2i_hip_pipeline* m_h = nullptr;
    uint32_t m_depth;
    std::vector<std::pair<uint32_t, uint32_t>> m_meta;
};

// ---- several devices of one node. The path shards by QUERY: every device holds a full replica of the
// index (a GOV2-scale index is ~1 % of one MI355X's 288 GB), a batch is cut into contiguous slices, one per replica,
// each slice runs on its device under its own host thread, and the answers are concatenated in order -- no exchange
// between devices, no collective. The host-side analogue in the reference is profile_queries.cpp:21-39 (one thread
// per core over disjoint query ranges). The same device may be listed more than once (two replicas on one GPU): the
// answers are the same, which is what the single-GPU test of this class relies on.
class gpu_index_set {
public:
    gpu_index_set(int index_kind, const void* image, size_t bytes, const void* wand, size_t wand_bytes,
                  std::vector<int> const& devices) {
        if (devices.empty()) throw std::invalid_argument("gpu_index_set: no devices");
        m_replicas.resize(devices.size());
        std::vector<std::string> errors(devices.size());
        std::vector<std::thread> pool; // replicas are uploaded in parallel (PCIe links are per device)
        for (size_t i = 0; i < devices.size(); ++i)
            pool.emplace_back([&, i] {
                try { m_replicas[i].reset(new gpu_index(index_kind, image, bytes, wand, wand_bytes, devices[i])); }
                catch (std::exception const& e) { errors[i] = e.what(); }
            });
        for (auto& t : pool) t.join();
        for (auto const& e : errors) if (!e.empty()) throw std::runtime_error(e);
        static std::atomic<uint64_t> generation(0);
        m_id = ++generation;
    }
    // identity of this set among all sets the process ever built: what a query operator keys its per-replica pipelines on
    // (the address of a set can be reused by a later one)
    uint64_t id() const { return m_id; }
    size_t replicas() const { return m_replicas.size(); }
    gpu_index const& replica(size_t i) const { return *m_replicas[i]; }
    size_t size() const { return m_replicas[0]->size(); }
    uint64_t num_docs() const { return m_replicas[0]->num_docs(); }
    void warmup(size_t) const {}
    gpu_index::document_enumerator operator[](size_t term) const { return (*m_replicas[0])[term]; }
    // contiguous slice of a batch of n queries owned by replica r (ds2i_amd/sharding.py: query_slice)
    static std::pair<size_t, size_t> slice(size_t n, size_t r, size_t parts) {
        const size_t base = n / parts, extra = n % parts;
        const size_t lo = r * base + std::min(r, extra);
        return {lo, lo + base + (r < extra ? 1 : 0)};
    }

private:
    std::vector<std::unique_ptr<gpu_index>> m_replicas;
    uint64_t m_id = 0;
};

// the query-operator concept over a replica set: same calls, same answers as gpu_query_op over one index.
// A batch is cut into TICKETS (contiguous runs of queries, several per replica) that the replicas' host threads take off a
// shared counter -- profile_queries.cpp:21-39 hands query i to thread i mod N; a counter does the same without assuming
// that all queries cost alike -- and every replica drives its tickets through its own pipeline (ds2i_hip_pipeline_*), two
// in flight: while the kernels of one ticket run, the next is being planned and uploaded. A batch that fits one ticket
// (the per-query latency loop of the `queries` driver) is answered inline on replica 0: no thread is created for it.
template <int OP>
class gpu_set_query_op {
public:
    explicit gpu_set_query_op(uint64_t k = 10) : m_k(k) {}
    template <class WandData> gpu_set_query_op(WandData const&, uint64_t k) : m_k(k) {}
    uint64_t operator()(gpu_index_set const& set, term_id_vec const& terms) {
        std::vector<term_id_vec> one(1, terms);
        return (*this)(set, one)[0];
    }
    std::vector<uint64_t> const& operator()(gpu_index_set const& set, std::vector<term_id_vec> const& queries) {
        const size_t parts = set.replicas(), n = queries.size();
        const size_t per = ticket_size(n, parts), ntickets = per ? (n + per - 1) / per : 0;
        m_stats = ds2i_hip_stats{};
        if (ntickets <= 1) { // one ticket: inline, through the one-shot call of replica 0
            if (m_ops.empty()) m_ops.assign(1, gpu_query_op<OP>(m_k));
            m_ops[0].collect_counters(m_counters);
            m_counts = m_ops[0](set.replica(0), queries);
            m_topk = m_ops[0].topk_batch();
            m_stats = m_ops[0].stats();
            return m_counts;
        }
        if (m_pipes.size() != parts || m_pipes_of != set.id()) {
            m_pipes.clear();
            for (size_t r = 0; r < parts; ++r) m_pipes.emplace_back(new gpu_pipeline(set.replica(r), 2));
            m_pipes_of = set.id();
        }
        m_counts.assign(n, 0);
        m_topk.assign(n, std::vector<float>());
        const uint32_t k = ranked() ? (uint32_t)m_k : 1u;
        std::atomic<size_t> next(0);
        std::vector<std::string> errors(parts);
        std::vector<ds2i_hip_stats> stats(parts, ds2i_hip_stats{});
        auto worker = [&](size_t r) {
            try {
                gpu_pipeline& pipe = *m_pipes[r];
                pipe.set_instrumented(m_counters);
                std::deque<std::pair<uint64_t, size_t>> inflight; // (pipeline ticket, first query)
                for (;;) {
                    while (inflight.size() < pipe.depth()) {
                        const size_t t = next.fetch_add(1);
                        if (t >= ntickets) break;
                        const size_t lo = t * per, hi = std::min(n, lo + per);
                        inflight.emplace_back(pipe.submit(OP, k, queries.begin() + lo, queries.begin() + hi), lo);
                    }
                    if (inflight.empty()) break;
                    const auto head = inflight.front();
                    inflight.pop_front();
                    gpu_pipeline::result res = pipe.wait(head.first);
                    for (size_t i = 0; i < res.counts.size(); ++i) {
                        m_counts[head.second + i] = res.counts[i];
                        if (ranked()) m_topk[head.second + i].assign(res.topk.begin() + i * k, res.topk.begin() + i * k + res.topk_len[i]);
                    }
                    stats[r].kernel_ms += res.stats.kernel_ms;
                    stats[r].docs_blocks_decoded += res.stats.docs_blocks_decoded;
                    stats[r].freqs_blocks_decoded += res.stats.freqs_blocks_decoded;
                    stats[r].block_max_examined += res.stats.block_max_examined;
                    stats[r].algorithmic_bytes += res.stats.algorithmic_bytes;
                    stats[r].postings_scored += res.stats.postings_scored;
                    stats[r].rounds += res.stats.rounds;
                }
            } catch (std::exception const& e) { errors[r] = e.what(); }
        };
        std::vector<std::thread> pool; // one host thread per replica: its tickets cross the C ABI on that device's streams
        for (size_t r = 1; r < parts; ++r) pool.emplace_back(worker, r);
        worker(0);
        for (auto& t : pool) t.join();
        for (auto const& e : errors)
            if (!e.empty()) {
                // a failed ticket leaves its replica's other tickets submitted and un-waited: the cached pipelines would answer
                // the next batch with DS2I_EBUSY. Drop them (destroying a pipeline waits for what it has in flight) and the
                // half-filled answers; the next call starts from fresh pipelines.
                release();
                m_counts.clear();
                m_topk.clear();
                throw std::runtime_error(e);
            }
        for (size_t r = 0; r < parts; ++r) {
            m_stats.kernel_ms = std::max(m_stats.kernel_ms, stats[r].kernel_ms);
            m_stats.docs_blocks_decoded += stats[r].docs_blocks_decoded;
            m_stats.freqs_blocks_decoded += stats[r].freqs_blocks_decoded;
            m_stats.block_max_examined += stats[r].block_max_examined;
            m_stats.algorithmic_bytes += stats[r].algorithmic_bytes;
            m_stats.postings_scored += stats[r].postings_scored;
            m_stats.rounds += stats[r].rounds;
        }
        return m_counts;
    }
    // scores of the last query of the last batch (reference shape); empty for an empty batch and for and / or
    std::vector<float> const& topk() const {
        static const std::vector<float> none;
        return m_topk.empty() ? none : m_topk.back();
    }
    std::vector<std::vector<float>> const& topk_batch() const { return m_topk; } // one entry per query (empty for and / or)
    ds2i_hip_stats const& stats() const { return m_stats; } // kernel_ms = busiest replica (sum of its tickets' windows), counters summed
    void collect_counters(bool on) { m_counters = on; }
    // drops the per-replica pipelines (they hold device buffers of the set's indexes): call it before the set is destroyed if
    // the operator is to outlive it; the next batch creates new ones
    void release() {
        m_pipes.clear();
        m_pipes_of = 0;
    }
    static constexpr bool ranked() { return OP >= DS2I_OP_RANKED_AND; }
    // Queries per ticket. A ticket is a set of kernel launches, and what a small one costs is span, not work: on one MI355X batches of
    // 512 / 1024 / 2048 queries run at 0.4-0.6 / 0.55-0.75 / 0.75 of the 4096-query rate, so a throughput caller gets about two
    // tickets per replica and never fewer than 512 queries; prefer_latency(true) restores the fine cut (about eight per replica, >= 64)
    // for callers that want the first answers early.
    void prefer_latency(bool on) { m_latency = on; }
    size_t ticket_size(size_t n, size_t parts) const {
        return m_latency ? std::max<size_t>(64, (n + parts * 8 - 1) / (parts * 8)) : std::max<size_t>(512, (n + parts * 2 - 1) / (parts * 2));
    }

private:
    uint64_t m_k;
    bool m_counters = false, m_latency = false;
    std::vector<gpu_query_op<OP>> m_ops;
    std::vector<std::unique_ptr<gpu_pipeline>> m_pipes;
    uint64_t m_pipes_of = 0; // gpu_index_set::id() the pipelines belong to (0 = none)
    std::vector<uint64_t> m_counts;
    std::vector<std::vector<float>> m_topk;
    ds2i_hip_stats m_stats{};
};

typedef gpu_query_op<DS2I_OP_AND> and_query;
typedef gpu_query_op<DS2I_OP_AND_FREQ> and_freq_query;
typedef gpu_query_op<DS2I_OP_OR> or_query;
typedef gpu_query_op<DS2I_OP_OR_FREQ> or_freq_query;
typedef gpu_query_op<DS2I_OP_RANKED_AND> ranked_and_query;
typedef gpu_query_op<DS2I_OP_WAND> wand_query;
typedef gpu_query_op<DS2I_OP_MAXSCORE> maxscore_query;
typedef gpu_query_op<DS2I_OP_RANKED_OR> ranked_or_query;
// the same operators over a replica set (one index per device)
template <int OP> using set_query = gpu_set_query_op<OP>;

} // namespace ds2i_hip
