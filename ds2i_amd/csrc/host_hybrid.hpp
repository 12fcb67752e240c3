// host_hybrid.hpp -- space/time optimiser for block_mixed indexes.
//
// Replaces, on the build side, reference optimal_hybrid_index.cpp (lambda-greedy over per-block (space, time)
// points, 60-115 and 367-396) + mixed_block::compute_space_time (mixed_block.hpp:119-150) + the learned CPU
// decode-time model dec_time_prediction.hpp. What changes for this framework:
//   * the decode-time model is the MI355X kernels' cost, not a CPU's: wave-level instructions per block, measured
//     with rocprofv3 (profiles/instr_probe.py). On the GPU OptPFor is the cheapest decoder, VarInt-G8IU is NOT
//     faster than it (the pshufb trick has no wave64 equivalent) and interpolative is a serial chain on one lane
//     whose cost grows with the number of tree nodes that carry bits -- the opposite ranking of the reference's CPU;
//   * the access counts come from the GPU kernels (ds2i_hip_batch_block_profile) instead of profile_queries.cpp;
//   * the global step solves the same Lagrangian min sum(access * time) s.t. sum(space) <= budget by bisection on
//     lambda over each block's lower convex hull (the reference sorts all lambda break points with stxxl and
//     applies them greedily; both pick argmin(time + lambda * space) per block, ties aside).
// Candidate set per full block exactly as mixed_block::compute_space_time: OptPFor with every usable b
// (mixed_block.hpp:77-89: skip b > max_b when the previous b already covers max_b; skip max_b - b > 28), VarInt-G8IU,
// interpolative; partial blocks are always interpolative. Laplace smoothing of the counts (+1,
// optimal_hybrid_index.cpp:84).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <thread>

#include "host_encode.hpp"
#include "host_index.hpp"

namespace ds2i_host {

struct hybrid_model {            // wave-level instructions (VALU + SALU) to decode one full 128-value block
    float pfor_base = 230.f;     // header, bit unpack, state update
    float pfor_exc = 90.f;       // + register-resident Simple16 exception path (<= 32 exceptions)
    float pfor_exc_many = 240.f; // + batched path through LDS instead (> 32 exceptions)
    float varint = 365.f;
    float interp_base = 300.f;   // + suffix-min fill
    float interp_node = 54.f;    // per tree node whose range is not degenerate (serial on lane 0)
};

struct hybrid_point { // one candidate encoding of one block
    float time;       // model time x (access + 1)
    uint16_t space;   // bytes, including the type byte
    uint8_t type;     // mixed_type
    int8_t b;         // OptPFor b, -1 otherwise
};

// number of interpolative tree nodes that read bits (range > 0): the serial work of the GPU decoder
inline uint32_t interp_live_nodes(const uint32_t* pre, size_t n, uint32_t low, uint32_t high) {
    if (!n || high == low) return 0;
    const size_t h = n / 2;
    const uint32_t val = pre[h];
    return 1 + interp_live_nodes(pre, h, low, val) + interp_live_nodes(pre + h + 1, n - h - 1, val, high);
}

// all (space, time) candidates of one block, then its lower convex hull sorted by increasing space
inline void hybrid_block_points(const uint32_t* in, uint32_t sum, size_t n, uint32_t access, hybrid_model const& m,
                                std::vector<hybrid_point>& hull) {
    hull.clear();
    std::vector<hybrid_point> pts;
    bytes_t buf;
    const float w = (float)access + 1.f;
    uint64_t total = 0;
    for (size_t i = 0; i < n; ++i) total += in[i];
    const bool interp_ok = total < 0xFFFFFFFFull;
    if (n < BLOCK) { // partial blocks: interpolative only, no time prediction (mixed_block.hpp:69-71,143)
        interpolative_encode(in, sum, n, buf);
        hull.push_back(hybrid_point{0.f, (uint16_t)buf.size(), (uint8_t)MIXED_INTERP, -1});
        return;
    }
    const uint32_t max_b = maxbits128(in);
    for (int i = 0; i < 17; ++i) {
        const uint32_t b = OPTPFOR_LOGS[i];
        if (b > max_b && i > 0 && OPTPFOR_LOGS[i - 1] >= max_b) continue; // useless
        if (max_b > b && max_b - b > 28) continue;                         // exception coder cannot hold it
        uint32_t nexc = 0;
        if (b < 32) for (size_t k = 0; k < BLOCK; ++k) nexc += (in[k] >> b) != 0;
        const uint32_t words = optpfor_try_b(b, in);            // payload words: packed values + Simple16 exceptions
        const uint32_t space = 1 + 4 * (1 + words);             // type byte + header word + payload
        const float t = m.pfor_base + (nexc == 0 ? 0.f : nexc <= 32 ? m.pfor_exc : m.pfor_exc_many);
        pts.push_back(hybrid_point{t * w, (uint16_t)space, (uint8_t)MIXED_PFOR, (int8_t)b});
    }
    buf.clear();
    varint_g8iu_encode(in, sum, n, buf);
    pts.push_back(hybrid_point{m.varint * w, (uint16_t)(1 + buf.size()), (uint8_t)MIXED_VARINT, -1});
    if (interp_ok) {
        buf.clear();
        interpolative_encode(in, sum, n, buf);
        uint32_t pre[BLOCK];
        pre[0] = in[0];
        for (size_t i = 1; i < n; ++i) pre[i] = pre[i - 1] + in[i];
        const uint32_t live = interp_live_nodes(pre, n - 1, 0, pre[n - 1]);
        pts.push_back(hybrid_point{(m.interp_base + m.interp_node * live) * w, (uint16_t)(1 + buf.size()), (uint8_t)MIXED_INTERP, -1});
    }
    // lower convex hull (optimal_hybrid_index.cpp:92-113): sort by (space, time); keep points that are faster than
    // every smaller one and whose exchange rate d(space)/d(time) keeps increasing
    std::sort(pts.begin(), pts.end(), [](hybrid_point const& a, hybrid_point const& b) {
        return a.space != b.space ? a.space < b.space : a.time < b.time;
    });
    std::vector<double> lam; // lam[i] = exchange rate at which hull[i] replaces hull[i-1]
    for (auto const& cur : pts) {
        if (hull.empty()) { hull.push_back(cur); lam.push_back(0.0); continue; }
        while (true) {
            auto const& prev = hull.back();
            if (cur.time >= prev.time) break; // dominated
            if (cur.space == prev.space) { hull.back() = cur; break; }
            const double l = double(cur.space - prev.space) / double(prev.time - cur.time);
            if (hull.size() > 1 && l < lam.back()) { hull.pop_back(); lam.pop_back(); continue; }
            hull.push_back(cur);
            lam.push_back(l);
            break;
        }
    }
}

// argmin over the hull of time + space / rate  (rate = bytes one is willing to pay per unit of time saved)
inline size_t hybrid_choose(const hybrid_point* h, size_t n, double rate) {
    size_t best = 0;
    for (size_t i = 1; i < n; ++i) {
        const double gain = double(h[best].time - h[i].time), cost = double(h[i].space - h[best].space);
        if (gain * rate >= cost) best = i;
    }
    return best;
}

class hybrid_index_builder {
public:
    hybrid_index_builder(uint64_t num_docs, hybrid_model const& m) : m_num_docs(num_docs), m_model(m) {}

    // access: 2 counters per block (docs decodes, freqs decodes) or null
    void add_posting_list(uint64_t n, const uint32_t* docs, const uint32_t* freqs, const uint32_t* access) {
        if (!n) throw std::invalid_argument("List must be nonempty");
        list_t L;
        L.docs.assign(docs, docs + n);
        L.freqs.assign(freqs, freqs + n);
        const uint64_t blocks = ceil_div(n, (uint64_t)BLOCK);
        if (access) L.access.assign(access, access + 2 * blocks);
        m_lists.push_back(std::move(L));
    }

    // A list that is not stored but regenerated on demand (large synthetic collections): provider(t, docs, freqs)
    // must return the same postings every time it is called for list t.
    typedef std::function<void(size_t, std::vector<uint32_t>&, std::vector<uint32_t>&)> provider_t;
    void add_virtual_list(const uint32_t* access, uint64_t blocks) {
        list_t L;
        L.is_virtual = true;
        if (access) L.access.assign(access, access + 2 * blocks);
        m_lists.push_back(std::move(L));
    }
    void set_provider(provider_t p) { m_provider = std::move(p); }

    // computes every block's hull (threaded). Afterwards min_space()/max_space() are known.
    void analyse(int threads) {
        const size_t V = m_lists.size();
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            std::vector<hybrid_point> hull;
            uint32_t dbuf[BLOCK], fbuf[BLOCK];
            for (;;) {
                const size_t t = next.fetch_add(1);
                if (t >= V) break;
                list_t& L = m_lists[t];
                std::vector<uint32_t> vd, vf;
                if (L.is_virtual) m_provider(t, vd, vf);
                const std::vector<uint32_t>& D = L.is_virtual ? vd : L.docs;
                const std::vector<uint32_t>& F = L.is_virtual ? vf : L.freqs;
                const uint64_t n = D.size(), blocks = ceil_div(n, (uint64_t)BLOCK);
                L.hull_off.assign(2 * blocks + 1, 0);
                L.hull.clear();
                uint32_t last_doc = uint32_t(-1), block_base = 0;
                size_t k = 0;
                for (uint64_t b = 0; b < blocks; ++b) {
                    const uint32_t cur = ((b + 1) * BLOCK <= n) ? BLOCK : (uint32_t)(n % BLOCK);
                    for (uint32_t i = 0; i < cur; ++i, ++k) {
                        dbuf[i] = D[k] - last_doc - 1;
                        last_doc = D[k];
                        fbuf[i] = F[k] - 1;
                    }
                    hybrid_block_points(dbuf, last_doc - block_base - (cur - 1), cur, L.access.empty() ? 0 : L.access[2 * b], m_model, hull);
                    L.hull.insert(L.hull.end(), hull.begin(), hull.end());
                    L.hull_off[2 * b + 1] = (uint32_t)L.hull.size();
                    hybrid_block_points(fbuf, uint32_t(-1), cur, L.access.empty() ? 0 : L.access[2 * b + 1], m_model, hull);
                    L.hull.insert(L.hull.end(), hull.begin(), hull.end());
                    L.hull_off[2 * b + 2] = (uint32_t)L.hull.size();
                    block_base = last_doc + 1;
                }
            }
        };
        run_threads(threads, worker);
        m_analysed = true;
    }

    // payload bytes (all blocks) and model time when every block takes argmin(time + space / rate)
    void evaluate(double rate, uint64_t& space, double& time) const {
        space = 0;
        time = 0;
        for (auto const& L : m_lists)
            for (size_t j = 0; j + 1 < L.hull_off.size(); ++j) {
                const hybrid_point* h = L.hull.data() + L.hull_off[j];
                const size_t c = hybrid_choose(h, L.hull_off[j + 1] - L.hull_off[j], rate);
                space += h[c].space;
                time += h[c].time;
            }
    }
    uint64_t min_space() const { uint64_t s; double t; evaluate(0.0, s, t); return s; }
    uint64_t max_space() const { uint64_t s; double t; evaluate(1e30, s, t); return s; }

    // the largest exchange rate whose total payload fits the budget (bisection; space(rate) is non-decreasing)
    double solve(uint64_t budget) const {
        if (max_space() <= budget) return 1e30;
        double lo = 0.0, hi = 1.0;
        uint64_t s;
        double t;
        for (int i = 0; i < 80; ++i) { evaluate(hi, s, t); if (s > budget) break; hi *= 4.0; }
        for (int i = 0; i < 60; ++i) {
            const double mid = 0.5 * (lo + hi);
            evaluate(mid, s, t);
            if (s <= budget) lo = mid; else hi = mid;
        }
        return lo;
    }

    // encodes the block_mixed index for that rate
    void freeze(double rate, int threads, bytes_t& image, uint64_t type_counts[6]) {
        if (!m_analysed) analyse(threads);
        const size_t V = m_lists.size();
        std::vector<bytes_t> enc(V);
        std::atomic<size_t> next(0);
        std::atomic<uint64_t> tc[6];
        for (auto& c : tc) c = 0;
        auto worker = [&]() {
            uint32_t dbuf[BLOCK], fbuf[BLOCK];
            for (;;) {
                const size_t t = next.fetch_add(1);
                if (t >= V) break;
                list_t const& L = m_lists[t];
                std::vector<uint32_t> vd, vf;
                if (L.is_virtual) m_provider(t, vd, vf);
                const std::vector<uint32_t>& D = L.is_virtual ? vd : L.docs;
                const std::vector<uint32_t>& F = L.is_virtual ? vf : L.freqs;
                bytes_t& out = enc[t];
                const uint32_t n = (uint32_t)D.size();
                vbyte_encode(n, out);
                const uint64_t blocks = ceil_div((uint64_t)n, (uint64_t)BLOCK);
                const size_t begin_maxs = out.size(), begin_endpoints = begin_maxs + 4 * blocks,
                             begin_blocks = begin_endpoints + 4 * (blocks - 1);
                out.resize(begin_blocks);
                uint32_t last_doc = uint32_t(-1), block_base = 0;
                size_t k = 0;
                for (uint64_t b = 0; b < blocks; ++b) {
                    const uint32_t cur = ((b + 1) * BLOCK <= n) ? BLOCK : (n % BLOCK);
                    for (uint32_t i = 0; i < cur; ++i, ++k) {
                        dbuf[i] = D[k] - last_doc - 1;
                        last_doc = D[k];
                        fbuf[i] = F[k] - 1;
                    }
                    std::memcpy(&out[begin_maxs + 4 * b], &last_doc, 4);
                    for (int side = 0; side < 2; ++side) {
                        const hybrid_point* h = L.hull.data() + L.hull_off[2 * b + side];
                        const hybrid_point& c = h[hybrid_choose(h, L.hull_off[2 * b + side + 1] - L.hull_off[2 * b + side], rate)];
                        mixed_encode_type((mixed_type)c.type, c.b, side ? fbuf : dbuf,
                                          side ? uint32_t(-1) : last_doc - block_base - (cur - 1), cur, out);
                        if (cur == BLOCK) ++tc[3 * side + c.type];
                    }
                    if (b != blocks - 1) {
                        const uint32_t ep = (uint32_t)(out.size() - begin_blocks);
                        std::memcpy(&out[begin_endpoints + 4 * b], &ep, 4);
                    }
                    block_base = last_doc + 1;
                }
            }
        };
        run_threads(threads, worker);
        block_index_builder builder(CODEC_MIXED, m_num_docs);
        for (size_t t = 0; t < V; ++t) {
            builder.add_encoded_list(enc[t].data(), enc[t].size());
            bytes_t().swap(enc[t]);
        }
        builder.freeze(image);
        if (type_counts) for (int i = 0; i < 6; ++i) type_counts[i] = tc[i].load();
    }

    uint64_t lists() const { return m_lists.size(); }
    bool analysed() const { return m_analysed; }

private:
    struct list_t {
        std::vector<uint32_t> docs, freqs, access;
        bool is_virtual = false;
        std::vector<hybrid_point> hull;  // concatenated hulls
        std::vector<uint32_t> hull_off;  // 2 * blocks + 1 offsets (docs block 0, freqs block 0, docs block 1, ...)
    };
    template <class F>
    static void run_threads(int threads, F& f) {
        if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(f);
        for (auto& th : pool) th.join();
    }
    uint64_t m_num_docs;
    hybrid_model m_model;
    std::vector<list_t> m_lists;
    provider_t m_provider;
    bool m_analysed = false;
};

} // namespace ds2i_host
