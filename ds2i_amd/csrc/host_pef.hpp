// Host-side (CPU, build-time) writers of the Elias-Fano index family ("opt" index):
//   integer codes            reference integer_codes.hpp:6-45
//   compact_ranked_bitvector reference compact_ranked_bitvector.hpp:14-113
//   indexed_sequence         reference indexed_sequence.hpp:23-87
//   strict_elias_fano        reference strict_elias_fano.hpp:13-36
//   strict_sequence          reference strict_sequence.hpp:24-97
//   optimal_partition        reference optimal_partition.hpp:67-121 ((1+eps) approximate DP)
//   partitioned_sequence     reference partitioned_sequence.hpp:21-120
//   positive_sequence        reference positive_sequence.hpp:15-29
//   freq_index builder/map   reference freq_index.hpp:18-112, 234-243; bitvector_collection.hpp:13-84
// plus the read-side WALKER the GPU upload uses to flatten an opt image into a chunk directory
// (the role SURVEY.md §8 a12 gives to "pre-decode ... at upload", extended to partitions).
// The succinct bit_vector conventions are restated from SURVEY.md Appendix B ("parity unpinned").
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "host_bits.hpp"
#include "host_index.hpp"

namespace ds2i_host {

// ---------------------------------------------------------------- bit helpers
inline void bv_append(bitvec_builder& dst, bitvec_builder const& src) {
    uint64_t n = src.size();
    auto const& w = src.words();
    for (uint64_t i = 0; i < n; i += 64) dst.append_bits(w[i >> 6], (unsigned)std::min<uint64_t>(64, n - i));
}
inline void write_gamma(bitvec_builder& bvb, uint64_t n) {
    uint64_t nn = n + 1;
    uint64_t l = msb64(nn);
    uint64_t hb = uint64_t(1) << l;
    bvb.append_bits(hb, (unsigned)l + 1);
    bvb.append_bits(nn ^ hb, (unsigned)l);
}
inline void write_gamma_nonzero(bitvec_builder& bvb, uint64_t n) { write_gamma(bvb, n - 1); }
inline void write_delta(bitvec_builder& bvb, uint64_t n) {
    uint64_t nn = n + 1;
    uint64_t l = msb64(nn);
    uint64_t hb = uint64_t(1) << l;
    write_gamma(bvb, l);
    bvb.append_bits(nn ^ hb, (unsigned)l);
}

// ---------------------------------------------------------------- ranked bitvector
struct rb_offsets {
    uint64_t universe, n, log_rank1_sampling, log_sampling1, rank1_sample_size, pointer_size, rank1_samples, pointers1,
        rank1_samples_offset, pointers1_offset, bits_offset, end;
    rb_offsets(uint64_t base, uint64_t u, uint64_t n_, global_parameters const& p)
        : universe(u), n(n_), log_rank1_sampling(p.rb_log_rank1_sampling), log_sampling1(p.rb_log_sampling1) {
        rank1_sample_size = ceil_log2(n + 1);
        pointer_size = ceil_log2(u);
        rank1_samples = log_rank1_sampling >= 64 ? 0 : (u >> log_rank1_sampling);
        pointers1 = n >> log_sampling1;
        rank1_samples_offset = base;
        pointers1_offset = rank1_samples_offset + rank1_samples * rank1_sample_size;
        bits_offset = pointers1_offset + pointers1 * pointer_size;
        end = bits_offset + u;
    }
};
inline uint64_t rb_bitsize(global_parameters const& p, uint64_t u, uint64_t n) { return rb_offsets(0, u, n, p).end; }
inline uint64_t ef_bitsize(global_parameters const& p, uint64_t u, uint64_t n) { return ef_offsets(0, u, n, p).end; }

template <class It>
inline void rb_write(bitvec_builder& bvb, It begin, uint64_t universe, uint64_t n, global_parameters const& params) {
    const uint64_t base = bvb.size();
    rb_offsets of(base, universe, n, params);
    bvb.zero_extend(of.end - base);
    auto set_rank1_samples = [&](uint64_t b, uint64_t e, uint64_t rank) {
        if (of.log_rank1_sampling >= 64) return;
        for (uint64_t sample = ceil_div(b, uint64_t(1) << of.log_rank1_sampling); (sample << of.log_rank1_sampling) < e; ++sample) {
            if (!sample) continue;
            bvb.set_bits(of.rank1_samples_offset + (sample - 1) * of.rank1_sample_size, rank, (unsigned)of.rank1_sample_size);
        }
    };
    const uint64_t sample1_mask = (uint64_t(1) << of.log_sampling1) - 1;
    uint64_t last = 0;
    It it = begin;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t v = *it++;
        if (i && v == last) throw std::runtime_error("Duplicate element");
        if (i && v < last) throw std::runtime_error("Sequence is not sorted");
        bvb.set(of.bits_offset + v, 1);
        if (i && (i & sample1_mask) == 0)
            bvb.set_bits(of.pointers1_offset + ((i >> of.log_sampling1) - 1) * of.pointer_size, v, (unsigned)of.pointer_size);
        set_rank1_samples(last + 1, v + 1, i);
        last = v;
    }
    set_rank1_samples(last + 1, universe, n);
}

// ---------------------------------------------------------------- indexed / strict sequences
enum seq_type : int { SEQ_EF = 0, SEQ_RB = 1, SEQ_ALL_ONES = 2 };
static const uint64_t SEQ_TYPE_BITS = 1;

inline global_parameters strict_params(global_parameters p) {
    p.ef_log_sampling0 = 63;
    p.rb_log_rank1_sampling = 63;
    return p;
}

template <bool STRICT>
inline uint64_t seq_bitsize(global_parameters const& params, uint64_t universe, uint64_t n, int* type_out = nullptr) {
    uint64_t best = (universe == n) ? 0 : uint64_t(-1);
    int type = SEQ_ALL_ONES;
    if (best) {
        const global_parameters sp = STRICT ? strict_params(params) : params;
        uint64_t ef = (STRICT ? ef_bitsize(sp, universe - n + 1, n) : ef_bitsize(sp, universe, n)) + SEQ_TYPE_BITS;
        if (ef < best) { best = ef; type = SEQ_EF; }
        uint64_t rb = rb_bitsize(sp, universe, n) + SEQ_TYPE_BITS;
        if (rb < best) { best = rb; type = SEQ_RB; }
    }
    if (type_out) *type_out = type;
    return best;
}

// values: sorted (strictly increasing for STRICT / ranked bitvector), < universe
template <bool STRICT>
inline void seq_write(bitvec_builder& bvb, const uint64_t* v, uint64_t universe, uint64_t n, global_parameters const& params) {
    int type;
    uint64_t cost = seq_bitsize<STRICT>(params, universe, n, &type);
    if (cost) bvb.append_bits((uint64_t)type, (unsigned)SEQ_TYPE_BITS);
    const global_parameters sp = STRICT ? strict_params(params) : params;
    switch (type) {
    case SEQ_EF:
        if (STRICT) {
            std::vector<uint64_t> shifted(n);
            for (uint64_t i = 0; i < n; ++i) shifted[i] = v[i] - i;
            ef_write(bvb, shifted.begin(), universe - n + 1, n, sp);
        } else {
            ef_write(bvb, v, universe, n, sp);
        }
        break;
    case SEQ_RB: rb_write(bvb, v, universe, n, sp); break;
    default: break; // all ones: zero bits
    }
}

// ---------------------------------------------------------------- optimal partition (approximate DP)
struct partition_config { double eps1 = 0.03, eps2 = 0.3; uint64_t fix_cost = 64; };

template <class CostFun>
inline std::vector<uint32_t> optimal_partition(const uint64_t* seq, uint64_t universe, uint64_t size, CostFun cost_fun,
                                               double eps1, double eps2) {
    typedef uint64_t cost_t;
    struct window { uint64_t start = 0, end = 0, min_p = 0, max_p = 0; cost_t cost_upper_bound = 0; };
    const cost_t single_block_cost = cost_fun(universe, size);
    std::vector<cost_t> min_cost(size + 1, single_block_cost);
    min_cost[0] = 0;
    std::vector<window> windows;
    const cost_t cost_lb = cost_fun(1, 1);
    cost_t cost_bound = cost_lb;
    while (eps1 == 0 || cost_bound < cost_lb / eps1) {
        window w;
        w.min_p = seq[0];
        w.cost_upper_bound = cost_bound;
        windows.push_back(w);
        if (cost_bound >= single_block_cost) break;
        cost_bound = (cost_t)(cost_bound * (1 + eps2));
    }
    std::vector<uint32_t> path(size + 1, 0);
    for (uint64_t i = 0; i < size; ++i) {
        uint64_t last_end = i + 1;
        for (auto& w : windows) {
            while (w.end < last_end) { w.max_p = seq[w.end]; ++w.end; }
            cost_t window_cost;
            while (true) {
                window_cost = cost_fun(w.max_p - w.min_p + 1, w.end - w.start);
                if (min_cost[i] + window_cost < min_cost[w.end]) {
                    min_cost[w.end] = min_cost[i] + window_cost;
                    path[w.end] = (uint32_t)i;
                }
                last_end = w.end;
                if (w.end == size) break;
                if (window_cost >= w.cost_upper_bound) break;
                w.max_p = seq[w.end];
                ++w.end;
            }
            w.min_p = seq[w.start] + 1;
            ++w.start;
        }
    }
    std::vector<uint32_t> partition;
    uint64_t cur = size;
    while (cur != 0) { partition.push_back((uint32_t)cur); cur = path[cur]; }
    std::reverse(partition.begin(), partition.end());
    return partition;
}

// ---------------------------------------------------------------- partitioned sequence
template <bool STRICT>
inline void partitioned_write(bitvec_builder& bvb, const uint64_t* seq, uint64_t universe, uint64_t n,
                              global_parameters const& params, partition_config const& conf = partition_config()) {
    auto cost_fun = [&](uint64_t u, uint64_t m) { return seq_bitsize<STRICT>(params, u, m) + conf.fix_cost; };
    std::vector<uint32_t> part = optimal_partition(seq, universe, n, cost_fun, conf.eps1, conf.eps2);
    const uint64_t partitions = part.size();
    write_gamma_nonzero(bvb, partitions);
    std::vector<uint64_t> cur;
    if (partitions == 1) {
        const uint64_t cur_base = seq[0];
        cur.resize(n);
        for (uint64_t i = 0; i < n; ++i) cur[i] = seq[i] - cur_base;
        bvb.append_bits(cur_base, (unsigned)ceil_log2(universe));
        if (n > 1) {
            if (cur_base + cur.back() + 1 == universe) write_delta(bvb, 0);
            else write_delta(bvb, cur.back());
        }
        seq_write<STRICT>(bvb, cur.data(), cur.back() + 1, n, params);
        return;
    }
    bitvec_builder bv_sequences;
    std::vector<uint64_t> endpoints, upper_bounds, sizes(part.begin(), part.end());
    uint64_t cur_i = 0, cur_base = seq[0];
    upper_bounds.push_back(cur_base);
    for (uint64_t p = 0; p < partitions; ++p) {
        cur.clear();
        uint64_t value = 0;
        for (; cur_i < part[p]; ++cur_i) {
            value = seq[cur_i];
            cur.push_back(value - cur_base);
        }
        seq_write<STRICT>(bv_sequences, cur.data(), cur.back() + 1, cur.size(), params);
        endpoints.push_back(bv_sequences.size());
        upper_bounds.push_back(value);
        cur_base = value + 1;
    }
    bitvec_builder bv_sizes, bv_ub;
    ef_write(bv_sizes, sizes.begin(), n, partitions - 1, params);
    ef_write(bv_ub, upper_bounds.begin(), universe, partitions + 1, params);
    const uint64_t endpoint_bits = ceil_log2(bv_sequences.size() + 1);
    write_gamma(bvb, endpoint_bits);
    bv_append(bvb, bv_sizes);
    bv_append(bvb, bv_ub);
    for (uint64_t p = 0; p + 1 < endpoints.size(); ++p) bvb.append_bits(endpoints[p], (unsigned)endpoint_bits);
    bv_append(bvb, bv_sequences);
}

// ---------------------------------------------------------------- freq_index<partitioned, positive<partitioned<strict>>>
class opt_index_builder {
public:
    opt_index_builder(uint64_t num_docs, global_parameters const& params = global_parameters())
        : m_num_docs(num_docs), m_params(params) {
        m_docs_endpoints.push_back(0);
        m_freqs_endpoints.push_back(0);
    }
    // one list -> (docs bits, freqs bits); thread-safe (no shared state), used by the parallel synth builder
    static void encode_list(uint64_t num_docs, global_parameters const& params, uint64_t n, const uint32_t* docs,
                            const uint32_t* freqs, bitvec_builder& docs_bits, bitvec_builder& freqs_bits) {
        if (!n) throw std::invalid_argument("List must be nonempty");
        uint64_t occurrences = 0;
        std::vector<uint64_t> d(n), cum(n);
        for (uint64_t i = 0; i < n; ++i) {
            d[i] = docs[i];
            occurrences += freqs[i];
            cum[i] = occurrences; // positive_sequence: strictly increasing prefix sums
        }
        write_gamma_nonzero(docs_bits, occurrences);
        if (occurrences > 1) docs_bits.append_bits(n, (unsigned)ceil_log2(occurrences + 1));
        partitioned_write<false>(docs_bits, d.data(), num_docs, n, params);
        partitioned_write<true>(freqs_bits, cum.data(), occurrences + 1, n, params);
    }
    void add_posting_list(uint64_t n, const uint32_t* docs, const uint32_t* freqs) {
        bitvec_builder db, fb;
        encode_list(m_num_docs, m_params, n, docs, freqs, db, fb);
        add_encoded(db, fb);
    }
    void add_encoded(bitvec_builder const& db, bitvec_builder const& fb) {
        bv_append(m_docs, db);
        m_docs_endpoints.push_back(m_docs.size());
        bv_append(m_freqs, fb);
        m_freqs_endpoints.push_back(m_freqs.size());
    }
    // image = 5 B params | u64 num_docs | collection(docs) | collection(freqs)
    // collection = u64 m_size | bit_vector m_endpoints | bit_vector m_bitvectors ; bit_vector = u64 bits | u64 nwords | words
    void freeze(bytes_t& out) const {
        out.push_back(m_params.ef_log_sampling0);
        out.push_back(m_params.ef_log_sampling1);
        out.push_back(m_params.rb_log_rank1_sampling);
        out.push_back(m_params.rb_log_sampling1);
        out.push_back(m_params.log_partition_size);
        put_pod<uint64_t>(out, m_num_docs);
        freeze_collection(out, m_docs, m_docs_endpoints);
        freeze_collection(out, m_freqs, m_freqs_endpoints);
    }

private:
    static void put_bv(bytes_t& out, bitvec_builder const& bv) {
        put_pod<uint64_t>(out, bv.size());
        put_pod<uint64_t>(out, bv.words().size());
        const uint8_t* w = (const uint8_t*)bv.words().data();
        out.insert(out.end(), w, w + 8 * bv.words().size());
    }
    void freeze_collection(bytes_t& out, bitvec_builder const& bits, std::vector<uint64_t> const& endpoints) const {
        const uint64_t size = endpoints.size() - 1;
        put_pod<uint64_t>(out, size);
        bitvec_builder ep;
        if (size) ef_write(ep, endpoints.begin(), bits.size(), size, m_params);
        put_bv(out, ep);
        put_bv(out, bits);
    }
    uint64_t m_num_docs;
    global_parameters m_params;
    bitvec_builder m_docs, m_freqs;
    std::vector<uint64_t> m_docs_endpoints, m_freqs_endpoints;
};

} // namespace ds2i_host
