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

// compact_ranked_bitvector image (layout: `rb_offsets`): rank1_samples | pointers1 | the `universe`-bit characteristic
// vector. As for Elias-Fano the auxiliary arrays are written from their meaning, in a second pass over the finished
// vector: rank1_samples[k-1] = number of ones before position k * 2^r (k >= 1), pointers1[k-1] = position of the one
// with k * 2^s1 ones before it.
template <class It>
inline void rb_write(bitvec_builder& bvb, It begin, uint64_t universe, uint64_t n, global_parameters const& params) {
    const rb_offsets of(bvb.size(), universe, n, params);
    bvb.zero_extend(of.end - of.rank1_samples_offset);
    uint64_t prev = 0;
    It it = begin;
    for (uint64_t i = 0; i < n; ++i, ++it) {
        const uint64_t v = *it;
        if (i && v == prev) throw std::runtime_error("Duplicate element");
        if (v < prev) throw std::runtime_error("Sequence is not sorted");
        prev = v;
        bvb.set(of.bits_offset + v, 1);
    }
    if (!of.rank1_samples && !of.pointers1) return;
    const uint64_t every1 = (uint64_t(1) << of.log_sampling1) - 1;
    const uint64_t rank_step = of.log_rank1_sampling < 64 ? uint64_t(1) << of.log_rank1_sampling : 0; // 0 = no samples
    scan_bits(bvb, of.bits_offset, universe, [&](uint64_t pos, uint64_t bits, uint64_t nbits, uint64_t ones_before) {
        // rank samples: rank1_samples[k-1] = number of 1s before position k * 2^r (k >= 1) -- the sampled positions inside
        // this chunk, each by one popcount
        if (of.rank1_samples && rank_step)
            for (uint64_t sp = pos ? (pos + rank_step - 1) / rank_step * rank_step : rank_step; sp < pos + nbits; sp += rank_step) {
                const uint64_t below = sp - pos; // < 64
                bvb.set_bits(of.rank1_samples_offset + ((sp >> of.log_rank1_sampling) - 1) * of.rank1_sample_size,
                             ones_before + (uint64_t)__builtin_popcountll(bits & ((uint64_t(1) << below) - 1)), (unsigned)of.rank1_sample_size);
            }
        if (of.pointers1 && bits)
            for_each_set_bit(bits, [&](uint64_t b, uint64_t k) {
                const uint64_t ones = ones_before + k;
                if (ones && !(ones & every1))
                    bvb.set_bits(of.pointers1_offset + ((ones >> of.log_sampling1) - 1) * of.pointer_size, pos + b, (unsigned)of.pointer_size);
            });
    });
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

// Shortest path over the sparsified partition DAG of "Partitioned Elias-Fano indexes" (Ottaviano & Venturini,
// SIGIR'14, section 5): node p = "the first p elements are partitioned", edge (i -> e) = one partition holding
// elements [i, e) at cost_fun(universe of the run, e - i). Only edges of geometrically growing cost classes are kept:
// class c may extend a partition starting at i while its cost stays below budget[c] = c_min * (1+eps2)^c, and classes
// stop at c_min / eps1. All classes start at the same node i, so the only per-class state is how far the class has
// reached (`reach`) and the largest element it covers; the run's smallest possible value follows from i alone.
// Parameters, edge order and the strict `<` of the relaxation match the reference builder
// (optimal_partition.hpp:67-121, eps1 = 0.03, eps2 = 0.3, configuration.hpp), so the chosen end points -- and with
// them the image -- are the ones ds2i's create_freq_index would choose.
template <class CostFun>
inline std::vector<uint32_t> optimal_partition(const uint64_t* seq, uint64_t universe, uint64_t size, CostFun cost_fun,
                                               double eps1, double eps2) {
    const uint64_t whole = cost_fun(universe, size);
    std::vector<uint64_t> dist(size + 1, whole); // cheapest known cost of partitioning the first p elements
    std::vector<uint32_t> parent(size + 1, 0);
    dist[0] = 0;
    std::vector<uint64_t> budget, reach, reach_top;
    const uint64_t c_min = cost_fun(1, 1);
    for (uint64_t bnd = c_min; eps1 == 0 || bnd < c_min / eps1; bnd = (uint64_t)(bnd * (1 + eps2))) {
        budget.push_back(bnd);
        if (bnd >= whole) break;
    }
    reach.assign(budget.size(), 0);
    reach_top.assign(budget.size(), 0);
    for (uint64_t i = 0; i < size; ++i) {
        const uint64_t floor_value = i ? seq[i - 1] + 1 : seq[0]; // smallest value a partition starting at i can hold
        const uint64_t here = dist[i];
        uint64_t at_least = i + 1; // a class never ends before the previous (cheaper) class did
        for (size_t c = 0; c < budget.size(); ++c) {
            uint64_t e = reach[c], top = reach_top[c];
            while (e < at_least) top = seq[e++];
            for (;;) {
                const uint64_t edge = cost_fun(top - floor_value + 1, e - i);
                if (here + edge < dist[e]) {
                    dist[e] = here + edge;
                    parent[e] = (uint32_t)i;
                }
                if (e == size || edge >= budget[c]) break;
                top = seq[e++];
            }
            reach[c] = e;
            reach_top[c] = top;
            at_least = e;
        }
    }
    std::vector<uint32_t> ends;
    for (uint64_t p = size; p != 0; p = parent[p]) ends.push_back((uint32_t)p);
    std::reverse(ends.begin(), ends.end());
    return ends;
}

// ---------------------------------------------------------------- partitioned sequences
// On-disk layout shared by partitioned_sequence (partitioned_sequence.hpp:131-178 reads it) and
// uniform_partitioned_sequence (uniform_partitioned_sequence.hpp:120-160), SURVEY.md Appendix A6:
//
//   gamma+(P)                                   P = number of partitions
//   P == 1:  first value in ceil_log2(universe) bits | n > 1: delta(span), span = last - first, or 0 when the run
//            reaches universe - 1 | base sequence of (v - first) over span + 1
//   P  > 1:  gamma(endpoint_bits) | [EF of the P-1 inner partition END INDEXES over universe n -- optimal partitions
//            only; fixed-size partitions need none] | EF of {first value, last value of partition 0, .., of partition
//            P-1} over `universe` | P-1 end offsets (bits, relative to the first partition's image) of
//            endpoint_bits each | the partitions' base sequences back to back; partition p stores v - (last value of
//            partition p-1) - 1 (partition 0: v - first value) over (its last stored value + 1)
//
// `ends` = the exclusive end index of every partition (ends.back() == n).
template <bool STRICT>
inline void write_partition_table(bitvec_builder& out, const uint64_t* seq, uint64_t universe, uint64_t n,
                                  std::vector<uint32_t> const& ends, bool with_sizes, global_parameters const& params) {
    const uint64_t P = ends.size();
    write_gamma_nonzero(out, P);
    std::vector<uint64_t> rel;
    // stores seq[from, to) relative to `origin` as one base sequence appended to `dst`
    auto emit_partition = [&](bitvec_builder& dst, uint64_t from, uint64_t to, uint64_t origin) {
        rel.resize(to - from);
        for (uint64_t i = from; i < to; ++i) rel[i - from] = seq[i] - origin;
        seq_write<STRICT>(dst, rel.data(), rel.back() + 1, to - from, params);
    };
    if (P == 1) {
        const uint64_t first = seq[0], span = seq[n - 1] - first;
        out.append_bits(first, (unsigned)ceil_log2(universe));
        if (n > 1) write_delta(out, seq[n - 1] + 1 == universe ? 0 : span);
        emit_partition(out, 0, n, first);
        return;
    }
    bitvec_builder body;
    std::vector<uint64_t> bounds(1, seq[0]), offsets_after, inner_ends(ends.begin(), ends.end());
    for (uint64_t p = 0, from = 0; p < P; from = ends[p++]) {
        emit_partition(body, from, ends[p], p ? seq[from - 1] + 1 : seq[0]);
        offsets_after.push_back(body.size());
        bounds.push_back(seq[ends[p] - 1]);
    }
    const unsigned endpoint_bits = (unsigned)ceil_log2(body.size() + 1);
    write_gamma(out, endpoint_bits);
    bitvec_builder table;
    if (with_sizes) ef_write(table, inner_ends.begin(), n, P - 1, params);
    ef_write(table, bounds.begin(), universe, P + 1, params);
    bv_append(out, table);
    for (uint64_t p = 0; p + 1 < P; ++p) out.append_bits(offsets_after[p], endpoint_bits);
    bv_append(out, body);
}

// partitioned_sequence: end points from the (1+eps)-approximate shortest path (optimal_partition above)
template <bool STRICT>
inline void partitioned_write(bitvec_builder& bvb, const uint64_t* seq, uint64_t universe, uint64_t n,
                              global_parameters const& params, partition_config const& conf = partition_config()) {
    auto cost_fun = [&](uint64_t u, uint64_t m) { return seq_bitsize<STRICT>(params, u, m) + conf.fix_cost; };
    write_partition_table<STRICT>(bvb, seq, universe, n, optimal_partition(seq, universe, n, cost_fun, conf.eps1, conf.eps2),
                                  true, params);
}

// uniform_partitioned_sequence: fixed partitions of 2^log_partition_size elements
template <bool STRICT>
inline void uniform_write(bitvec_builder& bvb, const uint64_t* seq, uint64_t universe, uint64_t n, global_parameters const& params) {
    const uint64_t psize = uint64_t(1) << params.log_partition_size;
    std::vector<uint32_t> ends;
    for (uint64_t e = psize; e < n; e += psize) ends.push_back((uint32_t)e);
    ends.push_back((uint32_t)n);
    write_partition_table<STRICT>(bvb, seq, universe, n, ends, false, params);
}

// The four freq_index instantiations of index_types.hpp:18-32; numbered like enum ds2i_hip_index_kind.
//   opt     : partitioned_sequence<indexed>          + positive<partitioned_sequence<strict_sequence>>
//   ef      : compact_elias_fano                     + positive<strict_elias_fano>
//   single  : indexed_sequence                       + positive<strict_sequence>
//   uniform : uniform_partitioned_sequence<indexed>  + positive<uniform_partitioned_sequence<strict_sequence>>
enum freq_layout : int { LAYOUT_OPT = 5, LAYOUT_EF = 6, LAYOUT_SINGLE = 7, LAYOUT_UNIFORM = 8 };
inline bool is_freq_layout(int kind) { return kind >= LAYOUT_OPT && kind <= LAYOUT_UNIFORM; }

template <bool STRICT>
inline void layout_write(int layout, bitvec_builder& bvb, const uint64_t* seq, uint64_t universe, uint64_t n,
                         global_parameters const& params) {
    switch (layout) {
    case LAYOUT_OPT: partitioned_write<STRICT>(bvb, seq, universe, n, params); break;
    case LAYOUT_UNIFORM: uniform_write<STRICT>(bvb, seq, universe, n, params); break;
    case LAYOUT_SINGLE: seq_write<STRICT>(bvb, seq, universe, n, params); break;
    case LAYOUT_EF:
        if (STRICT) { // strict_elias_fano.hpp:21-36: v_i - i over universe - n + 1, the index's own parameters
            std::vector<uint64_t> shifted(n);
            for (uint64_t i = 0; i < n; ++i) shifted[i] = seq[i] - i;
            ef_write(bvb, shifted.begin(), universe - n + 1, n, params);
        } else {
            ef_write(bvb, seq, universe, n, params);
        }
        break;
    default: throw std::invalid_argument("unknown freq_index layout");
    }
}

// ---------------------------------------------------------------- freq_index<DocsSequence, positive_sequence<...>>
class opt_index_builder {
public:
    opt_index_builder(uint64_t num_docs, global_parameters const& params = global_parameters(), int layout = LAYOUT_OPT)
        : m_num_docs(num_docs), m_params(params), m_layout(layout) {
        m_docs_endpoints.push_back(0);
        m_freqs_endpoints.push_back(0);
    }
    // one list -> (docs bits, freqs bits); thread-safe (no shared state), used by the parallel synth builder
    static void encode_list(uint64_t num_docs, global_parameters const& params, uint64_t n, const uint32_t* docs,
                            const uint32_t* freqs, bitvec_builder& docs_bits, bitvec_builder& freqs_bits,
                            int layout = LAYOUT_OPT) {
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
        layout_write<false>(layout, docs_bits, d.data(), num_docs, n, params);
        layout_write<true>(layout, freqs_bits, cum.data(), occurrences + 1, n, params);
    }
    void add_posting_list(uint64_t n, const uint32_t* docs, const uint32_t* freqs) {
        bitvec_builder db, fb;
        encode_list(m_num_docs, m_params, n, docs, freqs, db, fb, m_layout);
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
    int m_layout;
    bitvec_builder m_docs, m_freqs;
    std::vector<uint64_t> m_docs_endpoints, m_freqs_endpoints;
};

} // namespace ds2i_host

// =====================================================================================================
// Read side used at GPU upload: walk an opt image and flatten it into a per-list CHUNK DIRECTORY.
// A chunk = <= 128 consecutive postings that lie inside ONE docs partition and ONE freqs partition; the
// device decodes a chunk like a block of a block index (cmax[] plays block_max[]). The on-disk image is
// uploaded unchanged (the two bit vectors); the directory is auxiliary, like the list-offset table.
// =====================================================================================================
namespace ds2i_host {

struct bit_cursor { // succinct::bit_vector::enumerator
    bitview const* bv;
    uint64_t pos;
    uint64_t take(unsigned l) { uint64_t v = l ? bv->get_bits(pos, l) : 0; pos += l; return v; }
    uint64_t skip_zeros() { uint64_t z = 0; while (!bv->get(pos)) { ++pos; ++z; } ++pos; return z; }
};
inline uint64_t read_gamma(bit_cursor& it) { uint64_t l = it.skip_zeros(); return (it.take((unsigned)l) | (uint64_t(1) << l)) - 1; }
inline uint64_t read_delta(bit_cursor& it) { uint64_t l = read_gamma(it); return (it.take((unsigned)l) | (uint64_t(1) << l)) - 1; }

struct seq_partition {
    uint64_t begin, end;   // positions [begin, end) in the list
    uint64_t base;         // value added to the partition's local values
    uint64_t upper_bound;  // last value of the partition (absolute)
    int type;              // SEQ_EF / SEQ_RB / SEQ_ALL_ONES
    uint64_t lower_bits;   // EF
    uint64_t hi_offset;    // EF: higher_bits_offset ; RB: bits_offset (absolute bit positions)
    uint64_t lo_offset;    // EF: lower_bits_offset
};

template <bool STRICT>
inline void describe_raw_ef(uint64_t offset, uint64_t universe, uint64_t n, global_parameters const& params, seq_partition& p) {
    ef_offsets of(offset, STRICT ? universe - n + 1 : universe, n, params);
    p.type = SEQ_EF;
    p.lower_bits = of.lower_bits;
    p.hi_offset = of.higher_bits_offset;
    p.lo_offset = of.lower_bits_offset;
}

template <bool STRICT>
inline void describe_base(bitview const& bv, uint64_t offset, uint64_t universe, uint64_t n, global_parameters const& params,
                          seq_partition& p) {
    const global_parameters sp = STRICT ? strict_params(params) : params;
    p.lower_bits = 0;
    p.hi_offset = p.lo_offset = 0;
    if (universe == n) { p.type = SEQ_ALL_ONES; return; }
    p.type = (int)(bv.get_bits(offset, 1));
    if (p.type == SEQ_EF) {
        ef_offsets of(offset + 1, STRICT ? universe - n + 1 : universe, n, sp);
        p.lower_bits = of.lower_bits;
        p.hi_offset = of.higher_bits_offset;
        p.lo_offset = of.lower_bits_offset;
    } else {
        rb_offsets of(offset + 1, universe, n, sp);
        p.hi_offset = of.bits_offset;
    }
}

// partitioned_sequence header (partitioned_sequence.hpp:131-178) -> list of partitions
template <bool STRICT>
inline void walk_partitioned(bitview const& bv, uint64_t offset, uint64_t universe, uint64_t n, global_parameters const& params,
                             std::vector<seq_partition>& out) {
    out.clear();
    bit_cursor it{&bv, offset};
    const uint64_t partitions = read_gamma(it) + 1;
    if (partitions == 1) {
        seq_partition p;
        p.begin = 0;
        p.end = n;
        p.base = it.take((unsigned)ceil_log2(universe));
        uint64_t ub = 0;
        if (n > 1) {
            uint64_t ud = read_delta(it);
            ub = ud ? ud : (universe - p.base - 1);
        }
        p.upper_bound = p.base + ub;
        describe_base<STRICT>(bv, it.pos, ub + 1, n, params, p);
        out.push_back(p);
        return;
    }
    const uint64_t endpoint_bits = read_gamma(it);
    uint64_t cur = it.pos;
    std::vector<uint64_t> sizes, ubs;
    ef_decode_all(bv, cur, n, partitions - 1, params, sizes);
    cur += ef_bitsize(params, n, partitions - 1);
    ef_decode_all(bv, cur, universe, partitions + 1, params, ubs);
    cur += ef_bitsize(params, universe, partitions + 1);
    const uint64_t endpoints_offset = cur;
    const uint64_t sequences_offset = cur + endpoint_bits * (partitions - 1);
    out.reserve(partitions);
    for (uint64_t k = 0; k < partitions; ++k) {
        seq_partition p;
        p.begin = k ? sizes[k - 1] : 0;
        p.end = k + 1 < partitions ? sizes[k] : n;
        p.base = k ? ubs[k] + 1 : ubs[0];
        p.upper_bound = ubs[k + 1];
        uint64_t endpoint = k ? bv.get_bits(endpoints_offset + (k - 1) * endpoint_bits, (unsigned)endpoint_bits) : 0;
        describe_base<STRICT>(bv, sequences_offset + endpoint, p.upper_bound - p.base + 1, p.end - p.begin, params, p);
        out.push_back(p);
    }
}

// uniform_partitioned_sequence header (uniform_partitioned_sequence.hpp:121-166)
template <bool STRICT>
inline void walk_uniform(bitview const& bv, uint64_t offset, uint64_t universe, uint64_t n, global_parameters const& params,
                         std::vector<seq_partition>& out) {
    out.clear();
    bit_cursor it{&bv, offset};
    const uint64_t partitions = read_gamma(it) + 1;
    if (partitions == 1) {
        seq_partition p;
        p.begin = 0;
        p.end = n;
        p.base = it.take((unsigned)ceil_log2(universe));
        uint64_t ub = 0;
        if (n > 1) {
            uint64_t ud = read_delta(it);
            ub = ud ? ud : (universe - p.base - 1);
        }
        p.upper_bound = p.base + ub;
        describe_base<STRICT>(bv, it.pos, ub + 1, n, params, p);
        out.push_back(p);
        return;
    }
    if (partitions != ceil_div(n, uint64_t(1) << params.log_partition_size))
        throw std::runtime_error("corrupt uniform partitioned sequence");
    const uint64_t endpoint_bits = read_gamma(it);
    uint64_t cur = it.pos;
    std::vector<uint64_t> ubs;
    ef_decode_all(bv, cur, universe, partitions + 1, params, ubs);
    cur += ef_bitsize(params, universe, partitions + 1);
    const uint64_t endpoints_offset = cur;
    const uint64_t sequences_offset = cur + endpoint_bits * (partitions - 1);
    out.reserve(partitions);
    for (uint64_t k = 0; k < partitions; ++k) {
        seq_partition p;
        p.begin = k << params.log_partition_size;
        p.end = std::min(n, (k + 1) << params.log_partition_size);
        p.base = k ? ubs[k] + 1 : ubs[0];
        p.upper_bound = ubs[k + 1];
        uint64_t endpoint = k ? bv.get_bits(endpoints_offset + (k - 1) * endpoint_bits, (unsigned)endpoint_bits) : 0;
        describe_base<STRICT>(bv, sequences_offset + endpoint, p.upper_bound - p.base + 1, p.end - p.begin, params, p);
        out.push_back(p);
    }
}

// any of the four layouts -> list of partitions (ef / single are one partition with base 0)
template <bool STRICT>
inline void walk_layout(int layout, bitview const& bv, uint64_t offset, uint64_t universe, uint64_t n,
                        global_parameters const& params, std::vector<seq_partition>& out) {
    if (layout == LAYOUT_OPT) return walk_partitioned<STRICT>(bv, offset, universe, n, params, out);
    if (layout == LAYOUT_UNIFORM) return walk_uniform<STRICT>(bv, offset, universe, n, params, out);
    out.clear();
    seq_partition p;
    p.begin = 0;
    p.end = n;
    p.base = 0;
    p.upper_bound = universe - 1;
    if (layout == LAYOUT_EF) {
        p.lower_bits = 0;
        describe_raw_ef<STRICT>(offset, universe, n, params, p);
    } else {
        describe_base<STRICT>(bv, offset, universe, n, params, p);
    }
    out.push_back(p);
}

// sequential cursor over the set bits of one partition's high bits / bitmap
struct ones_cursor {
    bitview const* bv = nullptr;
    uint64_t pos = 0;   // bit position of the current (rank-th) one
    uint64_t rank = 0;  // index of the element at `pos`
    void start(bitview const& b, uint64_t from) { // positions on the first one at/after `from`
        bv = &b;
        pos = from;
        rank = 0;
        while (!bv->get(pos)) ++pos;
    }
    void advance_to(uint64_t r) { // r >= rank; afterwards pos = position of the r-th one
        while (rank < r) {
            uint64_t scan = pos + 1, w;
            for (;;) {
                w = bv->get_bits(scan, 56);
                if (w) break;
                scan += 56;
            }
            const uint64_t pc = (uint64_t)__builtin_popcountll(w);
            if (rank + pc <= r) { // take the whole window: land on its last one
                rank += pc;
                pos = scan + (63 - (uint64_t)__builtin_clzll(w));
            } else {
                for (uint64_t k = r - rank - 1; k; --k) w &= w - 1;
                pos = scan + (uint64_t)__builtin_ctzll(w);
                rank = r;
            }
        }
    }
};

// one side (docs or freqs) of a chunk, as the device consumes it (bit offsets relative to the list's sequence start)
struct chunk_side { uint32_t type, l, base, hi, hbias, lo, span; };

struct pef_chunk { // 12 dwords, see device_pef.hpp
    uint32_t gpos, packed, d_base, d_hi, d_hbias, d_lo, f_base, f_hi, f_hbias, f_lo, f_prev, spans;
};

struct pef_list_dir {
    uint32_t n = 0;
    uint64_t docs_bit0 = 0, freqs_bit0 = 0; // absolute bit offsets all rel fields are relative to
    std::vector<uint32_t> cmax;
    std::vector<pef_chunk> chunks;
};

struct opt_index_view {
    int layout = LAYOUT_OPT;
    global_parameters params;
    uint64_t num_docs = 0, size = 0;
    bitview docs_endpoints, docs_bits, freqs_endpoints, freqs_bits;
    std::vector<uint64_t> docs_off, freqs_off;
    static void take_bv(reader& r, bitview& bv) {
        bv.nbits = r.pod<uint64_t>();
        uint64_t nw = r.pod<uint64_t>();
        if (nw != ceil_div(bv.nbits, (uint64_t)64)) throw std::runtime_error("bad bit vector in opt image");
        bv.nbytes = 8 * nw;
        bv.bytes = r.take(8 * nw);
    }
    void parse(const void* image, size_t bytes) {
        reader r(image, bytes);
        params.ef_log_sampling0 = r.pod<uint8_t>();
        params.ef_log_sampling1 = r.pod<uint8_t>();
        params.rb_log_rank1_sampling = r.pod<uint8_t>();
        params.rb_log_sampling1 = r.pod<uint8_t>();
        params.log_partition_size = r.pod<uint8_t>();
        num_docs = r.pod<uint64_t>();
        size = r.pod<uint64_t>();
        take_bv(r, docs_endpoints);
        take_bv(r, docs_bits);
        uint64_t fsize = r.pod<uint64_t>();
        take_bv(r, freqs_endpoints);
        take_bv(r, freqs_bits);
        if (fsize != size) throw std::runtime_error("docs / freqs collections differ in size");
        if (num_docs > 0xFFFFFFFFull) throw std::runtime_error("num_docs exceeds 32 bits");
        if (size) {
            ef_decode_all(docs_endpoints, 0, docs_bits.nbits, size, params, docs_off);
            ef_decode_all(freqs_endpoints, 0, freqs_bits.nbits, size, params, freqs_off);
        }
    }

    // flatten list t into its chunk directory
    void build_dir(uint64_t t, pef_list_dir& dir) const {
        bit_cursor it{&docs_bits, docs_off[t]};
        const uint64_t occurrences = read_gamma(it) + 1;
        uint64_t n = 1;
        if (occurrences > 1) n = it.take((unsigned)ceil_log2(occurrences + 1));
        if (!n || n > 0xFFFFFFFFull) throw std::runtime_error("corrupt opt posting list header");
        std::vector<seq_partition> dp, fp;
        walk_layout<false>(layout, docs_bits, it.pos, num_docs, n, params, dp);
        walk_layout<true>(layout, freqs_bits, freqs_off[t], occurrences + 1, n, params, fp);
        dir.n = (uint32_t)n;
        dir.docs_bit0 = docs_off[t];
        dir.freqs_bit0 = freqs_off[t];
        dir.cmax.clear();
        dir.chunks.clear();
        size_t di = 0, fi = 0;
        ones_cursor dc, fc;
        auto start_cursor = [](ones_cursor& c, bitview const& bv, seq_partition const& p) {
            if (p.type == SEQ_EF) c.start(bv, p.hi_offset);
            else if (p.type == SEQ_RB) c.start(bv, p.hi_offset);
        };
        start_cursor(dc, docs_bits, dp[0]);
        start_cursor(fc, freqs_bits, fp[0]);
        uint64_t pos = 0, prev_s = 0;
        auto value_at = [&](bitview const& bv, seq_partition const& p, ones_cursor& c, uint64_t i, bool strict) -> uint64_t {
            // local element i of partition p (cursor already at rank i for EF / RB)
            if (p.type == SEQ_ALL_ONES) return p.base + i;
            if (p.type == SEQ_RB) return p.base + (c.pos - p.hi_offset);
            uint64_t high = c.pos - p.hi_offset - i - 1;
            uint64_t low = bv.get_bits(p.lo_offset + i * p.lower_bits, (unsigned)p.lower_bits);
            return p.base + ((high << p.lower_bits) | low) + (strict ? i : 0);
        };
        auto rel32 = [](uint64_t abs, uint64_t bit0) -> uint32_t {
            uint64_t r = abs - bit0;
            if (r > 0xFFFFFFFFull) throw std::runtime_error("posting list too large for the chunk directory");
            return (uint32_t)r;
        };
        while (pos < n) {
            while (dp[di].end <= pos) { ++di; start_cursor(dc, docs_bits, dp[di]); }
            while (fp[fi].end <= pos) { ++fi; start_cursor(fc, freqs_bits, fp[fi]); }
            const seq_partition& D = dp[di];
            const seq_partition& F = fp[fi];
            const uint64_t cnt = std::min<uint64_t>(128, std::min(D.end, F.end) - pos);
            const uint64_t di0 = pos - D.begin, fi0 = pos - F.begin;
            pef_chunk c{};
            c.gpos = (uint32_t)pos;
            uint32_t dspan = 0, fspan = 0;
            // ---- docs side
            if (D.type != SEQ_ALL_ONES) dc.advance_to(di0);
            const uint64_t d_first_hi = dc.pos;
            if (D.type == SEQ_ALL_ONES) {
                c.d_base = (uint32_t)(D.base + di0);
            } else {
                c.d_base = (uint32_t)D.base;
                c.d_hi = rel32(d_first_hi, dir.docs_bit0);
                c.d_hbias = D.type == SEQ_EF ? rel32(D.hi_offset + di0 + 1, dir.docs_bit0) : rel32(D.hi_offset, dir.docs_bit0);
                c.d_lo = D.type == SEQ_EF ? rel32(D.lo_offset + di0 * D.lower_bits, dir.docs_bit0) : 0;
            }
            // last element of the chunk: cmax + span of high bits
            if (D.type != SEQ_ALL_ONES) dc.advance_to(di0 + cnt - 1);
            const uint64_t last_doc = value_at(docs_bits, D, dc, di0 + cnt - 1, false);
            if (D.type != SEQ_ALL_ONES) dspan = (uint32_t)(dc.pos - d_first_hi + 1);
            // ---- freqs side (strictly increasing prefix sums S)
            if (F.type != SEQ_ALL_ONES) fc.advance_to(fi0);
            const uint64_t f_first_hi = fc.pos;
            if (F.type == SEQ_ALL_ONES) {
                c.f_base = (uint32_t)(F.base + fi0);
            } else {
                c.f_base = (uint32_t)(F.type == SEQ_EF ? F.base + fi0 : F.base);
                c.f_hi = rel32(f_first_hi, dir.freqs_bit0);
                c.f_hbias = F.type == SEQ_EF ? rel32(F.hi_offset + fi0 + 1, dir.freqs_bit0) : rel32(F.hi_offset, dir.freqs_bit0);
                c.f_lo = F.type == SEQ_EF ? rel32(F.lo_offset + fi0 * F.lower_bits, dir.freqs_bit0) : 0;
            }
            c.f_prev = (uint32_t)prev_s;
            if (F.type != SEQ_ALL_ONES) fc.advance_to(fi0 + cnt - 1);
            prev_s = value_at(freqs_bits, F, fc, fi0 + cnt - 1, true);
            if (F.type != SEQ_ALL_ONES) fspan = (uint32_t)(fc.pos - f_first_hi + 1);
            if (dspan > 0xFFFF || fspan > 0xFFFF) { dspan = std::min(dspan, 0xFFFFu); fspan = std::min(fspan, 0xFFFFu); } // 0xFFFF = "scan until count reached"
            c.packed = (uint32_t)cnt | ((uint32_t)D.type << 8) | ((uint32_t)D.lower_bits << 10) | ((uint32_t)F.type << 16) |
                       ((uint32_t)F.lower_bits << 18);
            c.spans = dspan | (fspan << 16);
            dir.cmax.push_back((uint32_t)last_doc);
            dir.chunks.push_back(c);
            pos += cnt;
        }
    }
};

} // namespace ds2i_host
