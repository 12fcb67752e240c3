// Host-side (CPU, build-time) index construction and image parsing.
//   block_posting_list::write      reference block_posting_list.hpp:13-53
//   compact_elias_fano::write      reference compact_elias_fano.hpp:69-136 (offsets 14-61)
//   block_freq_index builder/map   reference block_freq_index.hpp:18-70,124-134
//   wand_data                      reference wand_data.hpp:17-83
//   global_parameters              reference global_parameters.hpp:5-31
// The frozen image layout follows succinct::mapper::freeze as restated in SURVEY.md
// Appendix B (succinct is an empty submodule in /root/reference -> "parity unpinned").
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_bits.hpp"
#include "host_encode.hpp"

namespace ds2i_host {

struct global_parameters {
    uint8_t ef_log_sampling0 = 9, ef_log_sampling1 = 8, rb_log_rank1_sampling = 9, rb_log_sampling1 = 8,
            log_partition_size = 7;
};

// ------------------------------------------------------------ posting list (A2)
// vbyte(n) | u32 block_max[nb] | u32 block_endpoint[nb-1] | block0 docs | block0 freqs | ...
inline void write_posting_list(int codec, bytes_t& out, uint32_t n, const uint32_t* docs, const uint32_t* freqs) {
    vbyte_encode(n, out);
    const uint64_t blocks = ceil_div((uint64_t)n, (uint64_t)BLOCK);
    const size_t begin_maxs = out.size();
    const size_t begin_endpoints = begin_maxs + 4 * blocks;
    const size_t begin_blocks = begin_endpoints + 4 * (blocks - 1);
    out.resize(begin_blocks);
    uint32_t dbuf[BLOCK], fbuf[BLOCK];
    uint32_t last_doc = uint32_t(-1), block_base = 0;
    size_t k = 0;
    for (uint64_t b = 0; b < blocks; ++b) {
        uint32_t cur = ((b + 1) * BLOCK <= n) ? BLOCK : (n % BLOCK);
        for (uint32_t i = 0; i < cur; ++i, ++k) {
            dbuf[i] = docs[k] - last_doc - 1;
            last_doc = docs[k];
            fbuf[i] = freqs[k] - 1;
        }
        std::memcpy(&out[begin_maxs + 4 * b], &last_doc, 4);
        block_encode(codec, dbuf, last_doc - block_base - (cur - 1), cur, out, b);
        block_encode(codec, fbuf, uint32_t(-1), cur, out, b);
        if (b != blocks - 1) {
            uint32_t ep = (uint32_t)(out.size() - begin_blocks);
            std::memcpy(&out[begin_endpoints + 4 * b], &ep, 4);
        }
        block_base = last_doc + 1;
    }
}

// ------------------------------------------------------------ Elias-Fano (a13)
struct ef_offsets {
    uint64_t universe, n, log_sampling0, log_sampling1, lower_bits, mask, higher_bits_length, pointer_size,
        pointers0, pointers1, pointers0_offset, pointers1_offset, higher_bits_offset, lower_bits_offset, end;
    ef_offsets(uint64_t base, uint64_t u, uint64_t n_, global_parameters const& p)
        : universe(u), n(n_), log_sampling0(p.ef_log_sampling0), log_sampling1(p.ef_log_sampling1) {
        lower_bits = u > n ? msb64(u / n) : 0;
        mask = (uint64_t(1) << lower_bits) - 1;
        higher_bits_length = n + (u >> lower_bits) + 2;
        pointer_size = ceil_log2(higher_bits_length);
        pointers0 = (higher_bits_length - n) >> log_sampling0;
        pointers1 = n >> log_sampling1;
        pointers0_offset = base;
        pointers1_offset = pointers0_offset + pointers0 * pointer_size;
        higher_bits_offset = pointers1_offset + pointers1 * pointer_size;
        lower_bits_offset = higher_bits_offset + higher_bits_length;
        end = lower_bits_offset + n * lower_bits;
    }
};

// Sampled positions of a finished 0/1 array, straight from their definition (SURVEY.md Appendix A6, and what
// test_compact_elias_fano.cpp:45-80 / test_compact_ranked_bitvector.cpp:36-68 check by a direct scan): the `len` bits at
// `first` are walked 64 at a time; on_chunk(pos, bits, nbits, ones_before) gets the chunk that starts at relative
// position pos (bit i of `bits` = bit pos + i; nbits valid, the rest zero) and the number of 1s before it. Callers pick
// the set bits with ctz and ranks with popcount: the cost is per word and per sampled bit, not per bit of the universe.
template <class OnChunk>
inline void scan_bits(bitvec_builder const& bv, uint64_t first, uint64_t len, OnChunk on_chunk) {
    auto const& w = bv.words();
    uint64_t ones = 0;
    for (uint64_t pos = 0; pos < len; pos += 64) {
        const uint64_t at = first + pos, wi = at >> 6;
        const unsigned sh = (unsigned)(at & 63);
        uint64_t bits = w[wi] >> sh;
        if (sh && wi + 1 < w.size()) bits |= w[wi + 1] << (64 - sh);
        const uint64_t nbits = len - pos < 64 ? len - pos : 64;
        if (nbits < 64) bits &= (uint64_t(1) << nbits) - 1;
        on_chunk(pos, bits, nbits, ones);
        ones += (uint64_t)__builtin_popcountll(bits);
    }
}
// the positions (relative to the chunk's pos) of the chunk's set bits, in order: fn(pos_of_bit, index among the chunk's set bits)
template <class Fn>
inline void for_each_set_bit(uint64_t bits, Fn fn) {
    for (uint64_t k = 0; bits; bits &= bits - 1, ++k) fn((uint64_t)__builtin_ctzll(bits), k);
}

// compact_elias_fano image of a sorted sequence (layout: `ef_offsets`): pointers0 | pointers1 | high bits | low bits.
// Value i sets high bit (v_i >> l) + i + 1 and stores its l low bits at slot i. The two pointer arrays are filled in a
// second pass over the finished high bits from what they MEAN: pointers1[k-1] = position of the 1 with k * 2^s1 ones
// before it, pointers0[k-1] = position of the 0 with k * 2^s0 zeros before it (k >= 1; slots without such a bit stay 0).
template <class It>
inline void ef_write(bitvec_builder& bvb, It begin, uint64_t universe, uint64_t n, global_parameters const& params) {
    const ef_offsets of(bvb.size(), universe, n, params);
    bvb.zero_extend(of.end - of.pointers0_offset);
    uint64_t prev = 0;
    It it = begin;
    for (uint64_t i = 0; i < n; ++i, ++it) {
        const uint64_t v = *it;
        if (v < prev) throw std::runtime_error("Sequence is not sorted");
        prev = v;
        bvb.set(of.higher_bits_offset + (v >> of.lower_bits) + i + 1, 1);
        bvb.set_bits(of.lower_bits_offset + i * of.lower_bits, v & of.mask, (unsigned)of.lower_bits);
    }
    if (!of.pointers0 && !of.pointers1) return;
    const uint64_t every1 = (uint64_t(1) << of.log_sampling1) - 1;
    const uint64_t every0 = of.log_sampling0 < 64 ? (uint64_t(1) << of.log_sampling0) - 1 : ~uint64_t(0);
    scan_bits(bvb, of.higher_bits_offset, of.higher_bits_length, [&](uint64_t pos, uint64_t bits, uint64_t nbits, uint64_t ones_before) {
        const uint64_t c1 = (uint64_t)__builtin_popcountll(bits), zeros_before = pos - ones_before, c0 = nbits - c1;
        if (of.pointers1 && c1)
            for_each_set_bit(bits, [&](uint64_t b, uint64_t k) {
                const uint64_t ones = ones_before + k;
                if (ones && !(ones & every1))
                    bvb.set_bits(of.pointers1_offset + ((ones >> of.log_sampling1) - 1) * of.pointer_size, pos + b, (unsigned)of.pointer_size);
            });
        if (of.pointers0 && c0) {
            const uint64_t inv = ~bits & (nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0));
            for_each_set_bit(inv, [&](uint64_t b, uint64_t k) {
                const uint64_t zeros = zeros_before + k;
                if (zeros && !(zeros & every0))
                    bvb.set_bits(of.pointers0_offset + ((zeros >> of.log_sampling0) - 1) * of.pointer_size, pos + b, (unsigned)of.pointer_size);
            });
        }
    });
}

// sequential decode of all n values (upload-time flattening of m_endpoints, a12)
inline void ef_decode_all(bitview const& bv, uint64_t base, uint64_t universe, uint64_t n,
                          global_parameters const& params, std::vector<uint64_t>& out) {
    ef_offsets of(base, universe, n, params);
    if (of.end > bv.nbits) throw std::runtime_error("EF sequence exceeds bit vector");
    out.resize(n);
    uint64_t hp = of.higher_bits_offset;
    for (uint64_t i = 0; i < n; ++i) {
        while (!bv.get(hp)) {
            ++hp;
            if (hp >= of.lower_bits_offset) throw std::runtime_error("EF high bits truncated");
        }
        uint64_t high = hp - of.higher_bits_offset;
        ++hp;
        uint64_t low = bv.get_bits(of.lower_bits_offset + i * of.lower_bits, (unsigned)of.lower_bits);
        out[i] = ((high - i - 1) << of.lower_bits) | low;
    }
}

// ------------------------------------------------------------ freeze helpers
template <class T> inline void put_pod(bytes_t& out, T v) {
    const uint8_t* p = (const uint8_t*)&v;
    out.insert(out.end(), p, p + sizeof(T));
}
struct reader {
    const uint8_t* p;
    size_t n, pos = 0;
    reader(const void* d, size_t len) : p((const uint8_t*)d), n(len) {}
    template <class T> T pod() {
        if (pos + sizeof(T) > n) throw std::runtime_error("image truncated");
        T v;
        std::memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    const uint8_t* take(size_t len) {
        if (len > n - pos) throw std::runtime_error("image truncated");
        const uint8_t* r = p + pos;
        pos += len;
        return r;
    }
};

// ------------------------------------------------------------ block_freq_index
class block_index_builder {
public:
    block_index_builder(int codec, uint64_t num_docs, global_parameters const& params = global_parameters())
        : m_codec(codec), m_num_docs(num_docs), m_params(params) {
        m_endpoints.push_back(0);
    }
    void add_posting_list(uint64_t n, const uint32_t* docs, const uint32_t* freqs) {
        if (!n) throw std::invalid_argument("List must be nonempty");
        write_posting_list(m_codec, m_lists, (uint32_t)n, docs, freqs);
        m_endpoints.push_back(m_lists.size());
    }
    void add_encoded_list(const uint8_t* data, size_t len) {
        m_lists.insert(m_lists.end(), data, data + len);
        m_endpoints.push_back(m_lists.size());
    }
    // all lists at once, already encoded (the GPU encoder): `lists` = the concatenated list bytes, `ends` = the end
    // offset of every list
    void set_encoded_lists(bytes_t&& lists, std::vector<uint64_t> const& ends) {
        m_lists = std::move(lists);
        m_endpoints.assign(1, 0);
        m_endpoints.insert(m_endpoints.end(), ends.begin(), ends.end());
    }
    uint64_t lists() const { return m_endpoints.size() - 1; }
    // image = 5 B params | u64 m_size | u64 m_num_docs | bit_vector{u64 bits; u64 nwords; words} | u64 nbytes; bytes
    void freeze(bytes_t& out) const {
        const uint64_t size = m_endpoints.size() - 1;
        bitvec_builder bvb;
        if (size) ef_write(bvb, m_endpoints.begin(), m_lists.size(), size, m_params);
        out.push_back(m_params.ef_log_sampling0);
        out.push_back(m_params.ef_log_sampling1);
        out.push_back(m_params.rb_log_rank1_sampling);
        out.push_back(m_params.rb_log_sampling1);
        out.push_back(m_params.log_partition_size);
        put_pod<uint64_t>(out, size);
        put_pod<uint64_t>(out, m_num_docs);
        put_pod<uint64_t>(out, bvb.size());
        put_pod<uint64_t>(out, bvb.words().size());
        const uint8_t* w = (const uint8_t*)bvb.words().data();
        out.insert(out.end(), w, w + 8 * bvb.words().size());
        put_pod<uint64_t>(out, m_lists.size());
        out.insert(out.end(), m_lists.begin(), m_lists.end());
    }

private:
    int m_codec;
    uint64_t m_num_docs;
    global_parameters m_params;
    std::vector<uint64_t> m_endpoints;
    bytes_t m_lists;
};

// Parsed (zero-copy) view of a frozen block_freq_index image.
struct block_index_view {
    global_parameters params;
    uint64_t size = 0, num_docs = 0;
    bitview endpoints_bv;
    const uint8_t* lists = nullptr;
    uint64_t lists_bytes = 0;
    std::vector<uint64_t> list_offsets; // size+1 entries (flattened m_endpoints)

    void parse(const void* image, size_t bytes) {
        reader r(image, bytes);
        params.ef_log_sampling0 = r.pod<uint8_t>();
        params.ef_log_sampling1 = r.pod<uint8_t>();
        params.rb_log_rank1_sampling = r.pod<uint8_t>();
        params.rb_log_sampling1 = r.pod<uint8_t>();
        params.log_partition_size = r.pod<uint8_t>();
        size = r.pod<uint64_t>();
        num_docs = r.pod<uint64_t>();
        endpoints_bv.nbits = r.pod<uint64_t>();
        uint64_t nwords = r.pod<uint64_t>();
        if (nwords != ceil_div(endpoints_bv.nbits, (uint64_t)64)) throw std::runtime_error("bad endpoints bit vector");
        endpoints_bv.nbytes = 8 * nwords;
        endpoints_bv.bytes = r.take(8 * nwords);
        lists_bytes = r.pod<uint64_t>();
        lists = r.take(lists_bytes);
        if (num_docs > 0xFFFFFFFFull) throw std::runtime_error("num_docs exceeds 32 bits");
        if (size) ef_decode_all(endpoints_bv, 0, lists_bytes, size, params, list_offsets);
        list_offsets.push_back(lists_bytes);
        for (uint64_t i = 0; i + 1 < list_offsets.size(); ++i)
            if (list_offsets[i] >= list_offsets[i + 1]) throw std::runtime_error("list endpoints not increasing");
    }
};

// ------------------------------------------------------------ BM25 + wand_data
// reference bm25.hpp:7-25 -- float32 throughout, no contraction.
struct bm25 {
    static float doc_term_weight(uint64_t freq, float norm_len) {
        const float b = 0.5f, k1 = 1.2f;
        float f = (float)freq;
        return f / (f + k1 * (1.0f - b + b * norm_len));
    }
    static float query_term_weight(uint64_t freq, uint64_t df, uint64_t num_docs) {
        const float k1 = 1.2f;
        float f = (float)freq;
        float fdf = (float)df;
        float idf = std::log((float(num_docs) - fdf + 0.5f) / (fdf + 0.5f));
        static const float epsilon_score = 1.0E-6f;
        return f * std::max(epsilon_score, idf) * (1.0f + k1);
    }
};

// wand image = u64 N | float norm_lens[N] | u64 V | float max_term_weight[V]
inline void compute_norm_lens(const uint32_t* sizes, uint64_t num_docs, std::vector<float>& norm_lens) {
    norm_lens.resize(num_docs);
    double sum = 0;
    for (uint64_t i = 0; i < num_docs; ++i) {
        float len = (float)sizes[i];
        norm_lens[i] = len;
        sum += len;
    }
    float avg = float(sum / double(num_docs));
    for (uint64_t i = 0; i < num_docs; ++i) norm_lens[i] /= avg;
}
inline float list_max_weight(const float* norm_lens, uint64_t n, const uint32_t* docs, const uint32_t* freqs) {
    float mx = 0;
    for (uint64_t i = 0; i < n; ++i) mx = std::max(mx, bm25::doc_term_weight(freqs[i], norm_lens[docs[i]]));
    return mx;
}
inline void wand_freeze(std::vector<float> const& norm_lens, std::vector<float> const& max_w, bytes_t& out) {
    put_pod<uint64_t>(out, norm_lens.size());
    const uint8_t* p = (const uint8_t*)norm_lens.data();
    out.insert(out.end(), p, p + 4 * norm_lens.size());
    put_pod<uint64_t>(out, max_w.size());
    p = (const uint8_t*)max_w.data();
    out.insert(out.end(), p, p + 4 * max_w.size());
}
struct wand_view {
    uint64_t num_docs = 0, num_terms = 0;
    const uint8_t* norm_lens = nullptr;       // float[num_docs], unaligned
    const uint8_t* max_term_weight = nullptr; // float[num_terms], unaligned
    void parse(const void* image, size_t bytes) {
        reader r(image, bytes);
        num_docs = r.pod<uint64_t>();
        norm_lens = r.take(4 * num_docs);
        num_terms = r.pod<uint64_t>();
        max_term_weight = r.take(4 * num_terms);
    }
};

} // namespace ds2i_host
