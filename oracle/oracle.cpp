// ORACLE -- TEST INFRASTRUCTURE ONLY. CPU restatement of ds2i's query-processing read path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (ds2i_amd/) never links, imports or calls it.
//
// What it restates (reference file:line):
//   TightVariableByte::decode           block_codecs.hpp:84-98
//   bit_reader / read_interpolative     interpolative_coding.hpp:79-153
//   interpolative_block::decode         block_codecs.hpp:127-147
//   optpfor_block::decode               block_codecs.hpp:210-226  (+ FastPFor NewPFor::decodeBlock,
//                                       Simple16 unpack, fastunpack -- see "parity" below)
//   varint_G8IU_block::decode           block_codecs.hpp:239-258, 287-314
//   qmx_block::decode                   block_codecs.hpp:336-349, qmx_codec.hpp:636-6115
//   mixed_block::decode                 mixed_block.hpp:198-217
//   block_posting_list::document_enumerator   block_posting_list.hpp:84-354 (with the Profile=true
//                                       counters of block_profiler.hpp:9-62)
//   compact_elias_fano::enumerator::move      compact_elias_fano.hpp:155-182, 263-289
//   block_freq_index::operator[] / map  block_freq_index.hpp:85-94, 124-134
//   wand_data                           wand_data.hpp:55-63, 71-78
//   bm25                                bm25.hpp:7-25
//   and/or/ranked_and/wand/maxscore/ranked_or + topk_queue   queries.hpp:29-591
//   op_perftest timing                  queries.cpp:13-62
//
// PARITY STATUS
//   * vbyte, interpolative, QMX, posting-list header layout: pinned against the reference's own
//     outputs -- SURVEY.md Appendix C known-answer vectors (tests/golden/appendix_c.json) and, for
//     QMX, oracle/_ref/libqmx_ref.so compiled from /root/reference/qmx_codec.hpp (oracle/Makefile).
//   * OptPFor / Simple16 / bit-packing / VarInt-G8IU byte streams and the succinct::mapper::freeze
//     container layout: the code lives in ot/FastPFor and ot/succinct, which are EMPTY submodules
//     in /root/reference (.gitmodules:1-9, version unknowable). Restated from the published
//     FastPFor algorithm and SURVEY.md Appendix B: "parity unpinned" at byte level; pinned only
//     functionally (round trip + consumed-byte count, the same contract test_block_codecs.cpp:9-46
//     checks) and through query results, which are codec independent.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <numeric>
#include <stdexcept>
#include <sys/time.h>
#include <utility>
#include <vector>
#include <thread>
#include <atomic>

#include "oracle_pef.hpp"

namespace oracle {

static inline uint32_t msb32(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }
static inline uint32_t msb64(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }
static inline uint64_t ceil_log2(uint64_t x) { return x > 1 ? msb64(x - 1) + 1 : 0; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

static const uint32_t BLOCK = 128;
enum { CODEC_OPTPFOR = 0, CODEC_VARINT = 1, CODEC_INTERPOLATIVE = 2, CODEC_QMX = 3, CODEC_MIXED = 4 };

// ---------------------------------------------------------------- vbyte
static const uint8_t* vbyte_decode(const uint8_t* in, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t v = 0;
        for (unsigned shift = 0;; shift += 7) {
            uint8_t c = *in++;
            v += uint32_t(c & 127) << shift;
            if (c & 128) { *out++ = v; break; }
        }
    }
    return in;
}

// ---------------------------------------------------------------- interpolative
struct bit_reader {
    const uint8_t* in;
    uint32_t avail = 0;
    uint64_t buf = 0;
    size_t pos = 0;
    explicit bit_reader(const uint8_t* p) : in(p) {}
    uint32_t read(uint32_t len) {
        if (!len) return 0;
        if (avail < len) {
            buf |= uint64_t(ld32(in)) << avail;
            in += 4;
            avail += 32;
        }
        uint32_t val = (uint32_t)(buf & ((uint64_t(1) << len) - 1));
        buf >>= len;
        avail -= len;
        pos += len;
        return val;
    }
    uint32_t read_int(uint32_t u) {
        uint32_t b = msb32(u);
        uint64_t m = (uint64_t(1) << (b + 1)) - u;
        uint32_t val = read(b);
        if (val >= m) val = (val << 1) + read(1) - (uint32_t)m;
        return val;
    }
    void read_interpolative(uint32_t* out, size_t n, uint32_t low, uint32_t high) {
        size_t h = n / 2;
        uint32_t val = low + read_int(high - low + 1);
        out[h] = val;
        if (n == 1) return;
        if (h) read_interpolative(out, h, low, val);
        if (n - h - 1) read_interpolative(out + h + 1, n - h - 1, val, high);
    }
};

static const uint8_t* interpolative_decode(const uint8_t* in, uint32_t* out, uint32_t sum, size_t n) {
    if (sum == uint32_t(-1)) in = vbyte_decode(in, &sum, 1);
    out[n - 1] = sum;
    size_t bytes = 0;
    if (n > 1) {
        bit_reader br(in);
        br.read_interpolative(out, n - 1, 0, sum);
        for (size_t i = n - 1; i > 0; --i) out[i] -= out[i - 1];
        bytes = (br.pos + 7) / 8;
    }
    return in + bytes;
}

// ---------------------------------------------------------------- OptPFor (FastPFor NewPFor::decodeBlock)
template <int B> static void unpack32(const uint32_t* in, uint32_t* out) {
    if (B == 0) { for (int i = 0; i < 32; ++i) out[i] = 0; return; }
    if (B == 32) { for (int i = 0; i < 32; ++i) out[i] = in[i]; return; }
    const uint64_t mask = (uint64_t(1) << B) - 1;
    for (int i = 0; i < 32; ++i) {
        const int bit = i * B, w = bit >> 5, s = bit & 31;
        uint64_t x = in[w];
        if (s + B > 32) x |= uint64_t(in[w + 1]) << 32;
        out[i] = (uint32_t)((x >> s) & mask);
    }
}
typedef void (*unpack_fn)(const uint32_t*, uint32_t*);
template <int... Bs> struct unpack_table { static constexpr unpack_fn fns[sizeof...(Bs)] = {&unpack32<Bs>...}; };
template <int... Bs> constexpr unpack_fn unpack_table<Bs...>::fns[sizeof...(Bs)];
typedef unpack_table<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26,
                     27, 28, 29, 30, 31, 32> unpackers;

static const uint8_t S16_RUNS[16][6] = {
    {28, 1, 0, 0, 0, 0}, {7, 2, 14, 1, 0, 0}, {7, 1, 7, 2, 7, 1}, {14, 1, 7, 2, 0, 0},
    {14, 2, 0, 0, 0, 0}, {1, 4, 8, 3, 0, 0},  {1, 3, 4, 4, 3, 3}, {7, 4, 0, 0, 0, 0},
    {4, 5, 2, 4, 0, 0},  {2, 4, 4, 5, 0, 0},  {3, 6, 2, 5, 0, 0}, {2, 5, 3, 6, 0, 0},
    {4, 7, 0, 0, 0, 0},  {1, 10, 2, 9, 0, 0}, {2, 14, 0, 0, 0, 0}, {1, 28, 0, 0, 0, 0}};

// decodes whole words (like FastPFor: the last word may yield padding values)
static size_t simple16_decode(const uint32_t* in, size_t nwords, uint32_t* out) {
    uint32_t* o = out;
    for (size_t w = 0; w < nwords; ++w) {
        uint32_t word = ld32((const uint8_t*)(in + w));
        const uint8_t* r = S16_RUNS[word >> 28];
        uint32_t pos = 28;
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < r[2 * k]; ++c) {
                pos -= r[2 * k + 1];
                *o++ = (word >> pos) & ((1u << r[2 * k + 1]) - 1u);
            }
    }
    return o - out;
}

static const uint8_t* optpfor_decode_block(const uint8_t* in8, uint32_t* out) {
    // the block is a stream of little-endian u32 words at an arbitrary byte address
    uint32_t words[1 + 2 * BLOCK + 4 * 32 + 8];
    const uint32_t hdr = ld32(in8);
    const uint32_t b = hdr >> 26, nexc = (hdr >> 16) & 0x3FFu, ew = hdr & 0xFFFFu;
    const size_t total = 1 + (size_t)ew + (b <= 32 ? 4 * b : 128);
    if (b > 32 || ew > 2 * BLOCK) throw std::runtime_error("optpfor: corrupt block header");
    std::memcpy(words, in8, 4 * total);
    const uint32_t* in = words + 1;
    if (ew > 2 * nexc) throw std::runtime_error("optpfor: corrupt exception area");
    static thread_local std::vector<uint32_t> excbuf(2 * BLOCK * 28 + 64);
    uint32_t* exc = excbuf.data();
    if (ew) simple16_decode(in, ew, exc);
    in += ew;
    for (uint32_t g = 0; g < 4; ++g) unpackers::fns[b](in + g * b, out + 32 * g);
    for (uint32_t e = 0, lpos = uint32_t(-1); e < nexc; ++e) {
        lpos += exc[e] + 1;
        out[lpos & 127] |= (exc[e + nexc] + 1) << b;
    }
    return in8 + 4 * total;
}

// ---------------------------------------------------------------- VarInt-G8IU
static inline uint32_t g8iu_group(const uint8_t*& src, uint32_t* dst) {
    uint8_t desc = *src++;
    uint32_t n = 0, val = 0, k = 0;
    for (int j = 0; j < 8; ++j) {
        val |= uint32_t(src[j]) << (8 * k);
        ++k;
        if (!((desc >> j) & 1)) { dst[n++] = val; val = 0; k = 0; }
    }
    src += 8;
    return n;
}
static const uint8_t* varint_g8iu_decode_block(const uint8_t* in, uint32_t* out, size_t n) {
    size_t out_len = 0;
    const uint8_t* src = in;
    while (out_len <= n - 8) out_len += g8iu_group(src, out + out_len);
    while (out_len < n) {
        uint32_t buf[8];
        size_t read = g8iu_group(src, buf);
        size_t needed = std::min(read, n - out_len);
        std::memcpy(out + out_len, buf, 4 * needed);
        out_len += needed;
    }
    return src;
}

// ---------------------------------------------------------------- QMX
static const uint8_t QMX_BITS[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32};
static const uint16_t QMX_CAP[15] = {256, 128, 64, 40, 32, 24, 20, 36, 16, 28, 12, 20, 8, 12, 4};

// `to` must have room for 128 + 512 values (qmx_block::overflow, block_codecs.hpp:319)
static void qmx_decode_stream(uint32_t* to, const uint8_t* src, size_t len) {
    const uint8_t* in = src;
    const uint8_t* keys = src + len - 1;
    while (in <= keys) {
        uint8_t key = *keys--;
        uint32_t type = key >> 4, reps = 16u - (key & 15u);
        if (type == 15) { in += reps; continue; }
        const uint32_t bits = QMX_BITS[type], cap = QMX_CAP[type];
        for (uint32_t r = 0; r < reps; ++r) {
            if (type == 0) {
                for (uint32_t j = 0; j < 256; ++j) to[j] = 1;
            } else if (bits == 8) {
                for (uint32_t j = 0; j < 16; ++j) to[j] = in[j];
                in += 16;
            } else if (bits == 16) {
                for (uint32_t j = 0; j < 8; ++j) { uint16_t v; std::memcpy(&v, in + 2 * j, 2); to[j] = v; }
                in += 16;
            } else if (bits == 32) {
                for (uint32_t j = 0; j < 4; ++j) to[j] = ld32(in + 4 * j);
                in += 16;
            } else {
                uint32_t v1[4], v2[4] = {0, 0, 0, 0};
                std::memcpy(v1, in, 16);
                const bool two = (bits == 7 || bits == 9 || bits == 12 || bits == 21);
                if (two) std::memcpy(v2, in + 16, 16);
                const uint32_t mask = (1u << bits) - 1u;
                const uint32_t R1 = 32 / bits, lowpart = 32 - R1 * bits;
                const uint32_t off2 = bits == 12 ? 8 : bits == 21 ? 11 : bits - lowpart;
                for (uint32_t j = 0; j < cap; ++j) {
                    uint32_t l = j & 3, row = j >> 2, v;
                    if (!two || row < R1) v = (v1[l] >> (row * bits)) & mask;
                    else if (row == R1) v = ((v1[l] >> (row * bits)) | (v2[l] << lowpart)) & mask;
                    else v = (v2[l] >> ((row - R1 - 1) * bits + off2)) & mask;
                    to[j] = v;
                }
                in += two ? 32 : 16;
            }
            to += cap;
        }
    }
}

static const uint8_t* qmx_decode_block(const uint8_t* in, uint32_t* out) {
    uint32_t enc_len = 0;
    in = vbyte_decode(in, &enc_len, 1);
    qmx_decode_stream(out, in, enc_len);
    return in + enc_len;
}

// ---------------------------------------------------------------- codec dispatch
// out must hold block_size + overflow values (128 + 512)
static const uint8_t* block_decode(int codec, const uint8_t* in, uint32_t* out, uint32_t sum, size_t n) {
    if (codec == CODEC_MIXED) {
        int type = 2;
        if (n == BLOCK) type = *in++;
        if (type == 1) return n < BLOCK ? interpolative_decode(in, out, sum, n) : varint_g8iu_decode_block(in, out, n);
        if (type == 0) return n < BLOCK ? interpolative_decode(in, out, sum, n) : optpfor_decode_block(in, out);
        if (type == 2) return interpolative_decode(in, out, sum, n);
        throw std::runtime_error("mixed: unknown block type");
    }
    if (n < BLOCK || codec == CODEC_INTERPOLATIVE) return interpolative_decode(in, out, sum, n);
    switch (codec) {
    case CODEC_OPTPFOR: return optpfor_decode_block(in, out);
    case CODEC_VARINT: return varint_g8iu_decode_block(in, out, n);
    case CODEC_QMX: return qmx_decode_block(in, out);
    }
    throw std::runtime_error("unknown codec");
}

// ---------------------------------------------------------------- profile (block_profiler)
struct profile {
    uint64_t docs_blocks = 0, freqs_blocks = 0, block_max_examined = 0, algorithmic_bytes = 0, postings_scored = 0;
};

// ---------------------------------------------------------------- block posting list enumerator
class document_enumerator {
public:
    document_enumerator() {}
    document_enumerator(int codec, const uint8_t* data, uint64_t universe, profile* prof)
        : m_codec(codec), m_n(0), m_universe(universe), m_prof(prof) {
        m_base = vbyte_decode(data, &m_n, 1);
        m_blocks = (m_n + BLOCK - 1) / BLOCK;
        m_block_maxs = m_base;
        m_block_endpoints = m_block_maxs + 4 * m_blocks;
        m_blocks_data = m_block_endpoints + 4 * (m_blocks - 1);
        m_docs_buf.resize(BLOCK + 512);
        m_freqs_buf.resize(BLOCK + 512);
        if (m_prof) {
            m_prof->algorithmic_bytes += (m_base - data) + 8 + 4;
            m_prof->block_max_examined += 1;
        }
        reset();
    }
    void reset() { decode_docs_block(0); }
    void next() {
        ++m_pos_in_block;
        if (m_pos_in_block == m_cur_block_size) {
            if (m_cur_block + 1 == m_blocks) { m_cur_docid = (uint32_t)m_universe; return; }
            if (m_prof) { m_prof->block_max_examined += 1; m_prof->algorithmic_bytes += 4; }
            decode_docs_block(m_cur_block + 1);
        } else {
            m_cur_docid += m_docs_buf[m_pos_in_block] + 1;
        }
    }
    void next_geq(uint64_t lower_bound) {
        if (lower_bound > m_cur_block_max) {
            if (lower_bound > block_max(m_blocks - 1)) {
                if (m_prof) { m_prof->block_max_examined += 1; m_prof->algorithmic_bytes += 4; }
                m_cur_docid = (uint32_t)m_universe;
                return;
            }
            uint64_t block = m_cur_block + 1;
            while (block_max((uint32_t)block) < lower_bound) ++block;
            if (m_prof) {
                m_prof->block_max_examined += block - m_cur_block;
                m_prof->algorithmic_bytes += 4 * (block - m_cur_block);
            }
            decode_docs_block(block);
        }
        while (docid() < lower_bound) m_cur_docid += m_docs_buf[++m_pos_in_block] + 1;
    }
    void move(uint64_t pos) {
        uint64_t block = pos / BLOCK;
        if (block != m_cur_block) decode_docs_block(block);
        while (position() < pos) m_cur_docid += m_docs_buf[++m_pos_in_block] + 1;
    }
    uint64_t docid() const { return m_cur_docid; }
    uint64_t freq() {
        if (!m_freqs_decoded) decode_freqs_block();
        return m_freqs_buf[m_pos_in_block] + 1;
    }
    uint64_t position() const { return uint64_t(m_cur_block) * BLOCK + m_pos_in_block; }
    uint64_t size() const { return m_n; }
    uint64_t num_blocks() const { return m_blocks; }

private:
    uint32_t block_max(uint32_t b) const { return ld32(m_block_maxs + 4 * (size_t)b); }
    void decode_docs_block(uint64_t block) {
        uint32_t endpoint = block ? ld32(m_block_endpoints + 4 * (block - 1)) : 0;
        const uint8_t* block_data = m_blocks_data + endpoint;
        m_cur_block_size = ((block + 1) * BLOCK <= size()) ? BLOCK : (uint32_t)(size() % BLOCK);
        uint32_t cur_base = (block ? block_max((uint32_t)block - 1) : uint32_t(-1)) + 1;
        m_cur_block_max = block_max((uint32_t)block);
        m_freqs_block_data = block_decode(m_codec, block_data, m_docs_buf.data(),
                                          m_cur_block_max - cur_base - (m_cur_block_size - 1), m_cur_block_size);
        m_docs_buf[0] += cur_base;
        m_cur_block = (uint32_t)block;
        m_pos_in_block = 0;
        m_cur_docid = m_docs_buf[0];
        m_freqs_decoded = false;
        if (m_prof) {
            m_prof->docs_blocks += 1;
            m_prof->algorithmic_bytes += 4 + (m_freqs_block_data - block_data);
        }
    }
    void decode_freqs_block() {
        const uint8_t* next = block_decode(m_codec, m_freqs_block_data, m_freqs_buf.data(), uint32_t(-1), m_cur_block_size);
        m_freqs_decoded = true;
        if (m_prof) {
            m_prof->freqs_blocks += 1;
            m_prof->algorithmic_bytes += next - m_freqs_block_data;
        }
    }
    int m_codec = 0;
    uint32_t m_n = 0;
    const uint8_t* m_base = nullptr;
    uint32_t m_blocks = 0;
    const uint8_t *m_block_maxs = nullptr, *m_block_endpoints = nullptr, *m_blocks_data = nullptr;
    uint64_t m_universe = 0;
    uint32_t m_cur_block = 0, m_pos_in_block = 0, m_cur_block_max = 0, m_cur_block_size = 0, m_cur_docid = 0;
    const uint8_t* m_freqs_block_data = nullptr;
    bool m_freqs_decoded = false;
    std::vector<uint32_t> m_docs_buf, m_freqs_buf;
    profile* m_prof = nullptr;
};

// ---------------------------------------------------------------- bit vector + Elias-Fano move
struct bit_vector {
    const uint8_t* bytes = nullptr; // u64 words, possibly unaligned
    uint64_t nbits = 0, nbytes = 0;
    uint64_t word(uint64_t i) const { uint64_t w; std::memcpy(&w, bytes + 8 * i, 8); return w; }
    uint64_t get_word56(uint64_t pos) const {
        uint64_t byte = pos / 8, w = 0;
        std::memcpy(&w, bytes + byte, std::min<uint64_t>(8, nbytes - byte));
        return w >> (pos % 8);
    }
};
// succinct::bit_vector::unary_enumerator (restated)
struct unary_enumerator {
    const bit_vector* bv = nullptr;
    uint64_t pos = 0, buf = 0;
    unary_enumerator() {}
    unary_enumerator(const bit_vector& b, uint64_t p) : bv(&b), pos(p) {
        buf = bv->word(pos / 64) & (~uint64_t(0) << (pos % 64));
    }
    uint64_t next() {
        while (!buf) { pos += 64; buf = bv->word(pos / 64); }
        unsigned l = (unsigned)__builtin_ctzll(buf);
        buf &= buf - 1;
        pos = (pos & ~uint64_t(63)) + l;
        return pos;
    }
    void skip(uint64_t k) { // consume k ones
        uint64_t skipped = 0;
        unsigned w;
        while (skipped + (w = (unsigned)__builtin_popcountll(buf)) <= k) {
            skipped += w;
            pos += 64;
            buf = bv->word(pos / 64);
        }
        for (; skipped < k; ++skipped) buf &= buf - 1;
    }
};

struct ef_offsets {
    uint64_t universe, n, log_sampling0, log_sampling1, lower_bits, mask, higher_bits_length, pointer_size, pointers0,
        pointers1, pointers0_offset, pointers1_offset, higher_bits_offset, lower_bits_offset, end;
    ef_offsets() {}
    ef_offsets(uint64_t base, uint64_t u, uint64_t n_, uint8_t ls0, uint8_t ls1) : universe(u), n(n_), log_sampling0(ls0), log_sampling1(ls1) {
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

class ef_enumerator {
public:
    ef_enumerator(const bit_vector& bv, uint64_t offset, uint64_t universe, uint64_t n, uint8_t ls0, uint8_t ls1)
        : m_bv(&bv), m_of(offset, universe, n, ls0, ls1), m_position(n), m_value(universe) {}
    std::pair<uint64_t, uint64_t> move(uint64_t position) {
        if (position == m_position) return value();
        uint64_t skip = position - m_position;
        if (position > m_position && skip <= 8) {
            m_position = position;
            if (m_position == m_of.n) {
                m_value = m_of.universe;
            } else {
                unary_enumerator he = m_high;
                for (uint64_t i = 0; i < skip; ++i) he.next();
                m_value = ((he.pos - m_of.higher_bits_offset - m_position - 1) << m_of.lower_bits) | read_low();
                m_high = he;
            }
            return value();
        }
        return slow_move(position);
    }

private:
    std::pair<uint64_t, uint64_t> slow_move(uint64_t position) {
        if (position == m_of.n) { m_position = position; m_value = m_of.universe; return value(); }
        uint64_t skip = position - m_position, to_skip;
        if (position > m_position && (skip >> m_of.log_sampling1) == 0) {
            to_skip = skip - 1;
        } else {
            uint64_t ptr = position >> m_of.log_sampling1;
            uint64_t high_pos = ptr ? (m_bv->get_word56(m_of.pointers1_offset + (ptr - 1) * m_of.pointer_size) &
                                       ((uint64_t(1) << m_of.pointer_size) - 1)) : 0;
            uint64_t high_rank = ptr << m_of.log_sampling1;
            m_high = unary_enumerator(*m_bv, m_of.higher_bits_offset + high_pos);
            to_skip = position - high_rank;
        }
        m_high.skip(to_skip);
        m_position = position;
        uint64_t high = m_high.next() - m_of.higher_bits_offset;
        m_value = ((high - m_position - 1) << m_of.lower_bits) | read_low();
        return value();
    }
    uint64_t read_low() const { return m_bv->get_word56(m_of.lower_bits_offset + m_position * m_of.lower_bits) & m_of.mask; }
    std::pair<uint64_t, uint64_t> value() const { return {m_position, m_value}; }
    const bit_vector* m_bv;
    ef_offsets m_of;
    uint64_t m_position, m_value;
    unary_enumerator m_high;
};

// ---------------------------------------------------------------- index + wand views (mapper::map)
struct rd {
    const uint8_t* p; size_t n, pos = 0;
    template <class T> T pod() { if (pos + sizeof(T) > n) throw std::runtime_error("image truncated"); T v; std::memcpy(&v, p + pos, sizeof(T)); pos += sizeof(T); return v; }
    const uint8_t* take(size_t len) { if (len > n - pos) throw std::runtime_error("image truncated"); const uint8_t* r = p + pos; pos += len; return r; }
};

struct block_freq_index {
    typedef oracle::document_enumerator document_enumerator;
    int codec = 0;
    uint8_t params[5];
    uint64_t m_size = 0, m_num_docs = 0;
    bit_vector m_endpoints;
    const uint8_t* m_lists = nullptr;
    uint64_t m_lists_size = 0;
    mutable profile* prof = nullptr;
    void begin_profile(profile* p) const { prof = p; }
    void end_profile() const { prof = nullptr; }
    void map(int codec_, const void* image, size_t bytes) {
        codec = codec_;
        rd r{(const uint8_t*)image, bytes};
        for (auto& b : params) b = r.pod<uint8_t>();
        m_size = r.pod<uint64_t>();
        m_num_docs = r.pod<uint64_t>();
        m_endpoints.nbits = r.pod<uint64_t>();
        uint64_t nw = r.pod<uint64_t>();
        m_endpoints.nbytes = 8 * nw;
        m_endpoints.bytes = r.take(8 * nw);
        m_lists_size = r.pod<uint64_t>();
        m_lists = r.take(m_lists_size);
    }
    size_t size() const { return m_size; }
    uint64_t num_docs() const { return m_num_docs; }
    uint64_t endpoint(size_t i) const {
        ef_enumerator e(m_endpoints, 0, m_lists_size, m_size, params[0], params[1]);
        return e.move(i).second;
    }
    uint64_t list_offset(size_t i) const { return endpoint(i); }
    document_enumerator operator[](size_t i) const {
        if (i >= m_size) throw std::out_of_range("term id out of range");
        return document_enumerator(codec, m_lists + endpoint(i), m_num_docs, prof);
    }
};

// freq_index<DocsSequence, positive_sequence<FreqsBase>> (freq_index.hpp:12-249); the four instantiations of
// index_types.hpp:18-32 follow as typedefs
template <class DocsEnum, class FreqsBase>
struct freq_index_t {
    typedef freq_document_enumerator<DocsEnum, FreqsBase> document_enumerator;
    pef_params params;
    uint64_t m_num_docs = 0;
    bit_collection docs, freqs;
    mutable profile* prof = nullptr;
    mutable pef_profile pdocs, pfreqs;
    static void map_bv(rd& r, bitvec& bv) {
        bv.nbits = r.pod<uint64_t>();
        uint64_t nw = r.pod<uint64_t>();
        bv.nbytes = 8 * nw;
        bv.bytes = r.take(8 * nw);
    }
    void map(const void* image, size_t bytes) {
        rd r{(const uint8_t*)image, bytes};
        params.ef_log_sampling0 = r.pod<uint8_t>();
        params.ef_log_sampling1 = r.pod<uint8_t>();
        params.rb_log_rank1_sampling = r.pod<uint8_t>();
        params.rb_log_sampling1 = r.pod<uint8_t>();
        params.log_partition_size = r.pod<uint8_t>();
        m_num_docs = r.pod<uint64_t>();
        for (bit_collection* c : {&docs, &freqs}) {
            c->m_size = r.pod<uint64_t>();
            map_bv(r, c->endpoints);
            map_bv(r, c->bits);
        }
    }
    size_t size() const { return docs.m_size; }
    uint64_t num_docs() const { return m_num_docs; }
    uint64_t list_offset(size_t i) const { return docs.get(params, i); }
    document_enumerator operator[](size_t i) const { // freq_index.hpp:192-214
        if (i >= size()) throw std::out_of_range("term id out of range");
        bit_enumerator it(docs.bits, docs.get(params, i));
        const uint64_t start = it.position();
        uint64_t occurrences = read_gamma_nonzero(it);
        uint64_t n = 1;
        if (occurrences > 1) n = it.take(pef_ceil_log2(occurrences + 1));
        if (prof) prof->algorithmic_bytes += (it.position() - start + 7) / 8 + 16; // list header + two collection offsets
        DocsEnum de(docs.bits, it.position(), m_num_docs, n, params, prof ? &pdocs : nullptr);
        positive_enumerator<FreqsBase> fe(freqs.bits, freqs.get(params, i), occurrences + 1, n, params, prof ? &pfreqs : nullptr);
        return document_enumerator(de, fe);
    }
    void begin_profile(profile* p) const { prof = p; pdocs = pef_profile(); pfreqs = pef_profile(); }
    void end_profile() const {
        if (prof) {
            prof->docs_blocks += pdocs.partitions_entered;
            prof->freqs_blocks += pfreqs.partitions_entered;
            prof->algorithmic_bytes += pdocs.algorithmic_bytes + pfreqs.algorithmic_bytes;
        }
        prof = nullptr;
    }
};
typedef freq_index_t<partitioned_enumerator<false>, partitioned_enumerator<true>> opt_freq_index;  // index_types.hpp:29-32
typedef freq_index_t<plain_ef_enumerator, plain_sef_enumerator> ef_freq_index;                    // index_types.hpp:18-19
typedef freq_index_t<single_enumerator<false>, single_enumerator<true>> single_freq_index;        // index_types.hpp:21-22
typedef freq_index_t<uniform_enumerator<false>, uniform_enumerator<true>> uniform_freq_index;     // index_types.hpp:24-27

struct wand_data {
    uint64_t n_docs = 0, n_terms = 0;
    const uint8_t *nl = nullptr, *mw = nullptr;
    void map(const void* image, size_t bytes) {
        rd r{(const uint8_t*)image, bytes};
        n_docs = r.pod<uint64_t>();
        nl = r.take(4 * n_docs);
        n_terms = r.pod<uint64_t>();
        mw = r.take(4 * n_terms);
    }
    float norm_len(uint64_t d) const { float v; std::memcpy(&v, nl + 4 * d, 4); return v; }
    float max_term_weight(uint64_t t) const { float v; std::memcpy(&v, mw + 4 * t, 4); return v; }
};

struct bm25 {
    static float doc_term_weight(uint64_t freq, float norm_len) {
        const float b = 0.5f, k1 = 1.2f;
        float f = (float)freq;
        return f / (f + k1 * (1.0f - b + b * norm_len));
    }
    static float query_term_weight(uint64_t freq, uint64_t df, uint64_t num_docs) {
        const float k1 = 1.2f;
        float f = (float)freq, fdf = (float)df;
        float idf = std::log((float(num_docs) - fdf + 0.5f) / (fdf + 0.5f));
        static const float epsilon_score = 1.0E-6f;
        return f * std::max(epsilon_score, idf) * (1.0f + k1);
    }
};

// ---------------------------------------------------------------- queries.hpp
typedef std::vector<uint32_t> term_id_vec;

static void remove_duplicate_terms(term_id_vec& terms) {
    std::sort(terms.begin(), terms.end());
    terms.erase(std::unique(terms.begin(), terms.end()), terms.end());
}
static std::vector<std::pair<uint64_t, uint64_t>> query_freqs(term_id_vec terms) {
    std::vector<std::pair<uint64_t, uint64_t>> out;
    std::sort(terms.begin(), terms.end());
    for (size_t i = 0; i < terms.size(); ++i) {
        if (i == 0 || terms[i] != terms[i - 1]) out.emplace_back(terms[i], 1);
        else out.back().second += 1;
    }
    return out;
}

struct topk_queue {
    explicit topk_queue(uint64_t k) : m_k(k) {}
    bool insert(float score) {
        if (m_q.size() < m_k) {
            m_q.push_back(score);
            std::push_heap(m_q.begin(), m_q.end(), std::greater<float>());
            return true;
        } else if (score > m_q.front()) {
            std::pop_heap(m_q.begin(), m_q.end(), std::greater<float>());
            m_q.back() = score;
            std::push_heap(m_q.begin(), m_q.end(), std::greater<float>());
            return true;
        }
        return false;
    }
    bool would_enter(float score) const { return m_q.size() < m_k || score > m_q.front(); }
    void finalize() { std::sort_heap(m_q.begin(), m_q.end(), std::greater<float>()); }
    std::vector<float> const& topk() const { return m_q; }
    void clear() { m_q.clear(); }
    uint64_t m_k;
    std::vector<float> m_q;
};

template <class Index>
struct query_ctx {
    const Index* index;
    const wand_data* wdata;
    uint64_t k;
    std::vector<uint32_t>* matches; // optional doc-id list for and
    uint64_t freq_sum;               // and_freq / or_freq checksum
    topk_queue topk;
    query_ctx(const Index* i, const wand_data* w, uint64_t k_) : index(i), wdata(w), k(k_), matches(nullptr), freq_sum(0), topk(k_) {}
};

template <class Index>
static uint64_t and_query(query_ctx<Index>& c, term_id_vec terms, bool with_freqs) {
    typedef typename Index::document_enumerator document_enumerator;
    if (terms.empty()) return 0;
    remove_duplicate_terms(terms);
    std::vector<document_enumerator> enums;
    enums.reserve(terms.size());
    for (auto t : terms) enums.push_back((*c.index)[t]);
    std::sort(enums.begin(), enums.end(), [](document_enumerator const& l, document_enumerator const& r) { return l.size() < r.size(); });
    uint64_t results = 0, candidate = enums[0].docid();
    size_t i = 1;
    while (candidate < c.index->num_docs()) {
        for (; i < enums.size(); ++i) {
            enums[i].next_geq(candidate);
            if (enums[i].docid() != candidate) { candidate = enums[i].docid(); i = 0; break; }
        }
        if (i == enums.size()) {
            results += 1;
            if (c.matches) c.matches->push_back((uint32_t)candidate);
            if (with_freqs) for (i = 0; i < enums.size(); ++i) c.freq_sum += enums[i].freq();
            enums[0].next();
            candidate = enums[0].docid();
            i = 1;
        }
    }
    return results;
}

template <class Index>
static uint64_t or_query(query_ctx<Index>& c, term_id_vec terms, bool with_freqs) {
    typedef typename Index::document_enumerator document_enumerator;
    if (terms.empty()) return 0;
    remove_duplicate_terms(terms);
    std::vector<document_enumerator> enums;
    enums.reserve(terms.size());
    for (auto t : terms) enums.push_back((*c.index)[t]);
    uint64_t results = 0;
    uint64_t cur_doc = std::min_element(enums.begin(), enums.end(), [](document_enumerator const& l, document_enumerator const& r) { return l.docid() < r.docid(); })->docid();
    while (cur_doc < c.index->num_docs()) {
        results += 1;
        uint64_t next_doc = c.index->num_docs();
        for (size_t i = 0; i < enums.size(); ++i) {
            if (enums[i].docid() == cur_doc) {
                if (with_freqs) c.freq_sum += enums[i].freq();
                enums[i].next();
            }
            if (enums[i].docid() < next_doc) next_doc = enums[i].docid();
        }
        cur_doc = next_doc;
    }
    return results;
}

template <class Index>
struct scored_enum_t { typename Index::document_enumerator docs_enum; float q_weight; float max_weight; };

template <class Index>
static std::vector<scored_enum_t<Index>> make_scored(query_ctx<Index>& c, term_id_vec const& terms, bool need_max_weight) {
    typedef scored_enum_t<Index> scored_enum;
    auto qf = query_freqs(terms);
    std::vector<scored_enum> enums;
    enums.reserve(qf.size());
    uint64_t num_docs = c.index->num_docs();
    for (auto term : qf) {
        auto list = (*c.index)[term.first];
        float q_weight = bm25::query_term_weight(term.second, list.size(), num_docs);
        float max_weight = 0.f; // only wand / maxscore read m_max_term_weight (queries.hpp:232, 510)
        if (need_max_weight) {
            max_weight = q_weight * c.wdata->max_term_weight(term.first);
            if (c.index->prof) c.index->prof->algorithmic_bytes += 4;
        }
        enums.push_back(scored_enum{std::move(list), q_weight, max_weight});
    }
    return enums;
}
template <class Index>
static inline float norm_len(query_ctx<Index>& c, uint64_t d) {
    if (c.index->prof) { c.index->prof->algorithmic_bytes += 4; c.index->prof->postings_scored += 1; }
    return c.wdata->norm_len(d);
}

template <class Index>
static uint64_t ranked_and_query(query_ctx<Index>& c, term_id_vec terms) {
    typedef scored_enum_t<Index> scored_enum;
    c.topk.clear();
    if (terms.empty()) return 0;
    auto enums = make_scored(c, terms, false);
    std::sort(enums.begin(), enums.end(), [](scored_enum const& l, scored_enum const& r) { return l.docs_enum.size() < r.docs_enum.size(); });
    uint64_t candidate = enums[0].docs_enum.docid();
    size_t i = 1;
    while (candidate < c.index->num_docs()) {
        for (; i < enums.size(); ++i) {
            enums[i].docs_enum.next_geq(candidate);
            if (enums[i].docs_enum.docid() != candidate) { candidate = enums[i].docs_enum.docid(); i = 0; break; }
        }
        if (i == enums.size()) {
            float nl = norm_len(c, candidate), score = 0;
            for (i = 0; i < enums.size(); ++i) score += enums[i].q_weight * bm25::doc_term_weight(enums[i].docs_enum.freq(), nl);
            c.topk.insert(score);
            enums[0].docs_enum.next();
            candidate = enums[0].docs_enum.docid();
            i = 1;
        }
    }
    c.topk.finalize();
    return c.topk.topk().size();
}

template <class Index>
static uint64_t ranked_or_query(query_ctx<Index>& c, term_id_vec terms) {
    typedef scored_enum_t<Index> scored_enum;
    c.topk.clear();
    if (terms.empty()) return 0;
    auto enums = make_scored(c, terms, false);
    uint64_t cur_doc = std::min_element(enums.begin(), enums.end(), [](scored_enum const& l, scored_enum const& r) { return l.docs_enum.docid() < r.docs_enum.docid(); })->docs_enum.docid();
    while (cur_doc < c.index->num_docs()) {
        float score = 0, nl = norm_len(c, cur_doc);
        uint64_t next_doc = c.index->num_docs();
        for (size_t i = 0; i < enums.size(); ++i) {
            if (enums[i].docs_enum.docid() == cur_doc) {
                score += enums[i].q_weight * bm25::doc_term_weight(enums[i].docs_enum.freq(), nl);
                enums[i].docs_enum.next();
            }
            if (enums[i].docs_enum.docid() < next_doc) next_doc = enums[i].docs_enum.docid();
        }
        c.topk.insert(score);
        cur_doc = next_doc;
    }
    c.topk.finalize();
    return c.topk.topk().size();
}

template <class Index>
static uint64_t wand_query(query_ctx<Index>& c, term_id_vec const& terms) {
    typedef scored_enum_t<Index> scored_enum;
    c.topk.clear();
    if (terms.empty()) return 0;
    uint64_t num_docs = c.index->num_docs();
    auto enums = make_scored(c, terms, true);
    std::vector<scored_enum*> ordered;
    ordered.reserve(enums.size());
    for (auto& en : enums) ordered.push_back(&en);
    auto sort_enums = [&]() {
        std::sort(ordered.begin(), ordered.end(), [](scored_enum* l, scored_enum* r) { return l->docs_enum.docid() < r->docs_enum.docid(); });
    };
    sort_enums();
    while (true) {
        float upper_bound = 0;
        size_t pivot;
        bool found_pivot = false;
        for (pivot = 0; pivot < ordered.size(); ++pivot) {
            if (ordered[pivot]->docs_enum.docid() == num_docs) break;
            upper_bound += ordered[pivot]->max_weight;
            if (c.topk.would_enter(upper_bound)) { found_pivot = true; break; }
        }
        if (!found_pivot) break;
        uint64_t pivot_id = ordered[pivot]->docs_enum.docid();
        if (pivot_id == ordered[0]->docs_enum.docid()) {
            float score = 0, nl = norm_len(c, pivot_id);
            for (scored_enum* en : ordered) {
                if (en->docs_enum.docid() != pivot_id) break;
                score += en->q_weight * bm25::doc_term_weight(en->docs_enum.freq(), nl);
                en->docs_enum.next();
            }
            c.topk.insert(score);
            sort_enums();
        } else {
            uint64_t next_list = pivot;
            for (; ordered[next_list]->docs_enum.docid() == pivot_id; --next_list);
            ordered[next_list]->docs_enum.next_geq(pivot_id);
            for (size_t i = next_list + 1; i < ordered.size(); ++i) {
                if (ordered[i]->docs_enum.docid() < ordered[i - 1]->docs_enum.docid()) std::swap(ordered[i], ordered[i - 1]);
                else break;
            }
        }
    }
    c.topk.finalize();
    return c.topk.topk().size();
}

template <class Index>
static uint64_t maxscore_query(query_ctx<Index>& c, term_id_vec const& terms) {
    typedef scored_enum_t<Index> scored_enum;
    c.topk.clear();
    if (terms.empty()) return 0;
    auto enums = make_scored(c, terms, true);
    std::vector<scored_enum*> ordered;
    ordered.reserve(enums.size());
    for (auto& en : enums) ordered.push_back(&en);
    std::sort(ordered.begin(), ordered.end(), [](scored_enum* l, scored_enum* r) { return l->max_weight < r->max_weight; });
    std::vector<float> upper_bounds(ordered.size());
    upper_bounds[0] = ordered[0]->max_weight;
    for (size_t i = 1; i < ordered.size(); ++i) upper_bounds[i] = upper_bounds[i - 1] + ordered[i]->max_weight;
    uint64_t non_essential_lists = 0;
    uint64_t cur_doc = std::min_element(enums.begin(), enums.end(), [](scored_enum const& l, scored_enum const& r) { return l.docs_enum.docid() < r.docs_enum.docid(); })->docs_enum.docid();
    while (non_essential_lists < ordered.size() && cur_doc < c.index->num_docs()) {
        float score = 0, nl = norm_len(c, cur_doc);
        uint64_t next_doc = c.index->num_docs();
        for (size_t i = non_essential_lists; i < ordered.size(); ++i) {
            if (ordered[i]->docs_enum.docid() == cur_doc) {
                score += ordered[i]->q_weight * bm25::doc_term_weight(ordered[i]->docs_enum.freq(), nl);
                ordered[i]->docs_enum.next();
            }
            if (ordered[i]->docs_enum.docid() < next_doc) next_doc = ordered[i]->docs_enum.docid();
        }
        for (size_t i = non_essential_lists - 1; i + 1 > 0; --i) {
            if (!c.topk.would_enter(score + upper_bounds[i])) break;
            ordered[i]->docs_enum.next_geq(cur_doc);
            if (ordered[i]->docs_enum.docid() == cur_doc)
                score += ordered[i]->q_weight * bm25::doc_term_weight(ordered[i]->docs_enum.freq(), nl);
        }
        if (c.topk.insert(score)) {
            while (non_essential_lists < ordered.size() && !c.topk.would_enter(upper_bounds[non_essential_lists]))
                non_essential_lists += 1;
        }
        cur_doc = next_doc;
    }
    c.topk.finalize();
    return c.topk.topk().size();
}

enum { OP_AND = 0, OP_AND_FREQ, OP_OR, OP_OR_FREQ, OP_RANKED_AND, OP_WAND, OP_MAXSCORE, OP_RANKED_OR };

template <class Index>
static uint64_t run_op(query_ctx<Index>& c, int op, term_id_vec const& terms) {
    switch (op) {
    case OP_AND: return and_query(c, terms, false);
    case OP_AND_FREQ: return and_query(c, terms, true);
    case OP_OR: return or_query(c, terms, false);
    case OP_OR_FREQ: return or_query(c, terms, true);
    case OP_RANKED_AND: return ranked_and_query(c, terms);
    case OP_WAND: return wand_query(c, terms);
    case OP_MAXSCORE: return maxscore_query(c, terms);
    case OP_RANKED_OR: return ranked_or_query(c, terms);
    }
    throw std::invalid_argument("unknown op");
}

static double get_time_usecs() {
    timeval tv;
    gettimeofday(&tv, nullptr);
    return double(tv.tv_sec) * 1000000 + double(tv.tv_usec);
}

struct handle {
    int kind = 0; // 0..4 block codecs, 5 = opt, 6 = ef, 7 = single, 8 = uniform (freq_index layouts)
    block_freq_index index;
    opt_freq_index opt;
    ef_freq_index ef;
    single_freq_index single;
    uniform_freq_index uniform;
    wand_data wdata;
    bool has_wand = false;
};
// run f on whichever index the handle holds
template <class F> auto visit(handle* h, F&& f) -> decltype(f(h->index)) {
    switch (h->kind) {
    case 5: return f(h->opt);
    case 6: return f(h->ef);
    case 7: return f(h->single);
    case 8: return f(h->uniform);
    default: return f(h->index);
    }
}

} // namespace oracle

using namespace oracle;

extern "C" {

struct oracle_profile { uint64_t docs_blocks, freqs_blocks, block_max_examined, algorithmic_bytes, postings_scored; };

// -1 on error. consumed receives the number of bytes the decoder advanced over.
int oracle_decode_block(int codec, const uint8_t* in, uint32_t sum_of_values, uint32_t n, uint32_t* out, uint64_t* consumed) {
    try {
        std::vector<uint32_t> buf(BLOCK + 512);
        const uint8_t* end = block_decode(codec, in, buf.data(), sum_of_values, n);
        std::memcpy(out, buf.data(), 4 * n);
        if (consumed) *consumed = end - in;
        return 0;
    } catch (...) { return -1; }
}
int oracle_decode_vbyte(const uint8_t* in, uint32_t* value) {
    const uint8_t* e = vbyte_decode(in, value, 1);
    return (int)(e - in);
}
// raw QMX stream decode (for comparison with oracle/_ref): out must hold 128+512 values
void oracle_qmx_decode_stream(uint32_t* out, const uint8_t* src, uint64_t len) { qmx_decode_stream(out, src, len); }
// the restated scorer (bm25.hpp:7-25), element-wise: pinned against the reference's own bm25.hpp by tests/golden/bm25_reference.json
void oracle_bm25_doc_term_weight(const uint64_t* freq, const float* norm_len, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = bm25::doc_term_weight(freq[i], norm_len[i]);
}
void oracle_bm25_query_term_weight(const uint64_t* qtf, const uint64_t* df, uint64_t num_docs, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = bm25::query_term_weight(qtf[i], df[i], num_docs);
}

void* oracle_index_open(int kind, const void* image, uint64_t bytes, const void* wand, uint64_t wand_bytes) {
    try {
        handle* h = new handle;
        h->kind = kind;
        if (kind < 0 || kind > 8) throw std::invalid_argument("unknown index kind");
        if (kind <= 4) h->index.map(kind, image, bytes);
        else if (kind == 5) h->opt.map(image, bytes);
        else if (kind == 6) h->ef.map(image, bytes);
        else if (kind == 7) h->single.map(image, bytes);
        else h->uniform.map(image, bytes);
        if (wand) { h->wdata.map(wand, wand_bytes); h->has_wand = true; }
        return h;
    } catch (...) { return nullptr; }
}
void oracle_index_close(void* h) { delete (handle*)h; }
uint64_t oracle_index_size(void* hv) { return visit((handle*)hv, [](auto& idx) { return (uint64_t)idx.size(); }); }
uint64_t oracle_index_num_docs(void* hv) { return visit((handle*)hv, [](auto& idx) { return (uint64_t)idx.num_docs(); }); }
uint64_t oracle_list_offset(void* hv, uint64_t term) { return visit((handle*)hv, [&](auto& idx) { return (uint64_t)idx.list_offset(term); }); }

extern "C++" {
namespace {
template <class Index> int64_t list_size_t(const Index& idx, uint64_t term) { return (int64_t)idx[term].size(); }
template <class Index> int64_t list_enumerate_t(const Index& idx, uint64_t term, uint32_t* docs, uint32_t* freqs, uint64_t cap) {
    auto e = idx[term];
    uint64_t n = e.size();
    if (n > cap) return -1;
    for (uint64_t i = 0; i < n; ++i, e.next()) {
        if (e.position() != i) return -3;
        docs[i] = (uint32_t)e.docid();
        freqs[i] = (uint32_t)e.freq();
    }
    return (e.docid() == idx.num_docs()) ? (int64_t)n : -2;
}
template <class Index> void list_next_geq_t(const Index& idx, uint64_t term, const uint32_t* probes, uint64_t np, uint32_t* out_docid, uint32_t* out_freq) {
    auto e = idx[term];
    uint64_t N = idx.num_docs();
    for (uint64_t i = 0; i < np; ++i) {
        e.next_geq(probes[i]);
        out_docid[i] = (uint32_t)e.docid();
        out_freq[i] = e.docid() < N ? (uint32_t)e.freq() : 0;
    }
}
// random access: move(position) for each position, docid + freq
template <class Index> void list_move_t(const Index& idx, uint64_t term, const uint32_t* positions, uint64_t np, uint32_t* out_docid, uint32_t* out_freq) {
    auto e = idx[term];
    for (uint64_t i = 0; i < np; ++i) {
        e.move(positions[i]);
        out_docid[i] = (uint32_t)e.docid();
        out_freq[i] = (uint32_t)e.freq();
    }
}
void fill_profile(oracle_profile* dst, profile const& p) {
    dst->docs_blocks = p.docs_blocks; dst->freqs_blocks = p.freqs_blocks; dst->block_max_examined = p.block_max_examined;
    dst->algorithmic_bytes = p.algorithmic_bytes; dst->postings_scored = p.postings_scored;
}
template <class Index>
int64_t query_t(const Index& idx, handle* h, int op, uint32_t k, const uint32_t* terms, uint32_t nterms, float* topk, uint32_t* topk_len,
                uint32_t* matches, uint64_t match_cap, uint64_t* freq_sum, oracle_profile* prof) {
    profile p;
    idx.begin_profile(prof ? &p : nullptr);
    query_ctx<Index> c(&idx, &h->wdata, k ? k : 1);
    std::vector<uint32_t> m;
    if (matches) c.matches = &m;
    term_id_vec t(terms, terms + nterms);
    uint64_t r = run_op(c, op, t);
    idx.end_profile();
    if (topk_len) *topk_len = (uint32_t)c.topk.topk().size();
    if (topk) for (size_t i = 0; i < c.topk.topk().size() && i < k; ++i) topk[i] = c.topk.topk()[i];
    if (matches) std::memcpy(matches, m.data(), 4 * std::min<uint64_t>(m.size(), match_cap));
    if (freq_sum) *freq_sum = c.freq_sum;
    if (prof) fill_profile(prof, p);
    return (int64_t)r;
}
template <class Index>
int query_batch_t(const Index& idx, handle* h, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq,
                  uint64_t* out_count, float* out_topk, uint32_t* out_topk_len, uint64_t* out_freq_sum, oracle_profile* prof) {
    profile p;
    idx.begin_profile(prof ? &p : nullptr);
    query_ctx<Index> c(&idx, &h->wdata, k ? k : 1);
    for (uint32_t q = 0; q < nq; ++q) {
        term_id_vec t(terms + offs[q], terms + offs[q + 1]);
        c.freq_sum = 0;
        c.topk.clear();
        uint64_t r = run_op(c, op, t);
        if (out_count) out_count[q] = r;
        if (out_topk)
            for (uint32_t i = 0; i < k; ++i) out_topk[(size_t)q * k + i] = i < c.topk.topk().size() ? c.topk.topk()[i] : -INFINITY;
        if (out_topk_len) out_topk_len[q] = (uint32_t)c.topk.topk().size();
        if (out_freq_sum) out_freq_sum[q] = c.freq_sum;
    }
    idx.end_profile();
    if (prof) fill_profile(prof, p);
    return 0;
}
// The batch answered by `nthreads` host threads (the analogue of profile_queries.cpp:21-39: the index is immutable, every
// thread owns its query_ctx and takes queries off a shared counter). Profiling is off (the profile hook is per index).
// match_hash (optional, `and` / `and_freq`): per query, the order-sensitive checksum sum_i doc_i * (2 i + 1) mod 2^64 of
// its doc-id list -- the full lists of a 4096-query batch at 25 M docs are hundreds of MB, the checksums are not.
template <class Index>
int query_batch_mt_t(const Index& idx, handle* h, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq, uint32_t nthreads,
                     uint64_t* out_count, float* out_topk, uint32_t* out_topk_len, uint64_t* out_freq_sum, uint64_t* match_hash) {
    idx.begin_profile(nullptr);
    std::atomic<uint32_t> next(0);
    std::atomic<int> failed(0);
    auto worker = [&]() {
        try {
            query_ctx<Index> c(&idx, &h->wdata, k ? k : 1);
            std::vector<uint32_t> m;
            if (match_hash) c.matches = &m;
            for (;;) {
                const uint32_t q = next.fetch_add(1);
                if (q >= nq) break;
                term_id_vec t(terms + offs[q], terms + offs[q + 1]);
                c.freq_sum = 0;
                c.topk.clear();
                m.clear();
                const uint64_t r = run_op(c, op, t);
                if (out_count) out_count[q] = r;
                if (out_topk)
                    for (uint32_t i = 0; i < k; ++i) out_topk[(size_t)q * k + i] = i < c.topk.topk().size() ? c.topk.topk()[i] : -INFINITY;
                if (out_topk_len) out_topk_len[q] = (uint32_t)c.topk.topk().size();
                if (out_freq_sum) out_freq_sum[q] = c.freq_sum;
                if (match_hash) {
                    uint64_t hsh = 0;
                    for (size_t i = 0; i < m.size(); ++i) hsh += (uint64_t)m[i] * (2 * (uint64_t)i + 1);
                    match_hash[q] = hsh;
                }
            }
        } catch (...) { failed = 1; }
    };
    std::vector<std::thread> pool;
    for (uint32_t i = 0; i < std::max(1u, nthreads); ++i) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
    return failed ? -1 : 0;
}
// op_perftest (queries.cpp:13-62)
template <class Index>
int perftest_t(const Index& idx, handle* h, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq, uint32_t runs, double* stats_out) {
    idx.begin_profile(nullptr);
    query_ctx<Index> c(&idx, &h->wdata, k ? k : 1);
    std::vector<double> times;
    volatile uint64_t sink = 0;
    for (uint32_t run = 0; run <= runs; ++run) {
        for (uint32_t q = 0; q < nq; ++q) {
            term_id_vec t(terms + offs[q], terms + offs[q + 1]);
            double tick = get_time_usecs();
            uint64_t r = run_op(c, op, t);
            sink += r;
            double el = get_time_usecs() - tick;
            if (run != 0) times.push_back(el);
        }
    }
    if (times.empty()) return -1;
    std::sort(times.begin(), times.end());
    double total = std::accumulate(times.begin(), times.end(), 0.0);
    stats_out[0] = total / times.size();
    stats_out[1] = times[times.size() / 2];
    stats_out[2] = times[90 * times.size() / 100];
    stats_out[3] = times[95 * times.size() / 100];
    stats_out[4] = total * 1e-6;
    return 0;
}
} // namespace
} // extern "C++"


int64_t oracle_list_size(void* hv, uint64_t term) {
    handle* h = (handle*)hv;
    try { return visit(h, [&](auto& idx) { return list_size_t(idx, term); }); } catch (...) { return -1; }
}
// sequential enumeration with next(): docid()+freq()+position() of every posting; returns n or a negative code
int64_t oracle_list_enumerate(void* hv, uint64_t term, uint32_t* docs, uint32_t* freqs, uint64_t cap) {
    handle* h = (handle*)hv;
    try { return visit(h, [&](auto& idx) { return list_enumerate_t(idx, term, docs, freqs, cap); }); } catch (...) { return -1; }
}
// reset(); then next_geq(probes[i]) in order (probes non-decreasing): records docid() and freq()-or-0
int oracle_list_next_geq(void* hv, uint64_t term, const uint32_t* probes, uint64_t np, uint32_t* out_docid, uint32_t* out_freq) {
    handle* h = (handle*)hv;
    try {
        return visit(h, [&](auto& idx) { list_next_geq_t(idx, term, probes, np, out_docid, out_freq); return 0; });
    } catch (...) { return -1; }
}
// move(positions[i]) in order (positions non-decreasing): docid() and freq()
int oracle_list_move(void* hv, uint64_t term, const uint32_t* positions, uint64_t np, uint32_t* out_docid, uint32_t* out_freq) {
    handle* h = (handle*)hv;
    try {
        return visit(h, [&](auto& idx) { list_move_t(idx, term, positions, np, out_docid, out_freq); return 0; });
    } catch (...) { return -1; }
}

// One stand-alone sequence of the Elias-Fano family (bits = little-endian u64 words) run through the reference's own
// sequence tests (oracle_pef.hpp: test_generic_sequence.hpp / test_partitioned_sequence.cpp restated) with the oracle's
// enumerators as readers. Returns 0, or the code of the first failed requirement.
int oracle_sequence_selftest(int kind, const uint8_t* bits, uint64_t nbytes, uint64_t nbits, uint64_t universe, const uint64_t* seq,
                             uint64_t n, const uint8_t* params) {
    oracle::bitvec bv;
    bv.bytes = bits;
    bv.nbits = nbits;
    bv.nbytes = nbytes;
    oracle::pef_params p{params[0], params[1], params[2], params[3], params[4]};
    try { return oracle::sequence_selftest(kind, bv, universe, seq, n, p); } catch (...) { return -2; }
}

// one query; topk holds k floats, matches (optional) holds match_cap doc-ids. Returns the operator's value.
int64_t oracle_query(void* hv, int op, uint32_t k, const uint32_t* terms, uint32_t nterms, float* topk, uint32_t* topk_len,
                     uint32_t* matches, uint64_t match_cap, uint64_t* freq_sum, oracle_profile* prof) {
    handle* h = (handle*)hv;
    try {
        if (op >= OP_RANKED_AND && !h->has_wand) return -5;
        return visit(h, [&](auto& idx) { return query_t(idx, h, op, k, terms, nterms, topk, topk_len, matches, match_cap, freq_sum, prof); });
    } catch (...) { return -1; }
}

// batch form of oracle_query: per-query results + summed profile (one untimed pass)
int oracle_query_batch(void* hv, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq,
                       uint64_t* out_count, float* out_topk, uint32_t* out_topk_len, uint64_t* out_freq_sum, oracle_profile* prof) {
    handle* h = (handle*)hv;
    try {
        if (op >= OP_RANKED_AND && !h->has_wand) return -5;
        return visit(h, [&](auto& idx) { return query_batch_t(idx, h, op, k, terms, offs, nq, out_count, out_topk, out_topk_len, out_freq_sum, prof); });
    } catch (...) { return -1; }
}

// oracle_query_batch on nthreads host threads (no profile); match_hash: see query_batch_mt_t
int oracle_query_batch_mt(void* hv, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq, uint32_t nthreads,
                          uint64_t* out_count, float* out_topk, uint32_t* out_topk_len, uint64_t* out_freq_sum, uint64_t* match_hash) {
    handle* h = (handle*)hv;
    try {
        if (op >= OP_RANKED_AND && !h->has_wand) return -5;
        return visit(h, [&](auto& idx) { return query_batch_mt_t(idx, h, op, k, terms, offs, nq, nthreads, out_count, out_topk, out_topk_len, out_freq_sum, match_hash); });
    } catch (...) { return -1; }
}

// op_perftest (queries.cpp:13-62): `runs`+1 passes, first untimed, per-query gettimeofday microseconds.
// stats_out = {avg, q50, q90, q95, total_timed_seconds}. Single thread.
int oracle_perftest(void* hv, int op, uint32_t k, const uint32_t* terms, const uint32_t* offs, uint32_t nq, uint32_t runs, double* stats_out) {
    handle* h = (handle*)hv;
    try {
        if (op >= OP_RANKED_AND && !h->has_wand) return -5;
        return visit(h, [&](auto& idx) { return perftest_t(idx, h, op, k, terms, offs, nq, runs, stats_out); });
    } catch (...) { return -1; }
}

} // extern "C"
