// Host-side bit utilities shared by the index builder and the image parser.
// Product code (CPU, build side). Nothing here runs in the timed query path.
//
// Conventions follow what ds2i expects from ot/succinct (absent from
// /root/reference; restated from SURVEY.md Appendix B): a bit string is an
// array of u64 words, bit i = (words[i/64] >> (i%64)) & 1 (LSB first).
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ds2i_host {

inline uint32_t msb64(uint64_t x) { assert(x); return 63u - (uint32_t)__builtin_clzll(x); }
inline uint32_t msb32(uint32_t x) { assert(x); return 31u - (uint32_t)__builtin_clz(x); }
// util.hpp:30-33
inline uint64_t ceil_log2(uint64_t x) { return x > 1 ? msb64(x - 1) + 1 : 0; }
template <class A, class B> inline A ceil_div(A a, B b) { return (a + b - 1) / b; }

// Growable LSB-first bit string (the role of succinct::bit_vector_builder).
class bitvec_builder {
public:
    uint64_t size() const { return m_size; }
    void zero_extend(uint64_t n) {
        m_size += n;
        m_words.resize(ceil_div(m_size, (uint64_t)64), 0);
    }
    void set(uint64_t pos, bool b) {
        assert(pos < m_size);
        uint64_t& w = m_words[pos >> 6];
        uint64_t m = uint64_t(1) << (pos & 63);
        w = b ? (w | m) : (w & ~m);
    }
    // overwrite len (<=64) bits at pos
    void set_bits(uint64_t pos, uint64_t bits, unsigned len) {
        assert(pos + len <= m_size);
        if (!len) return;
        uint64_t mask = len == 64 ? ~uint64_t(0) : ((uint64_t(1) << len) - 1);
        bits &= mask;
        unsigned sh = pos & 63;
        uint64_t wi = pos >> 6;
        m_words[wi] = (m_words[wi] & ~(mask << sh)) | (bits << sh);
        if (sh + len > 64) {
            unsigned done = 64 - sh;
            m_words[wi + 1] = (m_words[wi + 1] & ~(mask >> done)) | (bits >> done);
        }
    }
    void append_bits(uint64_t bits, unsigned len) {
        uint64_t pos = m_size;
        zero_extend(len);
        set_bits(pos, bits, len);
    }
    std::vector<uint64_t> const& words() const { return m_words; }
    std::vector<uint64_t>& words() { return m_words; }

private:
    uint64_t m_size = 0;
    std::vector<uint64_t> m_words;
};

// Read-only view over an LSB-first bit string that may start at any byte.
struct bitview {
    const uint8_t* bytes = nullptr; // start of word array (may be unaligned)
    uint64_t nbits = 0;
    uint64_t nbytes = 0;
    bool get(uint64_t pos) const { return (bytes[pos >> 3] >> (pos & 7)) & 1; }
    // up to 57 bits starting at pos
    uint64_t get_bits(uint64_t pos, unsigned len) const {
        if (!len) return 0;
        uint64_t byte = pos >> 3;
        uint64_t w = 0;
        uint64_t avail = nbytes - byte;
        std::memcpy(&w, bytes + byte, avail >= 8 ? 8 : (size_t)avail);
        w >>= (pos & 7);
        return len >= 64 ? w : (w & ((uint64_t(1) << len) - 1));
    }
};

} // namespace ds2i_host
