// Host-side (CPU, build-time) block encoders for the ds2i block index formats.
// Product code: used by the index builder that bench.py / tests / the CLI use to
// materialise indexes. Decoding on the product path happens ONLY in HIP
// (device_codecs.hpp); the CPU decoders live in oracle/ and are test-only.
//
// Formats:
//   vbyte            reference block_codecs.hpp:17-99  (TightVariableByte)
//   interpolative    reference block_codecs.hpp:101-148 + interpolative_coding.hpp:10-77
//   optpfor          reference block_codecs.hpp:150-208; FastPFor OPTPFor<4,Simple16<false>>
//                    is NOT in /root/reference (empty submodule) -> restated from the
//                    published FastPFor algorithm, "parity unpinned" at byte level.
//   varint-G8IU      reference block_codecs.hpp:229-284; FastPFor VarIntG8IU restated.
//   qmx              reference block_codecs.hpp:317-333 + qmx_codec.hpp:176-634
//   mixed            reference mixed_block.hpp:38-66
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "host_bits.hpp"

namespace ds2i_host {

static const uint32_t BLOCK = 128;
typedef std::vector<uint8_t> bytes_t;

// ---------------------------------------------------------------- vbyte
// 7 data bits per byte, little-endian groups, terminator byte has bit 7 SET.
inline void vbyte_encode(uint32_t v, bytes_t& out) {
    while (v >= 128) {
        out.push_back((uint8_t)(v & 127));
        v >>= 7;
    }
    out.push_back((uint8_t)(v | 128));
}

// ---------------------------------------------------------------- interpolative
class bit_writer32 {
public:
    void write(uint32_t bits, uint32_t len) {
        if (!len) return;
        uint32_t pos = (uint32_t)(m_size & 31);
        m_size += len;
        if (pos == 0) {
            m_buf.push_back(bits);
        } else {
            m_buf.back() |= bits << pos;
            if (len > 32 - pos) m_buf.push_back(bits >> (32 - pos));
        }
    }
    // truncated binary code of val in [0,u)
    void write_int(uint32_t val, uint32_t u) {
        uint32_t b = msb32(u);
        uint64_t m = (uint64_t(1) << (b + 1)) - u;
        if (val < m) {
            write(val, b);
        } else {
            val += (uint32_t)m;
            write(val >> 1, b);
            write(val & 1, 1);
        }
    }
    void write_interpolative(const uint32_t* in, size_t n, uint32_t low, uint32_t high) {
        if (!n) return;
        size_t h = n / 2;
        uint32_t val = in[h];
        write_int(val - low, high - low + 1);
        write_interpolative(in, h, low, val);
        write_interpolative(in + h + 1, n - h - 1, val, high);
    }
    size_t size() const { return m_size; }
    const uint8_t* data() const { return (const uint8_t*)m_buf.data(); }

private:
    std::vector<uint32_t> m_buf;
    size_t m_size = 0;
};

inline void interpolative_encode(const uint32_t* in, uint32_t sum_of_values, size_t n, bytes_t& out) {
    uint32_t pre[BLOCK];
    pre[0] = in[0];
    for (size_t i = 1; i < n; ++i) pre[i] = pre[i - 1] + in[i];
    if (sum_of_values == uint32_t(-1)) {
        sum_of_values = pre[n - 1];
        vbyte_encode(sum_of_values, out);
    }
    bit_writer32 bw;
    bw.write_interpolative(pre, n - 1, 0, sum_of_values);
    out.insert(out.end(), bw.data(), bw.data() + ceil_div(bw.size(), (size_t)8));
}

// ---------------------------------------------------------------- Simple16
// 4-bit selector in the top bits, 28 payload bits. First value sits at the HIGH end
// of the payload (FastPFor's unpackers read (w>>27)&1 first). Byte-compat with
// upstream FastPFor is unverified (source absent); keep it in this one table.
struct s16_layout { uint8_t n; uint8_t bits[28]; };
inline const s16_layout* s16_table() {
    static s16_layout T[16];
    static bool init = false;
    if (!init) {
        // {count,width} runs per selector
        static const uint8_t runs[16][6] = {
            {28, 1, 0, 0, 0, 0}, {7, 2, 14, 1, 0, 0}, {7, 1, 7, 2, 7, 1}, {14, 1, 7, 2, 0, 0},
            {14, 2, 0, 0, 0, 0}, {1, 4, 8, 3, 0, 0},  {1, 3, 4, 4, 3, 3}, {7, 4, 0, 0, 0, 0},
            {4, 5, 2, 4, 0, 0},  {2, 4, 4, 5, 0, 0},  {3, 6, 2, 5, 0, 0}, {2, 5, 3, 6, 0, 0},
            {4, 7, 0, 0, 0, 0},  {1, 10, 2, 9, 0, 0}, {2, 14, 0, 0, 0, 0}, {1, 28, 0, 0, 0, 0}};
        for (int s = 0; s < 16; ++s) {
            int k = 0;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < runs[s][2 * r]; ++c) T[s].bits[k++] = runs[s][2 * r + 1];
            T[s].n = (uint8_t)k;
        }
        init = true;
    }
    return T;
}

// Returns number of words; appends them to out when out != nullptr.
inline uint32_t simple16_encode(const uint32_t* in, size_t n, std::vector<uint32_t>* out) {
    const s16_layout* T = s16_table();
    uint32_t words = 0;
    size_t i = 0;
    while (i < n) {
        size_t rem = n - i;
        int sel = -1;
        for (int s = 0; s < 16 && sel < 0; ++s) {
            size_t cnt = std::min<size_t>(rem, T[s].n);
            bool ok = true;
            for (size_t j = 0; j < cnt && ok; ++j) ok = (uint64_t(in[i + j]) >> T[s].bits[j]) == 0;
            if (ok) sel = s;
        }
        if (sel < 0) throw std::runtime_error("Simple16: value needs more than 28 bits");
        size_t cnt = std::min<size_t>(rem, T[sel].n);
        if (out) {
            uint32_t w = uint32_t(sel) << 28;
            uint32_t pos = 28;
            for (size_t j = 0; j < cnt; ++j) {
                pos -= T[sel].bits[j];
                w |= in[i + j] << pos;
            }
            out->push_back(w);
        }
        ++words;
        i += cnt;
    }
    return words;
}

// ---------------------------------------------------------------- OptPFor
static const uint32_t OPTPFOR_LOGS[17] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 20, 32};

inline uint32_t maxbits128(const uint32_t* in) {
    uint32_t acc = 0;
    for (uint32_t i = 0; i < BLOCK; ++i) acc |= in[i];
    return acc ? msb32(acc) + 1 : 0;
}

// exceptions array: nExc position deltas (pos0, pos_i - pos_{i-1} - 1) then nExc (v>>b)-1
inline uint32_t optpfor_exceptions(const uint32_t* in, uint32_t b, uint32_t* exc) {
    uint32_t pos[BLOCK], val[BLOCK], n = 0;
    for (uint32_t i = 0; i < BLOCK; ++i)
        if (b < 32 && (in[i] >> b) != 0) { pos[n] = i; val[n] = in[i] >> b; ++n; }
    for (uint32_t i = 0; i < n; ++i) {
        exc[i] = i ? pos[i] - pos[i - 1] - 1 : pos[0];
        exc[i + n] = val[i] - 1;
    }
    return n;
}

inline uint32_t optpfor_try_b(uint32_t b, const uint32_t* in) {
    if (b == 32) return BLOCK;
    uint32_t size = ceil_div(BLOCK * b, 32u);
    uint32_t exc[2 * BLOCK];
    uint32_t n = optpfor_exceptions(in, b, exc);
    if (n) size += simple16_encode(exc, 2 * n, nullptr);
    return size;
}

// reference block_codecs.hpp:156-182 (ds2i's early-stopping findBestB)
inline uint32_t optpfor_find_best_b(const uint32_t* in) {
    uint32_t b = 0, bsize = ~uint32_t(0);
    const uint32_t mb = maxbits128(in);
    uint32_t i = 0;
    while (mb > 28 + OPTPFOR_LOGS[i]) ++i;
    for (; i < 17; ++i) {
        if (OPTPFOR_LOGS[i] > mb) break;
        uint32_t csize = optpfor_try_b(OPTPFOR_LOGS[i], in);
        if (csize <= bsize) { b = OPTPFOR_LOGS[i]; bsize = csize; }
    }
    return b;
}

// 128 values -> header | Simple16 exceptions | 4 groups of b words (32 values each,
// value j of a group at bit j*b of the group's b-word little-endian stream).
inline void optpfor_encode_block(const uint32_t* in, bytes_t& out, int force_b = -1) {
    uint32_t b = force_b >= 0 ? (uint32_t)force_b : optpfor_find_best_b(in);
    std::vector<uint32_t> w;
    if (b < 32) {
        uint32_t exc[2 * BLOCK];
        uint32_t n = optpfor_exceptions(in, b, exc);
        std::vector<uint32_t> ew;
        if (n) simple16_encode(exc, 2 * n, &ew);
        w.push_back((b << 26) | (n << 16) | (uint32_t)ew.size());
        w.insert(w.end(), ew.begin(), ew.end());
        size_t base = w.size();
        w.resize(base + 4 * b, 0);
        if (b) {
            uint32_t mask = (uint32_t)((uint64_t(1) << b) - 1);
            for (uint32_t i = 0; i < BLOCK; ++i) {
                uint64_t bit = uint64_t(i) * b;
                uint64_t v = uint64_t(in[i] & mask) << (bit & 31);
                w[base + (bit >> 5)] |= (uint32_t)v;
                if ((bit & 31) + b > 32) w[base + (bit >> 5) + 1] |= (uint32_t)(v >> 32);
            }
        }
    } else {
        w.push_back(b << 26);
        w.insert(w.end(), in, in + BLOCK);
    }
    const uint8_t* p = (const uint8_t*)w.data();
    out.insert(out.end(), p, p + 4 * w.size());
}

inline void optpfor_encode(const uint32_t* in, uint32_t sum, size_t n, bytes_t& out, int force_b = -1) {
    if (n < BLOCK) { interpolative_encode(in, sum, n, out); return; }
    optpfor_encode_block(in, out, force_b);
}

// ---------------------------------------------------------------- VarInt-G8IU
// Group = 1 descriptor + 8 data bytes. Descriptor starts 0xFF; bit j is CLEARED iff
// data byte j is the last byte of an integer. Ints never straddle groups.
inline void varint_g8iu_encode(const uint32_t* in, uint32_t sum, size_t n, bytes_t& out) {
    if (n < BLOCK) { interpolative_encode(in, sum, n, out); return; }
    size_t i = 0;
    while (i < n) {
        uint8_t grp[9];
        std::memset(grp, 0, sizeof grp);
        uint8_t desc = 0xFF;
        uint32_t len = 0;
        while (i < n) {
            uint32_t v = in[i];
            uint32_t need = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < (1u << 24) ? 3 : 4;
            if (len + need > 8) break;
            for (uint32_t j = 0; j < need; ++j) grp[1 + len + j] = (uint8_t)(v >> (8 * j));
            len += need;
            desc &= (uint8_t)~(1u << (len - 1));
            ++i;
        }
        grp[0] = desc;
        out.insert(out.end(), grp, grp + 9);
    }
}

// ---------------------------------------------------------------- QMX
// Restatement of qmx_codec.hpp:176-634 (selector choice + 4-lane vertical packing).
// The decoder (device_codecs.hpp / oracle) is the source of truth for the format; the
// byte stream produced here is checked byte-for-byte against the reference encoder
// (oracle/_ref, built from /root/reference/qmx_codec.hpp) in tests.
struct qmx_class { uint8_t bits, type, ints; bool two; };
inline const qmx_class* qmx_classes() {
    // order == type id 0..14
    static const qmx_class C[15] = {{0, 0, 0 /*256*/, false}, {1, 1, 128, false}, {2, 2, 64, false},
                                    {3, 3, 40, false},        {4, 4, 32, false},  {5, 5, 24, false},
                                    {6, 6, 20, false},        {7, 7, 36, true},   {8, 8, 16, false},
                                    {9, 9, 28, true},         {10, 10, 12, false}, {12, 11, 20, true},
                                    {16, 12, 8, false},       {21, 13, 12, true},  {32, 14, 4, false}};
    return C;
}
inline uint32_t qmx_capacity(uint32_t type) { return type == 0 ? 256 : qmx_classes()[type].ints; }
inline int qmx_type_of_bits(uint32_t bits) {
    const qmx_class* C = qmx_classes();
    for (int t = 0; t < 15; ++t) if (C[t].bits == bits) return t;
    return -1;
}
inline uint8_t qmx_bits_needed(uint32_t v) {
    if (v == 1) return 0; // a run of ones costs zero bits
    static const uint8_t W[14] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32};
    for (int k = 0; k < 14; ++k)
        if (W[k] == 32 || v <= ((1u << W[k]) - 1)) return W[k];
    return 32;
}

// pack one vector (or vector pair) of class `type` from `src` (already zero padded)
inline void qmx_pack_vector(uint32_t type, const uint32_t* src, bytes_t& out) {
    const qmx_class& c = qmx_classes()[type];
    const uint32_t w = c.bits;
    if (type == 0) return;
    uint32_t v1[4] = {0, 0, 0, 0}, v2[4] = {0, 0, 0, 0};
    if (!c.two) {
        for (uint32_t j = 0; j < c.ints; ++j) v1[j & 3] |= src[j] << ((j >> 2) * w);
        const uint8_t* p = (const uint8_t*)v1;
        out.insert(out.end(), p, p + 16);
        return;
    }
    // two-vector classes: R1 full rows + one split row in the first vector; the second
    // vector starts with the split row's high part, further rows begin at bit `off2`.
    const uint32_t R1 = 32 / w;            // 7->4, 9->3, 12->2, 21->1
    const uint32_t lowpart = 32 - R1 * w;   // 7->4, 9->5, 12->8, 21->11
    const uint32_t off2 = (w == 12) ? 8 : (w == 21) ? 11 : (w - lowpart);
    const uint32_t rows = c.ints / 4;
    for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t l = 0; l < 4; ++l) {
            uint32_t v = src[4 * r + l];
            if (r < R1) v1[l] |= v << (r * w);
            else if (r == R1) { v1[l] |= v << (r * w); v2[l] |= v >> lowpart; }
            else v2[l] |= v << ((r - R1 - 1) * w + off2);
        }
    const uint8_t* p = (const uint8_t*)v1;
    out.insert(out.end(), p, p + 16);
    p = (const uint8_t*)v2;
    out.insert(out.end(), p, p + 16);
}

inline void qmx_write_run(const uint32_t* src, uint32_t raw_count, uint32_t bits, bytes_t& payload,
                          bytes_t& keys) {
    const int type = qmx_type_of_bits(bits);
    const uint32_t cap = qmx_capacity(type);
    uint32_t count = (raw_count + cap - 1) / cap;
    std::vector<uint32_t> padded(src, src + raw_count);
    padded.resize((size_t)count * cap, 0);
    const uint32_t* cur = padded.data();
    uint32_t remaining = raw_count;
    while (count > 0) {
        uint32_t batch = count > 16 ? 16 : count;
        keys.push_back((uint8_t)((type << 4) | (~(batch - 1) & 0x0F)));
        count -= batch;
        for (uint32_t k = 0; k < batch; ++k) {
            if (bits == 8 || bits == 16 || bits == 32) {
                // byte/short/word classes are WRITTEN truncated at the end of the run
                uint32_t m = std::min(cap, remaining);
                for (uint32_t j = 0; j < m; ++j) {
                    uint32_t v = cur[j];
                    for (uint32_t bb = 0; bb < bits / 8; ++bb) payload.push_back((uint8_t)(v >> (8 * bb)));
                }
            } else {
                qmx_pack_vector(type, cur, payload);
            }
            cur += cap;
            remaining = remaining > cap ? remaining - cap : 0;
        }
    }
}

inline size_t qmx_encode_block(const uint32_t* src, bytes_t& out) {
    const uint32_t W = 512;
    std::vector<uint8_t> len(BLOCK + W + 8, 0);
    for (uint32_t i = 0; i < BLOCK; ++i) len[i] = qmx_bits_needed(src[i]);
    for (uint32_t g = 0; g < BLOCK + 4; g += 4) {
        uint8_t m = std::max(std::max(len[g], len[g + 1]), std::max(len[g + 2], len[g + 3]));
        len[g] = len[g + 1] = len[g + 2] = len[g + 3] = m;
    }
    static const uint8_t NEXT[33] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 0, 16, 0, 0, 0, 21,
                                     0, 0, 0, 0, 32, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64};
    uint32_t cur = 0;
    while (cur < BLOCK) {
        uint32_t rem = BLOCK - cur;
        auto largest = [&](uint32_t k) { uint8_t m = 0; for (uint32_t j = 0; j < k; ++j) m = std::max(m, len[cur + j]); return m; };
        auto setall = [&](uint32_t k, uint8_t v) { for (uint32_t j = 0; j < k; ++j) len[cur + j] = v; };
        if (rem < 4) {
            uint8_t m = largest(8);
            if (m <= 8) setall(8, 8); else if (m <= 16) setall(8, 16); else if (m <= 32) setall(8, 32);
        } else if (rem < 8) {
            if (largest(8) <= 8) setall(8, 8);
        } else if (rem < 16) {
            if (largest(16) <= 8) setall(16, 8);
        }
        const uint8_t w = len[cur];
        const int type = qmx_type_of_bits(w);
        if (type < 0) throw std::runtime_error("QMX: bad width");
        const uint32_t cap = qmx_capacity(type);
        for (uint32_t blk = 0; blk < cap; blk += 4)
            if (len[cur + blk] > w) len[cur] = len[cur + 1] = len[cur + 2] = len[cur + 3] = NEXT[w];
        if (len[cur] == w) { setall(cap, w); cur += cap; }
    }
    bytes_t payload, keys;
    uint32_t rlen = 1, bits = len[0];
    for (uint32_t i = 1; i < BLOCK; ++i) {
        if (len[i] == bits) { ++rlen; continue; }
        qmx_write_run(src + i - rlen, rlen, bits, payload, keys);
        bits = len[i];
        rlen = 1;
    }
    qmx_write_run(src + BLOCK - rlen, rlen, bits, payload, keys);
    size_t before = out.size();
    out.insert(out.end(), payload.begin(), payload.end());
    out.insert(out.end(), keys.rbegin(), keys.rend());
    return out.size() - before;
}

inline void qmx_encode(const uint32_t* in, uint32_t sum, size_t n, bytes_t& out) {
    if (n < BLOCK) { interpolative_encode(in, sum, n, out); return; }
    bytes_t tmp;
    size_t l = qmx_encode_block(in, tmp);
    vbyte_encode((uint32_t)l, out);
    out.insert(out.end(), tmp.begin(), tmp.end());
}

// ---------------------------------------------------------------- mixed
enum mixed_type : uint8_t { MIXED_PFOR = 0, MIXED_VARINT = 1, MIXED_INTERP = 2 };

inline void mixed_encode_type(mixed_type t, int pfor_b, const uint32_t* in, uint32_t sum, size_t n, bytes_t& out) {
    if (n < BLOCK) {
        if (t != MIXED_INTERP) throw std::runtime_error("Partial blocks can only be encoded with interpolative");
    } else {
        out.push_back((uint8_t)t);
    }
    switch (t) {
    case MIXED_PFOR: optpfor_encode(in, sum, n, out, pfor_b); break;
    case MIXED_VARINT: varint_g8iu_encode(in, sum, n, out); break;
    case MIXED_INTERP: interpolative_encode(in, sum, n, out); break;
    }
}

// Deterministic per-block policy for block_mixed images that do not come out of the optimiser (host_hybrid.hpp), the
// one SURVEY.md section 8(d) fixes for the C5 configuration: every 16th block of a list (1, 17, ..) interpolative, otherwise
// VarInt-G8IU when every value fits 8 bits (the reference's fastest decoder on its CPU, mixed_block.hpp:198-217 lists it
// first) and OptPFor (findBestB) when not. An index written this way holds a substantial share of all three types.
inline void mixed_encode(const uint32_t* in, uint32_t sum, size_t n, bytes_t& out, uint64_t block_no = 0) {
    if (n < BLOCK) { mixed_encode_type(MIXED_INTERP, -1, in, sum, n, out); return; }
    uint64_t total = 0;
    uint32_t any = 0;
    for (size_t i = 0; i < n; ++i) { total += in[i]; any |= in[i]; }
    mixed_type t = any < 256u ? MIXED_VARINT : MIXED_PFOR;
    if ((block_no & 15u) == 1u && total < 0xFFFFFFFFull) t = MIXED_INTERP; // blocks 1, 17, 33, ..; interpolative codes u32 prefix sums
    mixed_encode_type(t, -1, in, sum, n, out);
}

enum codec_kind : int {
    CODEC_OPTPFOR = 0, CODEC_VARINT = 1, CODEC_INTERPOLATIVE = 2, CODEC_QMX = 3, CODEC_MIXED = 4
};

inline void block_encode(int codec, const uint32_t* in, uint32_t sum, size_t n, bytes_t& out, uint64_t block_no = 0) {
    switch (codec) {
    case CODEC_OPTPFOR: optpfor_encode(in, sum, n, out); break;
    case CODEC_VARINT: varint_g8iu_encode(in, sum, n, out); break;
    case CODEC_INTERPOLATIVE: interpolative_encode(in, sum, n, out); break;
    case CODEC_QMX: qmx_encode(in, sum, n, out); break;
    case CODEC_MIXED: mixed_encode(in, sum, n, out, block_no); break;
    default: throw std::invalid_argument("unknown codec");
    }
}

} // namespace ds2i_host
