// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.cpp). CPU restatement of the Elias-Fano index family
// read path ("opt" index = freq_index<partitioned_sequence<indexed_sequence>,
//                                      positive_sequence<partitioned_sequence<strict_sequence>>>):
//   integer codes (read)                 integer_codes.hpp:21-45
//   compact_elias_fano::enumerator       compact_elias_fano.hpp:138-417 (move / next / next_geq / prev_value,
//                                        slow_move 263-289, slow_next_geq 291-336, next_reader 359-388)
//   compact_ranked_bitvector::enumerator compact_ranked_bitvector.hpp:117-345
//   all_ones_sequence::enumerator        all_ones_sequence.hpp:25-75
//   indexed_sequence::enumerator         indexed_sequence.hpp:89-164
//   strict_elias_fano / strict_sequence  strict_elias_fano.hpp:38-80, strict_sequence.hpp:98-174
//   partitioned_sequence::enumerator     partitioned_sequence.hpp:122-347
//   positive_sequence::enumerator        positive_sequence.hpp:31-78
//   freq_index::operator[] / document_enumerator   freq_index.hpp:116-214 ; bitvector_collection.hpp:57-67
// succinct::bit_vector / enumerator / unary_enumerator are absent from /root/reference (empty submodule):
// restated from SURVEY.md Appendix B -> "parity unpinned" for the bit-level container, like the block indexes.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <utility>

namespace oracle {

typedef std::pair<uint64_t, uint64_t> value_type; // (position, value)

struct pef_params { uint8_t ef_log_sampling0, ef_log_sampling1, rb_log_rank1_sampling, rb_log_sampling1, log_partition_size; };

// ---------------------------------------------------------------- succinct::bit_vector (restated)
struct bitvec {
    const uint8_t* bytes = nullptr; // u64 words, possibly unaligned
    uint64_t nbits = 0, nbytes = 0;
    uint64_t word(uint64_t i) const {
        uint64_t w = 0;
        uint64_t off = 8 * i;
        if (off < nbytes) std::memcpy(&w, bytes + off, std::min<uint64_t>(8, nbytes - off));
        return w;
    }
    bool operator[](uint64_t pos) const { return (bytes[pos >> 3] >> (pos & 7)) & 1; }
    uint64_t get_word56(uint64_t pos) const {
        uint64_t byte = pos / 8, w = 0;
        if (byte < nbytes) std::memcpy(&w, bytes + byte, std::min<uint64_t>(8, nbytes - byte));
        return w >> (pos % 8);
    }
    uint64_t get_bits(uint64_t pos, uint64_t len) const {
        if (!len) return 0;
        uint64_t lo = word(pos / 64) >> (pos % 64);
        if (pos % 64 + len > 64) lo |= word(pos / 64 + 1) << (64 - pos % 64);
        return len == 64 ? lo : (lo & ((uint64_t(1) << len) - 1));
    }
    uint64_t predecessor1(uint64_t pos) const {
        uint64_t block = pos / 64, shift = 64 - pos % 64 - 1;
        uint64_t w = (word(block) << shift) >> shift;
        while (!w) w = word(--block);
        return block * 64 + (63 - (uint64_t)__builtin_clzll(w));
    }
};

struct bit_enumerator { // succinct::bit_vector::enumerator
    const bitvec* bv;
    uint64_t pos;
    bit_enumerator(const bitvec& b, uint64_t p) : bv(&b), pos(p) {}
    uint64_t take(uint64_t l) { uint64_t v = bv->get_bits(pos, l); pos += l; return v; }
    uint64_t skip_zeros() {
        uint64_t z = 0;
        while (!(*bv)[pos]) { ++pos; ++z; }
        ++pos; // consume the one
        return z;
    }
    uint64_t position() const { return pos; }
};
inline uint64_t read_gamma(bit_enumerator& it) { uint64_t l = it.skip_zeros(); return (it.take(l) | (uint64_t(1) << l)) - 1; }
inline uint64_t read_gamma_nonzero(bit_enumerator& it) { return read_gamma(it) + 1; }
inline uint64_t read_delta(bit_enumerator& it) { uint64_t l = read_gamma(it); return (it.take(l) | (uint64_t(1) << l)) - 1; }

static inline uint64_t select_in_word(uint64_t w, uint64_t k) {
    for (uint64_t i = 0; i < k; ++i) w &= w - 1;
    return (uint64_t)__builtin_ctzll(w);
}

struct unary_enum { // succinct::bit_vector::unary_enumerator
    const bitvec* bv = nullptr;
    uint64_t m_position = 0, m_buf = 0;
    unary_enum() {}
    unary_enum(const bitvec& b, uint64_t pos) : bv(&b), m_position(pos) { m_buf = bv->word(pos / 64) & (~uint64_t(0) << (pos % 64)); }
    uint64_t position() const { return m_position; }
    uint64_t next() {
        uint64_t buf = m_buf;
        while (!buf) { m_position += 64; buf = bv->word(m_position / 64); }
        uint64_t p = (uint64_t)__builtin_ctzll(buf);
        m_buf = buf & (buf - 1);
        m_position = (m_position & ~uint64_t(63)) + p;
        return m_position;
    }
    void skip(uint64_t k) { // position at the k-th one (0-based) at/after the current position
        uint64_t skipped = 0, buf = m_buf, w;
        while (skipped + (w = (uint64_t)__builtin_popcountll(buf)) <= k) { skipped += w; m_position += 64; buf = bv->word(m_position / 64); }
        uint64_t p = select_in_word(buf, k - skipped);
        m_buf = buf & (~uint64_t(0) << p);
        m_position = (m_position & ~uint64_t(63)) + p;
    }
    void skip0(uint64_t k) { // position at the k-th zero; the just-consumed one counts as a zero (compact_elias_fano.hpp:305-308)
        uint64_t skipped = 0, p = m_position % 64, w;
        uint64_t buf = ~m_buf & (~uint64_t(0) << p);
        while (skipped + (w = (uint64_t)__builtin_popcountll(buf)) <= k) { skipped += w; m_position += 64; buf = ~bv->word(m_position / 64); }
        p = select_in_word(buf, k - skipped);
        m_buf = ~buf & (~uint64_t(0) << p);
        m_position = (m_position & ~uint64_t(63)) + p;
    }
};

static inline uint64_t pef_ceil_log2(uint64_t x) { return x > 1 ? (63u - (uint64_t)__builtin_clzll(x - 1)) + 1 : 0; }
static inline uint64_t pef_msb(uint64_t x) { return 63u - (uint64_t)__builtin_clzll(x); }

// ---------------------------------------------------------------- compact_elias_fano
struct cef_offsets {
    uint64_t universe = 0, n = 0, log_sampling0 = 0, log_sampling1 = 0, lower_bits = 0, mask = 0, higher_bits_length = 0,
             pointer_size = 0, pointers0 = 0, pointers1 = 0, pointers0_offset = 0, pointers1_offset = 0, higher_bits_offset = 0,
             lower_bits_offset = 0, end = 0;
    cef_offsets() {}
    cef_offsets(uint64_t base, uint64_t u, uint64_t n_, pef_params const& p) : universe(u), n(n_), log_sampling0(p.ef_log_sampling0), log_sampling1(p.ef_log_sampling1) {
        lower_bits = u > n ? pef_msb(u / n) : 0;
        mask = (uint64_t(1) << lower_bits) - 1;
        higher_bits_length = n + (u >> lower_bits) + 2;
        pointer_size = pef_ceil_log2(higher_bits_length);
        pointers0 = log_sampling0 >= 64 ? 0 : ((higher_bits_length - n) >> log_sampling0);
        pointers1 = n >> log_sampling1;
        pointers0_offset = base;
        pointers1_offset = pointers0_offset + pointers0 * pointer_size;
        higher_bits_offset = pointers1_offset + pointers1 * pointer_size;
        lower_bits_offset = higher_bits_offset + higher_bits_length;
        end = lower_bits_offset + n * lower_bits;
    }
};
inline uint64_t cef_bitsize(pef_params const& p, uint64_t u, uint64_t n) { return cef_offsets(0, u, n, p).end; }

class cef_enumerator {
public:
    cef_enumerator() {}
    cef_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p)
        : m_bv(&bv), m_of(offset, universe, n, p), m_position(n), m_value(universe) {}
    value_type move(uint64_t position) {
        if (position == m_position) return value();
        uint64_t skip = position - m_position;
        if (position > m_position && skip <= 8) {
            m_position = position;
            if (m_position == size()) {
                m_value = m_of.universe;
            } else {
                unary_enum he = m_high;
                for (uint64_t i = 0; i < skip; ++i) he.next();
                m_value = ((he.position() - m_of.higher_bits_offset - m_position - 1) << m_of.lower_bits) | read_low();
                m_high = he;
            }
            return value();
        }
        return slow_move(position);
    }
    value_type next_geq(uint64_t lower_bound) {
        if (lower_bound == m_value) return value();
        uint64_t high_lower_bound = lower_bound >> m_of.lower_bits;
        uint64_t cur_high = m_value >> m_of.lower_bits;
        uint64_t high_diff = high_lower_bound - cur_high;
        if (lower_bound > m_value && high_diff <= 8) {
            next_reader nv(*this, m_position + 1);
            uint64_t val;
            do {
                m_position += 1;
                if (m_position < size()) val = nv();
                else { val = m_of.universe; break; }
            } while (val < lower_bound);
            m_value = val;
            return value();
        }
        return slow_next_geq(lower_bound);
    }
    uint64_t size() const { return m_of.n; }
    value_type next() {
        m_position += 1;
        if (m_position < size()) m_value = read_next();
        else m_value = m_of.universe;
        return value();
    }
    uint64_t prev_value() const {
        if (m_position == 0) return 0;
        uint64_t prev_high;
        if (m_position < size()) prev_high = m_bv->predecessor1(m_high.position() - 1);
        else prev_high = m_bv->predecessor1(m_of.lower_bits_offset - 1);
        prev_high -= m_of.higher_bits_offset;
        uint64_t prev_pos = m_position - 1;
        uint64_t prev_low = m_bv->get_word56(m_of.lower_bits_offset + prev_pos * m_of.lower_bits) & m_of.mask;
        return ((prev_high - prev_pos - 1) << m_of.lower_bits) | prev_low;
    }
    uint64_t position() const { return m_position; }

private:
    value_type slow_move(uint64_t position) {
        if (position == size()) { m_position = position; m_value = m_of.universe; return value(); }
        uint64_t skip = position - m_position, to_skip;
        if (position > m_position && (skip >> m_of.log_sampling1) == 0) {
            to_skip = skip - 1;
        } else {
            uint64_t ptr = position >> m_of.log_sampling1;
            uint64_t high_pos = pointer(m_of.pointers1_offset, ptr);
            uint64_t high_rank = ptr << m_of.log_sampling1;
            m_high = unary_enum(*m_bv, m_of.higher_bits_offset + high_pos);
            to_skip = position - high_rank;
        }
        m_high.skip(to_skip);
        m_position = position;
        m_value = read_next();
        return value();
    }
    value_type slow_next_geq(uint64_t lower_bound) {
        if (lower_bound >= m_of.universe) return move(size());
        uint64_t high_lower_bound = lower_bound >> m_of.lower_bits;
        uint64_t cur_high = m_value >> m_of.lower_bits;
        uint64_t high_diff = high_lower_bound - cur_high;
        uint64_t to_skip;
        if (lower_bound > m_value && (m_of.log_sampling0 >= 64 || (high_diff >> m_of.log_sampling0) == 0)) {
            to_skip = high_diff;
        } else {
            uint64_t ptr = m_of.log_sampling0 >= 64 ? 0 : (high_lower_bound >> m_of.log_sampling0);
            uint64_t high_pos = pointer(m_of.pointers0_offset, ptr);
            uint64_t high_rank0 = m_of.log_sampling0 >= 64 ? 0 : (ptr << m_of.log_sampling0);
            m_high = unary_enum(*m_bv, m_of.higher_bits_offset + high_pos);
            to_skip = high_lower_bound - high_rank0;
        }
        m_high.skip0(to_skip);
        m_position = m_high.position() - m_of.higher_bits_offset - high_lower_bound;
        next_reader rv(*this, m_position);
        while (true) {
            if (m_position == size()) { m_value = m_of.universe; return value(); }
            uint64_t val = rv();
            if (val >= lower_bound) { m_value = val; return value(); }
            m_position++;
        }
    }
    value_type value() const { return value_type(m_position, m_value); }
    uint64_t read_low() const { return m_bv->get_word56(m_of.lower_bits_offset + m_position * m_of.lower_bits) & m_of.mask; }
    uint64_t read_next() {
        uint64_t high = m_high.next() - m_of.higher_bits_offset;
        return ((high - m_position - 1) << m_of.lower_bits) | read_low();
    }
    struct next_reader {
        cef_enumerator& e;
        unary_enum he;
        uint64_t high_base, lower_bits, lower_base, mask;
        next_reader(cef_enumerator& en, uint64_t position)
            : e(en), he(en.m_high), high_base(en.m_of.higher_bits_offset + position + 1), lower_bits(en.m_of.lower_bits),
              lower_base(en.m_of.lower_bits_offset + position * en.m_of.lower_bits), mask(en.m_of.mask) {}
        ~next_reader() { e.m_high = he; }
        uint64_t operator()() {
            uint64_t high = he.next() - high_base;
            uint64_t low = e.m_bv->get_word56(lower_base) & mask;
            high_base += 1;
            lower_base += lower_bits;
            return (high << lower_bits) | low;
        }
    };
    uint64_t pointer(uint64_t offset, uint64_t i) const {
        if (i == 0) return 0;
        return m_bv->get_word56(offset + (i - 1) * m_of.pointer_size) & ((uint64_t(1) << m_of.pointer_size) - 1);
    }
    const bitvec* m_bv = nullptr;
    cef_offsets m_of;
    uint64_t m_position = 0, m_value = 0;
    unary_enum m_high;
};

// ---------------------------------------------------------------- compact_ranked_bitvector
struct crb_offsets {
    uint64_t universe = 0, n = 0, log_rank1_sampling = 0, log_sampling1 = 0, rank1_sample_size = 0, pointer_size = 0, rank1_samples = 0,
             pointers1 = 0, rank1_samples_offset = 0, pointers1_offset = 0, bits_offset = 0, end = 0;
    crb_offsets() {}
    crb_offsets(uint64_t base, uint64_t u, uint64_t n_, pef_params const& p) : universe(u), n(n_), log_rank1_sampling(p.rb_log_rank1_sampling), log_sampling1(p.rb_log_sampling1) {
        rank1_sample_size = pef_ceil_log2(n + 1);
        pointer_size = pef_ceil_log2(u);
        rank1_samples = log_rank1_sampling >= 64 ? 0 : (u >> log_rank1_sampling);
        pointers1 = n >> log_sampling1;
        rank1_samples_offset = base;
        pointers1_offset = rank1_samples_offset + rank1_samples * rank1_sample_size;
        bits_offset = pointers1_offset + pointers1 * pointer_size;
        end = bits_offset + u;
    }
};
inline uint64_t crb_bitsize(pef_params const& p, uint64_t u, uint64_t n) { return crb_offsets(0, u, n, p).end; }

class crb_enumerator {
public:
    crb_enumerator() {}
    crb_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p)
        : m_bv(&bv), m_of(offset, universe, n, p), m_position(n), m_value(universe) {}
    value_type move(uint64_t position) {
        if (position == m_position) return value();
        uint64_t skip = position - m_position;
        if (position > m_position && skip <= 8) {
            m_position = position;
            if (m_position == size()) {
                m_value = m_of.universe;
            } else {
                unary_enum he = m_enum;
                for (uint64_t i = 0; i < skip; ++i) he.next();
                m_value = he.position() - m_of.bits_offset;
                m_enum = he;
            }
            return value();
        }
        return slow_move(position);
    }
    value_type next_geq(uint64_t lower_bound) {
        if (lower_bound == m_value) return value();
        uint64_t diff = lower_bound - m_value;
        if (lower_bound > m_value && diff <= 8) {
            unary_enum he = m_enum;
            uint64_t val;
            do {
                m_position += 1;
                if (m_position < size()) val = he.next() - m_of.bits_offset;
                else { val = m_of.universe; break; }
            } while (val < lower_bound);
            m_value = val;
            m_enum = he;
            return value();
        }
        return slow_next_geq(lower_bound);
    }
    value_type next() {
        m_position += 1;
        if (m_position < size()) m_value = m_enum.next() - m_of.bits_offset;
        else m_value = m_of.universe;
        return value();
    }
    uint64_t size() const { return m_of.n; }
    uint64_t prev_value() const {
        if (m_position == 0) return 0;
        uint64_t pos;
        if (m_position < size()) pos = m_bv->predecessor1(m_enum.position() - 1);
        else pos = m_bv->predecessor1(m_of.end - 1);
        return pos - m_of.bits_offset;
    }

private:
    value_type slow_move(uint64_t position) {
        uint64_t skip = position - m_position;
        if (position == size()) { m_position = position; m_value = m_of.universe; return value(); }
        uint64_t to_skip;
        if (position > m_position && (skip >> m_of.log_sampling1) == 0) {
            to_skip = skip - 1;
        } else {
            uint64_t ptr = position >> m_of.log_sampling1;
            uint64_t ptr_pos = pointer(m_of.pointers1_offset, ptr, m_of.pointer_size);
            m_enum = unary_enum(*m_bv, m_of.bits_offset + ptr_pos);
            to_skip = position - (ptr << m_of.log_sampling1);
        }
        m_enum.skip(to_skip);
        m_position = position;
        m_value = m_enum.next() - m_of.bits_offset;
        return value();
    }
    value_type slow_next_geq(uint64_t lower_bound) {
        if (lower_bound >= m_of.universe) return move(size());
        uint64_t skip = lower_bound - m_value;
        m_enum = unary_enum(*m_bv, m_of.bits_offset + lower_bound);
        uint64_t begin;
        if (lower_bound > m_value && (m_of.log_rank1_sampling >= 64 || (skip >> m_of.log_rank1_sampling) == 0)) {
            begin = m_of.bits_offset + m_value;
        } else {
            uint64_t block = m_of.log_rank1_sampling >= 64 ? 0 : (lower_bound >> m_of.log_rank1_sampling);
            m_position = pointer(m_of.rank1_samples_offset, block, m_of.rank1_sample_size);
            begin = m_of.bits_offset + (m_of.log_rank1_sampling >= 64 ? 0 : (block << m_of.log_rank1_sampling));
        }
        uint64_t end = m_of.bits_offset + lower_bound;
        uint64_t begin_word = begin / 64, begin_shift = begin % 64, end_word = end / 64, end_shift = end % 64;
        uint64_t word = (m_bv->word(begin_word) >> begin_shift) << begin_shift;
        while (begin_word < end_word) {
            m_position += (uint64_t)__builtin_popcountll(word);
            word = m_bv->word(++begin_word);
        }
        if (end_shift) m_position += (uint64_t)__builtin_popcountll(word << (64 - end_shift));
        if (m_position < size()) m_value = m_enum.next() - m_of.bits_offset;
        else m_value = m_of.universe;
        return value();
    }
    value_type value() const { return value_type(m_position, m_value); }
    uint64_t pointer(uint64_t offset, uint64_t i, uint64_t size) const {
        if (i == 0) return 0;
        return m_bv->get_word56(offset + (i - 1) * size) & ((uint64_t(1) << size) - 1);
    }
    const bitvec* m_bv = nullptr;
    crb_offsets m_of;
    uint64_t m_position = 0, m_value = 0;
    unary_enum m_enum;
};

// ---------------------------------------------------------------- all_ones
class ao_enumerator {
public:
    ao_enumerator() {}
    ao_enumerator(uint64_t universe) : m_universe(universe), m_position(universe) {}
    value_type move(uint64_t position) { m_position = position; return value_type(m_position, m_position); }
    value_type next_geq(uint64_t lb) { m_position = lb; return value_type(m_position, m_position); }
    value_type next() { m_position += 1; return value_type(m_position, m_position); }
    uint64_t size() const { return m_universe; }
    uint64_t prev_value() const { return m_position == 0 ? 0 : m_position - 1; }

private:
    uint64_t m_universe = 0, m_position = 0;
};

// ---------------------------------------------------------------- strict_elias_fano
class sef_enumerator {
public:
    sef_enumerator() {}
    sef_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p) : m_ef(bv, offset, universe - n + 1, n, p) {}
    value_type move(uint64_t position) { auto v = m_ef.move(position); return value_type(v.first, v.second + v.first); }
    value_type next() { auto v = m_ef.next(); return value_type(v.first, v.second + v.first); }
    uint64_t size() const { return m_ef.size(); }
    uint64_t prev_value() const { return m_ef.position() ? m_ef.prev_value() + m_ef.position() - 1 : 0; }

private:
    cef_enumerator m_ef;
};

// ---------------------------------------------------------------- indexed_sequence / strict_sequence
enum { T_EF = 0, T_RB = 1, T_AO = 2 };

template <bool STRICT>
struct base_sequence {
    static pef_params seq_params(pef_params p) {
        if (STRICT) { p.ef_log_sampling0 = 63; p.rb_log_rank1_sampling = 63; }
        return p;
    }
    static uint64_t bitsize(pef_params const& params, uint64_t universe, uint64_t n) {
        uint64_t best = (universe == n) ? 0 : uint64_t(-1);
        pef_params sp = seq_params(params);
        uint64_t ef = (STRICT ? cef_bitsize(sp, universe - n + 1, n) : cef_bitsize(sp, universe, n)) + 1;
        if (ef < best) best = ef;
        uint64_t rb = crb_bitsize(sp, universe, n) + 1;
        if (rb < best) best = rb;
        return best;
    }
    class enumerator {
    public:
        enumerator() {}
        enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& params) {
            pef_params sp = seq_params(params);
            if (universe == n) m_type = T_AO;
            else m_type = (int)(bv.get_word56(offset) & 1);
            switch (m_type) {
            case T_EF:
                if (STRICT) m_sef = sef_enumerator(bv, offset + 1, universe, n, sp);
                else m_ef = cef_enumerator(bv, offset + 1, universe, n, sp);
                break;
            case T_RB: m_rb = crb_enumerator(bv, offset + 1, universe, n, sp); break;
            default: m_ao = ao_enumerator(universe); break;
            }
        }
        value_type move(uint64_t p) { return m_type == T_EF ? (STRICT ? m_sef.move(p) : m_ef.move(p)) : m_type == T_RB ? m_rb.move(p) : m_ao.move(p); }
        value_type next() { return m_type == T_EF ? (STRICT ? m_sef.next() : m_ef.next()) : m_type == T_RB ? m_rb.next() : m_ao.next(); }
        value_type next_geq(uint64_t lb) { // indexed_sequence only (strict sequences have no next_geq)
            return m_type == T_EF ? m_ef.next_geq(lb) : m_type == T_RB ? m_rb.next_geq(lb) : m_ao.next_geq(lb);
        }
        uint64_t size() const { return m_type == T_EF ? (STRICT ? m_sef.size() : m_ef.size()) : m_type == T_RB ? m_rb.size() : m_ao.size(); }
        uint64_t prev_value() const { return m_type == T_EF ? (STRICT ? m_sef.prev_value() : m_ef.prev_value()) : m_type == T_RB ? m_rb.prev_value() : m_ao.prev_value(); }
        int type() const { return m_type; }

    private:
        int m_type = T_AO;
        cef_enumerator m_ef;
        sef_enumerator m_sef;
        crb_enumerator m_rb;
        ao_enumerator m_ao;
    };
};

struct pef_profile { uint64_t partitions_entered = 0, algorithmic_bytes = 0; };

// ---------------------------------------------------------------- partitioned_sequence
struct partition_construction_test; // test_partitioned_sequence.cpp:13-44 restated (below)

template <bool STRICT>
class partitioned_enumerator {
    friend struct partition_construction_test;
public:
    typedef typename base_sequence<STRICT>::enumerator base_enum;
    partitioned_enumerator() {}
    partitioned_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& params, pef_profile* prof)
        : m_params(params), m_size(n), m_universe(universe), m_bv(&bv), m_prof(prof) {
        bit_enumerator it(bv, offset);
        m_partitions = read_gamma_nonzero(it);
        if (m_partitions == 1) {
            m_cur_partition = 0;
            m_cur_begin = 0;
            m_cur_end = n;
            uint64_t universe_bits = pef_ceil_log2(universe);
            m_cur_base = it.take(universe_bits);
            uint64_t ub = 0;
            if (n > 1) {
                uint64_t universe_delta = read_delta(it);
                ub = universe_delta ? universe_delta : (universe - m_cur_base - 1);
            }
            m_partition_enum = base_enum(*m_bv, it.position(), ub + 1, n, m_params);
            m_cur_upper_bound = m_cur_base + ub;
            if (m_prof) {
                m_prof->partitions_entered += 1;
                m_prof->algorithmic_bytes += (it.position() - offset + base_sequence<STRICT>::bitsize(m_params, ub + 1, n) + 7) / 8;
            }
        } else {
            m_endpoint_bits = read_gamma(it);
            uint64_t cur_offset = it.position();
            m_sizes = cef_enumerator(bv, cur_offset, n, m_partitions - 1, params);
            cur_offset += cef_bitsize(params, n, m_partitions - 1);
            m_upper_bounds = cef_enumerator(bv, cur_offset, universe, m_partitions + 1, params);
            cur_offset += cef_bitsize(params, universe, m_partitions + 1);
            m_endpoints_offset = cur_offset;
            cur_offset += m_endpoint_bits * (m_partitions - 1);
            m_sequences_offset = cur_offset;
        }
        m_position = size();
        slow_move();
    }
    value_type move(uint64_t position) {
        m_position = position;
        if (m_position >= m_cur_begin && m_position < m_cur_end) {
            uint64_t val = m_cur_base + m_partition_enum.move(m_position - m_cur_begin).second;
            return value_type(m_position, val);
        }
        return slow_move();
    }
    value_type next_geq(uint64_t lower_bound) {
        if (lower_bound >= m_cur_base && lower_bound <= m_cur_upper_bound) {
            auto val = m_partition_enum.next_geq(lower_bound - m_cur_base);
            m_position = m_cur_begin + val.first;
            return value_type(m_position, m_cur_base + val.second);
        }
        return slow_next_geq(lower_bound);
    }
    value_type next() {
        ++m_position;
        if (m_position < m_cur_end) {
            uint64_t val = m_cur_base + m_partition_enum.next().second;
            return value_type(m_position, val);
        }
        return slow_next();
    }
    uint64_t size() const { return m_size; }
    uint64_t prev_value() const {
        if (m_position == m_cur_begin) return m_cur_partition ? m_cur_base - 1 : 0;
        return m_cur_base + m_partition_enum.prev_value();
    }
    uint64_t num_partitions() const { return m_partitions; }

private:
    value_type slow_next() {
        if (m_position == m_size) {
            m_partition_enum.next();
            return value_type(m_position, m_universe);
        }
        switch_partition(m_cur_partition + 1);
        uint64_t val = m_cur_base + m_partition_enum.move(0).second;
        return value_type(m_position, val);
    }
    value_type slow_move() {
        if (m_position == size()) {
            if (m_partitions > 1) switch_partition(m_partitions - 1);
            m_partition_enum.move(m_partition_enum.size());
            return value_type(m_position, m_universe);
        }
        auto size_it = m_sizes.next_geq(m_position + 1);
        switch_partition(size_it.first);
        uint64_t val = m_cur_base + m_partition_enum.move(m_position - m_cur_begin).second;
        return value_type(m_position, val);
    }
    value_type slow_next_geq(uint64_t lower_bound) {
        if (m_partitions == 1) {
            if (lower_bound < m_cur_base) return move(0);
            return move(size());
        }
        auto ub_it = m_upper_bounds.next_geq(lower_bound);
        if (ub_it.first == 0) return move(0);
        if (ub_it.first == m_upper_bounds.size()) return move(size());
        switch_partition(ub_it.first - 1);
        return next_geq(lower_bound);
    }
    void switch_partition(uint64_t partition) {
        uint64_t endpoint = partition ? (m_bv->get_word56(m_endpoints_offset + (partition - 1) * m_endpoint_bits) & ((uint64_t(1) << m_endpoint_bits) - 1)) : 0;
        uint64_t partition_begin = m_sequences_offset + endpoint;
        m_cur_partition = partition;
        auto size_it = m_sizes.move(partition);
        m_cur_end = size_it.second;
        m_cur_begin = m_sizes.prev_value();
        auto ub_it = m_upper_bounds.move(partition + 1);
        m_cur_upper_bound = ub_it.second;
        m_cur_base = m_upper_bounds.prev_value() + (partition ? 1 : 0);
        m_partition_enum = base_enum(*m_bv, partition_begin, m_cur_upper_bound - m_cur_base + 1, m_cur_end - m_cur_begin, m_params);
        if (m_prof) { // partition bits + endpoint + 3 x 8 B of upper-level EF reads
            m_prof->partitions_entered += 1;
            m_prof->algorithmic_bytes += (base_sequence<STRICT>::bitsize(m_params, m_cur_upper_bound - m_cur_base + 1, m_cur_end - m_cur_begin) + 7) / 8 +
                                         (m_endpoint_bits + 7) / 8 + 24;
        }
    }
    pef_params m_params{};
    uint64_t m_partitions = 0, m_endpoints_offset = 0, m_endpoint_bits = 0, m_sequences_offset = 0, m_size = 0, m_universe = 0;
    uint64_t m_position = 0, m_cur_partition = 0, m_cur_begin = 0, m_cur_end = 0, m_cur_base = 0, m_cur_upper_bound = 0;
    const bitvec* m_bv = nullptr;
    cef_enumerator m_sizes, m_upper_bounds;
    base_enum m_partition_enum;
    pef_profile* m_prof = nullptr;
};

// ---------------------------------------------------------------- uniform_partitioned_sequence
// uniform_partitioned_sequence.hpp:113-315: partitions of exactly 2^log_partition_size elements (the last may be shorter),
// so the partition of a position is a shift and there is no `sizes` sequence.
template <bool STRICT>
class uniform_enumerator {
    friend struct partition_construction_test;
public:
    typedef typename base_sequence<STRICT>::enumerator base_enum;
    uniform_enumerator() {}
    uniform_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& params, pef_profile* prof)
        : m_params(params), m_size(n), m_universe(universe), m_bv(&bv), m_prof(prof) {
        bit_enumerator it(bv, offset);
        m_partitions = read_gamma_nonzero(it);
        if (m_partitions == 1) {
            m_cur_partition = 0;
            m_cur_begin = 0;
            m_cur_end = n;
            m_cur_base = it.take(pef_ceil_log2(universe));
            uint64_t ub = 0;
            if (n > 1) {
                uint64_t universe_delta = read_delta(it);
                ub = universe_delta ? universe_delta : (universe - m_cur_base - 1);
            }
            m_partition_enum = base_enum(*m_bv, it.position(), ub + 1, n, m_params);
            m_cur_upper_bound = m_cur_base + ub;
            if (m_prof) {
                m_prof->partitions_entered += 1;
                m_prof->algorithmic_bytes += (it.position() - offset + base_sequence<STRICT>::bitsize(m_params, ub + 1, n) + 7) / 8;
            }
        } else {
            m_endpoint_bits = read_gamma(it);
            uint64_t cur_offset = it.position();
            m_upper_bounds = cef_enumerator(bv, cur_offset, universe, m_partitions + 1, params);
            cur_offset += cef_bitsize(params, universe, m_partitions + 1);
            m_endpoints_offset = cur_offset;
            cur_offset += m_endpoint_bits * (m_partitions - 1);
            m_sequences_offset = cur_offset;
        }
        m_position = size();
        slow_move();
    }
    value_type move(uint64_t position) {
        m_position = position;
        if (m_position >= m_cur_begin && m_position < m_cur_end)
            return value_type(m_position, m_cur_base + m_partition_enum.move(m_position - m_cur_begin).second);
        return slow_move();
    }
    value_type next_geq(uint64_t lower_bound) {
        if (lower_bound >= m_cur_base && lower_bound <= m_cur_upper_bound) {
            auto val = m_partition_enum.next_geq(lower_bound - m_cur_base);
            m_position = m_cur_begin + val.first;
            return value_type(m_position, m_cur_base + val.second);
        }
        return slow_next_geq(lower_bound);
    }
    value_type next() {
        ++m_position;
        if (m_position < m_cur_end) return value_type(m_position, m_cur_base + m_partition_enum.next().second);
        return slow_next();
    }
    uint64_t size() const { return m_size; }
    uint64_t prev_value() const {
        if (m_position == m_cur_begin) return m_cur_partition ? m_cur_base - 1 : 0;
        return m_cur_base + m_partition_enum.prev_value();
    }

private:
    value_type slow_next() {
        if (m_position == m_size) {
            m_partition_enum.next();
            return value_type(m_position, m_universe);
        }
        switch_partition(m_cur_partition + 1);
        return value_type(m_position, m_cur_base + m_partition_enum.move(0).second);
    }
    value_type slow_move() {
        if (m_position == size()) {
            if (m_partitions > 1) switch_partition(m_partitions - 1);
            m_partition_enum.move(m_partition_enum.size());
            return value_type(m_position, m_universe);
        }
        switch_partition(m_position >> m_params.log_partition_size);
        return value_type(m_position, m_cur_base + m_partition_enum.move(m_position - m_cur_begin).second);
    }
    value_type slow_next_geq(uint64_t lower_bound) {
        if (m_partitions == 1) return lower_bound < m_cur_base ? move(0) : move(size());
        auto ub_it = m_upper_bounds.next_geq(lower_bound);
        if (ub_it.first == 0) return move(0);
        if (ub_it.first == m_upper_bounds.size()) return move(size());
        switch_partition(ub_it.first - 1);
        return next_geq(lower_bound);
    }
    void switch_partition(uint64_t partition) {
        uint64_t endpoint = partition ? (m_bv->get_word56(m_endpoints_offset + (partition - 1) * m_endpoint_bits) & ((uint64_t(1) << m_endpoint_bits) - 1)) : 0;
        m_cur_partition = partition;
        m_cur_begin = partition << m_params.log_partition_size;
        m_cur_end = std::min(size(), (partition + 1) << m_params.log_partition_size);
        auto ub_it = m_upper_bounds.move(partition + 1);
        m_cur_upper_bound = ub_it.second;
        m_cur_base = m_upper_bounds.prev_value() + (partition ? 1 : 0);
        m_partition_enum = base_enum(*m_bv, m_sequences_offset + endpoint, m_cur_upper_bound - m_cur_base + 1, m_cur_end - m_cur_begin, m_params);
        if (m_prof) { // partition bits + endpoint + 2 x 8 B of the upper_bounds EF
            m_prof->partitions_entered += 1;
            m_prof->algorithmic_bytes += (base_sequence<STRICT>::bitsize(m_params, m_cur_upper_bound - m_cur_base + 1, m_cur_end - m_cur_begin) + 7) / 8 +
                                         (m_endpoint_bits + 7) / 8 + 16;
        }
    }
    pef_params m_params{};
    uint64_t m_partitions = 0, m_endpoints_offset = 0, m_endpoint_bits = 0, m_sequences_offset = 0, m_size = 0, m_universe = 0;
    uint64_t m_position = 0, m_cur_partition = 0, m_cur_begin = 0, m_cur_end = 0, m_cur_base = 0, m_cur_upper_bound = 0;
    const bitvec* m_bv = nullptr;
    cef_enumerator m_upper_bounds;
    base_enum m_partition_enum;
    pef_profile* m_prof = nullptr;
};

// un-partitioned sequences behind the constructor shape the partitioned ones have (profile pointer ignored:
// A_skip is only defined for the block and partitioned layouts)
struct plain_ef_enumerator : cef_enumerator { // compact_elias_fano as a whole-list sequence (ef_index docs)
    plain_ef_enumerator() {}
    plain_ef_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p, pef_profile*)
        : cef_enumerator(bv, offset, universe, n, p) {}
};
struct plain_sef_enumerator : sef_enumerator { // strict_elias_fano with the index's own parameters (ef_index freqs)
    plain_sef_enumerator() {}
    plain_sef_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p, pef_profile*)
        : sef_enumerator(bv, offset, universe, n, p) {}
};
template <bool STRICT>
struct single_enumerator : base_sequence<STRICT>::enumerator { // indexed_sequence / strict_sequence (single_index)
    single_enumerator() {}
    single_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p, pef_profile*)
        : base_sequence<STRICT>::enumerator(bv, offset, universe, n, p) {}
};

// ---------------------------------------------------------------- the reference's own sequence tests, restated
// Each check returns 0 or the number of the first failed requirement (reported to the Python test as "line").
#define SEQ_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)

// test_partitioned_sequence.cpp:13-44: every partition entered through switch_partition() must report the base, the
// upper bound and the elements the plain sequence implies.
struct partition_construction_test {
    template <class Enum>
    static int run(Enum& r, const uint64_t* seq, uint64_t) {
        if (r.m_partitions == 1) return 0;
        for (uint64_t p = 0; p < r.m_partitions; ++p) {
            r.switch_partition(p);
            const uint64_t b = r.m_cur_begin, e = r.m_cur_end;
            SEQ_REQUIRE(e > b && e <= r.m_size, 101);
            SEQ_REQUIRE((p ? seq[b - 1] + 1 : seq[0]) == r.m_cur_base, 102);
            SEQ_REQUIRE(seq[e - 1] == r.m_cur_upper_bound, 103);
            for (uint64_t i = b; i < e; ++i) SEQ_REQUIRE(seq[i] == r.m_cur_base + r.m_partition_enum.move(i - b).second, 104);
        }
        return 0;
    }
};

// test_generic_sequence.hpp:28-88 (random access, enumeration, prev_value, small skips)
template <class Reader>
inline int sequence_test_move_next(Reader r, const uint64_t* seq, uint64_t n) {
    SEQ_REQUIRE(n == r.size(), 201);
    value_type val;
    for (uint64_t i = 0; i < n; ++i) {
        val = r.move(i);
        SEQ_REQUIRE(val.first == i, 202);
        SEQ_REQUIRE(val.second == seq[i], 203);
        SEQ_REQUIRE(r.prev_value() == (i ? seq[i - 1] : 0), 204);
    }
    r.move(n);
    SEQ_REQUIRE(r.prev_value() == seq[n - 1], 205);
    val = r.move(0);
    for (uint64_t i = 0; i < n; ++i) {
        SEQ_REQUIRE(val.second == seq[i], 206);
        SEQ_REQUIRE(r.prev_value() == (i ? seq[i - 1] : 0), 207);
        val = r.next();
    }
    SEQ_REQUIRE(val.first == r.size(), 208);
    SEQ_REQUIRE(r.prev_value() == seq[n - 1], 209);
    const uint64_t stride = n > 4096 ? n / 2048 : 1; // the reference tries every i; a stride keeps big inputs in seconds
    for (uint64_t i = 0; i < n; i += stride)
        for (uint64_t skip = 1; skip < n - i; skip <<= 1) {
            Reader rr = r;
            rr.move(i);
            val = rr.move(i + skip);
            SEQ_REQUIRE(val.first == i + skip, 210);
            SEQ_REQUIRE(val.second == seq[i + skip], 211);
        }
    return 0;
}

// test_generic_sequence.hpp:90-164 (successor of every gap position, beyond the last element, small skips)
template <class Reader>
inline int sequence_test_next_geq(Reader r, const uint64_t* seq, uint64_t n) {
    value_type val;
    uint64_t last = 0, rng = 0x9E3779B97F4A7C15ull;
    for (uint64_t i = 0; i < n; ++i) {
        if (seq[i] == last) continue;
        Reader rr = r;
        for (int t = 0; t < 10; ++t) {
            uint64_t p;
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            if (i == 0) p = last + 1;
            else if (i == 1) p = seq[i];
            else p = last + 1 + (rng >> 33) % (seq[i] - last);
            val = rr.next_geq(p);
            SEQ_REQUIRE(val.first == i, 301);
            SEQ_REQUIRE(val.second == seq[i], 302);
            SEQ_REQUIRE(rr.prev_value() == (val.first ? seq[val.first - 1] : 0), 303);
        }
        last = seq[i];
    }
    {
        Reader rr = r;
        val = rr.next_geq(seq[n - 1] + 1);
        SEQ_REQUIRE(val.first == rr.size(), 304);
        SEQ_REQUIRE(rr.prev_value() == seq[n - 1], 305);
        // beyond the universe. The reference asks for position == size() on the reader that already sits at the end; its
        // own small-skip loop (compact_elias_fano.hpp:194-209) steps the position once more before it notices, so for tiny
        // universes (singleton {1} in universe 2) it answers size() + 1, and all_ones_sequence answers the bound itself
        // (all_ones_sequence.hpp:47-53) -- restated as ">= size()".
        val = rr.next_geq(2 * seq[n - 1] + 1);
        SEQ_REQUIRE(val.first >= rr.size(), 306);
    }
    const uint64_t stride = n > 4096 ? n / 2048 : 1;
    for (uint64_t i = 0; i < n; i += stride)
        for (uint64_t skip = 1; skip < n - i; skip <<= 1) {
            uint64_t exp_pos = i + skip;
            // first of a run of equal values -- but never before the reader's own position: when seq[i] already equals
            // the bound the enumerator stays where it is (compact_elias_fano.hpp:186-188). The reference's loop walks
            // back past i for runs of three or more; SURVEY.md section 4 notes its next_geq test is never instantiated.
            while (exp_pos > i && seq[exp_pos - 1] == seq[i + skip]) --exp_pos;
            Reader rr = r;
            rr.move(i);
            val = rr.next_geq(seq[i + skip]);
            SEQ_REQUIRE(val.first == exp_pos, 307);
            SEQ_REQUIRE(val.second == seq[i + skip], 308);
        }
    return 0;
}

// kinds numbered like enum ds2i_sequence_kind (include/ds2i_build.h)
inline int sequence_selftest(int kind, const bitvec& bv, uint64_t universe, const uint64_t* seq, uint64_t n, pef_params const& p) {
    int rc = 0;
    switch (kind) {
    case 0: { cef_enumerator r(bv, 0, universe, n, p); rc = sequence_test_move_next(r, seq, n); return rc ? rc : sequence_test_next_geq(r, seq, n); }
    case 1: { crb_enumerator r(bv, 0, universe, n, p); rc = sequence_test_move_next(r, seq, n); return rc ? rc : sequence_test_next_geq(r, seq, n); }
    case 2: { base_sequence<false>::enumerator r(bv, 0, universe, n, p); rc = sequence_test_move_next(r, seq, n); return rc ? rc : sequence_test_next_geq(r, seq, n); }
    case 3: { base_sequence<true>::enumerator r(bv, 0, universe, n, p); return sequence_test_move_next(r, seq, n); }
    case 4: { partitioned_enumerator<false> r(bv, 0, universe, n, p, nullptr); rc = partition_construction_test::run(r, seq, n);
              if (!rc) rc = sequence_test_move_next(r, seq, n); return rc ? rc : sequence_test_next_geq(r, seq, n); }
    case 5: { partitioned_enumerator<true> r(bv, 0, universe, n, p, nullptr); rc = partition_construction_test::run(r, seq, n);
              return rc ? rc : sequence_test_move_next(r, seq, n); }
    case 6: { uniform_enumerator<false> r(bv, 0, universe, n, p, nullptr); rc = sequence_test_move_next(r, seq, n); return rc ? rc : sequence_test_next_geq(r, seq, n); }
    case 7: { uniform_enumerator<true> r(bv, 0, universe, n, p, nullptr); return sequence_test_move_next(r, seq, n); }
    default: return -1;
    }
}
#undef SEQ_REQUIRE

// ---------------------------------------------------------------- positive_sequence<Base> (positive_sequence.hpp:33-78)
template <class Base>
class positive_enumerator {
public:
    positive_enumerator() {}
    positive_enumerator(const bitvec& bv, uint64_t offset, uint64_t universe, uint64_t n, pef_params const& p, pef_profile* prof)
        : m_base(bv, offset, universe, n, p, prof), m_position(n) {}
    value_type move(uint64_t position) {
        uint64_t prev = m_cur;
        if (position != m_position + 1) {
            if (position == 0) {
                m_cur = m_base.move(0).second;
                m_position = 0;
                return value_type(m_position, m_cur);
            }
            prev = m_base.move(position - 1).second;
        }
        m_cur = m_base.next().second;
        m_position = position;
        return value_type(position, m_cur - prev);
    }

private:
    Base m_base;
    uint64_t m_position = 0, m_cur = 0;
};

// ---------------------------------------------------------------- freq_index::document_enumerator (freq_index.hpp:116-173)
struct opt_profile { pef_profile docs, freqs; uint64_t list_header_bytes = 0; };

template <class DocsEnum, class FreqsBase>
class freq_document_enumerator {
public:
    typedef positive_enumerator<FreqsBase> freqs_enum;
    freq_document_enumerator() {}
    freq_document_enumerator(DocsEnum d, freqs_enum f) : m_docs(d), m_freqs(f) { reset(); }
    void reset() { m_cur_pos = 0; m_cur_docid = m_docs.move(0).second; }
    void next() { auto v = m_docs.next(); m_cur_pos = v.first; m_cur_docid = v.second; }
    void next_geq(uint64_t lb) { auto v = m_docs.next_geq(lb); m_cur_pos = v.first; m_cur_docid = v.second; }
    void move(uint64_t p) { auto v = m_docs.move(p); m_cur_pos = v.first; m_cur_docid = v.second; }
    uint64_t docid() const { return m_cur_docid; }
    uint64_t freq() { return m_freqs.move(m_cur_pos).second; }
    uint64_t position() const { return m_cur_pos; }
    uint64_t size() const { return m_docs.size(); }

private:
    uint64_t m_cur_pos = 0, m_cur_docid = 0;
    DocsEnum m_docs;
    freqs_enum m_freqs;
};
typedef freq_document_enumerator<partitioned_enumerator<false>, partitioned_enumerator<true>> opt_document_enumerator;

struct bit_collection {
    uint64_t m_size = 0;
    bitvec endpoints, bits;
    uint64_t get(pef_params const& p, uint64_t i) const {
        cef_enumerator e(endpoints, 0, bits.nbits, m_size, p);
        return e.move(i).second;
    }
};

} // namespace oracle
