// CDNA4 device-side decode of one CHUNK of a partitioned Elias-Fano ("opt") list: <= 128 consecutive elements
// of one Elias-Fano / ranked-bitvector / all-ones partition (compact_elias_fano.hpp:14-61,
// compact_ranked_bitvector.hpp:14-50, all_ones_sequence.hpp:25-75, strict_elias_fano.hpp:38-80).
// Replaces the per-element unary scans of compact_elias_fano::enumerator::next / next_geq (184-232) and the
// popcount loop of compact_ranked_bitvector (256-302): one wave loads the chunk's high-bit words (one u64 per
// lane), popcounts + prefix-sums them, every lane locates its two elements by a 6-step search over the prefix
// counts (LDS) and a select-in-word, then reads its low bits with one unaligned 64-bit load.
// The chunk directory (cmax[], 12-dword entries) is built at upload by host_pef.hpp::opt_index_view::build_dir.
#pragma once
#include "device_codecs.hpp"

namespace ds2i_dev {

enum { CODEC_PEF = 5 };
enum { PEF_EF = 0, PEF_RB = 1, PEF_AO = 2 };
enum { PC_GPOS = 0, PC_PACKED, PC_D_BASE, PC_D_HI, PC_D_HBIAS, PC_D_LO, PC_F_BASE, PC_F_HI, PC_F_HBIAS, PC_F_LO, PC_F_PREV,
       PC_SPANS, PC_WORDS };

// k-th (0-based) set bit of a 64-bit word
DS2I_DEV uint32_t select64(uint64_t word, uint32_t k) {
    uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
    uint32_t pl = (uint32_t)__builtin_popcount(lo);
    uint32_t x = lo, off = 0;
    if (k >= pl) { k -= pl; x = hi; off = 32; }
#pragma unroll
    for (uint32_t w = 16; w; w >>= 1) {
        uint32_t p = (uint32_t)__builtin_popcount(x & ((1u << w) - 1u));
        if (k >= p) { k -= p; x >>= w; off += w; }
    }
    return off;
}

DS2I_DEV uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// Decodes `count` values of one side (docs: FREQ=false -> doc-ids; freqs: FREQ=true -> prefix sums S) of a chunk.
// bits = the collection's bit vector words; bit0 = absolute bit offset of this list's sequence; scr = >= 192 dwords LDS.
// v0 / v1 = values of elements lane / lane+64 (garbage for elements >= count).
template <bool FREQ>
DS2I_DEV void pef_decode_side(const uint8_t* bits, uint64_t bit0, uint32_t type, uint32_t l, uint32_t base, uint32_t hi,
                              uint32_t hbias, uint32_t lo, uint32_t span, uint32_t count, uint32_t* scr, uint32_t& v0,
                              uint32_t& v1) {
    const uint32_t lane = lane_id();
    if (type == PEF_AO) {
        v0 = base + lane;
        v1 = base + lane + 64;
        return;
    }
    const uint64_t hp0 = bit0 + hi;
    uint64_t wb = hp0 >> 6; // first word of the current 64-word window
    uint32_t done = 0;
    uint32_t h0 = 0, h1 = 0; // high-bit positions (relative to bit0) of my two elements
    bool first = true;
    const uint64_t* words = (const uint64_t*)bits;
    for (;;) {
        uint32_t nw = 64;
        if (span != 0xFFFFu && first) {
            uint32_t need = (uint32_t)(((hp0 & 63) + span + 63) >> 6);
            nw = need < 64 ? need : 64;
        }
        uint64_t word = lane < nw ? words[wb + lane] : 0;
        if (first && lane == 0) word &= ~0ull << (hp0 & 63);
        const uint32_t pc = (uint32_t)__builtin_popcountll(word);
        const uint32_t incl = wave_incl_scan(pc);
        const uint32_t total = bcast(incl, 63);
        scr[lane] = incl - pc;
        scr[64 + 2 * lane] = (uint32_t)word;
        scr[65 + 2 * lane] = (uint32_t)(word >> 32);
        wave_sync();
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const uint32_t r = lane + 64 * slot;
            if (r >= done && r < count && r - done < total) {
                const uint32_t rr = r - done;
                uint32_t idx = 0;
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1)
                    if (scr[idx + step] <= rr) idx += step;
                const uint64_t w = ((uint64_t)scr[65 + 2 * idx] << 32) | scr[64 + 2 * idx];
                const uint32_t bitpos = select64(w, rr - scr[idx]);
                const uint32_t hp = (uint32_t)(((wb + idx) << 6) + bitpos - bit0);
                if (slot) h1 = hp; else h0 = hp;
            }
        }
        wave_sync();
        done += total;
        if (done >= count) break;
        wb += 64;
        first = false;
    }
    if (type == PEF_RB) {
        v0 = base + (h0 - hbias);
        v1 = base + (h1 - hbias);
        return;
    }
    const uint32_t mask = l ? ((1u << l) - 1u) : 0u;
    uint32_t low0 = 0, low1 = 0;
    if (l) {
        const uint64_t lb0 = bit0 + lo + (uint64_t)lane * l, lb1 = lb0 + 64ull * l;
        if (lane < count) low0 = (uint32_t)(ld64(bits + (lb0 >> 3)) >> (lb0 & 7)) & mask;
        if (lane + 64 < count) low1 = (uint32_t)(ld64(bits + (lb1 >> 3)) >> (lb1 & 7)) & mask;
    }
    v0 = base + (((h0 - hbias - lane) << l) | low0) + (FREQ ? lane : 0u);
    v1 = base + (((h1 - hbias - (lane + 64)) << l) | low1) + (FREQ ? lane + 64 : 0u);
}

} // namespace ds2i_dev
