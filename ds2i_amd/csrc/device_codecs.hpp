// CDNA4 (gfx950, wave64) device-side block decoders for the ds2i block index formats.
// One wavefront decodes one 128-posting block cooperatively:
//   coalesced dword loads of the encoded block -> per-wave LDS staging window ->
//   per-lane bit extraction (v_alignbit/v_alignbyte) -> wave prefix sums (DPP).
// No MFMA: this is integer/bit work bound by HBM latency/bandwidth.
//
// Replaces (reference file:line):
//   TightVariableByte::decode        block_codecs.hpp:84-98
//   interpolative_block::decode      block_codecs.hpp:127-147, interpolative_coding.hpp:79-153
//   optpfor_block::decode            block_codecs.hpp:210-226 (+ FastPFor OPTPFor/Simple16, restated)
//   varint_G8IU_block::decode        block_codecs.hpp:239-258,287-314
//   qmx_block::decode                block_codecs.hpp:336-349, qmx_codec.hpp:636-6115
//   mixed_block::decode              mixed_block.hpp:198-217
//
// Value layout in registers ("layout A"): value i of a block lives in lane (i & 63),
// slot (i >> 6); a lane therefore returns two values v0 (index lane) and v1 (index lane+64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "abi_structs.hpp"

namespace ds2i_dev {

#define DS2I_DEV __device__ __forceinline__

enum { CODEC_OPTPFOR = 0, CODEC_VARINT = 1, CODEC_INTERPOLATIVE = 2, CODEC_QMX = 3, CODEC_MIXED = 4 };

static constexpr uint32_t STAGE_DW = 128; // staging window, dwords (512 B); larger blocks fall back to global reads
static constexpr uint32_t EXC_DW = 256;   // Simple16 exception scratch / generic out scratch
// The exception scratch is followed by a small constant table (u16 entries) built once per workgroup by
// s16_table_init(): [sel * 28 + k] = (low bit of field k) | (width << 8) for the 16 Simple16 layouts, then
// [448 + sel] = number of fields of layout sel.
static constexpr uint32_t S16_TAB_DW = 232;
static constexpr uint32_t EXC_LDS_DW = EXC_DW + S16_TAB_DW;

DS2I_DEV uint32_t lane_id() { return threadIdx.x & 63u; }

// All kernels run wave-uniform control flow with one wave per LDS region, so a
// wave-level fence is the only synchronisation ever needed between LDS phases.
DS2I_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

DS2I_DEV uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DS2I_DEV uint32_t bcast(uint32_t v, uint32_t src_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane); }
DS2I_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ---- wave64 inclusive prefix sum, DPP (row_shr 1/2/4/8 + row_bcast15 + row_bcast31)
template <int CTRL, int ROW_MASK>
DS2I_DEV uint32_t dpp_add(uint32_t x) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
    return x + t;
}
DS2I_DEV uint32_t wave_incl_scan(uint32_t x) {
    x = dpp_add<0x111, 0xF>(x); // row_shr:1
    x = dpp_add<0x112, 0xF>(x); // row_shr:2
    x = dpp_add<0x114, 0xF>(x); // row_shr:4
    x = dpp_add<0x118, 0xF>(x); // row_shr:8
    x = dpp_add<0x142, 0xA>(x); // row_bcast:15 -> rows 1,3
    x = dpp_add<0x143, 0xC>(x); // row_bcast:31 -> rows 2,3
    asm volatile("" : "+v"(x)); // opaque: otherwise "scan(x) - x" is re-associated into a second, unfused DPP chain
    return x;
}

template <int CTRL, int ROW_MASK>
DS2I_DEV uint32_t dpp_max(uint32_t x) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
    return x > t ? x : t;
}
// wave64 inclusive prefix maximum (unsigned), same DPP pattern as the prefix sum
DS2I_DEV uint32_t wave_incl_max_scan(uint32_t x) {
    x = dpp_max<0x111, 0xF>(x);
    x = dpp_max<0x112, 0xF>(x);
    x = dpp_max<0x114, 0xF>(x);
    x = dpp_max<0x118, 0xF>(x);
    x = dpp_max<0x142, 0xA>(x);
    x = dpp_max<0x143, 0xC>(x);
    return x;
}

template <int CTRL, int ROW_MASK>
DS2I_DEV uint32_t dpp_min(uint32_t x) {
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, ROW_MASK, 0xF, false);
    return x < t ? x : t;
}
// wave64 inclusive prefix minimum (unsigned)
DS2I_DEV uint32_t wave_incl_min_scan(uint32_t x) {
    x = dpp_min<0x111, 0xF>(x);
    x = dpp_min<0x112, 0xF>(x);
    x = dpp_min<0x114, 0xF>(x);
    x = dpp_min<0x118, 0xF>(x);
    x = dpp_min<0x142, 0xA>(x);
    x = dpp_min<0x143, 0xC>(x);
    return x;
}

// ---- unaligned global loads (lists are byte-aligned on disk; gfx950 under HSA runs
// in unaligned-access mode, the compiler emits one global_load_dword per memcpy)
DS2I_DEV uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
DS2I_DEV uint32_t ld8(const uint8_t* p) { return *p; }

// ---- staging window: a run of aligned dwords of the arena cached in LDS
struct Window {
    const uint8_t* gbase; // 4-byte aligned global address of st[0]
    uint32_t nbytes;      // staged bytes from gbase
    uint32_t* st;         // LDS, STAGE_DW dwords

    DS2I_DEV void load(const uint8_t* p, uint32_t want_bytes) {
        uintptr_t a = (uintptr_t)p;
        gbase = (const uint8_t*)(a & ~(uintptr_t)3);
        uint32_t sh = (uint32_t)(a & 3);
        uint32_t ndw = (sh + want_bytes + 3) >> 2;
        if (ndw > STAGE_DW) ndw = STAGE_DW;
        if (ndw == 0) ndw = 1;
        nbytes = ndw * 4;
        const uint32_t* g = (const uint32_t*)gbase;
        // branch-free: lanes past the end re-load (and re-store) the last dword instead of being masked off
        const uint32_t last = ndw - 1;
        for (uint32_t i0 = 0; i0 < ndw; i0 += 64) {
            uint32_t i = i0 + lane_id();
            i = i < last ? i : last;
            st[i] = g[i];
        }
        wave_sync();
    }
    DS2I_DEV bool covers(const uint8_t* p, uint32_t len) const {
        return p >= gbase && (uint32_t)(p - gbase) + len <= nbytes;
    }
    // 32 bits at arbitrary byte address
    DS2I_DEV uint32_t rd32(const uint8_t* p) const {
        uint32_t off = (uint32_t)(p - gbase);
        if (p >= gbase && off + 8 <= nbytes) {
            uint32_t w = off >> 2, r = off & 3;
            return __builtin_amdgcn_alignbyte(st[w + 1], st[w], r);
        }
        return ld32(p);
    }
    // 64 bits at arbitrary byte address
    DS2I_DEV uint64_t rd64(const uint8_t* p) const {
        uint32_t off = (uint32_t)(p - gbase);
        uint32_t lo, hi;
        if (p >= gbase && off + 12 <= nbytes) {
            uint32_t w = off >> 2, r = off & 3;
            uint32_t d0 = st[w], d1 = st[w + 1], d2 = st[w + 2];
            lo = __builtin_amdgcn_alignbyte(d1, d0, r);
            hi = __builtin_amdgcn_alignbyte(d2, d1, r);
        } else {
            lo = ld32(p);
            hi = ld32(p + 4);
        }
        return ((uint64_t)hi << 32) | lo;
    }
    DS2I_DEV uint32_t rd8(const uint8_t* p) const {
        uint32_t off = (uint32_t)(p - gbase);
        if (p >= gbase && off < nbytes) return (st[off >> 2] >> ((off & 3) * 8)) & 0xFF;
        return ld8(p);
    }
};

// ---- vbyte (terminator has bit 7 set); wave-uniform, returns bytes consumed
DS2I_DEV uint32_t vbyte_decode(const Window& w, const uint8_t* p, uint32_t& val) {
    uint32_t v = 0, shift = 0, i = 0;
    for (;;) {
        uint32_t c = uniform(w.rd8(p + i));
        ++i;
        v += (c & 127u) << shift;
        if ((c & 128u) || i == 5) break;
        shift += 7;
    }
    val = v;
    return i;
}

// ---- binary interpolative (serial bit stream; executed wave-uniformly, lane 0 stores)
struct BitReader {
    const Window* w;
    const uint8_t* in;
    uint64_t buf;
    uint32_t avail, pos;
    DS2I_DEV uint32_t read(uint32_t len) {
        if (!len) return 0;
        if (avail < len) {
            buf |= (uint64_t)uniform(w->rd32(in)) << avail;
            in += 4;
            avail += 32;
        }
        uint32_t val = (uint32_t)(buf & ((uint64_t(1) << len) - 1));
        buf >>= len;
        avail -= len;
        pos += len;
        return val;
    }
    DS2I_DEV uint32_t read_int(uint32_t u) { // u > 0
        uint32_t b = 31u - (uint32_t)__builtin_clz(u);
        uint64_t m = (uint64_t(1) << (b + 1)) - u;
        uint32_t val = read(b);
        if (val >= m) val = (val << 1) + read(1) - (uint32_t)m;
        return val;
    }
};

// Decodes n (<=128) values into out[] (LDS, index order) as PREFIX SUMS P_i
// (P_{n-1} = sum). Returns bytes consumed. The bit stream is inherently serial
// (interpolative_coding.hpp:124-146), so lane 0 runs it alone. `stk` (>= 24 dwords of LDS) is the
// explicit pre-order stack of pending right children (packed offset<<8|count, low, high); bounds
// stay in registers while walking down a left spine, so LDS is touched once per spine, not per value.
DS2I_DEV uint32_t interpolative_decode_prefix(const Window& w, const uint8_t* p, uint32_t sum, uint32_t n,
                                              uint32_t* out, uint32_t* stk) {
    uint32_t consumed = 0;
    const uint32_t lane = lane_id();
    // A subtree whose bounds coincide (low == high) costs the stream no bits and all its prefix sums equal the bound
    // (read_int(1) reads nothing, interpolative_coding.hpp:109-122): lane 0 skips it in O(1) and leaves its slots
    // "undefined"; afterwards every undefined slot takes the next defined value to its right (prefix sums are
    // non-decreasing, so that is a suffix minimum). Dense runs -- the blocks this codec is chosen for -- collapse to
    // a handful of serial steps.
    out[lane] = 0xFFFFFFFFu;
    out[lane + 64] = 0xFFFFFFFFu;
    wave_sync();
    if (lane == 0) {
        if (sum == 0xFFFFFFFFu) {
            consumed = vbyte_decode(w, p, sum);
            p += consumed;
        }
        out[n - 1] = sum;
        if (n > 1 && sum != 0) {
            BitReader br{&w, p, 0, 0, 0};
            int sp = 0;
            uint32_t o = 0, c = n - 1, low = 0, high = sum;
            for (;;) {
                while (c > 0 && high != low) {
                    uint32_t h = c >> 1;
                    uint32_t val = low + br.read_int(high - low + 1);
                    out[o + h] = val;
                    uint32_t rc = c - h - 1;
                    if (rc && high != val) {
                        stk[3 * sp] = ((o + h + 1) << 8) | rc;
                        stk[3 * sp + 1] = val;
                        stk[3 * sp + 2] = high;
                        ++sp;
                    }
                    c = h;
                    high = val;
                }
                if (!sp) break;
                --sp;
                uint32_t e = stk[3 * sp];
                low = stk[3 * sp + 1];
                high = stk[3 * sp + 2];
                o = e >> 8;
                c = e & 0xFFu;
            }
            consumed += (br.pos + 7) >> 3;
        }
    }
    wave_sync();
    // suffix minimum over out[0..127]: lane L scans element 127-L (upper half) and 63-L (lower half)
    uint32_t a = out[127u - lane], b = out[63u - lane];
    a = wave_incl_min_scan(a);
    b = wave_incl_min_scan(b);
    const uint32_t upper_min = bcast(a, 63);
    b = b < upper_min ? b : upper_min;
    wave_sync();
    out[127u - lane] = a;
    out[63u - lane] = b;
    wave_sync();
    return bcast(consumed, 0);
}

// Simple16 layout descriptors: three (count,width) runs packed 5 bits each: c0 | w0<<5 | c1<<10 | ...
#define S16D(c0, w0, c1, w1, c2, w2) ((c0) | ((w0) << 5) | ((c1) << 10) | ((w1) << 15) | ((c2) << 20) | ((w2) << 25))
__device__ static const uint32_t S16_DESC[16] = {
    S16D(28, 1, 0, 0, 0, 0), S16D(7, 2, 14, 1, 0, 0), S16D(7, 1, 7, 2, 7, 1), S16D(14, 1, 7, 2, 0, 0),
    S16D(14, 2, 0, 0, 0, 0), S16D(1, 4, 8, 3, 0, 0),  S16D(1, 3, 4, 4, 3, 3), S16D(7, 4, 0, 0, 0, 0),
    S16D(4, 5, 2, 4, 0, 0),  S16D(2, 4, 4, 5, 0, 0),  S16D(3, 6, 2, 5, 0, 0), S16D(2, 5, 3, 6, 0, 0),
    S16D(4, 7, 0, 0, 0, 0),  S16D(1, 10, 2, 9, 0, 0), S16D(2, 14, 0, 0, 0, 0), S16D(1, 28, 0, 0, 0, 0)};

// value k (0-based) of a Simple16 word with (wave-uniform) descriptor d: values fill the 28 payload
// bits from the high end (FastPFor unpack order).
DS2I_DEV uint32_t s16_value(uint32_t word, uint32_t d, uint32_t k) {
    const uint32_t c0 = d & 31, w0 = (d >> 5) & 31, c1 = (d >> 10) & 31, w1 = (d >> 15) & 31, w2 = d >> 25;
    uint32_t wdt, endbit;
    if (k < c0) { wdt = w0; endbit = (k + 1) * w0; }
    else if (k < c0 + c1) { wdt = w1; endbit = c0 * w0 + (k + 1 - c0) * w1; }
    else { wdt = w2; endbit = c0 * w0 + c1 * w1 + (k + 1 - c0 - c1) * w2; }
    return (word >> (28u - endbit)) & ((1u << wdt) - 1u);
}

DS2I_DEV uint16_t* s16_tab(uint32_t* exc) { return (uint16_t*)(exc + EXC_DW); }
DS2I_DEV void s16_table_init(uint32_t* exc) {
    uint16_t* t = s16_tab(exc);
    for (uint32_t i = lane_id(); i < 464u; i += 64u) {
        if (i < 448u) {
            const uint32_t sel = i / 28u, k = i - sel * 28u;
            const uint32_t d = S16_DESC[sel];
            const uint32_t c0 = d & 31, w0 = (d >> 5) & 31, c1 = (d >> 10) & 31, w1 = (d >> 15) & 31, c2 = (d >> 20) & 31, w2 = d >> 25;
            uint32_t wdt = 0, endbit = 0;
            if (k < c0) { wdt = w0; endbit = (k + 1) * w0; }
            else if (k < c0 + c1) { wdt = w1; endbit = c0 * w0 + (k + 1 - c0) * w1; }
            else if (k < c0 + c1 + c2) { wdt = w2; endbit = c0 * w0 + c1 * w1 + (k + 1 - c0 - c1) * w2; }
            t[i] = (uint16_t)((28u - endbit) | (wdt << 8)); // width 0 -> value 0 for fields a layout does not have
        } else {
            const uint32_t d = S16_DESC[i - 448u];
            t[i] = (uint16_t)((d & 31) + ((d >> 10) & 31) + ((d >> 20) & 31));
        }
    }
    wave_sync();
}

// ---- OptPFor full block (128 values). Returns bytes consumed; values in v0/v1 (layout A).
// `exc` (EXC_DW dwords) and `out` (128 dwords) are LDS scratch.
// Fast path: the block is dword aligned and lies inside the staging window (always the case for a
// block_optpfor index, whose lists are re-based at upload so that every full block is aligned):
// direct LDS dword indexing and v_alignbit extraction. Exceptions: one wave-uniform pass over the
// Simple16 words, one lane per VALUE of a word; positions by wave prefix sum.
DS2I_DEV uint32_t optpfor_decode(const Window& w, const uint8_t* p, uint32_t* exc, uint32_t* out, uint32_t& v0,
                                 uint32_t& v1) {
    const uint32_t lane = lane_id();
    const uint32_t hdr = uniform(w.rd32(p));
    const uint32_t b = hdr >> 26;
    uint32_t nexc = (hdr >> 16) & 0x3FFu;
    const uint32_t ew = hdr & 0xFFFFu;
    if (b >= 32) {
        const uint8_t* data = p + 4;
        v0 = w.rd32(data + 4 * lane);
        v1 = w.rd32(data + 4 * (lane + 64));
        return 4 * (1 + 128);
    }
    if (nexc > 128) nexc = 128; // corrupt header: stay inside the scratch arrays
    const uint32_t total = 4 * (1 + ew + 4 * b);
    const bool fast = (((uintptr_t)p & 3) == 0) && w.covers(p, total + 4);
    const uint32_t* blk = w.st + ((uint32_t)(p - w.gbase) >> 2); // only dereferenced when fast
    if (b == 0) {
        v0 = v1 = 0;
    } else {
        const uint32_t mask = (1u << b) - 1u;
        const uint32_t bit0 = lane * b, bit1 = bit0 + 64 * b;
        if (fast) {
            const uint32_t* data = blk + 1 + ew;
            const uint32_t i0 = bit0 >> 5, i1 = bit1 >> 5;
            v0 = __builtin_amdgcn_alignbit(data[i0 + 1], data[i0], bit0 & 31) & mask;
            v1 = __builtin_amdgcn_alignbit(data[i1 + 1], data[i1], bit1 & 31) & mask;
        } else {
            const uint8_t* data = p + 4 + 4 * ew;
            uint64_t x0 = w.rd64(data + 4 * (bit0 >> 5));
            uint64_t x1 = w.rd64(data + 4 * (bit1 >> 5));
            v0 = (uint32_t)(x0 >> (bit0 & 31)) & mask;
            v1 = (uint32_t)(x1 >> (bit1 & 31)) & mask;
        }
    }
    if (nexc && nexc <= 32 && ew <= 64) {
        // Common case (<= 32 exceptions, i.e. <= 64 Simple16 fields), kept in registers: one lane per Simple16 WORD
        // computes (count, offset); the start offsets are turned into a 64-bit mask through LDS, so the lane of field g
        // finds its word by a masked bit count; field geometry comes from the per-workgroup table. Lanes [0, nexc)
        // then hold the position deltas and lanes [nexc, 2 nexc) the high parts.
        const uint16_t* tab = s16_tab(exc);
        const uint32_t word = lane < ew ? (fast ? blk[1 + lane] : w.rd32(p + 4 + 4 * lane)) : 0u;
        const uint32_t cnt = lane < ew ? (uint32_t)tab[448u + (word >> 28)] : 0u;
        const uint32_t off = wave_incl_scan(cnt) - cnt; // index of my word's first field
        out[lane] = 0;
        if (lane < ew && off < 64u) out[off] = 1u; // words hold >= 1 field: starts are distinct
        wave_sync();
        const bool starts_here = out[lane] != 0u;
        const uint64_t starts = ballot(starts_here);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u));
        const uint32_t widx = below - (starts_here ? 0u : 1u); // word 0 starts at field 0, so widx >= 0
        const uint32_t wword = (uint32_t)__shfl((int)word, (int)(widx & 63u));
        const uint32_t woff = (uint32_t)__shfl((int)off, (int)(widx & 63u));
        const uint32_t k = lane - woff;
        const uint32_t fe = tab[(lane < 2 * nexc && k < 28u) ? (wword >> 28) * 28u + k : 0u];
        const uint32_t val = __builtin_amdgcn_ubfe(wword, fe & 0xFFu, fe >> 8);
        const uint32_t hi = (uint32_t)__shfl((int)val, (int)((lane + nexc) & 63u));
        const uint32_t lpos = wave_incl_scan(lane < nexc ? val + 1u : 0u) - 1u; // positions are delta coded
        wave_sync();
        out[lane] = 0;
        out[lane + 64] = 0;
        if (lane < nexc && lpos < 128u) out[lpos] = hi + 1u;
        wave_sync();
        v0 |= out[lane] << b;
        v1 |= out[lane + 64] << b;
        wave_sync();
    } else if (nexc && nexc <= 64 && ew <= 64) {
        // 33..64 exceptions (65..128 fields): the same scheme with two fields per lane (g and g + 64). The position
        // deltas are fields [0, nexc), all in the first slot; the high part of exception e is field e + nexc.
        const uint16_t* tab = s16_tab(exc);
        const uint32_t word = lane < ew ? (fast ? blk[1 + lane] : w.rd32(p + 4 + 4 * lane)) : 0u;
        const uint32_t cnt = lane < ew ? (uint32_t)tab[448u + (word >> 28)] : 0u;
        const uint32_t off = wave_incl_scan(cnt) - cnt;
        out[lane] = 0;
        out[lane + 64] = 0;
        if (lane < ew && off < 128u) out[off] = 1u;
        wave_sync();
        const bool st0 = out[lane] != 0u, st1 = out[lane + 64] != 0u;
        const uint64_t m0 = ballot(st0), m1 = ballot(st1);
        const uint32_t below0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
        const uint32_t below1 = (uint32_t)__builtin_popcountll(m0) +
                                __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
        const uint32_t widx0 = below0 - (st0 ? 0u : 1u), widx1 = below1 - (st1 ? 0u : 1u);
        const uint32_t wword0 = (uint32_t)__shfl((int)word, (int)(widx0 & 63u)), woff0 = (uint32_t)__shfl((int)off, (int)(widx0 & 63u));
        const uint32_t wword1 = (uint32_t)__shfl((int)word, (int)(widx1 & 63u)), woff1 = (uint32_t)__shfl((int)off, (int)(widx1 & 63u));
        const uint32_t k0 = lane - woff0, k1 = lane + 64u - woff1;
        const uint32_t fe0 = tab[k0 < 28u ? (wword0 >> 28) * 28u + k0 : 0u];
        const uint32_t fe1 = tab[(lane + 64u < 2 * nexc && k1 < 28u) ? (wword1 >> 28) * 28u + k1 : 0u];
        const uint32_t val0 = __builtin_amdgcn_ubfe(wword0, fe0 & 0xFFu, fe0 >> 8);
        const uint32_t val1 = __builtin_amdgcn_ubfe(wword1, fe1 & 0xFFu, fe1 >> 8);
        const uint32_t hidx = lane + nexc; // field holding the high part of exception `lane`
        const uint32_t h0 = (uint32_t)__shfl((int)val0, (int)(hidx & 63u)), h1 = (uint32_t)__shfl((int)val1, (int)(hidx & 63u));
        const uint32_t hi = hidx < 64u ? h0 : h1;
        const uint32_t lpos = wave_incl_scan(lane < nexc ? val0 + 1u : 0u) - 1u;
        wave_sync();
        out[lane] = 0;
        out[lane + 64] = 0;
        if (lane < nexc && lpos < 128u) out[lpos] = hi + 1u;
        wave_sync();
        v0 |= out[lane] << b;
        v1 |= out[lane + 64] << b;
        wave_sync();
    } else if (nexc) {
        const uint32_t need = 2 * nexc;
        if (ew <= 64) {
            // general form of the above for > 64 fields: batches of 64 through the exc[] scratch
            const uint32_t word = lane < ew ? (fast ? blk[1 + lane] : w.rd32(p + 4 + 4 * lane)) : 0u;
            const uint32_t d = S16_DESC[word >> 28];
            const uint32_t cnt = lane < ew ? (d & 31) + ((d >> 10) & 31) + ((d >> 20) & 31) : 0u;
            const uint32_t off = wave_incl_scan(cnt) - cnt; // first value index of my word
            for (uint32_t r0 = 0; r0 < need; r0 += 64) {
                out[lane] = 0;
                if (lane < ew && off - r0 < 64u) out[off - r0] = lane + 1; // words hold >= 1 value: starts are distinct
                const uint32_t carry = (uint32_t)__builtin_popcountll(ballot(lane < ew && off < r0)); // words begun before r0
                wave_sync();
                uint32_t mk = out[lane];
                wave_sync();
                mk = wave_incl_max_scan(mk);
                const uint32_t widx = (mk ? mk : carry) - 1u; // word holding value g = r0 + lane
                const uint32_t g = r0 + lane;
                const uint32_t wword = (uint32_t)__shfl((int)word, (int)(widx & 63u));
                const uint32_t wd = (uint32_t)__shfl((int)d, (int)(widx & 63u));
                const uint32_t woff = (uint32_t)__shfl((int)off, (int)(widx & 63u));
                if (g < need && g < EXC_DW) exc[g] = s16_value(wword, wd, g - woff);
            }
        } else {
            uint32_t off = 0;
            for (uint32_t j = 0; j < ew && off < need; ++j) {
                const uint32_t word = uniform(fast ? blk[1 + j] : w.rd32(p + 4 + 4 * j));
                const uint32_t d = S16_DESC[word >> 28];
                const uint32_t cnt = (d & 31) + ((d >> 10) & 31) + ((d >> 20) & 31);
                if (lane < cnt && off + lane < EXC_DW) exc[off + lane] = s16_value(word, d, lane);
                off += cnt;
            }
        }
        out[lane] = v0;
        out[lane + 64] = v1;
        wave_sync();
        // positions are delta coded: lpos_e = sum_{j<=e}(exc[j]+1) - 1
        uint32_t carry = 0;
        for (uint32_t base = 0; base < nexc; base += 64) {
            const uint32_t e = base + lane;
            const uint32_t dlt = (e < nexc) ? exc[e] + 1 : 0;
            const uint32_t incl = wave_incl_scan(dlt);
            const uint32_t lpos = carry + incl - 1;
            if (e < nexc && lpos < 128) out[lpos] |= (exc[e + nexc] + 1) << b;
            carry += bcast(incl, 63);
        }
        wave_sync();
        v0 = out[lane];
        v1 = out[lane + 64];
        wave_sync();
    }
    return total;
}

// ---- OptPFor full block that lies ENTIRELY in LDS (blk = its header dword, avail_dw = dwords staged from there on).
// Same value layout and result as optpfor_decode; differs in what it never does: touch global memory. A decode whose
// outputs may come from a global load (the unstaged fallbacks of optpfor_decode) makes the compiler drain vmcnt where
// its paths join -- and with it every unrelated load the caller has in flight (next block's bytes, range-table gathers).
// Serves b < 32 with <= 64 exceptions in <= 64 Simple16 words; returns false (nothing decoded) for anything else or
// when the block is not covered, and the caller takes the general path.
DS2I_DEV bool optpfor_decode_lds(const uint32_t* blk, uint32_t avail_dw, uint32_t* exc, uint32_t* out, uint32_t& v0, uint32_t& v1,
                                 uint32_t& consumed) {
    const uint32_t lane = lane_id();
    const uint32_t hdr = uniform(blk[0]);
    const uint32_t b = hdr >> 26, nexc = (hdr >> 16) & 0x3FFu, ew = hdr & 0xFFFFu;
    const uint32_t total_dw = 1u + ew + 4u * b;
    if (__builtin_expect(b >= 32u || nexc > 64u || ew > 64u || total_dw + 1u > avail_dw, 0)) return false;
    consumed = 4u * total_dw;
    const uint32_t* data = blk + 1 + ew;
    const uint32_t mask = (1u << b) - 1u; // b == 0: mask 0, every value 0
    const uint32_t bit0 = lane * b, bit1 = bit0 + 64u * b;
    const uint32_t i0 = bit0 >> 5, i1 = bit1 >> 5;
    v0 = __builtin_amdgcn_alignbit(data[i0 + 1], data[i0], bit0 & 31u) & mask;
    v1 = __builtin_amdgcn_alignbit(data[i1 + 1], data[i1], bit1 & 31u) & mask;
    if (!nexc) return true;
    const uint16_t* tab = s16_tab(exc);
    const uint32_t word = lane < ew ? blk[1 + lane] : 0u;
    const uint32_t cnt = lane < ew ? (uint32_t)tab[448u + (word >> 28)] : 0u;
    const uint32_t off = wave_incl_scan(cnt) - cnt; // index of my word's first field
    uint32_t hi, lpos;
    if (__builtin_expect(nexc <= 32u, 1)) { // <= 64 fields: one per lane
        out[lane] = 0; // (LDS operations of one wave are performed in issue order: no fence between the clear and the marks)
        if (lane < ew && off < 64u) out[off] = 1u; // words hold >= 1 field: starts are distinct
        wave_sync();
        const bool starts_here = out[lane] != 0u;
        const uint64_t starts = ballot(starts_here);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u));
        const uint32_t widx = below - (starts_here ? 0u : 1u);
        const uint32_t wword = (uint32_t)__shfl((int)word, (int)(widx & 63u));
        const uint32_t woff = (uint32_t)__shfl((int)off, (int)(widx & 63u));
        const uint32_t k = lane - woff;
        const uint32_t fe = tab[(lane < 2 * nexc && k < 28u) ? (wword >> 28) * 28u + k : 0u];
        const uint32_t val = __builtin_amdgcn_ubfe(wword, fe & 0xFFu, fe >> 8);
        hi = (uint32_t)__shfl((int)val, (int)((lane + nexc) & 63u));
        lpos = wave_incl_scan(lane < nexc ? val + 1u : 0u) - 1u; // positions are delta coded
    } else { // 33..64 exceptions: two fields per lane (g and g + 64)
        out[lane] = 0;
        out[lane + 64] = 0;
        if (lane < ew && off < 128u) out[off] = 1u;
        wave_sync();
        const bool st0 = out[lane] != 0u, st1 = out[lane + 64] != 0u;
        const uint64_t m0 = ballot(st0), m1 = ballot(st1);
        const uint32_t below0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
        const uint32_t below1 = (uint32_t)__builtin_popcountll(m0) +
                                __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
        const uint32_t widx0 = below0 - (st0 ? 0u : 1u), widx1 = below1 - (st1 ? 0u : 1u);
        const uint32_t wword0 = (uint32_t)__shfl((int)word, (int)(widx0 & 63u)), woff0 = (uint32_t)__shfl((int)off, (int)(widx0 & 63u));
        const uint32_t wword1 = (uint32_t)__shfl((int)word, (int)(widx1 & 63u)), woff1 = (uint32_t)__shfl((int)off, (int)(widx1 & 63u));
        const uint32_t k0 = lane - woff0, k1 = lane + 64u - woff1;
        const uint32_t fe0 = tab[k0 < 28u ? (wword0 >> 28) * 28u + k0 : 0u];
        const uint32_t fe1 = tab[(lane + 64u < 2 * nexc && k1 < 28u) ? (wword1 >> 28) * 28u + k1 : 0u];
        const uint32_t val0 = __builtin_amdgcn_ubfe(wword0, fe0 & 0xFFu, fe0 >> 8);
        const uint32_t val1 = __builtin_amdgcn_ubfe(wword1, fe1 & 0xFFu, fe1 >> 8);
        const uint32_t hidx = lane + nexc; // field holding the high part of exception `lane`
        const uint32_t h0 = (uint32_t)__shfl((int)val0, (int)(hidx & 63u)), h1 = (uint32_t)__shfl((int)val1, (int)(hidx & 63u));
        hi = hidx < 64u ? h0 : h1;
        lpos = wave_incl_scan(lane < nexc ? val0 + 1u : 0u) - 1u;
    }
    wave_sync();
    out[lane] = 0;
    out[lane + 64] = 0;
    if (lane < nexc && lpos < 128u) out[lpos] = hi + 1u;
    wave_sync();
    v0 |= out[lane] << b;
    v1 |= out[lane + 64] << b;
    wave_sync();
    return true;
}

// ---- OptPFor with the upload-time EXCEPTION SIDE SLOTS (block_optpfor indexes; abi_structs.hpp, BatchArgs::xslots).
// The on-disk block keeps its exceptions as two Simple16 streams (position deltas, high parts: block_codecs.hpp:210-226
// via FastPFor) -- on a wave that is ~8 dependent LDS round trips and half of a decode's instructions. At upload every
// full block of the index gets one 64-dword slot that holds the same information in the form a wave wants:
//   dwords 0..3   docs part:  128-bit mask of the positions that carry an exception (bit i = value i)
//   dwords 4..7   freqs part: the same
//   dword  8, 9   copies of the two parts' header dwords (b << 26 | exceptions << 16 | Simple16 words)
//   dword  10     0 in the common case (see XSLOT_FLAG), else XSLOT_SLOW | (1 + offset of the block's adds in the overflow area)
//   dwords 11..63 the "adds" = (high part + 1) << b of every exception, docs part first (in position order), then freqs part
// so that value i = low bits | (mask bit i ? adds[first + popcount(mask below i)] : 0). Everything a decode branches or
// computes addresses on sits at fixed places of the slot: a wave reads it in ONE LDS round trip, the packed low bits and the
// adds of BOTH parts in a second one, and the common case has no branch at all.

// the exceptions of one part: m = its mask (the same four dwords in every lane), XRD(i) = add i, first = index of the part's
// first add
template <class XRD>
DS2I_DEV void optpfor_apply_adds(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3, uint32_t first, XRD xrd, uint32_t& v0, uint32_t& v1) {
    const uint32_t lane = lane_id();
    const bool has0 = ((lane < 32u ? m0 >> lane : m1 >> (lane - 32u)) & 1u) != 0u;
    const bool has1 = ((lane < 32u ? m2 >> lane : m3 >> (lane - 32u)) & 1u) != 0u;
    const uint32_t r0 = __builtin_amdgcn_mbcnt_hi(m1, __builtin_amdgcn_mbcnt_lo(m0, first));
    const uint32_t r1 = __builtin_amdgcn_mbcnt_hi(m3, __builtin_amdgcn_mbcnt_lo(m2, first + (uint32_t)__builtin_popcount(m0) + (uint32_t)__builtin_popcount(m1)));
    const uint32_t a0 = xrd(has0 ? r0 : first), a1 = xrd(has1 ? r1 : first);
    v0 |= has0 ? a0 : 0u;
    v1 |= has1 ? a1 : 0u;
}
// the b-bit low parts of a full block (header dword hdr at dword 0 of the part; RD(i) = dword i of the part)
template <class RD>
DS2I_DEV void optpfor_low_bits(RD rd, uint32_t hdr, uint32_t& v0, uint32_t& v1) {
    const uint32_t lane = lane_id();
    const uint32_t b = hdr >> 26, ew = hdr & 0xFFFFu;
    const uint32_t mask = (1u << b) - 1u; // b == 0: mask 0, every value 0 (b < 32 here)
    const uint32_t bit0 = lane * b, bit1 = bit0 + 64u * b;
    const uint32_t i0 = 1u + ew + (bit0 >> 5), i1 = 1u + ew + (bit1 >> 5);
    v0 = __builtin_amdgcn_alignbit(rd(i0 + 1), rd(i0), bit0 & 31u) & mask;
    v1 = __builtin_amdgcn_alignbit(rd(i1 + 1), rd(i1), bit1 & 31u) & mask;
}

// The head of a staged slot: the two headers and the flag word, wave-uniform
struct SlotHead { uint32_t hd, hf, flag; };
DS2I_DEV SlotHead optpfor_slot_head(const uint32_t* slot) {
    const uint4 h = *(const uint4*)(slot + XSLOT_HDR);
    return SlotHead{uniform(h.x), uniform(h.y), uniform(h.z)};
}
// BOTH parts of a full block in the common case (h.flag == 0): st = the block's bytes (staged from its first dword on), slot = its
// staged side slot. No branch, no global memory; gaps-1 in (d0, d1), freqs-1 in (f0, f1), value i in lane i & 63, slot i >> 6.
// cons_d / cons_f = bytes of the two parts.
// (DOCS / FREQS: which parts the caller wants; the other one's outputs are left untouched)
template <bool DOCS = true, bool FREQS = true>
DS2I_DEV void optpfor_decode_pair(const uint32_t* st, const uint32_t* slot, const SlotHead& h, uint32_t& d0, uint32_t& d1, uint32_t& f0, uint32_t& f1,
                                  uint32_t& cons_d, uint32_t& cons_f) {
    const uint32_t nd = (h.hd >> 16) & 0x3FFu;
    const uint32_t tot_d = 1u + (h.hd & 0xFFFFu) + 4u * (h.hd >> 26);
    cons_d = 4u * tot_d;
    cons_f = 4u * (1u + (h.hf & 0xFFFFu) + 4u * (h.hf >> 26));
    // (a part without exceptions has an all-zero mask: its lanes read add `first` and discard it)
    if constexpr (DOCS) {
        const uint4 md = *(const uint4*)(slot);
        optpfor_low_bits([&](uint32_t i) { return st[i]; }, h.hd, d0, d1);
        optpfor_apply_adds(md.x, md.y, md.z, md.w, XSLOT_ADDS, [&](uint32_t i) { return slot[i & (XSLOT_DW - 1u)]; }, d0, d1);
    }
    if constexpr (FREQS) {
        const uint4 mf = *(const uint4*)(slot + 4);
        const uint32_t* const sf = st + tot_d;
        optpfor_low_bits([&](uint32_t i) { return sf[i]; }, h.hf, f0, f1);
        optpfor_apply_adds(mf.x, mf.y, mf.z, mf.w, XSLOT_ADDS + nd, [&](uint32_t i) { return slot[i & (XSLOT_DW - 1u)]; }, f0, f1);
    }
}

// One part (docs: part = 0, freqs: part = 1) of a full OptPFor block in EVERY case: its bytes staged in LDS from `blk` on
// (avail_dw dwords, 0 = not staged; blk -> the part's header) and its side slot staged at `slot`. `gpart` = the part's address
// in the arena and `xovf` = the overflow area, touched only by what the staging does not cover (a part beyond the staged
// bytes, a raw b = 32 block, adds in the overflow area); their loads are waited for before the function returns, so that
// nothing of it is "pending" for the compiler where the caller's paths join. nd = exceptions of the docs part (where the
// freqs part's adds start; ignored for part 0). Returns the bytes of the part.
DS2I_DEV uint32_t optpfor_decode_side(const uint32_t* blk, uint32_t avail_dw, const uint32_t* slot, const uint8_t* gpart, const uint32_t* xovf,
                                      uint32_t part, uint32_t nd, uint32_t& v0, uint32_t& v1, uint32_t* nexc_out = nullptr) {
    const uint32_t lane = lane_id();
    const uint32_t* const g = (const uint32_t*)gpart;
    const uint32_t hdr = uniform(slot[XSLOT_HDR + part]);
    const uint32_t b = hdr >> 26, nexc = (hdr >> 16) & 0x3FFu, ew = hdr & 0xFFFFu;
    if (nexc_out) *nexc_out = nexc;
    if (__builtin_expect(b >= 32u, 0)) { // raw block: 128 dwords behind the header
        uint32_t a0 = g[1 + lane], a1 = g[65 + lane];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");
        v0 = a0;
        v1 = a1;
        return 4u * 129u;
    }
    const uint32_t total_dw = 1u + ew + 4u * b;
    if (__builtin_expect(total_dw + 1u <= avail_dw, 1)) {
        optpfor_low_bits([&](uint32_t i) { return blk[i]; }, hdr, v0, v1);
    } else {
        uint32_t a0, a1;
        optpfor_low_bits([&](uint32_t i) { return g[i]; }, hdr, a0, a1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");
        v0 = a0;
        v1 = a1;
    }
    if (nexc) {
        const uint4 m = *(const uint4*)(slot + 4u * part);
        const uint32_t ovf = uniform(slot[XSLOT_FLAG]) & ~XSLOT_SLOW;
        const uint32_t first = part ? nd : 0u;
        if (__builtin_expect(ovf == 0u, 1)) {
            optpfor_apply_adds(m.x, m.y, m.z, m.w, XSLOT_ADDS + first, [&](uint32_t i) { return slot[i & (XSLOT_DW - 1u)]; }, v0, v1);
        } else {
            const uint32_t* const xo = xovf + (ovf - 1u);
            uint32_t a0 = 0, a1 = 0;
            optpfor_apply_adds(m.x, m.y, m.z, m.w, first, [&](uint32_t i) { return xo[i]; }, a0, a1);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");
            v0 |= a0;
            v1 |= a1;
        }
    }
    return 4u * total_dw;
}

// ---- VarInt-G8IU full block. Values are scattered to out[] (LDS) then re-read.
DS2I_DEV uint32_t varint_g8iu_decode(const Window& w, const uint8_t* p, uint32_t* out, uint32_t& v0, uint32_t& v1) {
    const uint32_t lane = lane_id();
    const uint8_t* g = p + 9 * lane;
    uint32_t desc = w.rd8(g);
    uint32_t cnt = (uint32_t)__builtin_popcount(~desc & 0xFFu);
    uint32_t incl = wave_incl_scan(cnt);
    uint32_t excl = incl - cnt;
    uint64_t active = ballot(excl < 128);
    uint32_t groups = (uint32_t)__builtin_popcountll(active);
    if (excl < 128) {
        uint64_t bytes = w.rd64(g + 1);
        uint32_t val = 0, k = 0, o = excl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            val |= (uint32_t)((bytes >> (8 * j)) & 0xFF) << (8 * k);
            ++k;
            if (!((desc >> j) & 1u)) {
                if (o < 128) out[o] = val;
                ++o;
                val = 0;
                k = 0;
            }
        }
    }
    wave_sync();
    v0 = out[lane];
    v1 = out[lane + 64];
    wave_sync();
    return 9 * groups;
}

// ---- QMX full block: vbyte(enc_len) | payload vectors | keys (reversed)
__device__ static const uint8_t QMX_BITS[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32};
__device__ static const uint16_t QMX_CAP[15] = {256, 128, 64, 40, 32, 24, 20, 36, 16, 28, 12, 20, 8, 12, 4};

DS2I_DEV uint32_t qmx_extract(const Window& w, const uint8_t* vec, uint32_t type, uint32_t j) {
    const uint32_t bits = QMX_BITS[type];
    if (type == 0) return 1u;
    if (type == 8) return w.rd8(vec + j);
    if (type == 12) return w.rd32(vec + 2 * j) & 0xFFFFu;
    if (type == 14) return w.rd32(vec + 4 * j);
    const uint32_t l = j & 3, r = j >> 2;
    const uint32_t mask = (1u << bits) - 1u;
    if (type != 7 && type != 9 && type != 11 && type != 13) return (w.rd32(vec + 4 * l) >> (r * bits)) & mask;
    const uint32_t R1 = 32u / bits;
    const uint32_t lowpart = 32u - R1 * bits;
    const uint32_t off2 = (bits == 12) ? 8u : (bits == 21) ? 11u : (bits - lowpart);
    if (r < R1) return (w.rd32(vec + 4 * l) >> (r * bits)) & mask;
    uint32_t second = w.rd32(vec + 16 + 4 * l);
    if (r == R1) return ((w.rd32(vec + 4 * l) >> (r * bits)) | (second << lowpart)) & mask;
    return (second >> ((r - R1 - 1) * bits + off2)) & mask;
}

DS2I_DEV uint32_t qmx_decode(Window& w, const uint8_t* p, uint32_t* out, uint32_t& v0, uint32_t& v1) {
    const uint32_t lane = lane_id();
    uint32_t enc_len;
    uint32_t vl = vbyte_decode(w, p, enc_len);
    const uint8_t* src = p + vl;
    if (!w.covers(src, enc_len + 16) && enc_len + 16 + 4 <= STAGE_DW * 4) w.load(src, enc_len + 16);
    uint32_t in = 0;
    int32_t kp = (int32_t)enc_len - 1;
    uint32_t outpos = 0;
    while ((int32_t)in <= kp) {
        uint32_t key = uniform(w.rd8(src + kp));
        --kp;
        uint32_t type = key >> 4, reps = 16u - (key & 15u);
        if (type == 15) { in += reps; continue; }
        const uint32_t cap = QMX_CAP[type];
        const uint32_t vbytes = (type == 0) ? 0u : (type == 7 || type == 9 || type == 11 || type == 13) ? 32u : 16u;
        for (uint32_t r = 0; r < reps; ++r) {
            if (outpos < 128) {
                for (uint32_t j = lane; j < cap; j += 64) {
                    uint32_t o = outpos + j;
                    if (o < 128) out[o] = qmx_extract(w, src + in, type, j);
                }
            }
            in += vbytes;
            outpos += cap;
        }
    }
    wave_sync();
    v0 = out[lane];
    v1 = out[lane + 64];
    wave_sync();
    return vl + enc_len;
}

// ---- generic block decode: n values (gap-1 / freq-1) -> v0,v1 (layout A; lanes >= n get 0).
// `out` = 128-dword LDS scratch (may be the destination buffer itself), `exc` = EXC_DW dwords.
// Returns bytes consumed.
// CODEC_T >= 0 fixes the codec at compile time (dead code of the other decoders disappears).
template <int CODEC_T>
DS2I_DEV uint32_t decode_block(int codec_rt, Window& w, const uint8_t* p, uint32_t sum, uint32_t n, uint32_t* out,
                               uint32_t* exc, uint32_t& v0, uint32_t& v1) {
    const uint32_t lane = lane_id();
    uint32_t consumed = 0;
    const int codec = CODEC_T >= 0 ? CODEC_T : codec_rt;
    int c = codec;
    if (n == 128 && codec == CODEC_MIXED) {
        uint32_t t = uniform(w.rd8(p));
        ++p;
        consumed = 1;
        c = (t == 0) ? CODEC_OPTPFOR : (t == 1) ? CODEC_VARINT : CODEC_INTERPOLATIVE;
    }
    if (n < 128 || c == CODEC_INTERPOLATIVE) {
        consumed += interpolative_decode_prefix(w, p, sum, n, out, exc);
        uint32_t a0 = (lane < n) ? out[lane] : 0, a1 = (lane + 64 < n) ? out[lane + 64] : 0;
        uint32_t b0 = (lane >= 1 && lane < n) ? out[lane - 1] : 0;
        uint32_t b1 = (lane + 64 < n) ? out[lane + 63] : 0;
        wave_sync();
        v0 = a0 - b0;
        v1 = a1 - b1;
        return consumed;
    }
    switch (c) {
    case CODEC_OPTPFOR: consumed += optpfor_decode(w, p, exc, out, v0, v1); break;
    case CODEC_VARINT: consumed += varint_g8iu_decode(w, p, out, v0, v1); break;
    default: consumed += qmx_decode(w, p, out, v0, v1); break;
    }
    return consumed;
}

} // namespace ds2i_dev
