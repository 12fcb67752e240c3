// or_query<with_freqs> on a block_optpfor index, the freqs half (gfx950 / CDNA4, wave64, no MFMA: integer work).
//
// or_query<true> (reference queries.hpp:88-131) walks the union of the lists and, for every list positioned on the current
// document, reads its freq (118-120): every posting of every list of the query is visited exactly once and its freq decoded.
// What the batch interface returns beside the union's size is a checksum of those freqs, their sum. The union itself has been
// a stream since round 3 (k_union: dense lists from their bitmaps, the others decoded once per block into an LDS bitmap) --
// but with the freqs it was bound by the latency of ONE block in flight per wave (47 k queries/s at GOV2 scale, 0.7 % of the
// roofline: a dependent round trip per block, 161 M blocks per batch). Which freqs are read does not depend on the union at
// all, so they are streamed here on their own: one wave per (query term, run of blocks), the run's bytes and exception side
// slots four blocks ahead by LDS-DMA with hand-counted waits, the freqs part decoded from the slot's copy of the headers
// (device_codecs.hpp, optpfor_decode_pair<false, true>) -- nothing but memory bandwidth and ~50 vector instructions per block.
// The union's size comes from k_union<false> as for `or`.
#include <hip/hip_runtime.h>

#include "device_enum.hpp"

using namespace ds2i_dev;

namespace {

constexpr int FS_RING = 4;   // blocks in flight per wave
constexpr uint32_t FS_RUN = 126; // blocks per wave (two windows of the skip table)
constexpr int FS_LOADS = 3;  // hand-issued loads per block (512 bytes of the block in two, its 256-byte side slot in one)

struct LdsFS {
    uint32_t stage[FS_RING][STAGE_DW];
    uint32_t xs[FS_RING][XSLOT_DW];
};

DS2I_DEV uint32_t fs_lds_offset(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// 512 bytes at g (4-byte aligned) -> LDS byte offset lds, 256 bytes at gx -> lds_x; voff = lane * 4 (ranked_stream.hip has the
// story of these statements: M0 is the DMA's LDS base and compiler-reserved, the instruction offset moves both addresses)
DS2I_DEV void fs_prefetch_blk(const uint8_t* g, uint32_t lds, const uint32_t* gx, uint32_t lds_x, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\tglobal_load_lds_dword %1, %2 offset:256\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(uniform(lds)), "s"(gx), "s"(uniform(lds_x)) : "memory");
}
template <int N> DS2I_DEV void fs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <class T> DS2I_DEV const T* fs_uniform_ptr(const T* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    return (const T*)(uintptr_t)(((unsigned long long)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v));
}

// grid = (runs of the longest list, query terms of the batch): wave (c, t) sums the freqs of blocks [c * FS_RUN, (c + 1) * FS_RUN) of term t's list
__global__ void __launch_bounds__(64, 8) k_freq_stream(FreqArgs a) {
    __shared__ LdsFS L;
    const uint32_t lane = lane_id();
    const QTerm* const qt = a.qterms + blockIdx.y;
    const uint32_t n = uniform(qt->n), nb = (n + 127u) >> 7;
    const uint32_t per = FS_RUN; // (a fixed run: a short list leaves most of its row of the grid idle, but every wave that works has a whole run)
    const uint32_t b0 = blockIdx.x * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    if (b0 >= b1) return;
    const uint32_t vl = 1u + (n >= (1u << 7)) + (n >= (1u << 14)) + (n >= (1u << 21)) + (n >= (1u << 28));
    const uint32_t bb = uniform(qt->blk_base);
    const unsigned long long list_off = ((unsigned long long)uniform((uint32_t)(qt->list_off >> 32)) << 32) | uniform((uint32_t)qt->list_off);
    const uint8_t* const data = a.arena + list_off + vl + 4ull * nb + 4ull * (nb - 1);
    const uint2* const tab = (const uint2*)a.skip + bb;
    const uint32_t* const xs0 = a.xslots + (size_t)XSLOT_DW * bb;
    const uint32_t nfull = n >> 7; // blocks [0, nfull) are full; block nfull (if n & 127) is the list's partial last block
    const uint32_t st_base = fs_lds_offset(&L.stage[0][0]), xs_base = fs_lds_offset(&L.xs[0][0]);
    const uint32_t voff = lane * 4u;
    unsigned long long acc = 0;
    const uint32_t e1 = b1 < nfull ? b1 : nfull; // full blocks of the run: [b0, e1)
    for (uint32_t w0 = b0; w0 < e1; w0 += 63u) {
        // rows w0 - 1 .. w0 + 62 of the list's skip table: lane j's row ends block first + j, i.e. starts block first + j + 1
        const uint32_t first = w0 ? w0 - 1u : 0u;
        const uint32_t ridx = first + lane;
        uint32_t rend = 0;
        if (ridx < nb) rend = tab[ridx].y;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rend)::"memory");
        const uint32_t cnt = (e1 - w0 < 63u) ? e1 - w0 : 63u;
        auto ep_of = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t { return b ? bcast(rend, b - 1u - first) : 0u; };
        auto issue = [&](uint32_t i) __attribute__((always_inline)) { // block w0 + i of the window into ring slot i % FS_RING
            const uint32_t b = w0 + i, slot = i & (FS_RING - 1);
            fs_prefetch_blk(fs_uniform_ptr(data + ep_of(b)), st_base + slot * (STAGE_DW * 4u), fs_uniform_ptr(xs0 + (size_t)XSLOT_DW * b), xs_base + slot * (XSLOT_DW * 4u), voff);
        };
        for (uint32_t i = 0; i < cnt && i < (uint32_t)FS_RING; ++i) issue(i);
        for (uint32_t i = 0; i < cnt; ++i) {
            // block i's loads are followed by those of the blocks already requested behind it
            const uint32_t behind = cnt - 1u - i < (uint32_t)(FS_RING - 1) ? cnt - 1u - i : (uint32_t)(FS_RING - 1);
            if (behind == 3u) fs_wait_vm<3 * FS_LOADS>();
            else if (behind == 2u) fs_wait_vm<2 * FS_LOADS>();
            else if (behind == 1u) fs_wait_vm<1 * FS_LOADS>();
            else fs_wait_vm<0>();
            const uint32_t slot = i & (FS_RING - 1);
            const uint32_t* const st = L.stage[slot];
            const uint32_t* const xs = L.xs[slot];
            const SlotHead h = optpfor_slot_head(xs);
            uint32_t d0 = 0, d1 = 0, f0, f1, cd, cf;
            if (__builtin_expect(h.flag == 0u, 1)) {
                optpfor_decode_pair<false, true>(st, xs, h, d0, d1, f0, f1, cd, cf);
            } else { // raw parts, parts beyond the staged bytes, adds in the overflow area: the general side-slot decoder
                const uint32_t hb = h.hd >> 26;
                const uint32_t docs_bytes = hb >= 32u ? 4u * 129u : 4u * (1u + (h.hd & 0xFFFFu) + 4u * hb);
                const uint32_t skip_dw = docs_bytes >> 2;
                optpfor_decode_side(st + (skip_dw < STAGE_DW ? skip_dw : 0u), skip_dw < STAGE_DW ? STAGE_DW - skip_dw : 0u, xs, data + ep_of(w0 + i) + docs_bytes, a.xovf, 1u,
                                    (h.hd >> 16) & 0x3FFu, f0, f1);
            }
            acc += (unsigned long long)(f0 + 1u) + (unsigned long long)(f1 + 1u);
            // (the slot is free again: every lane's reads of it are behind us -- the sums above depend on them)
            if (i + FS_RING < cnt) issue(i + FS_RING);
        }
    }
    if (b1 > nfull) { // the partial last block: plain freqs - 1 in the tail table (entry: sz gaps-1, sz freqs-1, two byte counts)
        const uint32_t sz = n & 127u;
        const uint32_t* const t = a.tails + (((unsigned long long)uniform((uint32_t)(qt->aux1 >> 32)) << 32) | uniform((uint32_t)qt->aux1));
        if (lane < sz) acc += (unsigned long long)t[sz + lane] + 1ull;
        if (lane + 64 < sz) acc += (unsigned long long)t[sz + lane + 64] + 1ull;
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && acc) atomicAdd(a.out_freq_sum + a.qterm_q[blockIdx.y], acc);
}

// ---------------------------------------------------------------- and / and_freq of all-dense queries
// and_query (reference queries.hpp:35-86: candidate = next posting of the shortest list, next_geq() on every other list) for
// queries whose lists all carry their exact bitmap (one document in 64 or denser: the queries that own most of the blocks).
// A posting of list i is a match iff its doc-id's bit is set in every other list's bitmap -- so the result count is a sum over
// the postings of the shortest list, and and_query<true>'s freqs of the matches are, list by list, a sum over that list's own
// postings. Every list that has to be read is a stream of its own (one wave per run of FS_RUN blocks, bytes + side slot four
// blocks ahead by LDS-DMA); the bit gathers of block i are issued when it is decoded and consumed behind the decode of block
// i + 1 (LDS-DMA as well: a compiler-issued load would drain the ring at every block).
constexpr int AS_MAXO = 3; // other lists per query (queries of up to 4 terms)
struct LdsAS {
    uint32_t stage[FS_RING][STAGE_DW];
    uint32_t xs[FS_RING][XSLOT_DW];
    uint32_t g[2][AS_MAXO][64]; // bitmap words of the block in flight: [half][other list][lane]
};
DS2I_DEV void as_gather_dword(const uint8_t* base, uint32_t byte_off, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(uniform(lds)) : "memory");
}
DS2I_DEV void as_wait(uint32_t n) { // s_waitcnt vmcnt(n) for a wave-uniform n <= 15
    switch (n) {
    case 0: fs_wait_vm<0>(); break;   case 1: fs_wait_vm<1>(); break;   case 2: fs_wait_vm<2>(); break;   case 3: fs_wait_vm<3>(); break;
    case 4: fs_wait_vm<4>(); break;   case 5: fs_wait_vm<5>(); break;   case 6: fs_wait_vm<6>(); break;   case 7: fs_wait_vm<7>(); break;
    case 8: fs_wait_vm<8>(); break;   case 9: fs_wait_vm<9>(); break;   case 10: fs_wait_vm<10>(); break; case 11: fs_wait_vm<11>(); break;
    case 12: fs_wait_vm<12>(); break; case 13: fs_wait_vm<13>(); break; case 14: fs_wait_vm<14>(); break; default: fs_wait_vm<15>(); break;
    }
}

template <bool WITH_FREQS>
__global__ void __launch_bounds__(64, 8) k_and_stream(AndStreamArgs a) {
    __shared__ LdsAS L;
    const uint32_t lane = lane_id();
    const StreamTerm* const st = a.terms + blockIdx.y;
    const uint32_t n = uniform(st->n), nb = (n + 127u) >> 7;
    const uint32_t b0 = blockIdx.x * FS_RUN, b1 = (b0 + FS_RUN < nb) ? b0 + FS_RUN : nb;
    if (b0 >= b1) return;
    const uint32_t vl = 1u + (n >= (1u << 7)) + (n >= (1u << 14)) + (n >= (1u << 21)) + (n >= (1u << 28));
    const uint32_t bb = uniform(st->blk_base), nother = uniform(st->nother), counts = uniform(st->counts);
    if (!WITH_FREQS && nother == 0u) {
        // and_query of ONE term (queries.hpp:58-84 walks the list and counts it): the answer is the list's length, which its header
        // holds -- nothing is decoded. (424 of the 4096 queries of the GOV2-scale batch; decoding their 5.3 M blocks to count them
        // was the longest kernel of the `and` step, 2.3 ms.) and_query<with_freqs> still streams the list: it needs every freq.
        if (b0 == 0u && lane == 0 && counts) atomicAdd(a.out_count + uniform(st->q), (unsigned long long)n);
        return;
    }
    const unsigned long long list_off = ((unsigned long long)uniform((uint32_t)(st->list_off >> 32)) << 32) | uniform((uint32_t)st->list_off);
    const uint8_t* const data = a.arena + list_off + vl + 4ull * nb + 4ull * (nb - 1);
    const uint2* const tab = (const uint2*)a.skip + bb;
    const uint32_t* const xs0 = a.xslots + (size_t)XSLOT_DW * bb;
    const uint8_t* bm[AS_MAXO];
#pragma unroll
    for (int j = 0; j < AS_MAXO; ++j) {
        const unsigned long long o = st->bm[j < (int)nother ? j : 0];
        bm[j] = a.rmw + (((unsigned long long)uniform((uint32_t)(o >> 32)) << 32) | uniform((uint32_t)o));
    }
    const uint32_t nfull = n >> 7;
    const uint32_t st_base = fs_lds_offset(&L.stage[0][0]), xs_base = fs_lds_offset(&L.xs[0][0]), g_base = fs_lds_offset(&L.g[0][0][0]);
    const uint32_t voff = lane * 4u;
    const uint32_t G = 2u * nother; // hand-issued gathers per block
    unsigned long long acc_f = 0;
    uint32_t acc_c = 0;
    // the postings of a decoded block against the words gathered for them
    auto consume = [&](uint32_t d0, uint32_t d1, uint32_t f0, uint32_t f1) __attribute__((always_inline)) {
        bool m0 = d0 != 0xFFFFFFFFu, m1 = d1 != 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < AS_MAXO; ++j) {
            if (j < (int)nother) {
                m0 = m0 && ((L.g[0][j][lane] >> (d0 & 31u)) & 1u);
                m1 = m1 && ((L.g[1][j][lane] >> (d1 & 31u)) & 1u);
            }
        }
        acc_c += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
        if constexpr (WITH_FREQS) acc_f += (unsigned long long)(m0 ? f0 : 0u) + (unsigned long long)(m1 ? f1 : 0u);
    };
    auto issue_gathers = [&](uint32_t d0, uint32_t d1) __attribute__((always_inline)) {
        const uint32_t o0 = (d0 != 0xFFFFFFFFu ? d0 >> 5 : 0u) * 4u, o1 = (d1 != 0xFFFFFFFFu ? d1 >> 5 : 0u) * 4u;
#pragma unroll
        for (int j = 0; j < AS_MAXO; ++j) {
            if (j < (int)nother) {
                as_gather_dword(bm[j], o0, g_base + (uint32_t)j * 256u);
                as_gather_dword(bm[j], o1, g_base + (uint32_t)(AS_MAXO + j) * 256u);
            }
        }
    };
    const uint32_t e1 = b1 < nfull ? b1 : nfull; // full blocks of the run: [b0, e1)
    for (uint32_t w0 = b0; w0 < e1; w0 += 63u) {
        const uint32_t first = w0 ? w0 - 1u : 0u;
        const uint32_t ridx = first + lane;
        uint2 row = make_uint2(0u, 0u); // {block_max, end offset} of block first + lane
        if (ridx < nb) row = tab[ridx];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(row.x), "+v"(row.y)::"memory");
        const uint32_t cnt = (e1 - w0 < 63u) ? e1 - w0 : 63u;
        auto ep_of = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t { return b ? bcast(row.y, b - 1u - first) : 0u; };
        auto base_of = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t { return b ? bcast(row.x, b - 1u - first) + 1u : 0u; };
        auto issue = [&](uint32_t i) __attribute__((always_inline)) {
            const uint32_t b = w0 + i, slot = i & (FS_RING - 1);
            fs_prefetch_blk(fs_uniform_ptr(data + ep_of(b)), st_base + slot * (STAGE_DW * 4u), fs_uniform_ptr(xs0 + (size_t)XSLOT_DW * b), xs_base + slot * (XSLOT_DW * 4u), voff);
        };
        for (uint32_t i = 0; i < cnt && i < (uint32_t)FS_RING; ++i) issue(i);
        uint32_t dP0 = 0xFFFFFFFFu, dP1 = 0xFFFFFFFFu, fP0 = 0, fP1 = 0; // the block whose gathers are in flight
        for (uint32_t i = 0; i < cnt; ++i) {
            // behind block i's bytes: the bytes of the blocks requested after it and the gathers of block i - 1
            const uint32_t pb = cnt - 1u - i < (uint32_t)(FS_RING - 1) ? cnt - 1u - i : (uint32_t)(FS_RING - 1);
            as_wait(FS_LOADS * pb + (i ? G : 0u));
            const uint32_t slot = i & (FS_RING - 1);
            const uint32_t* const stg = L.stage[slot];
            const uint32_t* const xs = L.xs[slot];
            const SlotHead h = optpfor_slot_head(xs);
            uint32_t v0, v1, f0 = 0, f1 = 0, cd, cf;
            if (__builtin_expect(h.flag == 0u, 1)) {
                optpfor_decode_pair<true, WITH_FREQS>(stg, xs, h, v0, v1, f0, f1, cd, cf);
            } else {
                uint32_t nd = 0;
                const uint8_t* const gblk = data + ep_of(w0 + i);
                cd = optpfor_decode_side(stg, STAGE_DW, xs, gblk, a.xovf, 0u, 0u, v0, v1, &nd);
                if constexpr (WITH_FREQS) {
                    const uint32_t skip_dw = cd >> 2;
                    optpfor_decode_side(stg + (skip_dw < STAGE_DW ? skip_dw : 0u), skip_dw < STAGE_DW ? STAGE_DW - skip_dw : 0u, xs, gblk + cd, a.xovf, 1u, nd, f0, f1);
                }
            }
            const uint32_t i0 = wave_incl_scan(v0 + 1u);
            const uint32_t i1 = wave_incl_scan(v1 + 1u) + bcast(i0, 63);
            const uint32_t base = base_of(w0 + i);
            const uint32_t d0 = base + i0 - 1u, d1 = base + i1 - 1u;
            // block i - 1: its gathers are followed only by the prefetch issued right after them
            if (i) {
                as_wait((i - 1u + FS_RING < cnt) ? (uint32_t)FS_LOADS : 0u);
                consume(dP0, dP1, fP0, fP1);
            }
            issue_gathers(d0, d1);
            if (i + FS_RING < cnt) issue(i + FS_RING);
            dP0 = d0;
            dP1 = d1;
            fP0 = f0 + 1u;
            fP1 = f1 + 1u;
        }
        fs_wait_vm<0>();
        consume(dP0, dP1, fP0, fP1);
    }
    if (b1 > nfull) { // the partial last block: plain gaps - 1 / freqs - 1 in the tail table; its bits by plain loads
        const uint32_t sz = n & 127u;
        const uint32_t* const t = a.tails + (((unsigned long long)uniform((uint32_t)(st->tail >> 32)) << 32) | uniform((uint32_t)st->tail));
        const uint32_t v0 = lane < sz ? t[lane] + 1u : 0u, v1 = lane + 64 < sz ? t[lane + 64] + 1u : 0u;
        const uint32_t f0 = lane < sz ? t[sz + lane] + 1u : 0u, f1 = lane + 64 < sz ? t[sz + lane + 64] + 1u : 0u;
        const uint32_t base = nfull ? tab[nfull - 1u].x + 1u : 0u;
        const uint32_t i0 = wave_incl_scan(v0);
        const uint32_t i1 = wave_incl_scan(v1) + bcast(i0, 63);
        const uint32_t d0 = lane < sz ? base + i0 - 1u : 0xFFFFFFFFu, d1 = lane + 64 < sz ? base + i1 - 1u : 0xFFFFFFFFu;
        bool m0 = d0 != 0xFFFFFFFFu, m1 = d1 != 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < AS_MAXO; ++j) {
            if (j < (int)nother) {
                const uint32_t* const w = (const uint32_t*)bm[j];
                const uint32_t x0 = m0 ? w[d0 >> 5] : 0u, x1 = m1 ? w[d1 >> 5] : 0u;
                m0 = m0 && ((x0 >> (d0 & 31u)) & 1u);
                m1 = m1 && ((x1 >> (d1 & 31u)) & 1u);
            }
        }
        acc_c += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
        if constexpr (WITH_FREQS) acc_f += (unsigned long long)(m0 ? f0 : 0u) + (unsigned long long)(m1 ? f1 : 0u);
    }
    for (int o = 32; o; o >>= 1) {
        acc_c += __shfl_xor(acc_c, o);
        if constexpr (WITH_FREQS) acc_f += __shfl_xor(acc_f, o);
    }
    const uint32_t q = uniform(st->q);
    if (lane == 0) {
        if (counts && acc_c) atomicAdd(a.out_count + q, (unsigned long long)acc_c);
        if (WITH_FREQS && a.out_freq_sum && acc_f) atomicAdd(a.out_freq_sum + q, acc_f);
    }
}

} // namespace

// longest = blocks of the longest list among the terms
extern "C" hipError_t ds2i_launch_freq_stream(const void* args, unsigned longest, unsigned nqterms, hipStream_t s) {
    const FreqArgs& a = *(const FreqArgs*)args;
    if (!nqterms) return hipSuccess;
    hipLaunchKernelGGL(k_freq_stream, dim3((longest + FS_RUN - 1) / FS_RUN, nqterms), dim3(64), 0, s, a);
    return hipGetLastError();
}

// and / and_freq of the all-dense queries: nterms records of AndStreamArgs::terms, longest = blocks of the longest of their lists
extern "C" hipError_t ds2i_launch_and_stream(const void* args, int with_freqs, unsigned longest, unsigned nterms, hipStream_t s) {
    const AndStreamArgs& a = *(const AndStreamArgs*)args;
    if (!nterms) return hipSuccess;
    const dim3 g((longest + FS_RUN - 1) / FS_RUN, nterms), b(64);
    if (with_freqs) hipLaunchKernelGGL((k_and_stream<true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_and_stream<false>), g, b, 0, s, a);
    return hipGetLastError();
}
