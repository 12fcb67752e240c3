// Pieces shared by the software-pipelined stream kernels (ranked_stream.hip: ranked_and / and; union_stream.hip: wand /
// maxscore / ranked_or): block search over a list's interleaved skip table, membership in a decoded block, explicit kernarg
// addressing, and the loads that are issued and waited for BY HAND (LDS-DMA block prefetch, LDS-DMA byte gathers, counted
// s_waitcnt). gfx950 / CDNA4, wave64. Everything here is DS2I_DEV (force-inlined device code).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_enum.hpp"
#include "device_score.hpp"

namespace ds2i_dev {
namespace stream {

// first block >= from of a list whose block_max >= lb, with its table words; rows = the list's interleaved skip table
// ({block_max, end offset} per block), wtab = its block weights. 64 rows per probe: the 64 after `from`, then a 64-ary
// search (the reference scans block_max linearly, block_posting_list.hpp:134-137).
struct Found { uint32_t blk, bmax, base, ep; float w; };
// the first probe's rows (from-1 .. from+62; lane 0 = the block before `from`, never a candidate itself): they do not depend on
// the doc-id searched for, so stage C requests them together with the candidates' norm_lens, one round trip earlier
struct Rows { uint2 e; float w; };
DS2I_DEV Rows rows_load(const uint2* tab, const float* wtab, uint32_t nb, uint32_t from) {
    const uint32_t idx = (from ? from - 1 : 0) + lane_id();
    Rows r{make_uint2(0xFFFFFFFFu, 0u), 0.f};
    if (idx < nb) { r.e = tab[idx]; r.w = wtab[idx]; }
    return r;
}
DS2I_DEV bool find_block_rows(const uint2* tab, const float* wtab, uint32_t nb, uint32_t from, uint32_t lb, Found& o, const Rows& first_rows) {
    const uint32_t lane = lane_id();
    if (from >= nb) return false;
    float wv = 0.f;
    auto finish = [&](uint2 e, uint32_t first_idx, uint64_t hit) __attribute__((always_inline)) {
        const uint32_t f = (uint32_t)__builtin_ctzll(hit);
        o.blk = first_idx + f;
        o.w = __uint_as_float(bcast(__float_as_uint(wv), f));
        o.bmax = bcast(e.x, f);
        const uint32_t pf = f ? f - 1 : 0;
        const uint32_t pmax = bcast(e.x, pf), pend = bcast(e.y, pf);
        o.base = o.blk ? pmax + 1u : 0u;
        o.ep = o.blk ? pend : 0u;
    };
    {
        const uint32_t first = from ? from - 1 : 0;
        const uint32_t idx = first + lane;
        const uint2 e = first_rows.e;
        wv = first_rows.w;
        const uint64_t hit = ballot(idx >= from && idx < nb && e.x >= lb);
        if (hit) { finish(e, first, hit); return true; }
        if (first + 64 >= nb) return false;
    }
    uint32_t lo = (from ? from - 1 : 0) + 64, hi = nb; // answer in [lo, hi) or none
    while (hi - lo > 63) {
        const uint32_t stride = (hi - lo + 63) / 64;
        uint32_t idx = lo + (lane + 1) * stride - 1;
        if (idx >= hi) idx = hi - 1;
        const uint32_t v = tab[idx].x;
        const uint64_t hit = ballot(v >= lb);
        if (!hit) return false;
        const uint32_t f = (uint32_t)__builtin_ctzll(hit);
        const uint32_t nhi = lo + (f + 1) * stride;
        hi = nhi < hi ? nhi : hi;
        lo = lo + f * stride;
    }
    const uint32_t first = lo - 1; // (lo >= 64 here)
    const uint32_t idx = first + lane;
    uint2 e = make_uint2(0xFFFFFFFFu, 0u);
    if (idx < hi) { e = tab[idx]; wv = wtab[idx]; }
    const uint64_t hit = ballot(idx >= lo && idx < hi && e.x >= lb);
    if (!hit) return false;
    finish(e, first, hit);
    return true;
}

// position of c in the sorted block d[128] (valid iff `want`): binary search per lane
DS2I_DEV bool rs_member(const uint32_t* d, uint32_t c, bool want, uint32_t& pos) {
    uint32_t idx = 0;
    if (want) {
#pragma unroll
        for (uint32_t step = 64; step; step >>= 1)
            if (d[idx + step - 1] < c) idx += step;
    }
    pos = idx;
    return want && d[idx] == c;
}

DS2I_DEV void store_topk_rs(float* topk, uint32_t* topk_len, uint32_t k, uint32_t slot, const TopK& tk) {
    const uint32_t lane = lane_id();
    if (lane < k) topk[(size_t)slot * k + lane] = tk.v;
    if (lane == 0) topk_len[slot] = tk.n;
}
template <int NK>
DS2I_DEV void store_topk_rs(float* topk, uint32_t* topk_len, uint32_t k, uint32_t slot, const TopKBig<NK>& tk) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int r = 0; r < NK; ++r)
        if ((uint32_t)r * 64u + lane < k) topk[(size_t)slot * k + (uint32_t)r * 64u + lane] = tk.v[r];
    if (lane == 0) topk_len[slot] = tk.n;
}

template <int I, int N, class F>
DS2I_DEV void rs_for(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rs_for<I + 1, N>(f);
    }
}
template <int I, int LO, class F>
DS2I_DEV void rs_for_down(F& f) { // I-1 down to LO
    if constexpr (I > LO) {
        f(std::integral_constant<int, I - 1>{});
        rs_for_down<I - 1, LO>(f);
    }
}

// The argument block is ~40 pointers and scalars. Read as a by-value kernel argument the compiler loads all of them at
// kernel entry and keeps them in SGPRs for the kernel's lifetime. Here the kernarg segment is addressed explicitly: the few
// hot fields are read where a unit starts, the cold ones at their use site through a pointer the optimiser cannot see through
// (so the loads stay where they are written instead of being hoisted above the loops).
typedef const BatchArgs __attribute__((address_space(4))) * KArgs; // (constant address space: uniform reads are s_load)
DS2I_DEV KArgs rs_args() {
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// a wave-uniform value / pointer the compiler could not prove uniform (anything loaded through a global pointer): through
// v_readfirstlane, so that what is computed from it is scalar arithmetic and the "s" constraints below get scalar registers
// (given a VGPR pair they assemble to nothing)
template <class T> DS2I_DEV const T* rs_uniform_ptr(const T* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    return (const T*)(uintptr_t)(((unsigned long long)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v));
}
DS2I_DEV unsigned long long rs_uniform64(unsigned long long v) { return ((unsigned long long)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v); }
DS2I_DEV float rs_uniformf(float v) { return __uint_as_float(uniform(__float_as_uint(v))); }

// ---- loads the compiler must not count. hipcc drains vmcnt to 0 wherever control flow joins with a load pending on
// some path, which would put every round trip back on the critical path; these are issued and waited for by hand.
// (i) block bytes + side slot: LDS-DMA, global -> LDS with no register in between (nothing the compiler could copy or spill
// early). 512 bytes at g (4-byte aligned) -> LDS byte offset `lds`, 256 bytes at gx -> lds_x; voff = lane * 4. M0 is the DMA's
// LDS base: compiler-reserved, so it is saved, set and restored inside the statement. (The instruction offset moves the global
// AND the LDS address: measured, profiles/probes/ldsdma_probe.hip.)
DS2I_DEV uint32_t rs_lds_offset(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
DS2I_DEV void rs_prefetch_blk(const uint8_t* g, uint32_t lds, const uint32_t* gx, uint32_t lds_x, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\tglobal_load_lds_dword %1, %2 offset:256\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %1, %4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(uniform(lds)), "s"(gx), "s"(uniform(lds_x)) : "memory");
}
static constexpr int PF_LOADS = 3; // hand-issued loads of one block prefetch
// (i') one dword at g, read past this CU's L1 (sc1: other CUs update it with atomics) -> the 64 dwords at LDS byte offset lds
DS2I_DEV void rs_fetch_word(const unsigned int* g, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(0u), "s"(g), "s"(uniform(lds)) : "memory");
}
// (i'') the same through this CU's L1: a stale copy is a LOWER floor, which is still a valid one (floors only rise), and the L1 of a CU
// running these kernels turns over within microseconds -- thousands of waves polling one word per query past the L1 every block made
// the word's L2 line the slowest load of every prefetch group
DS2I_DEV void rs_fetch_word_cached(const unsigned int* g, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(0u), "s"(g), "s"(uniform(lds)) : "memory");
}
// (ii) range-table bytes: LDS-DMA as well -- tab[off] of every lane lands, zero-extended, in the dword at LDS byte offset
// lds + 4 * lane (measured with the same probe). A hand-issued load into a VGPR is not an option: for the compiler the
// destination is written when the statement ends, and under register pressure it did copy the still-pending register
// (tests/asm_audit.py found it before the GPU did).
DS2I_DEV void rs_gather_u8(const uint8_t* tab, uint32_t off, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(tab), "s"(uniform(lds)) : "memory");
}
// one lane of a VGPR takes a wave-uniform value (v_writelane_b32; there is no builtin for it in this toolchain)
template <int LANE> DS2I_DEV void rs_writelane(uint32_t& dst, uint32_t v) {
    const uint32_t sv = uniform(v);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(sv), "n"(LANE));
}
// ... at a wave-uniform lane known only at run time: the lane select goes through M0 (a second SGPR operand would break the
// constant-bus limit of this encoding); M0 is compiler-reserved, so it is saved and restored inside the statement
DS2I_DEV void rs_writelane_at(uint32_t& dst, uint32_t v, uint32_t lane) {
    const uint32_t sv = uniform(v), sl = uniform(lane);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %1, %2, m0\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(dst) : "s"(sv), "s"(sl));
}
template <int N> DS2I_DEV void rs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// A wait the COMPILER sees (the builtin, not inline asm): placed where a rare path that issued compiler-visible loads joins the hot
// path again. hipcc's wait-count pass keeps every such load "possibly pending" on the joined path until a wait it knows about, and
// the next instruction of the hot path that merely REUSES one of their destination registers then gets an s_waitcnt vmcnt(0) -- which
// also drains the hand-issued prefetch that was meant to stay in flight. (Found in round 6: every iteration of k_ranked_stream /
// k_union_stream waited out the next block's prefetch right in front of its decode's first LDS read, because the general side-slot
// decoder -- one block in 10^4 -- loads through registers. 0x0F70 = vmcnt(0), expcnt / lgkmcnt untouched, gfx9 encoding.)
DS2I_DEV void rs_settle_vm() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// stage C: the 512 bytes from g (dword aligned) and the 256-byte side slot at gx -> LDS, by plain loads
DS2I_DEV void rs_stage_block(const uint32_t* g, const uint32_t* gx, uint32_t* st, uint32_t* xs) {
    const uint32_t lane = lane_id();
    const uint32_t w0 = g[lane], w1 = g[lane + 64], x = gx[lane];
    st[lane] = w0;
    st[lane + 64] = w1;
    xs[lane] = x;
    wave_sync();
}
// the partial last block of a list from the tail table (BatchArgs::tails; entry = sz gaps-1, sz freqs-1, bytes of the docs
// part, bytes of the freqs part). Rare (once per list and unit at most): plain loads, waited for here.
DS2I_DEV void rs_tail(const uint32_t* tails, unsigned long long entry, uint32_t sz, uint32_t& d0, uint32_t& d1, uint32_t& f0, uint32_t& f1, uint32_t& cons_d, uint32_t& cons_f) {
    const uint32_t lane = lane_id();
    const uint32_t* const t = tails + entry;
    uint32_t a0 = (lane < sz) ? t[lane] : 0u, a1 = (lane + 64 < sz) ? t[lane + 64] : 0u;
    uint32_t b0 = (lane < sz) ? t[sz + lane] : 0u, b1 = (lane + 64 < sz) ? t[sz + lane + 64] : 0u;
    uint32_t c0 = t[2u * sz], c1 = t[2u * sz + 1u];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1)::"memory");
    d0 = a0;
    d1 = a1;
    f0 = b0;
    f1 = b1;
    cons_d = uniform(c0);
    cons_f = uniform(c1);
    rs_settle_vm();
}
// docs (gaps-1) and freqs-1 of a full block staged at st / slot: the branch-free pair decoder, or -- a block in 10^4: raw parts,
// parts beyond the staged bytes, adds in the overflow area -- the general side-slot decoder part by part. gblk = the block's
// address in the arena.
DS2I_DEV void rs_decode_full(const uint32_t* st, const uint32_t* slot, const uint8_t* gblk, const uint32_t* xovf, uint32_t& d0, uint32_t& d1, uint32_t& f0,
                             uint32_t& f1, uint32_t& cons_d, uint32_t& cons_f) {
    const SlotHead h = optpfor_slot_head(slot);
    if (__builtin_expect(h.flag == 0u, 1)) {
        optpfor_decode_pair(st, slot, h, d0, d1, f0, f1, cons_d, cons_f);
    } else {
        uint32_t nd = 0;
        cons_d = optpfor_decode_side(st, STAGE_DW, slot, gblk, xovf, 0u, 0u, d0, d1, &nd);
        const uint32_t skip_dw = cons_d >> 2;
        cons_f = optpfor_decode_side(st + skip_dw, skip_dw < STAGE_DW ? STAGE_DW - skip_dw : 0u, slot, gblk + cons_d, xovf, 1u, nd, f0, f1);
        rs_settle_vm();
    }
}

// an UPPER bound of bm25 doc_term_weight(f, nl) = f / (f + k1 (1 - b + b nl)) (device_enum.hpp) for the pruning tests: the
// quotient through v_rcp_f32 (1 ulp) instead of the IEEE division sequence (11 instructions), widened by 2^-20 -- far more than
// the reciprocal's and the product's rounding can lose. Scores themselves are always computed with the exact division.
DS2I_DEV float rs_dtw_bound(uint32_t freq, float norm_len) {
    const float f = (float)freq;
    return f * __builtin_amdgcn_rcpf(f + 1.2f * (0.5f + 0.5f * norm_len)) * (1.0f + 1.0f / 1048576.0f);
}

} // namespace stream
} // namespace ds2i_dev
