// ranked_and on a block_optpfor index with the upload-time tables, as a software-pipelined stream
// (gfx950 / CDNA4, wave64; one wavefront per work unit, wave-uniform control flow, no MFMA: integer work).
//
// Replaces ranked_and_query (reference queries.hpp:322-401: candidate = next posting of the shortest list, next_geq() on
// every other list, score = sum of bm25 term scores in list order, topk_queue::insert 157-172) for queries of exactly
// NT = 2..8 distinct terms. Same results as k_conjunctive<true, ...> (kernels.hip), which stays the kernel of every
// other case (other codecs, no tables, 1 term, 5+ terms; block_mixed: ranked_stream_mixed.hip); what differs is how a unit
// is executed:
//
//   * a block of the driving list (list 0, the shortest) is handled exactly ONCE, in three stages that belong to
//     three different blocks at any moment:
//         stage N (block i+2)  chosen from the table window, its bytes and its exception side slot requested (LDS-DMA)
//         stage A (block i+1)  docs AND freqs decoded in one branch-free pass (device_codecs.hpp, optpfor_decode_pair); every
//                              posting gets a bound of its own list-0 term score from its freq alone
//         stage B (block i)    its range-table gathers -- issued only for the candidates whose own bound + the other lists'
//                              maxima could enter the heap -- are consumed: zero byte = the document is in no intersection;
//                              otherwise own bound + own bytes against the heap threshold, then the membership hints
//         stage C (block i)    only if somebody survived: norm_len, exact list-0 score, then list 1 .. NT-1 in order (locate
//                              block -> block-weight test -> decode -> membership -> score), heap insert
//     so the gather round trip of a block is covered by the decode of the next one and the block-bytes round trip by a
//     whole iteration;
//   * no enumerator object: the driving list's state is the 64-row table window in registers, a block's decoded doc-ids and
//     freqs stay in the registers of the lanes that own them (value i in lane i & 63, slot i >> 6), the other lists keep one
//     decoded block each (doc-ids + freqs) in LDS;
//   * one decoder: full blocks through their side slots (BatchArgs::xslots), the lists' partial last blocks from the tail table
//     (BatchArgs::tails) -- no Simple16, no interpolative walk, no scratch memory.
//
// Every pruning test is a true upper bound of the float32 score the scoring code would compute (device_score.hpp,
// BOUND_SLACK; doc_term_weight falls with norm_len, so the collection's shortest document bounds a term score from the freq
// alone), and topk_queue::insert is strict, so the heap ends with the same multiset of scores as the sequential traversal,
// bit for bit (tests/test_gpu.py: test_ranked_and_pruning_fuzz_bit_identical).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_enum.hpp"
#include "device_score.hpp"

using namespace ds2i_dev;

namespace {

#ifndef DS2I_RS_OCC2
#define DS2I_RS_OCC2 6
#endif
#ifndef DS2I_RS_OCC4
#define DS2I_RS_OCC4 6 // (round 5, on the leaner kernel: 5 -> 6 waves per SIMD for 3 / 4 lists +3.5 % end to end; 6 -> 7 / 8 for 2 lists: nothing)
#endif
// 5..8 lists: what fits -- one more decoded block (1 KB of LDS) and nine more parked scalars per list:
// 7 936 .. 11 008 bytes of LDS per wave
#define RS_WAVES(NT) ((NT) <= 2 ? DS2I_RS_OCC2 : (NT) <= 4 ? DS2I_RS_OCC4 : (NT) <= 7 ? 4 : 3)

template <int NT>
struct LdsRS {
    uint32_t stage[3][STAGE_DW]; // list 0: bytes of the blocks in stage B/C, in stage A and on their way in (LDS-DMA)
    uint32_t xs[3][XSLOT_DW];    // their exception side slots (LDS-DMA, with the bytes)
    uint32_t gb[2][64];          // list 1's range-table byte of every posting of the block in stage B (LDS-DMA, one dword per lane)
    uint32_t stb[STAGE_DW];      // stage C: bytes of the block of list j being decoded
    uint32_t xsb[XSLOT_DW];      // its side slot
    uint32_t dj[NT - 1][128];    // lists 1 .. NT-1: doc-ids of their current block
    uint32_t fj[NT - 1][128];    // and its freqs
    uint32_t fw[64];             // the query's shared floor word, as fetched an iteration ago (LDS-DMA: one copy per lane)
};

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
template <int N> DS2I_DEV void rs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

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
    }
}

// an UPPER bound of bm25 doc_term_weight(f, nl) = f / (f + k1 (1 - b + b nl)) (device_enum.hpp) for the pruning tests: the
// quotient through v_rcp_f32 (1 ulp) instead of the IEEE division sequence (11 instructions), widened by 2^-20 -- far more than
// the reciprocal's and the product's rounding can lose. Scores themselves are always computed with the exact division.
DS2I_DEV float rs_dtw_bound(uint32_t freq, float norm_len) {
    const float f = (float)freq;
    return f * __builtin_amdgcn_rcpf(f + 1.2f * (0.5f + 0.5f * norm_len)) * (1.0f + 1.0f / 1048576.0f);
}

#ifndef RS_HINT_FIRST
#define RS_HINT_FIRST(nt) ((nt) > 2)
#endif
template <int NT, bool STATS>
__global__ void __launch_bounds__(64, RS_WAVES(NT)) k_ranked_stream(BatchArgs a_unused) {
    static_assert(NT >= 2 && NT <= 8, "exact list counts 2..8");
    static_assert((NT - 2) * 9 + 8 < 64, "the per-list constants of lists 1.. are parked in the lanes of one VGPR");
    __shared__ LdsRS<NT> L;
    const uint32_t lane = lane_id();
    typename std::conditional<STATS, uint32_t, NullCounter>::type s_docs_blocks, s_freqs_blocks, s_bm_examined, s_scored, s_rounds;
    typename std::conditional<STATS, unsigned long long, NullCounter>::type s_bytes;
    s_docs_blocks = s_freqs_blocks = s_bm_examined = s_scored = s_rounds = 0;
    s_bytes = 0;
#ifdef DS2I_RS_PHASE
    // diagnostic build (-DDS2I_RS_PHASE, instrumented runs): shader cycles of a wave by where it spends them, reported through
    // Stats::phase_cycles (profiles/probes/rs_phase_probe.py). PT(slot) closes the interval since the previous PT and books it.
    unsigned long long pt[PH_COUNT] = {};
    unsigned long long pt_prev = __builtin_readcyclecounter();
#define PT(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); pt[slot] += t_ - pt_prev; pt_prev = t_; } while (0)
#else
#define PT(slot) ((void)0)
#endif
#ifdef DS2I_LINE_COUNT
    // diagnostic build: distinct 128-byte lines requested by the hand-placed gathers, by purpose (reported through Stats::phase_cycles)
    unsigned long long lc[PH_COUNT] = {};
    // lines touched by one wave instruction whose active lanes read `bytes` bytes (lanes in ascending address order)
    auto lines_of = [&](const void* addr, bool active, uint32_t bytes) -> uint32_t {
        const unsigned long long lo = (unsigned long long)(uintptr_t)addr >> 7, hi = ((unsigned long long)(uintptr_t)addr + bytes - 1) >> 7;
        const uint64_t act = ballot(active);
        unsigned long long prev_hi = ~0ull; // previous ACTIVE lane's last line
        uint32_t n = 0;
        for (uint64_t m = act; m; m &= m - 1) {
            const uint32_t src = (uint32_t)__builtin_ctzll(m);
            const unsigned long long l = ((unsigned long long)bcast((uint32_t)(lo >> 32), src) << 32) | bcast((uint32_t)lo, src);
            const unsigned long long h = ((unsigned long long)bcast((uint32_t)(hi >> 32), src) << 32) | bcast((uint32_t)hi, src);
            n += (uint32_t)(h - l + 1) - ((l == prev_hi) ? 1u : 0u);
            prev_hi = h;
        }
        return n;
    };
#define LC(slot, expr) lc[slot] += (expr)
#else
#define LC(slot, expr) ((void)0)
#endif
    const uint32_t nslice = rs_args()->nslice;
    for (uint32_t tkt = blockIdx.x; tkt < nslice; tkt += gridDim.x) {
        KArgs a = rs_args(); // (fields read below stay live for the unit; the cold ones are re-read at their use site)
        const UnitRec u = a->urec[tkt]; // (one 32-byte record: the unit, its query's terms, its histogram)
        const uint32_t uid = uniform(u.uid);
        if constexpr (STATS) { // diagnostic (DS2I_UNIT_CLOCK=1): when the unit started / ended
            unsigned long long* const clk = a->unit_clock;
            if (clk && lane == 0) clk[2ull * uid] = wall_clock64();
        }
        const uint32_t q = uniform(u.q), blk_begin = uniform(u.blk_begin), blk_end = uniform(u.blk_end);
        const bool whole = uniform(u.nparts) == 1u;
        const QTerm* const qt = rs_uniform_ptr(a->qterms + uniform(u.qt_off)); // exactly NT terms (the planner's launch groups)
        TopK tk;
        tk.init(a->k);
        // ---- list 0: the stream
        const uint32_t n0 = uniform(qt[0].n), nb0 = (n0 + 127u) >> 7;
        const uint32_t vl0 = 1u + (n0 >= (1u << 7)) + (n0 >= (1u << 14)) + (n0 >= (1u << 21)) + (n0 >= (1u << 28));
        const uint32_t bb0 = uniform(qt[0].blk_base);
        const uint8_t* const data0 = a->arena + rs_uniform64(qt[0].list_off) + vl0 + 4ull * nb0 + 4ull * (nb0 - 1);
        const float qw0 = rs_uniformf(qt[0].q_weight);
        const uint32_t* const xs0 = a->xslots + (size_t)XSLOT_DW * bb0; // list 0's side slots
        // ---- lists 1 .. NT-1: range table (hot), the rest of the QTerm is read when a candidate gets that far
        const uint8_t* rt[NT];
        uint32_t rsh[NT];
        float rsc[NT];
        auto bind_one = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            rt[j] = a->rmw + 64ull * uniform(qt[j].rmw_off64);
            rsh[j] = uniform(qt[j].rmw_shift);
            rsc[j] = rs_uniformf(qt[j].rmw_scale);
        };
        rs_for<1, NT>(bind_one);
        // what lists 1.. can add to any document at most (their list maxima), and the collection's shortest document: a posting of
        // list 0 with freq f scores at most qw0 * doc_term_weight(f, min_nl) there
        float rest_all = 0.f;
        {
            auto add_max = [&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; rest_all = rest_all + rsc[j] * 255.0f; };
            rs_for_down<NT, 1>(add_max);
        }
        const float min_nl = a->min_norm_len;
        const long long hdelta = a->rmh ? (long long)(a->rmh - a->rmw) : 0ll; // hint of an entry = the byte at the same offset of the parallel buffer
        // Three and four lists: the byte fetched ahead for every candidate is list 1's HINT, not its weight. A weight byte lets a
        // candidate through whenever its range holds any posting (one candidate in 4..6, each then costing a line per further
        // list); the hint also settles the ranges with a single posting, so that the further lists are asked about a few per cent
        // of the candidates only -- first their hints, then, for what is left, every list's weight for the threshold test.
        // (Two lists: the weight stays first, its threshold test removes more than the hint does.)
        const bool hint_first = RS_HINT_FIRST(NT) && hdelta != 0;
        const uint8_t* const gt1 = hint_first ? rt[1] + hdelta : rt[1];
        // block of list j whose doc-ids and freqs are in L.dj[j-1] / L.fj[j-1] (cur = ~0: none) and its block_max. Only stage C
        // touches them: they live in the lanes of one VGPR (v_readlane / v_writelane at a constant lane).
        // Together with the list's geometry (postings, first row in the per-block tables, arena offset, query weight): read once
        // per unit, here, so that stage C starts with its first memory request instead of a dependent read of the QTerm.
        uint32_t cold = 0xFFFFFFFFu;
        enum { C_CUR = 0, C_BMAX = 1, C_N = 2, C_BB = 3, C_LOLO = 4, C_LOHI = 5, C_QW = 6, C_TLLO = 7, C_TLHI = 8, C_PER = 9 };
#define cget(l) ((uint32_t)__builtin_amdgcn_readlane((int)cold, (l)))
#define cset(l, v) rs_writelane<(l)>(cold, (v))
        {
            auto park = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int CB = (j - 1) * C_PER;
                const unsigned long long lo = qt[j].list_off, tl = qt[j].aux1;
                cset(CB + C_N, qt[j].n);
                cset(CB + C_BB, qt[j].blk_base);
                cset(CB + C_LOLO, (uint32_t)lo);
                cset(CB + C_LOHI, (uint32_t)(lo >> 32));
                cset(CB + C_QW, __float_as_uint(qt[j].q_weight));
                cset(CB + C_TLLO, (uint32_t)tl);
                cset(CB + C_TLHI, (uint32_t)(tl >> 32));
            };
            rs_for<1, NT>(park);
        }
        // ---- pruning state: the parts of a split query share a score histogram (device_score.hpp)
        unsigned int* const q_hist = a->q_hist;
        const bool shared_floor = !whole && q_hist;
        ScoreHist sh;
        sh.init(shared_floor ? q_hist : nullptr, shared_floor ? uniform(u.hist_slot) : 0u, shared_floor ? rs_uniformf(qt[0].max_bmw + qt[0].suf_bmw) : 0.f,
                1.0f - 1.0f / 1048576.0f);
        // can a score enter the heap: s >= floor && (heap not full || s > k-th score) (TopK::would_enter), branch-free on two
        // wave-uniform values that are refreshed whenever the heap or the floor changes
        float e_floor = tk.floor, e_gt = -__builtin_inff();
        auto refresh = [&]() __attribute__((always_inline)) { e_floor = tk.floor; e_gt = tk.n < tk.k ? -__builtin_inff() : tk.thr; };
        auto enters = [&](float s) __attribute__((always_inline)) -> bool { return (s >= e_floor) & (s > e_gt); };
        // The floor the histogram implies is published in one word per split query (BatchArgs::q_floor, float bits: scores are
        // >= 0, so the bit patterns order like the values): whoever puts a score into its heap re-reads the histogram -- the only
        // moment the floor can have moved -- and raises the word; everybody else gets the word with every block, fetched an
        // iteration ahead by LDS-DMA, instead of a 256-counter scan every fourth block on the critical path.
        unsigned int* const fwp = shared_floor ? (unsigned int*)rs_uniform_ptr(a->q_floor + uniform(u.hist_slot)) : nullptr;
        auto adopt_word = [&](uint32_t bits) __attribute__((always_inline)) {
            const float f = __uint_as_float(bits);
            if (bits != 0u && f > tk.floor) { tk.floor = f; refresh(); }
        };
        if (shared_floor) adopt_word(uniform(__hip_atomic_load(fwp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        // ---- the 64-row window of list 0's table (lane j: row s_first + j; lane 0 is the row before the first block the window
        // can serve, unless that is block 0) and, per row, what a document of that block can score at most: block weight + for
        // each other list the largest range-table entry over the block's own doc-id span (read from the level whose entries are
        // wide enough for <= 16 of them to cover the span); -1 = no such row, or some other list has no posting in the span.
        // Before the first fill (s_valid == 0) no row is live and the refill starts at the unit's first block.
        uint32_t s_first = 0, s_valid = 0;
        uint2 s_e = make_uint2(0xFFFFFFFFu, 0u);
        float s_ub = -1.f;

        auto s_fill = [&](uint32_t first) __attribute__((always_inline)) {
            s_first = first;
            const uint32_t idx = first + lane;
            s_e = make_uint2(0xFFFFFFFFu, 0u);
            float s_w = 0.f;
            {   // (once per 63 blocks: the table pointers are re-derived here rather than carried through the loop)
                const uint2* const tab0 = (const uint2*)rs_args()->skip + bb0;
                const float* const w0tab = rs_args()->bmw + bb0;
                if (idx < blk_end) { s_e = tab0[idx]; s_w = w0tab[idx]; }
                LC(PH_PROLOG, lines_of(tab0 + idx, idx < blk_end, 8u) + lines_of(w0tab + idx, idx < blk_end, 4u));
            }
            const uint32_t prev_max = (uint32_t)__shfl_up((int)s_e.x, 1);
            const uint32_t base = (lane == 0) ? 0u : prev_max + 1u, top = s_e.x;
            const bool row = idx < blk_end && (lane > 0 || idx == 0) && top != 0xFFFFFFFFu && base <= top;
            const uint32_t b2 = row ? base : 0u, t2 = row ? top : 0u; // branch-free: a lane without a row reads entry 0 and discards it
            float acc = 0.f;
            bool dead = false;
            auto one_list = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const RmwLevels g(rs_args()->num_docs, rsh[j]);
                uint32_t lsh = rsh[j], lvl = 0;
                while (lvl < 2 && (t2 >> lsh) - (b2 >> lsh) >= 16u) { lsh += 6; ++lvl; }
                const uint32_t lo = b2 >> lsh, hi = t2 >> lsh;
                const bool fits = hi - lo < 16u;
                const uint64_t loff = lvl == 0 ? 0ull : lvl == 1 ? g.off[1] : g.off[2]; // (selects: a run-time index would put the array into scratch)
                const uint32_t m = max_of_bytes16(rt[j] + loff + (fits ? lo : 0u), fits ? hi - lo + 1u : 1u);
                LC(PH_PROLOG, lines_of(rt[j] + loff + (fits ? lo : 0u), true, 16u));
                const uint32_t best = (row && fits) ? m : 255u; // (255 = the list maximum)
                dead = dead || best == 0u;
                acc = acc + rsc[j] * (float)best;
            };
            rs_for_down<NT, 1>(one_list);
            s_ub = (dead || !row) ? -1.0f : (qw0 * s_w + acc) * BOUND_SLACK; // (scores are >= 0: -1 never enters)
        };
        // a block of list 0 on its way through the stages
        struct Blk { uint32_t blk, base, ep; };
        auto select = [&](uint32_t from, Blk& o) __attribute__((always_inline)) -> uint32_t { // first block >= from worth a visit
            for (;;) {
                if (from >= blk_end) return 0u;
                const uint32_t idx = s_first + lane;
                const uint64_t hit = ballot((idx >= from) & (s_ub >= 0.f) & enters(s_ub));
                if (__builtin_expect(hit != 0, 1)) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(hit), fp = f ? f - 1 : 0;
                    o.blk = s_first + f;
                    o.base = o.blk ? bcast(s_e.x, fp) + 1u : 0u;
                    o.ep = o.blk ? bcast(s_e.y, fp) : 0u;
                    return 1u;
                }
                if (s_valid && s_first + 64u >= blk_end) return 0u;
                s_fill(s_valid ? s_first + 63u : (blk_begin ? blk_begin - 1u : 0u)); // (once per 63 blocks)
                s_valid = 1u;
                s_bm_examined += 1;
                s_bytes += 4;
                const uint32_t lo = s_first ? s_first + 1u : 0u; // (lane 0 of a window is the row before its first block, unless that is block 0)
                from = from > lo ? from : lo;
            }
        };

        Blk A{}, B{};            // A: decoded this iteration; B: its gathers are consumed this iteration, then stage C if needed
        uint32_t haveA = 0, haveB = 0, finished = 0, from = blk_begin;
        uint32_t dA0 = 0xFFFFFFFFu, dA1 = 0xFFFFFFFFu, dB0 = 0xFFFFFFFFu, dB1 = 0xFFFFFFFFu; // doc-ids (value lane, lane + 64)
        // (a block's freqs are needed once more only if stage C scores it: they wait in the block's staging buffer, whose bytes
        // are dead once decoded, instead of in four registers)
        float boA0 = 0.f, boA1 = 0.f, boB0 = 0.f, boB1 = 0.f;                                  // freq-only bound of the list-0 term score
        // range-table weight bytes of a lane's two candidates, packed: byte j - 1 = list j (v_cvt_f32_ubyteN unpacks for free);
        // one dword holds lists 1..4, six and more lists take a pair
        using GP = typename std::conditional<(NT > 5), unsigned long long, uint32_t>::type;
        auto gbyte = [](GP g, int j) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(g >> (8 * (j - 1))) & 255u; };
        // staging buffers of list 0 (LDS byte offsets): the block in stage B/C, the block in stage A, the block on its way in
        const uint32_t st_base = rs_lds_offset(&L.stage[0][0]), gb_base = rs_lds_offset(&L.gb[0][0]), xs_base = rs_lds_offset(&L.xs[0][0]);
        const uint32_t fw_base = rs_lds_offset(&L.fw[0]);
        const uint32_t voff = lane * 4u;
        uint32_t bufB = 0, bufA = 1, bufN = 2;
        PT(PH_UNIT);
        for (;;) {
            // ---------------- stage N: the next block worth a visit as things stand now (the heap may still rule it out before its
            // turn), its bytes and side slot requested
            // A's bytes (and the floor word) were requested an iteration ago; the only loads issued after them are B's two gathers
            if (haveA) {
                ++s_rounds;
                if (haveB) rs_wait_vm<2>(); else rs_wait_vm<0>();
                PT(PH_PREFETCH);
                if (shared_floor) adopt_word(uniform(L.fw[0]));
            }
            PT(PH_FLOOR);
            Blk N{};
            const uint32_t haveN = select(from, N);
            if (!haveN) from = blk_end; // (the threshold only rises: what is not worth a visit now never will be)
            PT(PH_STREAM);
            if (haveN) {
                from = N.blk + 1u;
                const uint8_t* const g = rs_uniform_ptr(data0 + N.ep); // (full blocks of a block_optpfor list are dword aligned; a partial last block is not read from here)
                rs_prefetch_blk((const uint8_t*)((uintptr_t)g & ~(uintptr_t)3), st_base + bufN * (STAGE_DW * 4u), rs_uniform_ptr(xs0 + (size_t)XSLOT_DW * N.blk),
                                xs_base + bufN * (XSLOT_DW * 4u), voff);
                if (shared_floor) rs_fetch_word(fwp, fw_base);
                LC(PH_STREAM, lines_of((const uint8_t*)((uintptr_t)g & ~(uintptr_t)3) + 8u * lane, true, 8u));
            }
            if (haveA) {
                // ---------------- stage A: docs and freqs of block A; every posting gets a bound of its OWN list-0 term score from
                // its freq alone. Only the candidates that could enter the heap with that bound + the other lists' maxima ask list
                // 1's table at all (a gather is a cache line per candidate; with the heap warm five candidates in six fall here),
                // and the tests of stage B use the candidate's own bound where round 4 used the block's weight.
                const uint32_t szA = ((A.blk + 1u) * 128u <= n0) ? 128u : (n0 & 127u);
                uint32_t v0, v1, fv0, fv1, consA, consF;
                if (__builtin_expect(szA == 128u, 1)) rs_decode_full(L.stage[bufA], L.xs[bufA], data0 + A.ep, rs_args()->xovf, v0, v1, fv0, fv1, consA, consF);
                else rs_tail(rs_args()->tails, rs_uniform64(qt[0].aux1), szA, v0, v1, fv0, fv1, consA, consF);
                const uint32_t g0 = (lane < szA) ? v0 + 1u : 0u, g1 = (lane + 64 < szA) ? v1 + 1u : 0u;
                const uint32_t i0 = wave_incl_scan(g0);
                const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
                dA0 = (lane < szA) ? A.base + i0 - 1u : 0xFFFFFFFFu;
                dA1 = (lane + 64 < szA) ? A.base + i1 - 1u : 0xFFFFFFFFu;
                boA0 = qw0 * rs_dtw_bound(fv0 + 1u, min_nl);
                boA1 = qw0 * rs_dtw_bound(fv1 + 1u, min_nl);
                L.stage[bufA][lane] = fv0 + 1u; // (same lanes write and read: no fence needed before stage C's read an iteration later)
                L.stage[bufA][lane + 64] = fv1 + 1u;
                ++s_docs_blocks;
                ++s_freqs_blocks;
                s_bm_examined += 1;
                s_bytes += 8 + consA + consF; // block_max + endpoint + both parts
            }
            PT(PH_DOCS);
            if (haveB) {
                // ---------------- stage B: the gathers of block B, issued before stage A ran; the only loads issued after them are
                // those of the prefetch above
                if (haveN) { if (shared_floor) rs_wait_vm<PF_LOADS + 1>(); else rs_wait_vm<PF_LOADS>(); } else rs_wait_vm<0>();
                PT(PH_TOPK);
                const uint32_t x0 = L.gb[0][lane], x1 = L.gb[1][lane]; // the byte fetched ahead: list 1's weight (2 lists) or hint (3, 4 lists)
                GP gP0 = x0, gP1 = x1;
                // (the threshold only rises: a candidate alive now was alive when the gathers were issued, so its byte is there)
                bool ok0 = (dB0 != 0xFFFFFFFFu) & enters((boB0 + rest_all) * BOUND_SLACK) & (x0 != 0u);
                bool ok1 = (dB1 != 0xFFFFFFFFu) & enters((boB1 + rest_all) * BOUND_SLACK) & (x1 != 0u);
                if (hint_first) {
                    ok0 = ok0 & ((x0 == 255u) | (x0 == rmh_code(dB0, rsh[1])));
                    ok1 = ok1 & ((x1 == 255u) | (x1 == rmh_code(dB1, rsh[1])));
                    gP0 = gP1 = 0u;
                    LC(PH_C_SURV1, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                    if constexpr (NT > 2) {
                        if (ballot(ok0) | ballot(ok1)) { // the further lists' hints, all requested before any is tested
                            uint32_t h0[NT] = {}, h1[NT] = {};
                            auto hload = [&](auto jc) __attribute__((always_inline)) {
                                constexpr int j = decltype(jc)::value;
                                const uint8_t* const ht = rt[j] + hdelta;
                                h0[j] = ok0 ? (uint32_t)ht[dB0 >> rsh[j]] : 0u;
                                h1[j] = ok1 ? (uint32_t)ht[dB1 >> rsh[j]] : 0u;
                                LC(PH_MEMBER, lines_of(ht + (dB0 >> rsh[j]), ok0, 1u) + lines_of(ht + (dB1 >> rsh[j]), ok1, 1u));
                            };
                            rs_for<2, NT>(hload);
                            auto htest = [&](auto jc) __attribute__((always_inline)) {
                                constexpr int j = decltype(jc)::value;
                                ok0 = ok0 & (h0[j] != 0u) & ((h0[j] == 255u) | (h0[j] == rmh_code(dB0, rsh[j])));
                                ok1 = ok1 & (h1[j] != 0u) & ((h1[j] == 255u) | (h1[j] == rmh_code(dB1, rsh[j])));
                            };
                            rs_for<2, NT>(htest);
                        }
                    }
                    if (ballot(ok0) | ballot(ok1)) { // every list's weight byte for what is left
                        auto wload = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            gP0 |= (GP)(ok0 ? (uint32_t)rt[j][dB0 >> rsh[j]] : 0u) << (8 * (j - 1));
                            gP1 |= (GP)(ok1 ? (uint32_t)rt[j][dB1 >> rsh[j]] : 0u) << (8 * (j - 1));
                            LC(PH_FREQS, lines_of(rt[j] + (dB0 >> rsh[j]), ok0, 1u) + lines_of(rt[j] + (dB1 >> rsh[j]), ok1, 1u));
                        };
                        rs_for<1, NT>(wload);
                    }
                } else if constexpr (NT > 2) {
                    // lists 2.. : their bytes only for the candidates list 1's byte lets through (list maxima for the others)
                    float rest = 0.f;
                    auto add_max = [&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; rest = rest + rsc[j] * 255.0f; };
                    rs_for_down<NT, 2>(add_max);
                    ok0 = ok0 & enters((boB0 + (rest + rsc[1] * (float)x0)) * BOUND_SLACK);
                    ok1 = ok1 & enters((boB1 + (rest + rsc[1] * (float)x1)) * BOUND_SLACK);
                    if (ballot(ok0) | ballot(ok1)) {
                        auto load_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            gP0 |= (GP)(uint32_t)rt[j][(ok0 ? dB0 : 0u) >> rsh[j]] << (8 * (j - 1));
                            gP1 |= (GP)(uint32_t)rt[j][(ok1 ? dB1 : 0u) >> rsh[j]] << (8 * (j - 1));
                        };
                        rs_for<2, NT>(load_one);
                        auto test_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            ok0 = ok0 & (gbyte(gP0, j) != 0u);
                            ok1 = ok1 & (gbyte(gP1, j) != 0u);
                        };
                        rs_for<2, NT>(test_one);
                    }
                }
                // what the lists after list `after` can add to this lane's two candidates, from their own bytes (summed from the
                // last list down, so that the value for `after` is a prefix of the same chain whatever `after` is)
                auto rest_of = [&](GP g, int after) __attribute__((always_inline)) -> float {
                    float r = 0.f;
                    auto add_one = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        if (j > after) r = r + rsc[j] * (float)gbyte(g, j);
                    };
                    rs_for_down<NT, 1>(add_one);
                    return r;
                };
                float r0 = rest_of(gP0, 0), r1 = rest_of(gP1, 0);
                ok0 = ok0 & enters((boB0 + r0) * BOUND_SLACK);
                ok1 = ok1 & enters((boB1 + r1) * BOUND_SLACK);
                LC(PH_C_VISIT, __builtin_popcountll(ballot(dB0 != 0xFFFFFFFFu)) + __builtin_popcountll(ballot(dB1 != 0xFFFFFFFFu)));
                if (!hint_first) LC(PH_C_SURV1, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                if (!hint_first && hdelta && (ballot(ok0) | ballot(ok1))) {
                    // membership hints (BatchArgs::rmh): a weight byte only says that SOME posting of list j lies in the candidate's
                    // range; where that range holds exactly one posting its hint byte says which. A candidate at another offset is
                    // not in the list -- settled here, by one more byte, instead of by a block search and a block decode in stage C.
                    auto hint_one = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        const uint8_t* const ht = rt[j] + hdelta;
                        const bool hashint = rsh[j] != 0u; // (one doc-id per entry: the weight byte was the answer)
                        const uint32_t h0 = (ok0 & hashint) ? (uint32_t)ht[dB0 >> rsh[j]] : 255u, h1 = (ok1 & hashint) ? (uint32_t)ht[dB1 >> rsh[j]] : 255u;
                        LC(PH_MEMBER, lines_of(ht + (dB0 >> rsh[j]), ok0 & hashint, 1u) + lines_of(ht + (dB1 >> rsh[j]), ok1 & hashint, 1u));
                        ok0 = ok0 & ((h0 == 255u) | (h0 == rmh_code(dB0, rsh[j])));
                        ok1 = ok1 & ((h1 == 255u) | (h1 == rmh_code(dB1, rsh[j])));
                    };
                    rs_for<1, NT>(hint_one);
                }
                LC(PH_C_SURV2, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                PT(PH_MEMBER);
                if (__builtin_expect((ballot(ok0) | ballot(ok1)) != 0, 0)) {
                    LC(PH_C_LIVEROUNDS, 1);
                    // ---------------- stage C: somebody of block B may enter the heap: norm_len, exact list-0 score
                    const float* const norm_lens = rs_args()->norm_lens;
                    const uint8_t* const arena = rs_args()->arena;
                    // (list 1's rows after its current block go out together with the norm_lens: the search they serve comes first
                    // in the probe below and does not depend on which candidate it is for)
                    const Rows rows1 = rows_load((const uint2*)rs_args()->skip + cget(C_BB), rs_args()->bmw + cget(C_BB), (cget(C_N) + 127u) >> 7, cget(C_CUR) + 1u);
                    const float nl0 = ok0 ? norm_lens[dB0] : 1.f, nl1 = ok1 ? norm_lens[dB1] : 1.f;
                    LC(PH_SCORE, lines_of(norm_lens + dB0, ok0, 4u) + lines_of(norm_lens + dB1, ok1, 4u));
                    const uint32_t fB0 = L.stage[bufB][lane], fB1 = L.stage[bufB][lane + 64];
                    float pa0 = qw0 * doc_term_weight(fB0, nl0), pa1 = qw0 * doc_term_weight(fB1, nl1);
                    {
                        const uint32_t nv = (uint32_t)(__builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                        s_scored += nv;
                        s_bytes += 4ull * nv;
                    }
                    ok0 = ok0 & enters((pa0 + r0) * BOUND_SLACK);
                    ok1 = ok1 & enters((pa1 + r1) * BOUND_SLACK);
                    // lists 1 .. NT-1 in order: a candidate moves on only while partial score + what the later lists can add to IT
                    // can still enter the heap
                    auto probe = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int CB = (j - 1) * C_PER; // this list's lanes of `cold`
                        uint64_t todo0 = ballot(ok0), todo1 = ballot(ok1);
                        if (!(todo0 | todo1)) return;
                        const uint32_t nj = cget(CB + C_N), nbj = (nj + 127u) >> 7, bbj = cget(CB + C_BB);
                        const uint32_t vlj = 1u + (nj >= (1u << 7)) + (nj >= (1u << 14)) + (nj >= (1u << 21)) + (nj >= (1u << 28));
                        const uint8_t* const dataj = arena + (((unsigned long long)cget(CB + C_LOHI) << 32) | cget(CB + C_LOLO)) + vlj + 4ull * nbj + 4ull * (nbj - 1);
                        const uint2* const tabj = (const uint2*)rs_args()->skip + bbj;
                        const float* const wtabj = rs_args()->bmw + bbj;
                        const float qwj = __uint_as_float(cget(CB + C_QW));
                        bool rows_fresh = j == 1; // (rows1 was loaded for from = list 1's current block + 1)
                        const float rj0 = rest_of(gP0, j), rj1 = rest_of(gP1, j); // the lists after j
                        const float bj0 = rsc[j] * (float)gbyte(gP0, j), bj1 = rsc[j] * (float)gbyte(gP1, j);
                        uint32_t* const dj = L.dj[j - 1];
                        uint32_t* const fj = L.fj[j - 1];
                        bool mem0 = false, mem1 = false;
                        uint32_t curj = cget(CB + C_CUR), bmj = cget(CB + C_BMAX);
                        while (todo0 | todo1) {
                            const uint32_t amin = todo0 ? bcast(dB0, (uint32_t)__builtin_ctzll(todo0)) : bcast(dB1, (uint32_t)__builtin_ctzll(todo1));
                            if (curj == 0xFFFFFFFFu || amin > bmj) {
                                Found fb;
                                const bool found = find_block_rows(tabj, wtabj, nbj, curj + 1u, amin, fb, rows_fresh ? rows1 : rows_load(tabj, wtabj, nbj, curj + 1u));
                                LC(PH_FIND, 1);
                                if (!found) { // list j has nothing >= amin: no later document of list 0 can be a result either
                                    s_bm_examined += 1;
                                    s_bytes += 4;
                                    finished = 1;
                                    break;
                                }
                                s_bm_examined += (curj == 0xFFFFFFFFu) ? 1u : fb.blk - curj;
                                s_bytes += 4ull * ((curj == 0xFFFFFFFFu) ? 1u : fb.blk - curj);
                                // before the block is decoded: partial + min(block weight, own byte) + later lists, per candidate inside it
                                const float cbw = qwj * fb.w;
                                const bool in0 = ((todo0 >> lane) & 1) && dB0 <= fb.bmax, in1 = ((todo1 >> lane) & 1) && dB1 <= fb.bmax;
                                const float t0 = bj0 < cbw ? bj0 : cbw, t1 = bj1 < cbw ? bj1 : cbw;
                                const bool can0 = in0 & enters(((pa0 + t0) + rj0) * BOUND_SLACK);
                                const bool can1 = in1 & enters(((pa1 + t1) + rj1) * BOUND_SLACK);
                                if (!(ballot(can0) | ballot(can1))) { // nobody inside the block can enter: it is not decoded
                                    todo0 &= ~ballot(in0);
                                    todo1 &= ~ballot(in1);
                                    continue; // (the list stays where it was: the next search restarts there, with the same rows)
                                }
                                rows_fresh = false;
                                const uint8_t* pb = dataj + fb.ep;
                                LC(PH_C_BDOCS, 1);
                                LC(PH_DOCS, lines_of(pb + 8u * lane, true, 8u));
                                const uint32_t szb = ((fb.blk + 1) * 128u <= nj) ? 128u : (nj & 127u);
                                uint32_t v0, v1, w0, w1, consD, consF2;
                                if (__builtin_expect(szb == 128u, 1)) { // (full blocks of a block_optpfor list are dword aligned)
                                    rs_stage_block((const uint32_t*)pb, rs_args()->xslots + (size_t)XSLOT_DW * (bbj + fb.blk), L.stb, L.xsb);
                                    rs_decode_full(L.stb, L.xsb, pb, rs_args()->xovf, v0, v1, w0, w1, consD, consF2);
                                } else {
                                    rs_tail(rs_args()->tails, ((unsigned long long)cget(CB + C_TLHI) << 32) | cget(CB + C_TLLO), szb, v0, v1, w0, w1, consD, consF2);
                                }
                                const uint32_t g0 = (lane < szb) ? v0 + 1u : 0u, g1 = (lane + 64 < szb) ? v1 + 1u : 0u;
                                const uint32_t i0 = wave_incl_scan(g0);
                                const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
                                dj[lane] = (lane < szb) ? fb.base + i0 - 1u : 0xFFFFFFFFu;
                                dj[lane + 64] = (lane + 64 < szb) ? fb.base + i1 - 1u : 0xFFFFFFFFu;
                                fj[lane] = w0 + 1u;
                                fj[lane + 64] = w1 + 1u;
                                wave_sync();
                                curj = fb.blk;
                                bmj = fb.bmax;
                                cset(CB + C_CUR, curj);
                                cset(CB + C_BMAX, bmj);
                                ++s_docs_blocks;
                                ++s_freqs_blocks;
                                s_bytes += 4 + consD + consF2;
                            }
                            // candidates inside the block: members or not, settled now
                            const bool in0 = ((todo0 >> lane) & 1) && dB0 <= bmj, in1 = ((todo1 >> lane) & 1) && dB1 <= bmj;
                            const uint64_t ib0 = ballot(in0), ib1 = ballot(in1);
                            uint32_t q0 = 0, q1 = 0;
                            bool m0, m1;
                            if (__builtin_popcountll(ib0) + __builtin_popcountll(ib1) > 16) {
                                m0 = rs_member(dj, dB0, in0, q0);
                                m1 = rs_member(dj, dB1, in1, q1);
                            } else { // few candidates: broadcast each, two equality ballots over the block
                                const uint32_t e0 = dj[lane], e1 = dj[lane + 64];
                                uint64_t r0m = 0, r1m = 0;
                                for (int half = 0; half < 2; ++half) {
                                    uint64_t td = half ? ib1 : ib0;
                                    while (td) {
                                        const uint32_t src = (uint32_t)__builtin_ctzll(td);
                                        td &= td - 1;
                                        const uint32_t c = bcast(half ? dB1 : dB0, src);
                                        const uint64_t h0 = ballot(e0 == c), h1 = ballot(e1 == c);
                                        if (h0 | h1) {
                                            const uint32_t pp = h0 ? (uint32_t)__builtin_ctzll(h0) : 64u + (uint32_t)__builtin_ctzll(h1);
                                            if (half) { r1m |= 1ull << src; if (lane == src) q1 = pp; }
                                            else { r0m |= 1ull << src; if (lane == src) q0 = pp; }
                                        }
                                    }
                                }
                                m0 = (r0m >> lane) & 1;
                                m1 = (r1m >> lane) & 1;
                            }
                            todo0 &= ~ib0;
                            todo1 &= ~ib1;
                            // members take list j's term score at once
                            if (m0) { pa0 = pa0 + qwj * doc_term_weight(fj[q0], nl0); mem0 = true; }
                            if (m1) { pa1 = pa1 + qwj * doc_term_weight(fj[q1], nl1); mem1 = true; }
                        }
                        // members whose score can still enter go on to the next list (a candidate the loop left unsettled -- list j
                        // ended below it -- is not a member)
                        ok0 = ok0 & mem0 & enters((pa0 + rj0) * BOUND_SLACK);
                        ok1 = ok1 & mem1 & enters((pa1 + rj1) * BOUND_SLACK);
                    };
                    rs_for<1, NT>(probe);
                    // pa0 / pa1 are complete scores of documents of the intersection now
                    uint32_t inserted = 0;
                    for (int half = 0; half < 2; ++half) {
                        const float sc = half ? pa1 : pa0;
                        uint64_t todo = ballot((half ? ok1 : ok0) & enters(sc));
                        while (todo) {
                            const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                            todo &= todo - 1;
                            const float v = __uint_as_float(bcast(__float_as_uint(sc), src));
                            LC(PH_C_HEAP, 1);
                            if (tk.insert(v)) {
                                refresh();
                                inserted = 1;
                                if (shared_floor && lane == 0) sh.add(v);
                            }
                        }
                    }
                    if (shared_floor && inserted) { // the histogram moved: what floor does it imply now
                        const float f = sh.floor(tk.k);
                        if (f > 0.f) {
                            if (lane == 0) __hip_atomic_fetch_max(fwp, __float_as_uint(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (f > tk.floor) { tk.floor = f; refresh(); }
                        }
                    }
                }
            }
            PT(PH_SCORE);
            if (__builtin_expect(finished, 0)) break;
            // ---------------- rotate: A becomes B (its gathers are issued now and consumed an iteration later, behind the next
            // block's decode), the block whose bytes were requested becomes A
            B = A;
            haveB = haveA;
            dB0 = dA0;
            dB1 = dA1;
            boB0 = boA0;
            boB1 = boA1;
            if (haveB) {
                // only the candidates whose own bound + the other lists' maxima can still enter the heap ask list 1's table (the others
                // read entry 0: one shared line); a block without any such candidate is done. One byte per candidate from list 1 (the
                // other lists' bytes are fetched in stage B for the candidates inside list 1's ranges only: most die at list 1).
                const bool al0 = (dB0 != 0xFFFFFFFFu) & enters((boB0 + rest_all) * BOUND_SLACK), al1 = (dB1 != 0xFFFFFFFFu) & enters((boB1 + rest_all) * BOUND_SLACK);
                haveB = (ballot(al0) | ballot(al1)) != 0 ? 1u : 0u;
                if (haveB) {
                    LC(PH_C_ALIVE, __builtin_popcountll(ballot(al0)) + __builtin_popcountll(ballot(al1)));
                    LC(PH_C_GBLOCKS, 1);
                    rs_gather_u8(gt1, (al0 ? dB0 : 0u) >> rsh[1], gb_base);
                    rs_gather_u8(gt1, (al1 ? dB1 : 0u) >> rsh[1], gb_base + 256u);
                    LC(PH_TOPK, lines_of(gt1 + (dB0 >> rsh[1]), al0, 1u) + lines_of(gt1 + (dB1 >> rsh[1]), al1, 1u) + 2u);
                }
            }
            PT(PH_PROBE);
            A = N;
            haveA = haveN;
            const uint32_t t = bufB;
            bufB = bufA;
            bufA = bufN;
            bufN = t;
            if (!(haveA | haveB)) break;
        }
        PT(PH_TOTAL);
        rs_wait_vm<0>(); // (a unit left early -- list exhausted -- may still have a prefetch or gathers in flight)
#undef cget
#undef cset
        KArgs r = rs_args();
        if constexpr (STATS) {
            unsigned long long* const clk = r->unit_clock;
            if (clk && lane == 0) clk[2ull * uid + 1] = wall_clock64();
        }
        if (whole) {
            if (lane == 0) { r->out_count[q] = tk.n; if (r->out_freq_sum) r->out_freq_sum[q] = 0; }
            store_topk_rs(r->out_topk, r->out_topk_len, tk.k, q, tk);
        } else {
            if (lane == 0) { r->unit_count[uid] = tk.n; r->unit_freq_sum[uid] = 0; }
            store_topk_rs(r->unit_topk, r->unit_topk_len, tk.k, uid, tk);
        }
    }
    Stats* const stats = rs_args()->stats;
    if (STATS && stats && lane == 0) {
        atomicAdd(&stats->docs_blocks, (unsigned long long)s_docs_blocks);
        atomicAdd(&stats->freqs_blocks, (unsigned long long)s_freqs_blocks);
        atomicAdd(&stats->block_max_examined, (unsigned long long)s_bm_examined);
        atomicAdd(&stats->algorithmic_bytes, (unsigned long long)s_bytes);
        atomicAdd(&stats->postings_scored, (unsigned long long)s_scored);
        atomicAdd(&stats->rounds, (unsigned long long)s_rounds);
#ifdef DS2I_LINE_COUNT
        for (int i = 0; i < PH_COUNT; ++i) if (lc[i]) atomicAdd(&stats->phase_cycles[i], lc[i]);
#endif
#ifdef DS2I_RS_PHASE
        for (int i = 0; i < PH_COUNT; ++i) if (pt[i]) atomicAdd(&stats->phase_cycles[i], pt[i]);
#endif
    }
#undef LC
#undef PT
}

} // namespace

extern "C" {
// nt = exact number of distinct terms of every query of the launch (2..8; the planner's DS2I_STREAM_NT_MAX caps it); the caller has checked that the index is
// block_optpfor with skip table, block weights, range tables and side slots, and that k <= 64
hipError_t ds2i_launch_ranked_stream(int nt, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    const bool st = a.stats != nullptr;
    switch (nt) {
    case 2: if (st) hipLaunchKernelGGL((k_ranked_stream<2, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<2, false>), g, b, 0, s, a); break;
    case 3: if (st) hipLaunchKernelGGL((k_ranked_stream<3, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<3, false>), g, b, 0, s, a); break;
    case 4: if (st) hipLaunchKernelGGL((k_ranked_stream<4, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<4, false>), g, b, 0, s, a); break;
    // 5..8: the 5..8-term class (capi_batch.cpp: up to DS2I_STREAM_NT_MAX lists, default 8)
    case 5: if (st) hipLaunchKernelGGL((k_ranked_stream<5, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<5, false>), g, b, 0, s, a); break;
    case 6: if (st) hipLaunchKernelGGL((k_ranked_stream<6, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<6, false>), g, b, 0, s, a); break;
    case 7: if (st) hipLaunchKernelGGL((k_ranked_stream<7, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<7, false>), g, b, 0, s, a); break;
    case 8: if (st) hipLaunchKernelGGL((k_ranked_stream<8, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<8, false>), g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}
