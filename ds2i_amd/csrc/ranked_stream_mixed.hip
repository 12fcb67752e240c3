// ranked_and on a block_MIXED index with the upload-time pruning tables, as a software-pipelined stream (round 4's kernel, kept
// for the index kind whose blocks need the general decoders; block_optpfor has its own, ranked_stream.hip)
// (gfx950 / CDNA4, wave64; one wavefront per work unit, wave-uniform control flow, no MFMA: integer work).
//
// Replaces ranked_and_query (reference queries.hpp:322-401: candidate = next posting of the shortest list, next_geq() on
// every other list, score = sum of bm25 term scores in list order, topk_queue::insert 157-172) for queries of exactly
// NT = 2..4 distinct terms. Same results as k_conjunctive<true, ...> (kernels.hip), which stays the kernel of every
// other case (other codecs, no tables, 1 term, 5+ terms); what differs is how a unit is executed:
//
//   * a block of the driving list (list 0, the shortest) is handled exactly ONCE, in three stages that belong to
//     three different blocks at any moment:
//         stage A (block i+1)  bytes already requested -> LDS; bytes of block i+2 requested; OptPFor docs decode
//         stage B (block i)    its range-table gathers, issued before stage A ran, are consumed: zero byte = the document
//                              is in no intersection; otherwise block weight + own bytes against the heap threshold
//         stage C (block i)    only if somebody survived: freqs of the block, freq-only bound, norm_len, exact list-0
//                              score, then list 1 .. NT-1 in order (locate block -> block-weight test -> decode ->
//                              membership -> freq -> score), heap insert
//         then the gathers of block i+1 are issued and the roles rotate,
//     so the gather round trip of a block is covered by the decode of the next one and the block-bytes round trip by a
//     whole iteration -- two blocks of the driving list are in flight per wave, not one;
//   * no enumerator object: the driving list's state is the 64-row table window in registers (as in k_conjunctive's
//     stream), a block's decoded doc-ids and freqs stay in the registers of the lanes that own them (value i in lane
//     i & 63, slot i >> 6), the other lists keep one decoded block each in LDS with a three-scalar tag.
//
// Every pruning test is a true upper bound of the float32 score the scoring code would compute (device_score.hpp,
// BOUND_SLACK), and topk_queue::insert is strict, so the heap ends with the same multiset of scores as the sequential
// traversal, bit for bit (tests/test_gpu.py: test_ranked_and_pruning_fuzz_bit_identical).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_enum.hpp"
#include "device_score.hpp"

using namespace ds2i_dev;

namespace {

#ifndef DS2I_RS_FLOOR_EVERY
#define DS2I_RS_FLOOR_EVERY 4 // power of two: the shared histogram is consulted every n-th visited block
#endif
static_assert(DS2I_RS_FLOOR_EVERY > 0 && (DS2I_RS_FLOOR_EVERY & (DS2I_RS_FLOOR_EVERY - 1)) == 0, "DS2I_RS_FLOOR_EVERY is used as a mask: power of two");
#ifndef DS2I_RS_OCC2
#define DS2I_RS_OCC2 6
#endif
#ifndef DS2I_RS_OCC4
#define DS2I_RS_OCC4 5
#endif
#define RS_WAVES(NT) ((NT) <= 2 ? DS2I_RS_OCC2 : DS2I_RS_OCC4)

template <int NT>
struct LdsRS {
    uint32_t stage[3][STAGE_DW]; // list 0: bytes of the blocks in stage B/C, in stage A and on their way in (LDS-DMA)
    uint32_t gb[2][64];          // list 1's range-table byte of every posting of the block in stage B (LDS-DMA, one dword per lane)
    uint32_t stb[STAGE_DW];      // bytes of the other lists' block decoded last
    uint32_t dj[NT - 1][128];    // lists 1 .. NT-1: doc-ids of their current block
    uint32_t fj[128];            // freqs of the block of the list that decoded its freqs last (f_owner)
    uint32_t out[128];           // decoder scratch (OptPFor exception scatter, interpolative prefix sums)
    uint32_t exc[EXC_LDS_DW];    // Simple16 scratch + field table
};

// first block >= from of a list whose block_max >= lb, with its table words; rows = the list's interleaved skip table
// ({block_max, end offset} per block), wtab = its block weights. 64 rows per probe: the 64 after `from`, then a 64-ary
// search (the reference scans block_max linearly, block_posting_list.hpp:134-137).
struct Found { uint32_t blk, bmax, base, ep; float w; };
DS2I_DEV bool find_block_rows(const uint2* tab, const float* wtab, uint32_t nb, uint32_t from, uint32_t lb, Found& o) {
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
    {   // rows from-1 .. from+62 (lane 0 = the block before `from`, never a candidate itself)
        const uint32_t first = from ? from - 1 : 0;
        const uint32_t idx = first + lane;
        uint2 e = make_uint2(0xFFFFFFFFu, 0u);
        if (idx < nb) { e = tab[idx]; wv = wtab[idx]; }
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
// kernel entry and keeps them in SGPRs for the kernel's lifetime (the old kernels sit at the 102-SGPR limit with three
// VGPRs of spilled scalars because of it). Here the kernarg segment is addressed explicitly: the few hot fields are read
// where a unit starts, the cold ones at their use site through a pointer the optimiser cannot see through (so the loads
// stay where they are written instead of being hoisted above the loops).
typedef const BatchArgs __attribute__((address_space(4))) * KArgs; // (constant address space: uniform reads are s_load)
DS2I_DEV KArgs rs_args() {
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// ---- loads the compiler must not count. hipcc drains vmcnt to 0 wherever control flow joins with a load pending on
// some path, which would put every round trip back on the critical path; these are issued and waited for by hand.
// (i) block bytes: LDS-DMA, global -> LDS with no register in between (nothing the compiler could copy or spill early);
DS2I_DEV uint32_t rs_lds_offset(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
// 512 bytes at g (4-byte aligned) -> LDS byte offset `lds`; voff = lane * 4. M0 is the DMA's LDS base: compiler-reserved,
// so it is saved, set and restored inside each statement.
DS2I_DEV void rs_prefetch512(const uint8_t* g, uint32_t lds, uint32_t voff) {
    uint32_t keep;
    // (the instruction offset moves the global AND the LDS address: measured, profiles/probes/ldsdma_probe.hip)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\tglobal_load_lds_dword %1, %2 offset:256\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(uniform(lds)) : "memory");
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

// optpfor / interpolative block -> gaps or freqs-1 in (v0, v1), value i in lane i & 63, slot i >> 6. The common case
// (full block inside the staged 512 bytes) never touches global memory; anything else takes the general decoder and is
// made opaque, so that no output of this function is ever "pending on vmcnt" for the compiler: the caller's prefetches
// and gathers stay in flight across it.
template <int CODEC>
DS2I_DEV uint32_t rs_decode(uint32_t* st, const uint8_t* p, uint32_t sum, uint32_t n, uint32_t* out, uint32_t* exc, uint32_t& v0, uint32_t& v1) {
    uint32_t consumed = 0;
    const uint32_t woff = (uint32_t)((uintptr_t)p & 3u);
    if constexpr (CODEC == CODEC_OPTPFOR) {
        if (__builtin_expect(n == 128u && woff == 0u && optpfor_decode_lds(st, STAGE_DW, exc, out, v0, v1, consumed), 1)) return consumed;
    }
    Window w{(const uint8_t*)((uintptr_t)p & ~(uintptr_t)3), STAGE_DW * 4u, st};
    uint32_t a0, a1;
    consumed = uniform(decode_block<CODEC>(CODEC, w, p, sum, n, out, exc, a0, a1));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");
    v0 = a0;
    v1 = a1;
    return consumed;
}

// CODEC: CODEC_OPTPFOR (block_optpfor) or CODEC_MIXED (block_mixed: a type byte in front of every full block; its OptPFor
// blocks are then not dword aligned and, like its VarInt-G8IU and interpolative blocks, take the general decoders)
#ifndef RS_HINT_FIRST
#define RS_HINT_FIRST(nt) ((nt) > 2)
#endif
template <int NT, bool STATS, int CODEC = CODEC_OPTPFOR>
__global__ void __launch_bounds__(64, RS_WAVES(NT)) k_ranked_stream_mixed(BatchArgs a_unused) {
    static_assert(NT >= 2 && NT <= 4, "exact list counts 2..4");
    __shared__ LdsRS<NT> L;
    const uint32_t lane = lane_id();
    s16_table_init(L.exc);
    typename std::conditional<STATS, uint32_t, NullCounter>::type s_docs_blocks, s_freqs_blocks, s_bm_examined, s_scored, s_rounds;
    typename std::conditional<STATS, unsigned long long, NullCounter>::type s_bytes;
    s_docs_blocks = s_freqs_blocks = s_bm_examined = s_scored = s_rounds = 0;
    s_bytes = 0;
#ifdef DS2I_LINE_COUNT
    // diagnostic build: distinct 128-byte lines requested by the hand-placed gathers, by purpose (reported through Stats::phase_cycles)
    unsigned long long lc[PH_COUNT] = {};
    // lines touched by one wave instruction whose active lanes read `bytes` bytes at ascending addresses
    auto lines_of = [&](const void* addr, bool active, uint32_t bytes) -> uint32_t {
        const unsigned long long lo = (unsigned long long)(uintptr_t)addr >> 7, hi = ((unsigned long long)(uintptr_t)addr + bytes - 1) >> 7;
        const uint64_t act = ballot(active);
        // previous ACTIVE lane's last line
        unsigned long long prev_hi = ~0ull;
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
        const uint32_t uid = a->order[tkt];
        const Unit u = a->units[uid];
        if constexpr (STATS) { // diagnostic (DS2I_UNIT_CLOCK=1): when the unit started / ended
            unsigned long long* const clk = a->unit_clock;
            if (clk && lane == 0) clk[2ull * uid] = wall_clock64();
        }
        const uint32_t q = u.q;
        const bool whole = u.nparts == 1;
        const QTerm* const qt = a->qterms + a->q_off[q]; // exactly NT terms (the planner's launch groups)
        TopK tk;
        tk.init(a->k);
        // ---- list 0: the stream
        const uint32_t n0 = qt[0].n, nb0 = (n0 + 127u) >> 7;
        const uint32_t vl0 = 1u + (n0 >= (1u << 7)) + (n0 >= (1u << 14)) + (n0 >= (1u << 21)) + (n0 >= (1u << 28));
        const uint8_t* const data0 = a->arena + qt[0].list_off + vl0 + 4ull * nb0 + 4ull * (nb0 - 1);
        const float qw0 = qt[0].q_weight;
        // ---- lists 1 .. NT-1: range table (hot), the rest of the QTerm is read when a candidate gets that far
        const uint8_t* rt[NT];
        uint32_t rsh[NT];
        float rsc[NT];
        auto bind_one = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            rt[j] = a->rmw + 64ull * qt[j].rmw_off64;
            rsh[j] = qt[j].rmw_shift;
            rsc[j] = qt[j].rmw_scale;
        };
        rs_for<1, NT>(bind_one);
        const long long hdelta = a->rmh ? (long long)(a->rmh - a->rmw) : 0ll; // hint of an entry = the byte at the same offset of the parallel buffer
        // Three and four lists: the byte fetched ahead for every candidate is list 1's HINT, not its weight. A weight byte lets a
        // candidate through whenever its range holds any posting (one candidate in 4..6, each then costing a line per further
        // list); the hint also settles the ranges with a single posting, so that the further lists are asked about a few per cent
        // of the candidates only -- first their hints, then, for what is left, every list's weight for the threshold test.
        // (Two lists: the weight stays first, its threshold test removes more than the hint does.)
        const bool hint_first = RS_HINT_FIRST(NT) && hdelta != 0;
        const uint8_t* const gt1 = hint_first ? rt[1] + hdelta : rt[1];
        // block of list j whose doc-ids are in L.dj[j-1] (cur = ~0: none), its block_max, size and the arena offset of its
        // freqs part; f_owner = list whose current block's freqs are in L.fj (0 = nobody); stb_owner = list whose current
        // block's bytes are in L.stb (from stb_base on). Only stage C touches them: they live in the lanes of one VGPR
        // (v_readlane / v_writelane at a constant lane) instead of ~20 SGPRs the hot loop would have to carry.
        uint32_t cold = 0xFFFFFFFFu;
        enum { C_CUR = 0, C_BMAX = 1, C_SZ = 2, C_FOLO = 3, C_FOHI = 4, C_PER = 5, C_FOWNER = 60, C_SOWNER = 61, C_SBLO = 62, C_SBHI = 63 };
#define cget(l) ((uint32_t)__builtin_amdgcn_readlane((int)cold, (l)))
#define cset(l, v) rs_writelane<(l)>(cold, (v))
        cset(C_FOWNER, 0u);
        cset(C_SOWNER, 0u);
        // ---- pruning state: the parts of a split query share a score histogram (device_score.hpp)
        unsigned int* const q_hist = a->q_hist;
        const bool shared_floor = !whole && q_hist;
        ScoreHist sh;
        sh.init(shared_floor ? q_hist : nullptr, shared_floor ? a->q_hist_slot[q] : 0u, shared_floor ? qt[0].max_bmw + qt[0].suf_bmw : 0.f,
                1.0f - 1.0f / 1048576.0f);
        // can a score enter the heap: s >= floor && (heap not full || s > k-th score) (TopK::would_enter), branch-free on two
        // wave-uniform values that are refreshed whenever the heap or the floor changes
        float e_floor = tk.floor, e_gt = -__builtin_inff();
        auto refresh = [&]() __attribute__((always_inline)) { e_floor = tk.floor; e_gt = tk.n < tk.k ? -__builtin_inff() : tk.thr; };
        auto enters = [&](float s) __attribute__((always_inline)) -> bool { return (s >= e_floor) & (s > e_gt); };
        auto adopt_floor = [&]() __attribute__((always_inline)) {
            const float f = sh.floor(tk.k);
            if (f > tk.floor) tk.floor = f;
            refresh();
        };
        if (shared_floor) adopt_floor();
        uint32_t floor_tick = 1;
        // ---- the 64-row window of list 0's table (lane j: row s_first + j; lane 0 is the row before the first block the window
        // can serve, unless that is block 0), its block weights and, per row, what the other lists can add to a document of
        // that block: for each of them the largest range-table entry over the block's own doc-id span (read from the level
        // whose entries are wide enough for <= 16 of them to cover the span). s_dead: some other list has no posting in the span.
        uint32_t s_first = 0;
        uint2 s_e = make_uint2(0xFFFFFFFFu, 0u);
        float s_wq = 0.f, s_ub = 0.f; // q_weight x block weight; (that + the other lists' span maxima) x slack, -1 = nothing to intersect

        auto s_fill = [&](uint32_t first) __attribute__((always_inline)) {
            s_first = first;
            const uint32_t idx = first + lane;
            s_e = make_uint2(0xFFFFFFFFu, 0u);
            float s_w = 0.f;
            {   // (once per 63 blocks: the table pointers are re-derived here rather than carried through the loop)
                const uint32_t bb = qt[0].blk_base;
                const uint2* const tab0 = (const uint2*)rs_args()->skip + bb;
                const float* const w0tab = rs_args()->bmw + bb;
                if (idx < u.blk_end) { s_e = tab0[idx]; s_w = w0tab[idx]; }
                LC(PH_PROLOG, lines_of(tab0 + idx, idx < u.blk_end, 8u) + lines_of(w0tab + idx, idx < u.blk_end, 4u));
            }
            const uint32_t prev_max = (uint32_t)__shfl_up((int)s_e.x, 1);
            const uint32_t base = (lane == 0) ? 0u : prev_max + 1u, top = s_e.x;
            const bool row = idx < u.blk_end && (lane > 0 || idx == 0) && top != 0xFFFFFFFFu && base <= top;
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
                const uint32_t m = max_of_bytes16(rt[j] + g.off[lvl] + (fits ? lo : 0u), fits ? hi - lo + 1u : 1u);
                LC(PH_PROLOG, lines_of(rt[j] + g.off[lvl] + (fits ? lo : 0u), true, 16u));
                const uint32_t best = (row && fits) ? m : 255u; // (255 = the list maximum)
                dead = dead || best == 0u;
                acc = acc + rsc[j] * (float)best;
            };
            rs_for_down<NT, 1>(one_list);
            s_wq = qw0 * s_w;
            s_ub = dead ? -1.0f : (s_wq + acc) * BOUND_SLACK; // (scores are >= 0: -1 never enters)
        };
        auto s_live = [&](uint32_t from) __attribute__((always_inline)) -> uint64_t {
            const uint32_t idx = s_first + lane;
            const bool ok = (idx >= from) & (idx < u.blk_end) & ((lane > 0) | (idx == 0)) & (s_ub >= 0.f) & enters(s_ub);
            return ballot(ok);
        };
        // a block of list 0 on its way through the stages
        struct Blk { uint32_t blk, bmax, base, ep; float wq; };
        auto select = [&](uint32_t from, Blk& o) __attribute__((always_inline)) -> bool { // first block >= from worth a visit
            for (;;) {
                if (from >= u.blk_end) return false;
                const uint64_t hit = s_live(from);
                if (__builtin_expect(hit != 0, 1)) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(hit), fp = f ? f - 1 : 0;
                    o.blk = s_first + f;
                    o.bmax = bcast(s_e.x, f);
                    o.base = o.blk ? bcast(s_e.x, fp) + 1u : 0u;
                    o.ep = o.blk ? bcast(s_e.y, fp) : 0u;
                    o.wq = __uint_as_float(bcast(__float_as_uint(s_wq), f));
                    return true;
                }
                if (s_first + 64 >= u.blk_end) return false;
                s_fill(s_first + 63); // (once per 63 blocks)
                from = from > s_first + 1 ? from : s_first + 1;
            }
        };
        s_fill(u.blk_begin ? u.blk_begin - 1 : 0);
        s_bm_examined += 1;
        s_bytes += 4;

        Blk A{}, B{};            // A: decoded this iteration; B: its gathers are consumed this iteration, then stage C if needed
        bool haveA, haveB = false;
        uint32_t dA0 = 0xFFFFFFFFu, dA1 = 0xFFFFFFFFu, dB0 = 0xFFFFFFFFu, dB1 = 0xFFFFFFFFu; // doc-ids (value lane, lane + 64)
        uint32_t gB0[NT] = {}, gB1[NT] = {};                                                    // range-table bytes of lists 1..
        uint32_t consA = 0, consB = 0, szA = 0, szB = 0; // bytes of the docs part, postings of the block
        // staging buffers of list 0 (LDS byte offsets): the block in stage B/C, the block in stage A, the block on its way in
        const uint32_t st_base = rs_lds_offset(&L.stage[0][0]), gb_base = rs_lds_offset(&L.gb[0][0]);
        uint32_t bufB = 0, bufA = 1, bufN = 2;
        const uint32_t voff = lane * 4u;
        bool finished = false;
        haveA = select(u.blk_begin, A);
        if (haveA) rs_prefetch512((const uint8_t*)((uintptr_t)(data0 + A.ep) & ~(uintptr_t)3), st_base + bufA * (STAGE_DW * 4u), voff);
        while (haveA || haveB) {
            Blk N{};
            bool haveN = false;
            if (haveA) {
                // ---------------- stage A: the bytes of the block after A requested, A's docs decoded
                ++s_rounds;
                if (shared_floor && (floor_tick++ & (DS2I_RS_FLOOR_EVERY - 1)) == 0) adopt_floor();
                haveN = select(A.blk + 1, N); // as things stand now: the heap may still rule it out before its turn
                // A's bytes were requested an iteration ago; the only loads issued after them are B's two gathers
                if (haveB) rs_wait_vm<2>(); else rs_wait_vm<0>();
                if (haveN) rs_prefetch512((const uint8_t*)((uintptr_t)(data0 + N.ep) & ~(uintptr_t)3), st_base + bufN * (STAGE_DW * 4u), voff);
                if (haveN) LC(PH_STREAM, lines_of((const uint8_t*)((uintptr_t)(data0 + N.ep) & ~(uintptr_t)3) + 8u * lane, true, 8u));
                szA = ((A.blk + 1) * 128u <= n0) ? 128u : (n0 & 127u);
                uint32_t v0, v1;
                consA = rs_decode<CODEC>(L.stage[bufA], data0 + A.ep, A.bmax - A.base - (szA - 1), szA, L.out, L.exc, v0, v1);
                const uint32_t g0 = (lane < szA) ? v0 + 1u : 0u, g1 = (lane + 64 < szA) ? v1 + 1u : 0u;
                const uint32_t i0 = wave_incl_scan(g0);
                const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
                dA0 = (lane < szA) ? A.base + i0 - 1u : 0xFFFFFFFFu;
                dA1 = (lane + 64 < szA) ? A.base + i1 - 1u : 0xFFFFFFFFu;
                ++s_docs_blocks;
                s_bm_examined += 1;
                s_bytes += 8 + consA; // block_max + endpoint + docs part
            }
            if (haveB) {
                // ---------------- stage B: the gathers of block B, issued before stage A ran; the only loads issued after them are
                // the two of the prefetch above
                if (haveA && haveN) rs_wait_vm<2>(); else rs_wait_vm<0>();
                gB0[1] = L.gb[0][lane];
                gB1[1] = L.gb[1][lane];
                bool ok0 = (dB0 != 0xFFFFFFFFu) & (gB0[1] != 0u), ok1 = (dB1 != 0xFFFFFFFFu) & (gB1[1] != 0u);
                if (hint_first) {
                    ok0 = ok0 & ((gB0[1] == 255u) | (gB0[1] == rmh_code(dB0, rsh[1])));
                    ok1 = ok1 & ((gB1[1] == 255u) | (gB1[1] == rmh_code(dB1, rsh[1])));
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
                            gB0[j] = ok0 ? (uint32_t)rt[j][dB0 >> rsh[j]] : 0u;
                            gB1[j] = ok1 ? (uint32_t)rt[j][dB1 >> rsh[j]] : 0u;
                            LC(PH_FREQS, lines_of(rt[j] + (dB0 >> rsh[j]), ok0, 1u) + lines_of(rt[j] + (dB1 >> rsh[j]), ok1, 1u));
                        };
                        rs_for<1, NT>(wload);
                    }
                } else if constexpr (NT > 2) {
                    // lists 2.. : their bytes only for the candidates list 1's byte lets through (list maxima for the others)
                    float rest = 0.f;
                    auto add_max = [&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; rest = rest + rsc[j] * 255.0f; };
                    rs_for_down<NT, 2>(add_max);
                    ok0 = ok0 & enters((B.wq + (rest + rsc[1] * (float)gB0[1])) * BOUND_SLACK);
                    ok1 = ok1 & enters((B.wq + (rest + rsc[1] * (float)gB1[1])) * BOUND_SLACK);
                    if (ballot(ok0) | ballot(ok1)) {
                        auto load_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            gB0[j] = (uint32_t)rt[j][(ok0 ? dB0 : 0u) >> rsh[j]];
                            gB1[j] = (uint32_t)rt[j][(ok1 ? dB1 : 0u) >> rsh[j]];
                        };
                        rs_for<2, NT>(load_one);
#ifdef DS2I_LINE_COUNT
                        auto cnt_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            LC(PH_FREQS, lines_of(rt[j] + ((ok0 ? dB0 : 0u) >> rsh[j]), true, 1u) + lines_of(rt[j] + ((ok1 ? dB1 : 0u) >> rsh[j]), true, 1u));
                        };
                        rs_for<2, NT>(cnt_one);
#endif
                        auto test_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            ok0 = ok0 & (gB0[j] != 0u);
                            ok1 = ok1 & (gB1[j] != 0u);
                        };
                        rs_for<2, NT>(test_one);
                    }
                }
                // what the lists after list `after` can add to this lane's two candidates, from their own bytes (summed from the
                // last list down, so that the value for `after` is a prefix of the same chain whatever `after` is)
                auto rest_of = [&](const uint32_t (&g)[NT], int after) __attribute__((always_inline)) -> float {
                    float r = 0.f;
                    auto add_one = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        if (j > after) r = r + rsc[j] * (float)g[j];
                    };
                    rs_for_down<NT, 1>(add_one);
                    return r;
                };
                float r0 = rest_of(gB0, 0), r1 = rest_of(gB1, 0);
                ok0 = ok0 & enters((B.wq + r0) * BOUND_SLACK);
                ok1 = ok1 & enters((B.wq + r1) * BOUND_SLACK);
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
                if (__builtin_expect((ballot(ok0) | ballot(ok1)) != 0, 0)) {
                    LC(PH_C_LIVEROUNDS, 1);
                    // ---------------- stage C: somebody of block B may enter the heap
                    // freqs of the block (its bytes are still staged), freq-only bound (doc_term_weight falls with norm_len, so the
                    // collection's shortest document bounds the term score from the freq alone), norm_len, exact list-0 score
                    const float min_nl = rs_args()->min_norm_len;
                    const float* const norm_lens = rs_args()->norm_lens;
                    const uint8_t* const arena = rs_args()->arena;
                    uint32_t fv0, fv1, consF;
                    {
                        const uint8_t* p = data0 + B.ep; // (full blocks are dword aligned and a multiple of 4 bytes long)
                        uint32_t* const stB = L.stage[bufB];
                        const uint32_t skip_dw = consB >> 2;
                        if (CODEC == CODEC_OPTPFOR && szB == 128u && ((uintptr_t)p & 3u) == 0u && (consB & 3u) == 0u && skip_dw < STAGE_DW &&
                            optpfor_decode_lds(stB + skip_dw, STAGE_DW - skip_dw, L.exc, L.out, fv0, fv1, consF)) {
                        } else {
                            Window w{(const uint8_t*)((uintptr_t)p & ~(uintptr_t)3), STAGE_DW * 4u, stB};
                            uint32_t a0, a1;
                            consF = uniform(decode_block<CODEC>(CODEC, w, p + consB, 0xFFFFFFFFu, szB, L.out, L.exc, a0, a1));
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");
                            fv0 = a0;
                            fv1 = a1;
                        }
                    }
                    ++s_freqs_blocks;
                    s_bytes += consF;
                    const uint32_t f0 = fv0 + 1u, f1 = fv1 + 1u;
#ifndef DS2I_RS_NO_FREQ_BOUND
                    ok0 = ok0 & enters((qw0 * doc_term_weight(f0, min_nl) + r0) * BOUND_SLACK);
                    ok1 = ok1 & enters((qw0 * doc_term_weight(f1, min_nl) + r1) * BOUND_SLACK);
#endif
                    const float nl0 = ok0 ? norm_lens[dB0] : 1.f, nl1 = ok1 ? norm_lens[dB1] : 1.f;
                    LC(PH_SCORE, lines_of(norm_lens + dB0, ok0, 4u) + lines_of(norm_lens + dB1, ok1, 4u));
                    float pa0 = qw0 * doc_term_weight(f0, nl0), pa1 = qw0 * doc_term_weight(f1, nl1);
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
                        const QTerm* const tj = qt + j;
                        const uint32_t nj = tj->n, nbj = (nj + 127u) >> 7;
                        const uint32_t vlj = 1u + (nj >= (1u << 7)) + (nj >= (1u << 14)) + (nj >= (1u << 21)) + (nj >= (1u << 28));
                        const uint8_t* const dataj = arena + tj->list_off + vlj + 4ull * nbj + 4ull * (nbj - 1);
                        const uint2* const tabj = (const uint2*)rs_args()->skip + tj->blk_base;
                        const float* const wtabj = rs_args()->bmw + tj->blk_base;
                        const float qwj = tj->q_weight;
                        const float rj0 = rest_of(gB0, j), rj1 = rest_of(gB1, j); // the lists after j
                        const float bj0 = rsc[j] * (float)gB0[j], bj1 = rsc[j] * (float)gB1[j];
                        uint32_t* const dj = L.dj[j - 1];
                        bool mem0 = false, mem1 = false;
                        uint32_t curj = cget(CB + C_CUR), bmj = cget(CB + C_BMAX);
                        while (todo0 | todo1) {
                            const uint32_t amin = todo0 ? bcast(dB0, (uint32_t)__builtin_ctzll(todo0)) : bcast(dB1, (uint32_t)__builtin_ctzll(todo1));
                            if (curj == 0xFFFFFFFFu || amin > bmj) {
                                Found fb;
                                const bool found = find_block_rows(tabj, wtabj, nbj, curj + 1u, amin, fb);
                                LC(PH_FIND, 1);
                                if (!found) { // list j has nothing >= amin: no later document of list 0 can be a result either
                                    s_bm_examined += 1;
                                    s_bytes += 4;
                                    finished = true;
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
                                    continue; // (the list stays where it was: the next search restarts there)
                                }
                                const uint8_t* pb = dataj + fb.ep;
                                Window wb{nullptr, 0, L.stb};
                                wb.load(pb, STAGE_DW * 4u - 4u);
                                LC(PH_C_BDOCS, 1);
                                LC(PH_DOCS, lines_of(pb + 8u * lane, true, 8u));
                                const uint32_t szb = ((fb.blk + 1) * 128u <= nj) ? 128u : (nj & 127u);
                                uint32_t v0, v1;
                                const uint32_t consD = uniform(decode_block<CODEC>(CODEC, wb, pb, fb.bmax - fb.base - (szb - 1), szb, dj, L.exc, v0, v1));
                                const uint32_t g0 = (lane < szb) ? v0 + 1u : 0u, g1 = (lane + 64 < szb) ? v1 + 1u : 0u;
                                const uint32_t i0 = wave_incl_scan(g0);
                                const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
                                dj[lane] = (lane < szb) ? fb.base + i0 - 1u : 0xFFFFFFFFu;
                                dj[lane + 64] = (lane + 64 < szb) ? fb.base + i1 - 1u : 0xFFFFFFFFu;
                                wave_sync();
                                curj = fb.blk;
                                bmj = fb.bmax;
                                const unsigned long long fo = (unsigned long long)(pb + consD - arena);
                                const unsigned long long sb = (unsigned long long)(uintptr_t)wb.gbase;
                                cset(CB + C_CUR, curj);
                                cset(CB + C_BMAX, bmj);
                                cset(CB + C_SZ, szb);
                                cset(CB + C_FOLO, (uint32_t)fo);
                                cset(CB + C_FOHI, (uint32_t)(fo >> 32));
                                cset(C_SOWNER, (uint32_t)j);
                                cset(C_SBLO, (uint32_t)sb);
                                cset(C_SBHI, (uint32_t)(sb >> 32));
                                if (cget(C_FOWNER) == (uint32_t)j) cset(C_FOWNER, 0u);
                                ++s_docs_blocks;
                                s_bytes += 4 + consD;
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
                            if (ballot(m0) | ballot(m1)) { // members take list j's term score at once
                                if (cget(C_FOWNER) != (uint32_t)j) {
                                    const uint8_t* pf = arena + (((unsigned long long)cget(CB + C_FOHI) << 32) | cget(CB + C_FOLO));
                                    const uint8_t* sbase = (const uint8_t*)(uintptr_t)(((unsigned long long)cget(C_SBHI) << 32) | cget(C_SBLO));
                                    Window wf{sbase, STAGE_DW * 4u, L.stb};
                                    if (cget(C_SOWNER) != (uint32_t)j || !wf.covers(pf, 64)) {
                                        wf.load(pf, 256u);
                                        LC(PH_C_BFREQS, 1);
                                        LC(PH_PROBE, lines_of(pf + 4u * lane, true, 4u));
                                        cset(C_SOWNER, 0u); // (the window no longer starts at the block)
                                    }
                                    uint32_t v0, v1;
                                    const uint32_t consF2 = decode_block<CODEC>(CODEC, wf, pf, 0xFFFFFFFFu, cget(CB + C_SZ), L.fj, L.exc, v0, v1);
                                    L.fj[lane] = v0 + 1u;
                                    L.fj[lane + 64] = v1 + 1u;
                                    wave_sync();
                                    cset(C_FOWNER, (uint32_t)j);
                                    ++s_freqs_blocks;
                                    s_bytes += consF2;
                                }
                                if (m0) { pa0 = pa0 + qwj * doc_term_weight(L.fj[q0], nl0); mem0 = true; }
                                if (m1) { pa1 = pa1 + qwj * doc_term_weight(L.fj[q1], nl1); mem1 = true; }
                            }
                        }
                        // members whose score can still enter go on to the next list (a candidate the loop left unsettled -- list j
                        // ended below it -- is not a member)
                        ok0 = ok0 & mem0 & enters((pa0 + rj0) * BOUND_SLACK);
                        ok1 = ok1 & mem1 & enters((pa1 + rj1) * BOUND_SLACK);
                    };
                    rs_for<1, NT>(probe);
                    // pa0 / pa1 are complete scores of documents of the intersection now
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
                                if (shared_floor && lane == 0) sh.add(v);
                            }
                        }
                    }
                }
            }
            if (__builtin_expect(finished, 0)) break;
            // ---------------- rotate: A becomes B (its gathers are issued now and consumed an iteration later, behind the next
            // block's decode), the block whose bytes were requested becomes A
            B = A;
            haveB = haveA;
            dB0 = dA0;
            dB1 = dA1;
            consB = consA;
            szB = szA;
            if (haveB) {
                // one byte per candidate from list 1's table (the other lists' bytes are fetched in stage B for the candidates inside
                // list 1's ranges only: a gather is one cache-line request per lane, and most candidates die at list 1)
                rs_gather_u8(gt1, (dB0 != 0xFFFFFFFFu ? dB0 : 0u) >> rsh[1], gb_base);
                rs_gather_u8(gt1, (dB1 != 0xFFFFFFFFu ? dB1 : 0u) >> rsh[1], gb_base + 256u);
                LC(PH_TOPK, lines_of(gt1 + ((dB0 != 0xFFFFFFFFu ? dB0 : 0u) >> rsh[1]), true, 1u) + lines_of(gt1 + ((dB1 != 0xFFFFFFFFu ? dB1 : 0u) >> rsh[1]), true, 1u));
#ifdef DS2I_LINE_COUNT
                {   // the same lines by the width of list 1's ranges (shift 0 = one doc-id per byte: the densest lists)
                    const uint32_t nl = lines_of(gt1 + ((dB0 != 0xFFFFFFFFu ? dB0 : 0u) >> rsh[1]), true, 1u) + lines_of(gt1 + ((dB1 != 0xFFFFFFFFu ? dB1 : 0u) >> rsh[1]), true, 1u);
                    const uint32_t shv = rsh[1];
                    if (shv == 0) lc[PH_TOTAL] += nl; else if (shv == 1) lc[PH_INSERT] += nl; else if (shv == 2) lc[PH_PREFETCH] += nl; else if (shv <= 4) lc[PH_FLOOR] += nl; else lc[PH_UNIT] += nl;
                }
#endif
            }
            A = N;
            haveA = haveN;
            const uint32_t t = bufB;
            bufB = bufA;
            bufA = bufN;
            bufN = t;
        }
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
    }
#undef LC
}

} // namespace

extern "C" {
// nt = exact number of distinct terms of every query of the launch (2..4); block_mixed index with skip table, block weights
// and range tables, k <= 64 (instrumented and uninstrumented runs share the instantiation with counters)
hipError_t ds2i_launch_ranked_stream_mixed(int nt, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    switch (nt) {
    case 2: hipLaunchKernelGGL((k_ranked_stream_mixed<2, true, CODEC_MIXED>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_ranked_stream_mixed<3, true, CODEC_MIXED>), g, b, 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_ranked_stream_mixed<4, true, CODEC_MIXED>), g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}
