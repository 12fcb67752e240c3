// HIP kernels (gfx950 / CDNA4, wave64) for the ds2i batched query path.
// One wavefront per WORK UNIT (a piece of a query: a block range of its shortest list, or a doc-id range), one unit
// per single-wave workgroup; every memory operation is wave-cooperative, control flow is wave-uniform. No MFMA
// (integer work).
//
//   k_conjunctive   and_query / ranked_and_query   reference queries.hpp:35-86, 322-401
//                   block-synchronous intersection: each round intersects the window
//                   [lo, min_i block_max_i] of all lists' current blocks at once (up to
//                   128 candidates, two per lane) instead of one candidate per step; ranked_and scores
//                   progressively and prunes with exact bounds from the per-block max-weight table
//   k_disjunctive   wand / maxscore / ranked_or (top-k of the union) and or / or_freq, block-synchronous
//                   reference queries.hpp:88-131, 200-319, 404-476, 478-591
//   k_daat          every operator in the reference's one-document-per-step order (DS2I_OP_REFERENCE_ORDER)
//   k_daat_long     the same traversals for queries with more than 16 terms (state in global scratch)
//   k_merge         partial results of split queries
//   k_decode_list   full decode of one list (Index::operator[] + enumeration)
//   k_block_max_weights / k_list_top_bmw   upload-time block-max BM25 weights
//   k_selftest*     primitives (scan, bm25) used by the GPU unit tests
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_enum.hpp"
#include "device_score.hpp"

using namespace ds2i_dev;

namespace {

template <int TMAX, bool META_IN_LDS = true, bool WITH_POS = true, bool WITH_S16 = true, int NF = TMAX>
struct Lds {
    uint32_t docs[TMAX][128];
    uint32_t freqs[NF][128]; // NF < TMAX: the lists after list 0 share slot 1 (CtxT SHARE_F)
    uint32_t meta[META_IN_LDS ? TMAX : 1][META_IN_LDS ? M_WORDS : 1];
    // + the Simple16 field table (device_codecs.hpp); kernels compiled for the Elias-Fano layouts never read it and leave
    // it out: 928 B per wave, which takes the 3-4-list ranked kernel from 21 to 24 resident workgroups per CU
    uint32_t exc[WITH_S16 ? EXC_LDS_DW : EXC_DW];
    uint32_t st[STAGE_DW];
    uint8_t pos[WITH_POS ? TMAX : 1][WITH_POS ? 128 : 4]; // match position of candidate c in list i (and_freq; row 0
                            // unused by the conjunctive kernel and reused as ord/ub by the daat kernel)
    DS2I_DEV uint32_t* ord() { return (uint32_t*)&pos[0][0]; }       // daat: ordered_enums [TMAX<=16]
    DS2I_DEV float* ub() { return (float*)&pos[0][64]; }             // maxscore upper_bounds [TMAX<=16]
};

DS2I_DEV void bind_meta(MetaLds& m, uint32_t* lds_meta) { m.p = lds_meta; }
template <int T> DS2I_DEV void bind_meta(MetaReg<T>&, uint32_t*) {}

template <int CODEC_T, class META, bool STATS = true, bool SHARE_F = false, class LDS>
DS2I_DEV CtxT<CODEC_T, META, STATS, SHARE_F> make_ctx(LDS& L, const BatchArgs& a, uint32_t* docs, uint32_t* freqs) {
    CtxT<CODEC_T, META, STATS, SHARE_F> c;
    c.docs = docs;
    c.freqs = freqs;
    bind_meta(c.meta, &L.meta[0][0]);
    c.exc = L.exc;
    if constexpr (CODEC_T != CODEC_PEF && CODEC_T != CODEC_OPTPFOR) s16_table_init(L.exc); // (block_optpfor kernels decode through the side slots)
    c.win.st = L.st;
    c.win.gbase = a.arena;
    c.win.nbytes = 0;
    c.arena = a.arena;
    c.bits0 = a.bits0;
    c.bits1 = a.bits1;
    c.codec = a.codec;
    c.num_docs = a.num_docs;
    c.block_profile = a.block_profile;
    c.skip = (const uint2*)a.skip;
    c.xslots = a.xslots;
    c.xovf = a.xovf;
    c.tails = a.tails;
    c.init_stats();
    return c;
}
template <int CODEC_T, class META, bool STATS = true, bool SHARE_F = false, class LDS>
DS2I_DEV CtxT<CODEC_T, META, STATS, SHARE_F> make_ctx(LDS& L, const BatchArgs& a) {
    return make_ctx<CODEC_T, META, STATS, SHARE_F>(L, a, &L.docs[0][0], &L.freqs[0][0]);
}

template <int NK>
DS2I_DEV void store_topk(float* topk, uint32_t* topk_len, uint32_t k, uint32_t slot, const TopKBig<NK>& tk) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int r = 0; r < NK; ++r)
        if ((uint32_t)r * 64u + lane < k) topk[(size_t)slot * k + (uint32_t)r * 64u + lane] = tk.v[r];
    if (lane == 0) topk_len[slot] = tk.n;
}
DS2I_DEV void store_topk(float* topk, uint32_t* topk_len, uint32_t k, uint32_t slot, const TopK& tk) {
    const uint32_t lane = lane_id();
    if (lane < k) topk[(size_t)slot * k + lane] = tk.v;
    if (lane == 0) topk_len[slot] = tk.n;
}

// ------------------------------------------------------------------ conjunctive
// Candidate membership of c (held by this lane, valid iff `want`) in the sorted LDS block d[128].
DS2I_DEV bool member_bsearch(const uint32_t* d, uint32_t c, bool want, uint32_t& pos) {
    uint32_t idx = 0;
    if (want) {
#pragma unroll
        for (uint32_t step = 64; step; step >>= 1)
            if (d[idx + step - 1] < c) idx += step;
    }
    pos = idx;
    return want && d[idx] == c;
}

// Runs body(i) for i in [FROM, nt); body returns false to break. With REG the loop is expanded at compile time
// (template recursion), so i is a constant inside the body and the register-resident enumerator state is never
// indexed dynamically; otherwise it is a plain loop.
template <int I, int N, class F>
DS2I_DEV bool static_list_loop(uint32_t nt, F& f) {
    if constexpr (I < N) {
        if ((uint32_t)I >= nt) return true;
        if (!f(std::integral_constant<uint32_t, (uint32_t)I>{})) return false;
        return static_list_loop<I + 1, N>(nt, f);
    }
    return true;
}
// body(integral_constant i) for i = HI-1 down to LO, compile-time expanded
template <int HI, int LO, class F>
DS2I_DEV void static_loop_down(F& f) {
    if constexpr (HI > LO) {
        f(std::integral_constant<uint32_t, (uint32_t)(HI - 1)>{});
        static_loop_down<HI - 1, LO>(f);
    }
}
#define DS2I_LIST_LOOP(FROM, body)                                   \
    if constexpr (REG) {                                             \
        static_list_loop<(FROM), TMAX>(nt, body);                    \
    } else {                                                         \
        for (uint32_t i_ = (FROM); i_ < nt; ++i_)                    \
            if (!body(i_)) break;                                    \
    }

// Waves per SIMD the conjunctive kernels are compiled for. <=2 lists: 6 (80 VGPRs, 12 B/lane of scratch; measured on the
// GOV2-scale batch: 6 / 7 / 8 waves = 210 / 209 / 203 k queries/s -- the spills of the tighter budgets cost what the
// extra waves hide); beyond that LDS caps the residency anyway (8 / 13 / 23 KiB per wave of the 160 KiB per CU), and
// without a bound the register allocator lets the unrolled list loops balloon (237 VGPRs, 2 waves/SIMD for the 4-list
// kernel when left alone).
#ifndef DS2I_FLOOR_EVERY
#define DS2I_FLOOR_EVERY 4 // power of two
#endif
static_assert(DS2I_FLOOR_EVERY > 0 && (DS2I_FLOOR_EVERY & (DS2I_FLOOR_EVERY - 1)) == 0, "DS2I_FLOOR_EVERY is used as a mask: power of two");
#ifndef DS2I_OCC2
#define DS2I_OCC2 6
#endif
#define CONJ_WAVES(T) ((T) <= 2 ? DS2I_OCC2 : (T) <= 4 ? 5 : (T) <= 8 ? 3 : 1)
// and / and_freq carry no scoring state: their <=2-list kernels fit 8 waves/SIMD (59-61 VGPRs, no scratch)
#define CONJ_WAVES_R(RANKED, T) (!(RANKED) && (T) <= 2 ? 8 : CONJ_WAVES(T))

// LDS of the conjunctive kernels: the shared layout plus, for ranked_and, the norm_len of every posting of list 0's
// current block (-1 = the posting was dropped by the freq-only bound and its norm_len never fetched) and its list-0 term
// score; ranked_and has no use for the match positions (it scores progressively). 5540 B for <=2 lists: 29 waves per
// CU fit, 24 (6 per SIMD) are used.
template <int TMAX, bool META_IN_LDS, bool RANKED, bool WITH_S16 = true>
struct LdsConj : Lds<TMAX, META_IN_LDS, !RANKED, WITH_S16, (RANKED && TMAX > 2) ? 2 : TMAX> {
    float nl[RANKED ? 128 : 1];
    float part0[RANKED ? 128 : 1]; // list-0 term score of each posting of the block (-inf = dropped): read every round
    // range-table bytes of each posting of list 0's block in the other lists, packed: byte i-1 of qb = list i (1..4),
    // byte i-5 of qb2 = list i (5..7)
    uint32_t qb[RANKED ? 128 : 1];
    uint32_t qb2[RANKED && (TMAX > 4) ? 128 : 1];
};

template <bool RANKED, bool WITH_FREQS, int TMAX, int CODEC_T, bool STATS = true>
__global__ void __launch_bounds__(64, CONJ_WAVES_R(RANKED, TMAX)) k_conjunctive(BatchArgs a) {
    // <=4 lists: every list loop below is fully unrolled, so the enumerator state is addressed with constants
    // and lives in registers (MetaReg); 8/16 lists keep it in LDS (code size)
    constexpr bool REG = TMAX <= 4;
    typedef typename std::conditional<REG, MetaReg<TMAX>, MetaLds>::type META;
    __shared__ LdsConj<TMAX, !REG, RANKED, CODEC_T != CODEC_PEF && CODEC_T != CODEC_OPTPFOR> L; // (no Simple16 field table for the Elias-Fano layouts and for block_optpfor through its side slots)
    const uint32_t lane = lane_id();
    // ranked_and with 3+ lists: one freqs buffer for list 0, one shared by the others (each is used where it is decoded)
    constexpr bool SHARE_F = RANKED && TMAX > 2;
    CtxT<CODEC_T, META, STATS, SHARE_F> cx = make_ctx<CODEC_T, META, STATS, SHARE_F>(L, a);
    cx.want_freqs = WITH_FREQS && !RANKED; // and_freq reads the freq of every match: both parts of a block in one pass (side slots)
    // ranked_and only: per-block max doc_term_weight table (null = no pruning). Three levels, all exact (the bounds
    // are true upper bounds of the float32 score and topk_queue::insert is strict, queries.hpp:157-172):
    //   * blocks of list 0 whose bound cannot enter the heap are skipped without being decoded (skip_list0);
    //   * SCORE-FIRST rounds, once the heap is full or a floor is known: the list-0 term score of every candidate is
    //     computed first; a candidate goes on to list i only while its partial score + the later lists' list maxima can
    //     still enter the heap, so the long lists are probed -- and their blocks decoded -- for few candidates;
    //   * before list i's next block is decoded, the best alive partial score + that block's max weight is tested.
    const float* const bmw = RANKED ? a.bmw : nullptr;
    // Doc-id-range tables (BatchArgs::rmw): one byte gather per candidate and other list, no search. A zero byte proves
    // the candidate is in no intersection with that list; otherwise the bytes bound its score in the other lists far
    // tighter than the list maxima (M_SUF) do. Lists 1..7 of a query are covered (the 9-16-term class keeps M_SUF).
    constexpr int RL = TMAX < 8 ? TMAX : 8; // lists 1 .. RL-1 have their bytes packed per candidate
    const uint8_t* const rmw = TMAX <= 8 ? a.rmw : nullptr;
    // one work unit per (single-wave) workgroup, costliest units first: the hardware dispatcher
    // interleaves the workgroups of the concurrently running LDS classes as resources free up
    for (uint32_t tkt = blockIdx.x; tkt < a.nslice; tkt += gridDim.x) {
        const uint32_t uid = a.order[tkt];
        const unsigned long long t_unit = (STATS && a.unit_clock) ? wall_clock64() : 0ull;
        const Unit u = a.units[uid];
#ifdef DS2I_PHASE_TIMING
        const unsigned long long unit_t0 = __builtin_readcyclecounter();
#endif
        const uint32_t q = u.q;
        const bool whole = u.nparts == 1;
        const uint32_t t0 = a.q_off[q], nt = a.q_off[q + 1] - t0;
        unsigned long long count = 0, fsum = 0;
        TopK tk;
        tk.init(a.k);
        if (nt == 0 || nt > (uint32_t)TMAX) { // empty query -> 0 results (queries.hpp:41,335)
            if (lane == 0) { a.out_count[q] = 0; if (a.out_freq_sum) a.out_freq_sum[q] = 0; }
            if (RANKED) store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
            continue;
        }
        // list 0 (shortest) drives; the unit owns its blocks [blk_begin, blk_end). The other lists are
        // bound lazily: their first block is located by the first candidate (no block-0 decode).
        auto bind_one = [&](auto ic) __attribute__((always_inline)) { const uint32_t i = ic; cx.bind(i, a.qterms[t0 + i]); return true; };
        DS2I_LIST_LOOP(0, bind_one)
        const unsigned long long mbase = a.out_matches ? a.match_off[q] + 128ull * u.blk_begin : 0;
        const unsigned long long mcap = a.out_matches ? 128ull * (u.blk_end - u.blk_begin) : 0;
        // ---- pruning state (ranked_and with a bmw table)
        // the parts of a split query share a score histogram (ScoreHist): the floor it yields is exact to compare against,
        // since all parts add a document's terms in the same order (a dropped document scores <= floor <= final threshold)
        const bool shared_floor = RANKED && bmw && !whole && a.q_hist;
        ScoreHist sh;
        sh.init(shared_floor ? a.q_hist : nullptr, shared_floor ? a.q_hist_slot[q] : 0u,
                shared_floor ? __uint_as_float(uniform(__float_as_uint(a.qterms[t0].max_bmw + a.qterms[t0].suf_bmw))) : 0.f,
                1.0f - 1.0f / 1048576.0f);
        auto adopt_floor = [&]() __attribute__((always_inline)) {
            const float f = sh.floor(tk.k);
            if (f > tk.floor) tk.floor = f;
        };
        if (RANKED && bmw && nt == 1) // k blocks of the list hold a document reaching floor1 (computed at upload)
            tk.floor = __uint_as_float(uniform(__float_as_uint(a.qterms[t0].floor1)));
        if (shared_floor) adopt_floor();
        auto can_prune = [&]() { return tk.n >= tk.k || tk.floor > 0.f; };
        const float* const w0tab = bmw ? bmw + cx.m(0, M_PBASE) : nullptr;
        // first block >= blk of list 0, inside the unit, whose bound (its own max weight + the other lists' list maxima)
        // can enter the heap: 64 table entries per probe, nothing decoded
        auto skip_list0 = [&](uint32_t blk) __attribute__((always_inline)) -> uint32_t {
            if (!RANKED || !bmw || !can_prune()) return blk;
            const float qw0 = __uint_as_float(cx.m(0, M_QW)), suf0 = __uint_as_float(cx.m(0, M_SUF));
            while (blk < u.blk_end) {
                const uint32_t idx = blk + lane;
                float w = 0.f;
                if (idx < u.blk_end) w = w0tab[idx];
                const float ub = (qw0 * w + suf0) * BOUND_SLACK;
                const uint64_t hit = ballot(idx < u.blk_end && tk.would_enter(ub));
                if (hit) return blk + (uint32_t)__builtin_ctzll(hit);
                blk += 64;
            }
            return u.blk_end;
        };
        const bool use_rmw = rmw && nt > 1 && (!RANKED || bmw); // wave-uniform
        // the range-table bytes of candidate c in lists 1..nt-1 (all gathers are issued before the first is consumed);
        // false = some list has no posting in c's range, so c cannot be a match
        auto rmw_gather = [&](uint32_t c, bool valid, uint32_t& qlo, uint32_t& qhi) __attribute__((always_inline)) -> bool {
            uint32_t e[RL] = {};
            auto load_one = [&](auto ic) __attribute__((always_inline)) {
                constexpr uint32_t i = decltype(ic)::value;
                const uint8_t* tab = rmw + 64ull * cx.m(i, M_RBASE);
                e[i] = valid ? (uint32_t)tab[c >> cx.m(i, M_RSHIFT)] : 0u;
                return true;
            };
            // two trips: list 1 first, the other lists only for the candidates inside list 1's ranges (a table holds 4..8
            // entries per posting, so a random document is outside with probability ~0.85 whatever the list's length).
            // A gather is one cache-line request per lane, and with 3+ lists the L1's request rate, not the latency, was
            // what the gathers cost: 500 k -> 525 k queries/s on the GOV2-scale batch.
            load_one(std::integral_constant<uint32_t, 1>{});
            valid = valid && e[1] != 0;
            static_list_loop<2, RL>(nt, load_one);
            bool ok = valid;
            qlo = qhi = 0;
            auto pack_one = [&](auto ic) __attribute__((always_inline)) {
                constexpr uint32_t i = decltype(ic)::value;
                ok = ok && e[i] != 0;
                if constexpr (i <= 4) qlo |= e[i] << (8 * (i - 1)); else qhi |= e[i] << (8 * (i - 5));
                return true;
            };
            static_list_loop<1, RL>(nt, pack_one);
            return ok;
        };
        // bound of the candidate's term score in list i, from its packed byte
        auto rmw_term = [&](uint32_t qlo, uint32_t qhi, auto ic) __attribute__((always_inline)) -> float {
            constexpr uint32_t i = decltype(ic)::value;
            const uint32_t b = i <= 4 ? (qlo >> (8 * (i - 1))) & 255u : (qhi >> (8 * ((i - 5) & 3))) & 255u;
            return __uint_as_float(cx.m(i, M_RSCALE)) * (float)b;
        };
        // sum of those bounds over the lists j > after (added from the last list down, so that the value for `after` is a
        // prefix of the same chain whatever `after` is)
        auto rmw_rest = [&](uint32_t qlo, uint32_t qhi, uint32_t after) __attribute__((always_inline)) -> float {
            float r = 0.f;
            auto add_one = [&](auto jc) __attribute__((always_inline)) {
                constexpr uint32_t j = decltype(jc)::value;
                if (j < nt && j > after) r = r + rmw_term(qlo, qhi, jc);
            };
            static_loop_down<RL, 1>(add_one);
            return r;
        };
        // and_query only: lists dense enough to carry an exact bitmap (bit i of bm_lists) are tested by a bit gather and then
        // never probed or decoded -- membership is all and_query wants from them; the others go by their range-table byte
        // (zero = not a member for sure) and are verified by the usual probe
        uint32_t bm_lists = 0;
        if constexpr (!RANKED) { // (and_freq filters by the bitmap too -- exact, where a byte covering 1..4 doc-ids of a dense list
                                 // mostly is not zero -- but still probes the list: it needs the matching postings' freqs)
            if (use_rmw && a.rmw_bitmaps) {
                auto mark = [&](auto ic) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(ic)::value;
                    if (RmwLevels::has_bitmap(cx.m(i, M_N), a.num_docs)) bm_lists |= 1u << i;
                    return true;
                };
                static_list_loop<1, RL>(nt, mark);
            }
        }
        auto and_filter = [&](uint32_t c, bool valid) __attribute__((always_inline)) -> bool {
            uint32_t e[RL] = {};
            auto load_one = [&](auto ic) __attribute__((always_inline)) {
                constexpr uint32_t i = decltype(ic)::value;
                const uint8_t* tab = rmw + 64ull * cx.m(i, M_RBASE);
                if ((bm_lists >> i) & 1u) {
                    const uint32_t* bm = (const uint32_t*)(tab + RmwLevels(a.num_docs, cx.m(i, M_RSHIFT)).bytes());
                    e[i] = valid ? (bm[c >> 5] >> (c & 31u)) & 1u : 0u;
                } else {
                    e[i] = valid ? (uint32_t)tab[c >> cx.m(i, M_RSHIFT)] : 0u;
                }
                return true;
            };
            static_list_loop<1, RL>(nt, load_one);
            bool ok = valid;
            auto test_one = [&](auto ic) __attribute__((always_inline)) { ok = ok && e[decltype(ic)::value] != 0; return true; };
            static_list_loop<1, RL>(nt, test_one);
            if (a.rmh && ballot(ok)) {
                // membership hints (BatchArgs::rmh; block_optpfor indexes): for the candidates every list's byte lets through, the
                // lists without a bitmap say WHICH document of the candidate's range is theirs (where it is the only one): a
                // candidate elsewhere in that range is not a member and is never probed. All hint loads first, then the tests.
                const long long hd = (long long)(a.rmh - a.rmw);
                uint32_t h[RL] = {};
                auto load_hint = [&](auto ic) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(ic)::value;
                    h[i] = 255u;
                    if (!((bm_lists >> i) & 1u)) {
                        const uint8_t* ht = rmw + 64ull * cx.m(i, M_RBASE) + hd;
                        if (ok && cx.m(i, M_RSHIFT) != 0u) h[i] = (uint32_t)ht[c >> cx.m(i, M_RSHIFT)]; // (one doc-id per entry: the weight byte was the answer)
                    }
                    return true;
                };
                static_list_loop<1, RL>(nt, load_hint);
                auto test_hint = [&](auto ic) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(ic)::value;
                    ok = ok && ((h[i] == 255u) | (h[i] == rmh_code(c, cx.m(i, M_RSHIFT))));
                    return true;
                };
                static_list_loop<1, RL>(nt, test_hint);
            }
            return ok;
        };
        cx.s_bytes += 4;
        ++cx.s_bm_examined;
        uint32_t lo = 0, floor_tick = 1;
        uint64_t okm0 = ~0ull, okm1 = ~0ull; // and / and_freq: candidates of list 0's block the range tables have not ruled out
        uint32_t part_blk = 0xFFFFFFFFu; // block of list 0 whose norm_lens / list-0 scores are in L.nl
        // ---- list 0 as a STREAM (block indexes with the interleaved skip table). The driving list is walked front to
        // back, most of its blocks only to find that none of their documents can be a result, so what a step costs is its
        // dependent memory round trips. The stream keeps a 64-entry window of the list's table rows in registers (lane j:
        // {block_max, end offset} and the block weight of entry s_first + j; lane 0 is the row BEFORE the first block the
        // window can serve, whose block_max / end offset give that block's base / start): "which block is next" is a ballot
        // over registers, a block's table words cost no load, and the bytes of the block that will be taken after the
        // current one are requested while the current one is still being worked on (pf_d0 / pf_d1: 512 B, two dwords per lane).
        // The freq_index layouts stream the same way over their chunk directory: the window holds cmax[] (a chunk's last
        // doc-id) and the chunk weights, and what is requested ahead is the next chunk's 12-dword directory entry (its bit
        // positions: the first of the two dependent loads a chunk decode starts with).
        const bool pstream = cx.is_pef();
        const bool stream0 = pstream || cx.skip;
        const uint2* const tab0 = (stream0 && !pstream) ? cx.skip + cx.m(0, M_PBASE) : nullptr;
        const uint32_t* const cmax0 = pstream ? (const uint32_t*)cx.ptr(0, M_MAXS_LO) : nullptr;
        const uint32_t* const ent0 = pstream ? (const uint32_t*)cx.ptr(0, M_END_LO) : nullptr;
        const uint8_t* data0 = nullptr;
        uint32_t s_first = 0, pf_blk = 0xFFFFFFFFu, pf_d0 = 0, pf_d1 = 0, pf_x = 0;
        uint2 s_e = make_uint2(0xFFFFFFFFu, 0u);
        float s_w = 0.f, s_rb = 0.f;
        bool s_none = false; // some other list has no posting at all inside the block's doc-id span: nothing to intersect
        auto s_fill = [&](uint32_t first) __attribute__((always_inline)) {
            s_first = first;
            const uint32_t idx = first + lane;
            s_e = make_uint2(0xFFFFFFFFu, 0u);
            s_w = 0.f;
            if (idx < u.blk_end) {
                if (pstream) s_e.x = cmax0[idx]; else s_e = tab0[idx];
                if (RANKED && w0tab) s_w = w0tab[idx];
            }
            s_none = false;
            {
                s_rb = RANKED ? __uint_as_float(cx.m(0, M_SUF)) : 0.f; // what the other lists can add to a document of the block: their list maxima, or
                if (use_rmw) {
                    // ... with range tables the largest entry each of them has over the block's own doc-id span [base, block_max],
                    // read from the level of the table whose entries are wide enough for <= 16 of them to cover the span
                    // (a lane serves the block of its table row; 16 independent byte loads per list, once per 63 blocks)
                    const uint32_t prev_max = (uint32_t)__shfl_up((int)s_e.x, 1);
                    const uint32_t base = (lane == 0) ? 0u : prev_max + 1u, top = s_e.x;
                    const bool row = idx < u.blk_end && (lane > 0 || idx == 0) && top != 0xFFFFFFFFu && base <= top;
                    float acc = 0.f;
                    auto one_list = [&](auto jc) __attribute__((always_inline)) {
                        constexpr uint32_t j = decltype(jc)::value;
                        if (j >= nt) return;
                        const uint32_t sh = cx.m(j, M_RSHIFT);
                        const RmwLevels g(a.num_docs, sh);
                        const uint8_t* tb = rmw + 64ull * cx.m(j, M_RBASE);
                        uint32_t best = 255u; // (the list maximum)
                        {   // branch-free: a lane without a row reads entry 0 and discards it
                            const uint32_t b2 = row ? base : 0u, t2 = row ? top : 0u;
                            uint32_t lsh = sh, lvl = 0;
                            while (lvl < 2 && (t2 >> lsh) - (b2 >> lsh) >= 16u) { lsh += 6; ++lvl; }
                            const uint32_t lo = b2 >> lsh, hi = t2 >> lsh;
                            const bool fits = hi - lo < 16u;
                            const uint32_t m = max_of_bytes16(tb + g.off[lvl] + (fits ? lo : 0u), fits ? hi - lo + 1u : 1u);
                            if (row && fits) best = m;
                        }
                        s_none = s_none || best == 0u;
                        if constexpr (RANKED) acc = acc + __uint_as_float(cx.m(j, M_RSCALE)) * (float)best;
                    };
                    static_loop_down<RL, 1>(one_list);
                    s_rb = acc;
                }
            }
        };
        // blocks >= from of the window that are worth a visit: inside the unit and (ranked, once bounds can prune) able to
        // hold a document that enters the heap going by the block's weight + the other lists' list maxima
        auto s_live = [&](uint32_t from) __attribute__((always_inline)) -> uint64_t {
            const uint32_t idx = s_first + lane;
            bool ok = idx >= from && idx < u.blk_end && (lane > 0 || idx == 0) && !s_none;
            if constexpr (RANKED) {
                if (bmw && can_prune()) {
                    const float qw0 = __uint_as_float(cx.m(0, M_QW));
                    ok = ok && tk.would_enter((qw0 * s_w + s_rb) * BOUND_SLACK);
                }
            }
            return ballot(ok);
        };
        auto s_next = [&](uint32_t from) __attribute__((always_inline)) -> uint32_t { // first block >= from worth a visit, or blk_end
            for (;;) {
                if (from >= u.blk_end) return u.blk_end;
                const uint64_t hit = s_live(from);
                if (hit) return s_first + (uint32_t)__builtin_ctzll(hit);
                if (s_first + 64 >= u.blk_end) return u.blk_end;
                s_fill(s_first + 63);
                from = from > s_first + 1 ? from : s_first + 1;
            }
        };
        if (stream0) {
            if (!pstream) {
                const uint8_t* maxs0 = cx.ptr(0, M_MAXS_LO);
                const uint32_t nb0 = cx.m(0, M_NB);
                data0 = maxs0 + 4ull * nb0 + 4ull * (nb0 - 1);
            }
            s_fill(u.blk_begin ? u.blk_begin - 1 : 0);
        }
        // list 0 moves on: `want` = first block with block_max >= lo (or the unit's first block)
        uint32_t want = u.blk_begin;
        bool have_bi = false;
        typename decltype(cx)::BlockInfo bi0;
        bool finished = false;
        bool need0 = true;
#ifdef DS2I_PHASE_TIMING
        cx.s_phase[PH_UNIT] += __builtin_readcyclecounter() - unit_t0;
#endif
        while (!finished) {
            ++cx.s_rounds;
#ifdef DS2I_PHASE_TIMING
            const unsigned long long round_t0 = __builtin_readcyclecounter(); // PH_PROLOG = rounds that end at the range-table test
#endif
            if (stream0 && (need0 || lo > cx.m(0, M_BMAX))) { // list 0 supplies the candidates of this round
                // (lo never exceeds block_max + 1 of list 0's block -- the window is cut at it -- so the next block with
                // block_max >= lo is simply the next one)
                const uint32_t from = need0 ? u.blk_begin : cx.m(0, M_CUR) + 1;
                if (from >= u.blk_end) break;
                cx.s_bm_examined += 1;
                cx.s_bytes += 4;
                // (the shared histogram costs an L2 round trip: consulted every DS2I_FLOOR_EVERY-th block of list 0; every
                // block: -3 %, every 16th: -6 % -- the staler floor costs decodes)
                if (shared_floor && (floor_tick++ & (DS2I_FLOOR_EVERY - 1)) == 0) { PT_BEGIN(cx); adopt_floor(); PT_END(cx, PH_FLOOR); }
                PT_BEGIN(cx);
                const uint32_t blk2 = s_next(from);
                if (blk2 >= u.blk_end) break;
                const uint32_t f = blk2 - s_first, fp = f ? f - 1 : 0;
                bi0.bmax = bcast(s_e.x, f);
                bi0.next_ep = bcast(s_e.y, f);
                bi0.base = blk2 ? bcast(s_e.x, fp) + 1u : 0u;
                bi0.ep = blk2 ? bcast(s_e.y, fp) : 0u;
                const bool staged = pf_blk == blk2;
                if (staged && !pstream) { // the block's bytes were requested a block ago: from registers into the staging window
                    const uint8_t* p = data0 + bi0.ep;
                    cx.win.gbase = (const uint8_t*)((uintptr_t)p & ~(uintptr_t)3);
                    cx.win.nbytes = 512;
                    cx.win.st[lane] = pf_d0;
                    cx.win.st[lane + 64] = pf_d1;
                    if (cx.side()) { cx.exc[lane] = pf_x; cx.slot_blk = cx.m(0, M_PBASE) + blk2; } // (its side slot came with them)
                    wave_sync();
                }
                PT_END(cx, PH_STREAM);
                if (pstream) cx.decode_docs_pef(0, blk2, staged ? &pf_d0 : nullptr);
                else cx.decode_docs(0, blk2, &bi0, staged);
#ifdef DS2I_PHASE_TIMING
                cx.s_phase[PH_C_VISIT] += 1;
#endif
                need0 = false;
                { // request the bytes of the block that is next as things stand (the heap may still rule it out later)
                    PT_BEGIN(cx);
                    const uint64_t nx = s_live(blk2 + 1);
                    pf_blk = 0xFFFFFFFFu;
                    if (nx) {
                        const uint32_t fn = (uint32_t)__builtin_ctzll(nx);
                        if (pstream) { // the chunk's directory entry (lanes 0..11) and its cmax (lane 12), as decode_docs_pef reads them
                            const uint32_t nb2 = s_first + fn, cm = bcast(s_e.x, fn);
                            pf_d0 = lane == PC_WORDS ? cm : 0u;
                            if (lane < PC_WORDS) pf_d0 = ent0[(size_t)nb2 * PC_WORDS + lane];
                        } else {
                            const uint32_t* g = (const uint32_t*)((uintptr_t)(data0 + bcast(s_e.y, fn - 1)) & ~(uintptr_t)3);
                            pf_d0 = g[lane];
                            pf_d1 = g[lane + 64];
                            if (cx.side()) pf_x = cx.xslots[(size_t)XSLOT_DW * (cx.m(0, M_PBASE) + s_first + fn) + lane];
                        }
                        pf_blk = s_first + fn;
                    }
                    PT_END(cx, PH_PREFETCH);
                }
                if constexpr (!RANKED) {
                    if (use_rmw) { // once per block of list 0: who can be a member of every other list at all
                        const uint32_t n0c = L.docs[0][lane], n1c = L.docs[0][lane + 64];
                        okm0 = ballot(and_filter(n0c, n0c != 0xFFFFFFFFu));
                        okm1 = ballot(and_filter(n1c, n1c != 0xFFFFFFFFu));
                    }
                }
            } else if (need0 || lo > cx.m(0, M_BMAX)) { // (freq_index layouts, or no skip table: the search-based form)
                const bool tabbed = META::SKIPTAB && !cx.is_pef() && cx.skip;
                if (!need0) {
                    const uint32_t cur = cx.m(0, M_CUR);
                    { PT_BEGIN(cx); want = tabbed ? cx.find_block_info(0, cur + 1, lo, bi0) : cx.find_block(0, cur + 1, lo); PT_END(cx, PH_FIND); }
                    have_bi = tabbed;
                    cx.s_bm_examined += want - cur;
                    cx.s_bytes += 4ull * (want - cur);
                }
                if (want >= u.blk_end) break;
                // (the shared histogram costs an L2 round trip: consulted every DS2I_FLOOR_EVERY-th block of list 0)
                if (shared_floor && (floor_tick++ & (DS2I_FLOOR_EVERY - 1)) == 0) adopt_floor();
                const uint32_t blk2 = skip_list0(want);
                if (blk2 >= u.blk_end) break;
                cx.decode_docs(0, blk2, (have_bi && blk2 == want) ? &bi0 : nullptr);
                need0 = false;
                if constexpr (!RANKED) {
                    if (use_rmw) { // once per block of list 0: who can be a member of every other list at all
                        const uint32_t n0c = L.docs[0][lane], n1c = L.docs[0][lane + 64];
                        okm0 = ballot(and_filter(n0c, n0c != 0xFFFFFFFFu));
                        okm1 = ballot(and_filter(n1c, n1c != 0xFFFFFFFFu));
                    }
                }
            }
            uint32_t hi = cx.m(0, M_BMAX);
            const uint32_t c0 = L.docs[0][lane], c1 = L.docs[0][lane + 64];
            bool al0 = c0 >= lo && c0 != 0xFFFFFFFFu, al1 = c1 >= lo && c1 != 0xFFFFFFFFu;
            if constexpr (!RANKED) {
                al0 = al0 && ((okm0 >> lane) & 1);
                al1 = al1 && ((okm1 >> lane) & 1);
            }
            // ranked_and scores PROGRESSIVELY: pa0 / pa1 are the running float32 sums of this lane's two candidates in
            // list order (queries.hpp:372-380). `sf` (wave-uniform) = the heap is full or a floor is known, so bounds
            // can prune: then the list-0 term scores of the whole block are computed up front (once per block, kept in
            // LDS); otherwise they are computed for the candidates that survive list 1, when they are first needed.
            const bool sf = RANKED && bmw && (use_rmw || can_prune());
            float pa0 = 0.f, pa1 = 0.f;
            uint32_t ql0 = 0, qh0 = 0, ql1 = 0, qh1 = 0; // this lane's two candidates: packed range-table bytes
            bool have_p = false;
            if constexpr (RANKED) {
                if (sf || nt == 1) {
                    const uint32_t cur0 = cx.m(0, M_CUR);
                    if (part_blk != cur0) { // once per block of list 0: its freqs, the norm_lens and the list-0 term scores
                        const float qw0 = __uint_as_float(cx.m(0, M_QW));
                        bool v0 = c0 != 0xFFFFFFFFu, v1 = c1 != 0xFFFFFFFFu;
                        float r0 = __uint_as_float(cx.m(0, M_SUF)), r1 = r0;
                        if (sf && use_rmw) {
                            // Range tables first, with the BLOCK's weight standing in for the candidates' list-0 scores: most
                            // blocks of the driving list hold no document that is both inside every other list's ranges and
                            // able to enter the heap -- those are left without decoding their freqs or touching a norm_len.
                            PT_BEGIN(cx);
                            const float wblk = qw0 * (stream0 ? __uint_as_float(bcast(__float_as_uint(s_w), cur0 - s_first))
                                                              : w0tab[cur0]); // (uniform load, in flight with the gathers)
                            v0 = rmw_gather(c0, v0, ql0, qh0);
                            v1 = rmw_gather(c1, v1, ql1, qh1);
                            r0 = rmw_rest(ql0, qh0, 0);
                            r1 = rmw_rest(ql1, qh1, 0);
                            v0 = v0 && tk.would_enter((wblk + r0) * BOUND_SLACK);
                            v1 = v1 && tk.would_enter((wblk + r1) * BOUND_SLACK);
                            if (a.rmh && (ballot(v0) | ballot(v1))) {
                                // membership hints (BatchArgs::rmh): for the candidates the weight bytes let through, every other
                                // list says WHICH document of the candidate's range is its own (where it is the only one there):
                                // a candidate elsewhere in that range is in no intersection with the list -- settled by one more
                                // byte per list instead of a block search and a block decode. All loads first, then the tests.
                                const long long hd = (long long)(a.rmh - a.rmw);
                                uint32_t h0[RL] = {}, h1[RL] = {};
                                auto load_hint = [&](auto ic) __attribute__((always_inline)) {
                                    constexpr uint32_t i = decltype(ic)::value;
                                    const uint8_t* ht = rmw + 64ull * cx.m(i, M_RBASE) + hd;
                                    const uint32_t sh = cx.m(i, M_RSHIFT);
                                    h0[i] = (v0 && sh != 0u) ? (uint32_t)ht[c0 >> sh] : 255u; // (one doc-id per entry: the weight byte was the answer)
                                    h1[i] = (v1 && sh != 0u) ? (uint32_t)ht[c1 >> sh] : 255u;
                                    return true;
                                };
                                static_list_loop<1, RL>(nt, load_hint);
                                auto test_hint = [&](auto ic) __attribute__((always_inline)) {
                                    constexpr uint32_t i = decltype(ic)::value;
                                    const uint32_t sh = cx.m(i, M_RSHIFT);
                                    v0 = v0 && ((h0[i] == 255u) | (h0[i] == rmh_code(c0, sh)));
                                    v1 = v1 && ((h1[i] == 255u) | (h1[i] == rmh_code(c1, sh)));
                                    return true;
                                };
                                static_list_loop<1, RL>(nt, test_hint);
                            }
                            PT_END(cx, PH_PROBE);
#ifdef DS2I_PHASE_TIMING
                            cx.s_phase[PH_C_SURV1] += __builtin_popcountll(ballot(v0)) + __builtin_popcountll(ballot(v1));
#endif
                            if (!(ballot(v0) | ballot(v1))) {
#ifdef DS2I_PHASE_TIMING
                                cx.s_phase[PH_PROLOG] += __builtin_readcyclecounter() - round_t0;
                                cx.s_phase[PH_INSERT] += 1; // (count of such rounds)
#endif
                                if (hi == 0xFFFFFFFFu) break;
                                lo = hi + 1;
                                continue;
                            }
                            L.qb[lane] = ql0;
                            L.qb[lane + 64] = ql1;
                            if constexpr (TMAX > 4) { L.qb2[lane] = qh0; L.qb2[lane + 64] = qh1; }
                        }
                        if (!cx.freqs_ready(0)) cx.decode_freqs(0);
                        PT_BEGIN(cx);
                        const uint32_t f0 = L.freqs[0][lane], f1 = L.freqs[0][lane + 64];
                        if (sf) {
                            // The 4-byte norm_len gather is the path's largest source of memory traffic (a 64-byte
                            // request each). doc_term_weight falls with norm_len, so the freq alone bounds the term
                            // score: a posting whose bound (shortest document of the collection) cannot reach the heap
                            // is dropped before its norm_len is fetched. The heap only tightens, so the verdict holds
                            // for every later round of this block (-inf marks the dropped postings).
                            // With range tables the other lists' part of the bound is per candidate (r0 / r1 above).
                            v0 = v0 && tk.would_enter((qw0 * doc_term_weight(f0, a.min_norm_len) + r0) * BOUND_SLACK);
                            v1 = v1 && tk.would_enter((qw0 * doc_term_weight(f1, a.min_norm_len) + r1) * BOUND_SLACK);
                        }
                        const float n0 = v0 ? a.norm_lens[c0] : -1.f, n1 = v1 ? a.norm_lens[c1] : -1.f;
                        L.nl[lane] = n0;
                        L.nl[lane + 64] = n1;
                        L.part0[lane] = v0 ? qw0 * doc_term_weight(f0, n0) : -__builtin_inff();
                        L.part0[lane + 64] = v1 ? qw0 * doc_term_weight(f1, n1) : -__builtin_inff();
                        part_blk = cur0;
#ifdef DS2I_PHASE_TIMING
                        {
                            const float p0_ = L.part0[lane], p1_ = L.part0[lane + 64];
                            cx.s_phase[PH_C_SURV2] += __builtin_popcountll(ballot(v0 && tk.would_enter((p0_ + r0) * BOUND_SLACK))) +
                                                      __builtin_popcountll(ballot(v1 && tk.would_enter((p1_ + r1) * BOUND_SLACK)));
                        }
#endif
                        const uint32_t nv = (uint32_t)(__builtin_popcountll(ballot(v0)) + __builtin_popcountll(ballot(v1)));
                        cx.s_bytes += 4ull * nv;
                        cx.s_scored += nv;
                        wave_sync();
                        PT_END(cx, PH_SCORE);
                    }
                    pa0 = L.part0[lane]; // list-0 term scores of this lane's two candidates (-inf: dropped at block init)
                    pa1 = L.part0[lane + 64];
                    have_p = true;
                    if (sf) {
                        float r0 = __uint_as_float(cx.m(0, M_SUF)), r1 = r0;
                        if (use_rmw) {
                            ql0 = L.qb[lane];
                            ql1 = L.qb[lane + 64];
                            if constexpr (TMAX > 4) { qh0 = L.qb2[lane]; qh1 = L.qb2[lane + 64]; }
                            r0 = rmw_rest(ql0, qh0, 0);
                            r1 = rmw_rest(ql1, qh1, 0);
                        }
                        // (pa >= 0 excludes the dropped postings explicitly: while the heap is not full would_enter(-inf) holds)
                        al0 = al0 && pa0 >= 0.f && tk.would_enter((pa0 + r0) * BOUND_SLACK);
                        al1 = al1 && pa1 >= 0.f && tk.would_enter((pa1 + r1) * BOUND_SLACK);
                    }
                }
            }
#ifdef DS2I_PHASE_TIMING
            cx.s_phase[PH_C_LIVEROUNDS] += 1;
#endif
            bool pruned = false;
            auto probe_list = [&](auto ic) __attribute__((always_inline)) -> bool {
                const uint32_t i = ic;
                if constexpr (!RANKED && !WITH_FREQS) { if ((bm_lists >> i) & 1u) return true; } // settled by its bitmap
                uint64_t b0 = ballot(al0), b1 = ballot(al1);
                if (!(b0 | b1)) return false;
                uint32_t amin = b0 ? bcast(c0, (uint32_t)__builtin_ctzll(b0)) : bcast(c1, (uint32_t)__builtin_ctzll(b1));
                if (cx.m(i, M_CUR) == 0xFFFFFFFFu || amin > cx.m(i, M_BMAX)) {
                    uint32_t cur = cx.m(i, M_CUR);
                    uint32_t blk, nbmax = 0;
                    float wnew = 0.f;
                    typename decltype(cx)::BlockInfo bi;
                    const bool tabbed = META::SKIPTAB && !cx.is_pef() && cx.skip;
                    const float* wtab = (RANKED && sf) ? bmw + cx.m(i, M_PBASE) : nullptr;
                    {
                        PT_BEGIN(cx);
                        if (tabbed) { blk = cx.find_block_info(i, cur + 1, amin, bi, wtab, wnew); nbmax = bi.bmax; }
                        else blk = cx.find_block(i, cur + 1, amin, nbmax, wtab, wnew);
                        PT_END(cx, PH_FIND);
                    }
                    if (blk >= cx.m(i, M_NB)) { // list i has nothing >= amin: no further match exists
                        cx.s_bm_examined += 1;
                        cx.s_bytes += 4;
                        al0 = al1 = false;
                        finished = true;
                        return false;
                    }
                    // a lazily bound list is positioned by one 64-ary search, not a scan from block 0
                    cx.s_bm_examined += (cur == 0xFFFFFFFFu) ? 1u : blk - cur;
                    cx.s_bytes += 4ull * ((cur == 0xFFFFFFFFu) ? 1u : blk - cur);
                    bool skip_decode = false;
                    if constexpr (RANKED) {
                        if (sf) {
                            // best alive partial score inside [lo, min(hi, block_max)] + this block's max weight + the
                            // later lists' maxima: if that cannot enter the heap the block is not even decoded
                            const uint32_t wh = nbmax < hi ? nbmax : hi;
                            const float cbw = __uint_as_float(cx.m(i, M_QW)) * wnew;
                            float pm, p1, tail = __uint_as_float(cx.m(i, M_SUF));
                            if (use_rmw) { // per candidate: its partial + min(block weight, its own byte bound in list i) + its later lists
                                float t0 = cbw, t1 = cbw;
                                if constexpr (std::is_same<decltype(ic), uint32_t>::value) {
                                    // (run-time list slot: the 5-8-list class; keep the block weight for list i)
                                } else {
                                    const float b0 = rmw_term(ql0, qh0, ic), b1 = rmw_term(ql1, qh1, ic);
                                    t0 = b0 < cbw ? b0 : cbw;
                                    t1 = b1 < cbw ? b1 : cbw;
                                }
                                pm = (al0 && c0 <= wh) ? (pa0 + t0) + rmw_rest(ql0, qh0, i) : 0.f;
                                p1 = (al1 && c1 <= wh) ? (pa1 + t1) + rmw_rest(ql1, qh1, i) : 0.f;
                                tail = 0.f;
                            } else {
                                pm = (al0 && c0 <= wh) ? pa0 + cbw : cbw;
                                p1 = (al1 && c1 <= wh) ? pa1 + cbw : cbw;
                            }
                            pm = p1 > pm ? p1 : pm; // scores are >= 0: their bit patterns order like the values
                            pm = __uint_as_float(bcast(wave_incl_max_scan(__float_as_uint(pm)), 63));
                            if (!tk.would_enter((pm + tail) * BOUND_SLACK)) {
                                hi = wh;
                                skip_decode = true;
                            }
                        }
                    }
                    if (skip_decode) {
                        pruned = true;
                        return false;
                    }
                    cx.decode_docs(i, blk, tabbed ? &bi : nullptr);
#ifdef DS2I_PHASE_TIMING
                    cx.s_phase[PH_C_BDOCS] += 1;
#endif
                }
                PT_BEGIN(cx);
                uint32_t bm = cx.m(i, M_BMAX);
                hi = bm < hi ? bm : hi;
                bool w0 = al0 && c0 <= hi, w1 = al1 && c1 <= hi;
                const uint32_t* d = L.docs[i];
                uint64_t wb0 = ballot(w0), wb1 = ballot(w1);
                uint32_t nw = (uint32_t)(__builtin_popcountll(wb0) + __builtin_popcountll(wb1));
                uint32_t p0 = 0, p1 = 0;
                if (nw > 24) {
                    al0 = member_bsearch(d, c0, w0, p0);
                    al1 = member_bsearch(d, c1, w1, p1);
                } else {
                    // few candidates: broadcast each, two equality ballots over the block
                    const uint32_t d0 = d[lane], d1 = d[lane + 64];
                    uint64_t r0 = 0, r1 = 0;
                    for (int half = 0; half < 2; ++half) {
                        uint64_t todo = half ? wb1 : wb0;
                        while (todo) {
                            uint32_t src = (uint32_t)__builtin_ctzll(todo);
                            todo &= todo - 1;
                            uint32_t c = bcast(half ? c1 : c0, src);
                            uint64_t e0 = ballot(d0 == c), e1 = ballot(d1 == c);
                            if (e0 | e1) {
                                uint32_t pp = e0 ? (uint32_t)__builtin_ctzll(e0) : 64u + (uint32_t)__builtin_ctzll(e1);
                                if (half) { r1 |= 1ull << src; if (lane == src) p1 = pp; }
                                else { r0 |= 1ull << src; if (lane == src) p0 = pp; }
                            }
                        }
                    }
                    al0 = (r0 >> lane) & 1;
                    al1 = (r1 >> lane) & 1;
                }
                PT_END(cx, PH_MEMBER);
                if constexpr (RANKED) { // members take list i's term score at once
                    if (ballot(al0) | ballot(al1)) {
                        if (!have_p) { // (only possible at list 1) list-0 term scores of the surviving candidates
                            if (!cx.freqs_ready(0)) cx.decode_freqs(0);
                            const float qw0 = __uint_as_float(cx.m(0, M_QW));
                            const float n0 = al0 ? a.norm_lens[c0] : 0.f, n1 = al1 ? a.norm_lens[c1] : 0.f;
                            L.nl[lane] = n0;
                            L.nl[lane + 64] = n1;
                            pa0 = al0 ? qw0 * doc_term_weight(L.freqs[0][lane], n0) : 0.f;
                            pa1 = al1 ? qw0 * doc_term_weight(L.freqs[0][lane + 64], n1) : 0.f;
                            part_blk = 0xFFFFFFFFu; // L.nl now holds this round's survivors only
                            have_p = true;
                            const uint32_t nv = (uint32_t)(__builtin_popcountll(ballot(al0)) + __builtin_popcountll(ballot(al1)));
                            cx.s_bytes += 4ull * nv;
                            cx.s_scored += nv;
                        }
#ifdef DS2I_PHASE_TIMING
                        if (!cx.freqs_ready(i)) cx.s_phase[PH_C_BFREQS] += 1;
#endif
                        if (!cx.freqs_ready(i)) cx.decode_freqs(i);
                        const float qw = __uint_as_float(cx.m(i, M_QW));
                        const uint32_t* f = cx.F(i);
                        if (al0) pa0 = pa0 + qw * doc_term_weight(f[p0], L.nl[lane]);
                        if (al1) pa1 = pa1 + qw * doc_term_weight(f[p1], L.nl[lane + 64]);
                        if (sf) { // who cannot reach the heap any more drops out before the next list is touched
                            float r0 = __uint_as_float(cx.m(i, M_SUF)), r1 = r0;
                            if (use_rmw) { r0 = rmw_rest(ql0, qh0, i); r1 = rmw_rest(ql1, qh1, i); }
                            al0 = al0 && tk.would_enter((pa0 + r0) * BOUND_SLACK);
                            al1 = al1 && tk.would_enter((pa1 + r1) * BOUND_SLACK);
                        }
                    }
                    return true;
                }
                if constexpr (WITH_FREQS && !RANKED) {
                    if (al0) L.pos[i][lane] = (uint8_t)p0;
                    if (al1) L.pos[i][lane + 64] = (uint8_t)p1;
                }
                return true;
            };
            DS2I_LIST_LOOP(1, probe_list)
            if (RANKED && pruned) { // the window [lo, hi] holds no document that could enter the heap
                if (hi == 0xFFFFFFFFu) break;
                lo = hi + 1;
                continue;
            }
            // candidates that survived every list and lie inside the window are matches
            al0 = al0 && c0 <= hi;
            al1 = al1 && c1 <= hi;
            const uint64_t s0 = ballot(al0), s1 = ballot(al1);
            const uint32_t ns = (uint32_t)(__builtin_popcountll(s0) + __builtin_popcountll(s1));
            if (ns) {
                if (a.out_matches) {
                    const uint64_t lt = (1ull << lane) - 1;
                    unsigned long long i0 = count + __builtin_popcountll(s0 & lt);
                    unsigned long long i1 = count + __builtin_popcountll(s0) + __builtin_popcountll(s1 & lt);
                    if (al0 && i0 < mcap) a.out_matches[mbase + i0] = c0;
                    if (al1 && i1 < mcap) a.out_matches[mbase + i1] = c1;
                }
                count += ns;
                if constexpr (WITH_FREQS && !RANKED) { // and_query<with_freqs> touches every matching posting's freq
                    wave_sync(); // L.pos writes visible
                    unsigned long long fs = 0; // freqs are u32: the checksum must not wrap at 2^32
                    auto freq_list = [&](auto ic) __attribute__((always_inline)) -> bool {
                        const uint32_t i = ic;
                        if (!cx.m(i, M_FDEC)) cx.decode_freqs(i);
                        const uint32_t* f = L.freqs[i];
                        uint32_t f0 = 0, f1 = 0;
                        if (al0) f0 = f[i ? L.pos[i][lane] : lane];
                        if (al1) f1 = f[i ? L.pos[i][lane + 64] : lane + 64];
                        fs += (unsigned long long)f0 + f1;
                        return true;
                    };
                    DS2I_LIST_LOOP(0, freq_list)
                    fsum += fs; // (this lane's share: reduced over the wave once, when the unit ends)
                }
                if constexpr (RANKED) {
                    // pa0 / pa1 are complete scores now; only those that can enter the heap are inserted (serial, rare once warm)
#ifdef DS2I_PHASE_TIMING
                    const unsigned long long tk_t0 = __builtin_readcyclecounter();
#endif
                    for (int half = 0; half < 2; ++half) {
                        const bool al = half ? al1 : al0;
                        const float sc = half ? pa1 : pa0;
                        uint64_t todo = ballot(al && tk.would_enter(sc));
                        while (todo) {
                            uint32_t src = (uint32_t)__builtin_ctzll(todo);
                            todo &= todo - 1;
                            const float v = __uint_as_float(bcast(__float_as_uint(sc), src));
#ifdef DS2I_PHASE_TIMING
                            cx.s_phase[PH_C_HEAP] += 1;
#endif
                            if (tk.insert(v) && shared_floor && lane == 0) sh.add(v);
                        }
                    }
#ifdef DS2I_PHASE_TIMING
                    cx.s_phase[PH_TOPK] += __builtin_readcyclecounter() - tk_t0;
#endif
                }
            }
            if (hi == 0xFFFFFFFFu) break;
            lo = hi + 1;
        }
#ifdef DS2I_PHASE_TIMING
        cx.s_phase[PH_TOTAL] += __builtin_readcyclecounter() - unit_t0;
#endif
        if constexpr (WITH_FREQS && !RANKED) {
            for (int o = 32; o; o >>= 1) fsum += __shfl_xor(fsum, o);
        }
        if (whole) {
            if (lane == 0) {
                a.out_count[q] = RANKED ? tk.n : count;
                if (a.out_freq_sum) a.out_freq_sum[q] = fsum;
            }
            if (RANKED) store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
        } else {
            if (lane == 0) {
                a.unit_count[uid] = RANKED ? tk.n : count;
                a.unit_freq_sum[uid] = fsum;
            }
            if (RANKED) store_topk(a.unit_topk, a.unit_topk_len, a.k, uid, tk);
        }
        if (STATS && a.unit_clock && lane == 0) { a.unit_clock[2ull * uid] = t_unit; a.unit_clock[2ull * uid + 1] = wall_clock64(); }
    }
    cx.flush_stats(a.stats);
}

// Merges the partial results of split queries: counts add up, the top-k of a union is the top-k of
// the parts' top-ks (scores are per-document, so the merged multiset equals the sequential one).
__global__ void __launch_bounds__(64) k_merge(MergeArgs a) {
    const uint32_t lane = lane_id();
    for (uint32_t w = blockIdx.x; w < a.nsplit; w += gridDim.x) {
        const uint32_t q = a.split_queries[w];
        const uint32_t u0 = a.q_unit_off[q], u1 = a.q_unit_off[q + 1];
        unsigned long long count = 0, fsum = 0;
        for (uint32_t u = u0 + lane; u < u1; u += 64) { count += a.unit_count[u]; fsum += a.unit_freq_sum[u]; }
        for (int o = 32; o; o >>= 1) {
            count += __shfl_xor(count, o);
            fsum += __shfl_xor(fsum, o);
        }
        TopK tk;
        tk.init(a.k);
        if (a.ranked) {
            for (uint32_t u = u0; u < u1; ++u) {
                const uint32_t len = a.unit_topk_len[u];
                float v = lane < len ? a.unit_topk[(size_t)u * a.k + lane] : -__builtin_inff();
                uint64_t todo = ballot(lane < len && tk.would_enter(v));
                while (todo) {
                    uint32_t src = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    tk.insert(__uint_as_float(bcast(__float_as_uint(v), src)));
                }
            }
            store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
        }
        if (lane == 0) {
            a.out_count[q] = a.ranked ? tk.n : count;
            if (a.out_freq_sum) a.out_freq_sum[q] = fsum;
        }
    }
}

// ------------------------------------------------------------------ document-at-a-time
template <class CX>
DS2I_DEV float score_of(CX& cx, uint32_t s, float norm_len) {
    return __uint_as_float(cx.m(s, M_QW)) * doc_term_weight(cx.freq(s), norm_len);
}

// stable insertion sort of ord[0..n) by key(slot) (== libstdc++ std::sort for n <= 16)
template <class Key>
DS2I_DEV void sort_ord(uint32_t* ord, uint32_t n, Key key) {
    if (lane_id() == 0) {
        for (uint32_t i = 1; i < n; ++i) {
            uint32_t v = ord[i];
            auto kv = key(v);
            uint32_t j = i;
            while (j > 0 && kv < key(ord[j - 1])) { ord[j] = ord[j - 1]; --j; }
            ord[j] = v;
        }
    }
    wave_sync();
}

// One unit of a reference-order operator. The per-list enumerator state (`meta`: M_WORDS dwords per slot, the same
// memory cx.meta points to), the list order `ord` and the maxscore upper bounds `ub` live wherever the caller keeps
// them: LDS for the <=16-term classes (k_daat), a global scratch area for longer queries (k_daat_long).
template <int OP, class TK = TopK, class CX>
DS2I_DEV void daat_unit(CX& cx, const BatchArgs& a, const uint32_t uid, uint32_t* meta, uint32_t* ord, float* ubs, const uint32_t tmax) {
    const uint32_t lane = lane_id();
    constexpr bool RANKED = OP >= OP_RANKED_AND;
    {
        // a unit of these operators is a doc-id range [blk_begin, blk_end) of the query (whole range when
        // nparts == 1); inside the unit blk_end plays the role of num_docs (the exhaustion sentinel)
        const Unit u = a.units[uid];
        const uint32_t q = u.q;
        const bool whole = u.nparts == 1;
        const uint32_t N = whole ? a.num_docs : u.blk_end;
        cx.num_docs = N;
        const uint32_t t0 = a.q_off[q], nt = a.q_off[q + 1] - t0;
        unsigned long long count = 0, fsum = 0;
        TK tk;
        tk.init(a.k);
        if (nt == 0 || nt > tmax) {
            if (lane == 0) { a.out_count[q] = 0; if (a.out_freq_sum) a.out_freq_sum[q] = 0; }
            if (RANKED) store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
            return;
        }
        if (whole) {
            for (uint32_t i = 0; i < nt; ++i) cx.open(i, a.qterms[t0 + i]);
        } else {
            for (uint32_t i = 0; i < nt; ++i) { cx.bind(i, a.qterms[t0 + i]); cx.next_geq(i, u.blk_begin); }
        }
        if (OP == OP_WAND || OP == OP_MAXSCORE) {
            cx.s_bytes += 4ull * nt; // max_term_weight[term]
            if (a.seed_topk && a.seed_len[q] >= a.k) {
                // the ranked_and pass found >= k documents: its k-th score bounds the final k-th score from below.
                // The two operators sum a document's terms in different orders (size- vs docid-sorted lists), so the
                // floor is relaxed by 1e-5 relative -- far above float32 re-association noise, far below any pruning loss
                const float kth = a.seed_topk[(size_t)q * a.k + a.k - 1];
                tk.floor = __uint_as_float(uniform(__float_as_uint(kth * (1.0f - 1.0e-5f))));
            }
        }
        auto norm_len = [&](uint32_t d) {
            cx.s_bytes += 4;
            ++cx.s_scored;
            return __uint_as_float(uniform(__float_as_uint(a.norm_lens[d])));
        };

        if (OP == OP_AND || OP == OP_AND_FREQ || OP == OP_RANKED_AND) {
            // reference traversal, one candidate at a time (queries.hpp:58-84 / 362-387)
            const unsigned long long mbase = a.out_matches ? a.match_off[q] : 0;
            const unsigned long long mcap = a.out_matches ? a.match_off[q + 1] - mbase : 0;
            uint32_t cand = cx.docid(0);
            uint32_t i = 1;
            while (cand < N) {
                for (; i < nt; ++i) {
                    cx.next_geq(i, cand);
                    uint32_t d = cx.docid(i);
                    if (d != cand) { cand = d; i = 0; break; }
                }
                if (i == nt) {
                    if (OP == OP_RANKED_AND) {
                        float nl = norm_len(cand), score = 0.f;
                        for (i = 0; i < nt; ++i) score += score_of(cx, i, nl);
                        tk.insert(score);
                    } else {
                        if (a.out_matches && lane == 0 && count < mcap) a.out_matches[mbase + count] = cand;
                        ++count;
                        if (OP == OP_AND_FREQ) for (i = 0; i < nt; ++i) fsum += cx.freq(i);
                    }
                    cx.next(0);
                    cand = cx.docid(0);
                    i = 1;
                }
            }
        } else if (OP == OP_OR || OP == OP_OR_FREQ || OP == OP_RANKED_OR) {
            // queries.hpp:105-127 / 438-462
            uint32_t cur = N;
            for (uint32_t i = 0; i < nt; ++i) { uint32_t d = cx.docid(i); cur = d < cur ? d : cur; }
            while (cur < N) {
                float score = 0.f, nl = 0.f;
                if (OP == OP_RANKED_OR) nl = norm_len(cur);
                uint32_t nxt = N;
                for (uint32_t i = 0; i < nt; ++i) {
                    if (cx.docid(i) == cur) {
                        if (OP == OP_RANKED_OR) score += score_of(cx, i, nl);
                        if (OP == OP_OR_FREQ) fsum += cx.freq(i);
                        cx.next(i);
                    }
                    uint32_t d = cx.docid(i);
                    nxt = d < nxt ? d : nxt;
                }
                if (OP == OP_RANKED_OR) tk.insert(score); else ++count;
                cur = nxt;
            }
        } else if (OP == OP_WAND) {
            // queries.hpp:236-305
            if (lane == 0) for (uint32_t i = 0; i < nt; ++i) ord[i] = i;
            wave_sync();
            auto by_docid = [&](uint32_t s) { return meta[s * M_WORDS + M_DOCID]; };
            sort_ord(ord, nt, by_docid);
            for (;;) {
                float upper = 0.f;
                uint32_t pivot = 0;
                bool found = false;
                for (pivot = 0; pivot < nt; ++pivot) {
                    uint32_t s = uniform(ord[pivot]);
                    if (cx.docid(s) == N) break;
                    upper += __uint_as_float(cx.m(s, M_MAXW));
                    if (tk.would_enter(upper)) { found = true; break; }
                }
                if (!found) break;
                const uint32_t pivot_id = cx.docid(uniform(ord[pivot]));
                if (pivot_id == cx.docid(uniform(ord[0]))) {
                    float score = 0.f, nl = norm_len(pivot_id);
                    for (uint32_t j = 0; j < nt; ++j) {
                        uint32_t s = uniform(ord[j]);
                        if (cx.docid(s) != pivot_id) break;
                        score += score_of(cx, s, nl);
                        cx.next(s);
                    }
                    tk.insert(score);
                    sort_ord(ord, nt, by_docid);
                } else {
                    uint32_t nl_ = pivot;
                    while (cx.docid(uniform(ord[nl_])) == pivot_id) --nl_;
                    cx.next_geq(uniform(ord[nl_]), pivot_id);
                    if (lane == 0) {
                        for (uint32_t j = nl_ + 1; j < nt; ++j) {
                            uint32_t x = ord[j], y = ord[j - 1];
                            if (meta[x * M_WORDS + M_DOCID] < meta[y * M_WORDS + M_DOCID]) { ord[j] = y; ord[j - 1] = x; }
                            else break;
                        }
                    }
                    wave_sync();
                }
            }
        } else { // OP_MAXSCORE, queries.hpp:514-577
            if (lane == 0) for (uint32_t i = 0; i < nt; ++i) ord[i] = i;
            wave_sync();
            auto by_maxw = [&](uint32_t s) { return __uint_as_float(meta[s * M_WORDS + M_MAXW]); };
            sort_ord(ord, nt, by_maxw);
            if (lane == 0) {
                float acc = 0.f;
                for (uint32_t i = 0; i < nt; ++i) {
                    float mw = __uint_as_float(meta[ord[i] * M_WORDS + M_MAXW]);
                    acc = i ? acc + mw : mw;
                    ubs[i] = acc;
                }
            }
            wave_sync();
            uint32_t non_ess = 0, cur = N;
            // with a seeded floor some lists are non-essential before the first document (the reference only updates
            // this after a successful insert, queries.hpp:568-574 -- same rule, applied to the initial bound)
            while (non_ess < nt && !tk.would_enter(__uint_as_float(uniform(__float_as_uint(ubs[non_ess])))))
                ++non_ess;
            for (uint32_t i = 0; i < nt; ++i) { uint32_t d = cx.docid(i); cur = d < cur ? d : cur; }
            while (non_ess < nt && cur < N) {
                float score = 0.f, nl = norm_len(cur);
                uint32_t nxt = N;
                for (uint32_t i = non_ess; i < nt; ++i) {
                    uint32_t s = uniform(ord[i]);
                    if (cx.docid(s) == cur) {
                        score += score_of(cx, s, nl);
                        cx.next(s);
                    }
                    uint32_t d = cx.docid(s);
                    nxt = d < nxt ? d : nxt;
                }
                for (uint32_t i = non_ess; i-- > 0;) {
                    float ub = __uint_as_float(uniform(__float_as_uint(ubs[i])));
                    if (!tk.would_enter(score + ub)) break;
                    uint32_t s = uniform(ord[i]);
                    cx.next_geq(s, cur);
                    if (cx.docid(s) == cur) score += score_of(cx, s, nl);
                }
                if (tk.insert(score)) {
                    while (non_ess < nt && !tk.would_enter(__uint_as_float(uniform(__float_as_uint(ubs[non_ess])))))
                        ++non_ess;
                }
                cur = nxt;
            }
        }
        if (whole) {
            if (lane == 0) {
                a.out_count[q] = RANKED ? tk.n : count;
                if (a.out_freq_sum) a.out_freq_sum[q] = fsum;
            }
            if (RANKED) store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
        } else {
            if (lane == 0) {
                a.unit_count[uid] = RANKED ? tk.n : count;
                a.unit_freq_sum[uid] = fsum;
            }
            if (RANKED) store_topk(a.unit_topk, a.unit_topk_len, a.k, uid, tk);
        }
    }
}

template <int OP, int TMAX, int CODEC_T = -1>
__global__ void __launch_bounds__(64) k_daat(BatchArgs a) {
    __shared__ Lds<TMAX> L;
    CtxT<CODEC_T, MetaLds> cx = make_ctx<CODEC_T, MetaLds>(L, a);
    for (uint32_t tkt = blockIdx.x; tkt < a.nslice; tkt += gridDim.x)
        daat_unit<OP>(cx, a, a.order[tkt], &L.meta[0][0], L.ord(), L.ub(), (uint32_t)TMAX);
    cx.flush_stats(a.stats);
}

// Queries with more than 16 distinct terms (the reference has no limit, queries.hpp:35-86): the same traversals with
// the per-list state -- 128 doc-ids + 128 freqs + M_WORDS dwords per list, the list order and the upper bounds -- in a
// global scratch area of long_stride dwords per unit instead of LDS; only the decoders' staging stays in LDS.
struct LdsLong {
    uint32_t exc[EXC_LDS_DW];
    uint32_t st[STAGE_DW];
};
template <int OP, class TK = TopK>
__global__ void __launch_bounds__(64) k_daat_long(BatchArgs a) {
    __shared__ LdsLong L;
    CtxT<-1, MetaLds> cx;
    cx.docs = cx.freqs = nullptr;
    cx.meta.p = nullptr;
    cx.exc = L.exc;
    s16_table_init(L.exc);
    cx.win.st = L.st;
    cx.win.gbase = a.arena;
    cx.win.nbytes = 0;
    cx.arena = a.arena;
    cx.bits0 = a.bits0;
    cx.bits1 = a.bits1;
    cx.codec = a.codec;
    cx.num_docs = a.num_docs;
    cx.block_profile = a.block_profile;
    cx.skip = (const uint2*)a.skip;
    cx.init_stats();
    for (uint32_t tkt = blockIdx.x; tkt < a.nslice; tkt += gridDim.x) {
        const uint32_t uid = a.order[tkt];
        const uint32_t q = a.units[uid].q;
        const uint32_t nt = a.q_off[q + 1] - a.q_off[q];
        uint32_t* base = a.long_scratch + (size_t)tkt * a.long_stride;
        cx.docs = base;
        cx.freqs = base + 128u * nt;
        cx.meta.p = base + 256u * nt;
        uint32_t* ord = cx.meta.p + (uint32_t)M_WORDS * nt;
        float* ub = (float*)(ord + nt);
        daat_unit<OP, TK>(cx, a, uid, cx.meta.p, ord, ub, 0xFFFFFFFFu);
    }
    cx.flush_stats(a.stats);
}

// Order-independent score accumulation for the block-synchronous disjunctive kernel. Which of a document's lists are
// essential when it is met -- hence the order its term scores would be added in -- depends on how far the pruning
// threshold has risen, i.e. on the query's split into parts and on timing. Term scores (float32, computed exactly as
// the reference computes them) are therefore summed in fixed point: integer addition is associative, so a
// document's score is the same bits whatever the order, run to run and for wand / maxscore / ranked_or alike. The
// fixed point is RELATIVE to the query: its unit is 2^-62 of the power of two above the query's score bound (the sum of
// its lists' max scores, the same bits in every part of a split query), so the sum of <= 16 terms fits 63 bits and a term
// loses nothing unless it is 2^-38 of the bound -- scores of 1e-6 (terms in most of the documents) are as exact as
// scores of 20. The result differs from the reference's sequential float sum by a few ulps at most (tests hold 1e-5;
// the reference's own ranked test holds 1e-3, test_ranked_queries.cpp:52).
struct FxScale {
    double to_fx, from_fx; // powers of two
    DS2I_DEV void init(float bound) { // bound >= 0, wave-uniform
        const unsigned long long eb = (__float_as_uint(bound) >> 23) & 0xFFu; // bound < 2^(eb - 126)
        to_fx = __longlong_as_double((long long)((1211ull - eb) << 52));       // 2^(62 - (eb - 126))
        from_fx = __longlong_as_double((long long)((835ull + eb) << 52));
    }
    DS2I_DEV unsigned long long of(float term) const { return (unsigned long long)((double)term * to_fx); } // term >= 0
    DS2I_DEV float value(unsigned long long acc) const { return (float)((double)acc * from_fx); }
};

// ------------------------------------------------------------------ block-synchronous disjunctive top-k
// wand / maxscore / ranked_or all return the top-k of the UNION of the query's lists (queries.hpp:200-319, 404-476,
// 478-591; the reference's own ranked test holds them equal, test_ranked_queries.cpp:39-57). The document-at-a-time
// traversals of the reference (k_daat above) advance one document per step; here a step is a WINDOW of doc-ids:
//   * lists are ordered by max score, upper_bounds[] are the prefix sums and the first `non_ess` lists are
//     non-essential exactly as in maxscore_query (queries.hpp:529-547): a document that occurs in them only cannot
//     enter the heap. The threshold starts at the ranked_and seed (every AND result is an OR result).
//   * every essential list keeps one decoded block; the window is [lo, min of their block_max], so all postings of
//     the essential lists inside the window sit in LDS. Each posting is a candidate, owned by the first essential list
//     that contains it (later lists mark their copy as a duplicate);
//   * a candidate's max-score bound (lists it was found in + all non-essential lists) is tested first, the survivors
//     are scored exactly: essential lists by position, non-essential lists probed from the highest bound down while
//     score + upper_bound can still enter (queries.hpp:553-564), 128 candidates at a time.
// Position (0..127) of one candidate in each of the query's lists, 7 bits per list slot, in registers: the union kernels
// are capped by LDS per wave (residency hides their dependent round trips), so what a lane knows about its own two
// candidates stays out of LDS. <=4 lists fit one dword, <=8 one qword, 16 two.
template <int TMAX>
struct PosPack {
    typedef typename std::conditional<(TMAX <= 4), uint32_t, unsigned long long>::type word_t;
    static constexpr int PER = TMAX <= 4 ? 4 : 9, NW = (TMAX + PER - 1) / PER;
    word_t w[NW];
    DS2I_DEV void clear() {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = 0;
    }
    DS2I_DEV void set(uint32_t x, uint32_t pos) { // x wave-uniform
        if (NW == 1 || x < (uint32_t)PER) w[0] |= (word_t)pos << (7u * x);
        else w[NW - 1] |= (word_t)pos << (7u * (x - (uint32_t)PER));
    }
    DS2I_DEV uint32_t get(uint32_t x) const {
        if (NW == 1 || x < (uint32_t)PER) return (uint32_t)(w[0] >> (7u * x)) & 127u;
        return (uint32_t)(w[NW - 1] >> (7u * (x - (uint32_t)PER))) & 127u;
    }
};

template <int TMAX>
struct LdsOr { // (the decoded blocks are in dynamic shared memory, see k_disjunctive)
    uint32_t meta[TMAX][M_WORDS];
    uint32_t exc[EXC_LDS_DW]; // + the Simple16 field table (device_codecs.hpp)
    uint32_t st[STAGE_DW];
    uint32_t lord[16];   // list slots by increasing max score
    float lub[16];       // upper_bounds (prefix sums of max scores in that order)
    float wub[16];       // the same prefix sums for the current window: essential lists by their current block's max weight
    uint32_t nomore[16]; // list has no posting >= this doc-id
    uint32_t dupw[TMAX][4]; // bit i of list x: posting i of its current block is owned by an earlier list in this window
};

// MODE 0: top-k (wand / maxscore / ranked_or). MODE 1: or_query (count of the union, queries.hpp:88-131): every list is
// essential, owned candidates are counted. MODE 2: or_query<with_freqs>: additionally every freq of the window is
// summed (the reference touches them all).
// waves per SIMD the top-k instantiations are compiled for: the kernel waits on dependent round trips most of the time, so
// residency is what hides them. <=2 lists: LDS (5.5 KiB per wave) would allow 7; beyond 4 lists LDS caps the residency first
#ifndef DS2I_BLOCKMAX_TMAX
#define DS2I_BLOCKMAX_TMAX 2 // block-max pruning in the top-k union kernels of up to this many lists (see k_disjunctive)
#endif
#ifndef DS2I_DISJ_WAVES2
#define DS2I_DISJ_WAVES2 6
#endif
#ifndef DS2I_DISJ_WAVES4
#define DS2I_DISJ_WAVES4 5
#endif
constexpr int DISJ_WAVES(int tmax, int mode) { return mode != 0 ? 1 : tmax <= 2 ? DS2I_DISJ_WAVES2 : tmax <= 4 ? DS2I_DISJ_WAVES4 : 1; }
template <int TMAX, int CODEC_T, bool STATS = true, int MODE = 0>
__global__ void __launch_bounds__(64, DISJ_WAVES(TMAX, MODE)) k_disjunctive(BatchArgs a) {
    __shared__ LdsOr<TMAX> L;
    // the decoded blocks live in DYNAMIC shared memory, sized by the launch for the longest query it contains
    // (a.dyn_lists <= TMAX list slots: docs[dyn_lists][128] then freqs[dyn_lists][128]). Residency hides this kernel's
    // dependent round trips and LDS per wave caps residency, so a 5-term query should not pay for 8 lists.
    extern __shared__ uint32_t dyn_lds[];
    uint32_t* const Ldocs = dyn_lds;
    uint32_t* const Lfreqs = dyn_lds + 128u * a.dyn_lists;
    const uint32_t lane = lane_id();
    CtxT<CODEC_T, MetaLds, STATS> cx = make_ctx<CODEC_T, MetaLds, STATS>(L, a, Ldocs, Lfreqs);
    for (uint32_t tkt = blockIdx.x; tkt < a.nslice; tkt += gridDim.x) {
        const uint32_t uid = a.order[tkt];
        const unsigned long long t_unit = (STATS && a.unit_clock) ? wall_clock64() : 0ull;
#ifdef DS2I_PHASE_TIMING
        const unsigned long long pt_unit0 = __builtin_readcyclecounter();
#endif
        const Unit u = a.units[uid];
        const uint32_t q = u.q;
        const bool whole = u.nparts == 1;
        const uint32_t N = whole ? a.num_docs : u.blk_end; // the unit's doc-id range is [lo, N)
        uint32_t lo = whole ? 0u : u.blk_begin;
        const uint32_t t0 = a.q_off[q], nt = a.q_off[q + 1] - t0;
        TopK tk;
        tk.init(a.k);
        unsigned long long count = 0, fsum = 0;
        if (nt == 0 || nt > (uint32_t)TMAX || nt > a.dyn_lists || N == 0) {
            if (whole) {
                if (lane == 0) { a.out_count[q] = 0; if (a.out_freq_sum) a.out_freq_sum[q] = 0; }
                if (MODE == 0) store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
            } else {
                if (lane == 0) { a.unit_count[uid] = 0; a.unit_freq_sum[uid] = 0; }
                if (MODE == 0) store_topk(a.unit_topk, a.unit_topk_len, a.k, uid, tk);
            }
            continue;
        }
        for (uint32_t i = 0; i < nt; ++i) cx.bind(i, a.qterms[t0 + i]);
        if (MODE == 0) cx.s_bytes += 4ull * nt; // max_term_weight[term]
        if (MODE == 0 && a.seed_topk && a.seed_len[q] >= a.k) { // see k_daat: ranked_and's k-th score, relaxed by 1e-5
            const float kth = a.seed_topk[(size_t)q * a.k + a.k - 1];
            tk.floor = __uint_as_float(uniform(__float_as_uint(kth * (1.0f - 1.0e-5f))));
        }
        if (MODE == 0) {
            // static floor (host, from the upload-time block weights): some term of the query has k blocks whose best
            // posting alone scores >= floor1, and a document's score is >= any one of its term scores
            const float f1 = __uint_as_float(uniform(__float_as_uint(a.qterms[t0].floor1))) * (1.0f - 1.0e-5f);
            if (f1 > tk.floor) tk.floor = f1;
        }
        if (lane == 0) {
            for (uint32_t i = 0; i < nt; ++i) { L.lord[i] = i; L.nomore[i] = 0xFFFFFFFFu; }
            for (uint32_t i = 1; i < nt; ++i) { // stable insertion sort by max score (queries.hpp:529-533)
                const uint32_t v = L.lord[i];
                const float kv = __uint_as_float(L.meta[v][M_MAXW]);
                uint32_t j = i;
                while (j > 0 && kv < __uint_as_float(L.meta[L.lord[j - 1]][M_MAXW])) { L.lord[j] = L.lord[j - 1]; --j; }
                L.lord[j] = v;
            }
            float acc = 0.f;
            for (uint32_t i = 0; i < nt; ++i) {
                const float mw = __uint_as_float(L.meta[L.lord[i]][M_MAXW]);
                acc = i ? acc + mw : mw;
                L.lub[i] = acc;
            }
        }
        wave_sync();
        auto ubf = [&](uint32_t i) { return __uint_as_float(uniform(__float_as_uint(L.lub[i]))); };
        auto ubw = [&](uint32_t i) { // this window's bounds (block-max pruning) or the list-level upper_bounds
            return __uint_as_float(uniform(__float_as_uint((MODE == 0 && TMAX <= DS2I_BLOCKMAX_TMAX) ? L.wub[i] : L.lub[i])));
        };
        auto slot_at = [&](uint32_t p) { return uniform(L.lord[p]); };
        auto maxw = [&](uint32_t x) { return __uint_as_float(cx.m(x, M_MAXW)); };
        auto qw = [&](uint32_t x) { return __uint_as_float(cx.m(x, M_QW)); };
        FxScale fx;
        fx.init(ubf(nt - 1));
        uint32_t non_ess = 0;
        auto update_non_ess = [&]() __attribute__((always_inline)) { if (MODE == 0) while (non_ess < nt && !tk.would_enter(ubf(non_ess))) ++non_ess; };
        update_non_ess();
        auto cbw = [&](uint32_t x) { return __uint_as_float(cx.m(x, M_CBW)); }; // q_weight * bmw of the block list x is positioned on
        // POSITIONS list x on the first block whose block_max >= d; false when the list has no posting >= d. With the
        // interleaved skip table the block is only located (table words + its max weight into the list's state) and decoded
        // later, by ensure_docs(), if it can still matter; without it (Elias-Fano layouts) it is decoded at once.
        // `skipping`: the list moves on from its current block after a window that could not hold a result; blocks whose
        // own best posting, added to `rest` (the bound of all the other lists, valid up to doc-id hi2), cannot enter are
        // passed over unexamined (find_block_where).
        // always_inline: left to the inliner, the lambda becomes a real call once it grows, and the call ABI costs this
        // kernel half its throughput
        // block codecs compiled for one codec always come with the table (launch_batch passes it to these kernels
        // unconditionally), so the decode-at-once path is not even compiled into them: one copy less of the decoder
        constexpr bool ALWAYS_TABBED = CODEC_T >= 0 && CODEC_T != CODEC_PEF;
        const bool tabbed = ALWAYS_TABBED || (!cx.is_pef() && cx.skip);
        // Block-max pruning (position first, decode if the block can matter; per-window bounds from the blocks' max weights)
        // pays for queries of <= 2 lists: half the blocks are never decoded. With more lists the sum of the block maxima
        // is rarely below the threshold (measured on the GOV2-scale batch: 3-4 lists decode 6 % fewer blocks, 5+ none) and
        // the bookkeeping costs more than it saves, so those classes keep list-level bounds and decode as they position.
        constexpr bool BLOCKMAX = MODE == 0 && TMAX <= DS2I_BLOCKMAX_TMAX;
        auto position = [&](uint32_t x, uint32_t d, bool may_go_back, bool skipping, float rest, uint32_t hi2, bool eager_freqs) __attribute__((always_inline)) -> bool {
            if (d >= uniform(L.nomore[x])) return false;
            const uint32_t cur = cx.m(x, M_CUR);
            uint32_t from;
            if (cur == 0xFFFFFFFFu) from = 0;
            else if (d > cx.m(x, M_BMAX)) from = cur + 1;
            else if (may_go_back && d < (tabbed ? cx.m(x, M_BASE) : uniform((Ldocs + 128u * x)[0]))) from = 0; // a non-essential list may have been moved ahead
            else return true;
            uint32_t blk, bmax_u;
            float w = 0.f;
            typename decltype(cx)::BlockInfo bi;
            const float* wtab = (BLOCKMAX && a.bmw) ? a.bmw + cx.m(x, M_PBASE) : nullptr;
            {
                PT_BEGIN(cx);
                if (!ALWAYS_TABBED && !tabbed) blk = cx.find_block(x, from, d, bmax_u, wtab, w);
                else if (BLOCKMAX && skipping && wtab) {
                    const float qx = qw(x);
                    blk = cx.find_block_where(x, from, d, bi, wtab, w, [&](uint32_t bm, float wv) { return bm >= hi2 || tk.would_enter((rest + qx * wv) * BOUND_SLACK); });
                } else blk = cx.find_block_info(x, from, d, bi, wtab, w);
                PT_END(cx, PH_FIND);
            }
            if (blk >= cx.m(x, M_NB)) {
                if (lane == 0) L.nomore[x] = d;
                wave_sync();
                return false;
            }
            cx.s_bm_examined += skipping ? blk - from + 1 : 1u;
            cx.s_bytes += 4;
            if (blk != cur) {
                if (BLOCKMAX && tabbed) {
                    cx.setm(x, M_CUR, blk);
                    cx.setm(x, M_BMAX, bi.bmax);
                    cx.setm(x, M_EP, bi.ep);
                    cx.setm(x, M_NEXTEP, bi.next_ep);
                    cx.setm(x, M_BASE, bi.base);
                    cx.setm(x, M_DDEC, 0);
                    cx.setm(x, M_FDEC, 0);
                } else if constexpr (!(BLOCKMAX && ALWAYS_TABBED)) {
                    cx.decode_docs(x, blk, tabbed ? &bi : nullptr);
                    if (MODE == 0 && tabbed) cx.setm(x, M_BASE, bi.base); // (only the lower-list lookups of the top-k modes go back)
                    // the path is bound by dependent round trips, not by instructions: an owner's freqs are wanted for the
                    // freq-only bound of its first candidate, and decoding them now -- while the block's bytes are still
                    // in the staging window -- saves the reload a later decode would wait for
                    if (eager_freqs) cx.decode_freqs(x);
                }
                if (MODE == 0) {
                    cx.setm(x, M_CBW, __float_as_uint(wtab ? qw(x) * w : maxw(x)));
                    wave_sync();
                }
            }
            return true;
        };
        auto ensure_docs = [&](uint32_t x) __attribute__((always_inline)) {
            if constexpr (!BLOCKMAX) return;
            if (cx.m(x, M_DDEC)) return;
            typename decltype(cx)::BlockInfo bi;
            bi.ep = cx.m(x, M_EP);
            bi.next_ep = cx.m(x, M_NEXTEP);
            bi.bmax = cx.m(x, M_BMAX);
            bi.base = cx.m(x, M_BASE);
            cx.decode_docs(x, cx.m(x, M_CUR), &bi);
        };
        // the parts of a split query share a score histogram (ScoreHist). Parts may add a document's term scores in
        // different orders (which lists are essential depends on each part's threshold), hence the 1e-5 relaxation
        const bool shared_floor = MODE == 0 && !whole && a.q_hist;
        ScoreHist sh;
        sh.init(shared_floor ? a.q_hist : nullptr, shared_floor ? a.q_hist_slot[q] : 0u, shared_floor ? ubf(nt - 1) : 0.f, 1.0f - 1.0e-5f);
        ScoreHist::Snapshot hsnap = {0u, 0u, 0u, 0u};
        if (shared_floor) hsnap = sh.load();
        auto adopt_floor = [&]() __attribute__((always_inline)) { // the other parts of this query may have raised the bar
            const float f = sh.floor(hsnap, tk.k);
            hsnap = sh.load(); // resolved a round from now
            if (f > tk.floor) { tk.floor = f; update_non_ess(); }
        };
        uint32_t skip_x = 0xFFFFFFFFu, skip_hi2 = 0; // the list that moves on after a window that was passed over
        float skip_rest = 0.f;
        while (non_ess < nt && lo < N) {
            ++cx.s_rounds;
            // (measured: consulting the histogram every 4th window instead saves the trips and loses as much to the staler floor)
            if (shared_floor) { PT_BEGIN(cx); adopt_floor(); PT_END(cx, PH_FLOOR); }
            if (non_ess >= nt) break;
            // ---- window: every essential list is positioned on its block at lo; [lo, hi] ends with the first of them
#ifdef DS2I_PHASE_TIMING
            const unsigned long long pt_prolog0 = __builtin_readcyclecounter();
#endif
            uint32_t hi = N - 1, hi2 = N - 1, xmin = 0xFFFFFFFFu, live = 0; // live: bit x = essential list x has postings in or after the window
            for (uint32_t p = non_ess; p < nt; ++p) {
                const uint32_t x = slot_at(p);
                if (!position(x, lo, false, x == skip_x, skip_rest, skip_hi2, MODE == 0)) continue;
                live |= 1u << x;
                const uint32_t bm = cx.m(x, M_BMAX);
                if (bm < hi || xmin == 0xFFFFFFFFu) { hi2 = hi; hi = bm < hi ? bm : hi; xmin = x; }
                else if (bm < hi2) hi2 = bm;
                if (!BLOCKMAX && lane < 4) L.dupw[x][lane] = 0; // (block-max windows clear the owners' flags below)
            }
            skip_x = 0xFFFFFFFFu;
            if (!live) break;
            // ---- bounds of this window. wub[p] = the most the lists lord[0..p] can add to a document of the window: the
            // non-essential lists by their list maxima, the essential ones by the max weight of the block they are
            // positioned on (block-max maxscore). The first `ps` lists cannot lift a document into the heap on their own:
            // inside this window they are treated like non-essential lists (looked up for the candidates of the others,
            // decoded only if a candidate needs them); if that is all of them the window is passed over undecoded.
            uint32_t ps = non_ess;
            if constexpr (BLOCKMAX) {
                float acc = 0.f, rest = 0.f;
                ps = nt;
                for (uint32_t p = 0; p < nt; ++p) {
                    const uint32_t x = slot_at(p);
                    const float bnd = p < non_ess ? maxw(x) : (((live >> x) & 1u) ? cbw(x) : 0.f);
                    acc = p ? acc + bnd : bnd;
                    if (x != xmin) rest += bnd;
                    if (lane == 0) L.wub[p] = acc;
                    if (ps == nt && p >= non_ess && tk.would_enter(acc * BOUND_SLACK)) ps = p;
                }
                wave_sync();
                if (ps == nt) { // no document of [lo, hi] can enter: move the list that ends the window, skipping by weight
                    skip_x = xmin;
                    skip_rest = rest;
                    skip_hi2 = hi2;
#ifdef DS2I_PHASE_TIMING
                    cx.s_phase[PH_PROLOG] += __builtin_readcyclecounter() - pt_prolog0;
#endif
                    if (hi == 0xFFFFFFFFu) break;
                    lo = hi + 1;
                    continue;
                }
            }
            if constexpr (BLOCKMAX) {
                for (uint32_t p = ps; p < nt; ++p) {
                    const uint32_t x = slot_at(p);
                    if (!((live >> x) & 1u)) continue;
                    if (!cx.m(x, M_DDEC)) { // docs, then freqs while the block's bytes are in the staging window (see position())
                        ensure_docs(x);
                        cx.decode_freqs(x);
                    }
                    if (lane < 4) L.dupw[x][lane] = 0;
                }
            }
            wave_sync();
#ifdef DS2I_PHASE_TIMING
            cx.s_phase[PH_PROLOG] += __builtin_readcyclecounter() - pt_prolog0;
#endif
            if (MODE == 2) { // every freq of every list inside the window
                unsigned long long fs = 0;
                for (uint32_t p = 0; p < nt; ++p) {
                    const uint32_t x = slot_at(p);
                    if (!((live >> x) & 1u)) continue;
                    const uint32_t c0 = (Ldocs + 128u * x)[lane], c1 = (Ldocs + 128u * x)[lane + 64];
                    const bool i0 = c0 >= lo && c0 <= hi, i1 = c1 >= lo && c1 <= hi;
                    if (!(ballot(i0) | ballot(i1))) continue;
                    if (!cx.m(x, M_FDEC)) cx.decode_freqs(x);
                    fs += (unsigned long long)(i0 ? (Lfreqs + 128u * x)[lane] : 0u) + (i1 ? (Lfreqs + 128u * x)[lane + 64] : 0u);
                }
                for (int o = 32; o; o >>= 1) fs += __shfl_xor(fs, o);
                fsum += fs;
            }
            for (uint32_t p = ps; p < nt; ++p) { // ---- owner list e: its postings in [lo, hi] not owned earlier
                if (p < non_ess) continue;             // became non-essential during this window
                const uint32_t e = slot_at(p);
                if (!((live >> e) & 1u)) continue;
                const uint32_t c0 = (Ldocs + 128u * e)[lane], c1 = (Ldocs + 128u * e)[lane + 64];
                bool v0 = c0 >= lo && c0 <= hi && !((L.dupw[e][lane >> 5] >> (lane & 31u)) & 1u);
                bool v1 = c1 >= lo && c1 <= hi && !((L.dupw[e][2u + (lane >> 5)] >> (lane & 31u)) & 1u);
                if (!(ballot(v0) | ballot(v1))) continue;
                if (MODE != 0) { // or_query: count the owned candidates, mark their copies in the later lists
                    count += (unsigned long long)(__builtin_popcountll(ballot(v0)) + __builtin_popcountll(ballot(v1)));
                    for (uint32_t p2 = p + 1; p2 < nt; ++p2) {
                        const uint32_t x = slot_at(p2);
                        if (!((live >> x) & 1u)) continue;
                        uint32_t q0, q1;
                        if (member_bsearch((Ldocs + 128u * x), c0, v0, q0)) atomicOr(&L.dupw[x][q0 >> 5], 1u << (q0 & 31u));
                        if (member_bsearch((Ldocs + 128u * x), c1, v1, q1)) atomicOr(&L.dupw[x][q1 >> 5], 1u << (q1 & 31u));
                    }
                    wave_sync();
                    continue;
                }
                uint32_t fm0 = 0, fm1 = 0; // lists (beyond e) each candidate occurs in
                PosPack<TMAX> pp0, pp1;   // ... and where
                pp0.clear();
                pp1.clear();
                // lists below the first owner (non-essential, or unable to lift a document of this window): looked up last
                const uint32_t fo = ps > non_ess ? ps : non_ess;
                const float ub_low = fo ? ubw(fo - 1) : 0.f;
                // Range tables (BatchArgs::rmw): what the lower lists can add to THIS candidate, one byte gather per lower
                // list and candidate instead of their list maxima -- and a zero byte says the candidate is not in that list
                // at all, so it is never looked up there. lb*[j] = byte of the list at position j of the max-score order,
                // packed four to a dword.
                constexpr int NL = TMAX - 1, NLW = (NL + 3) / 4;
                const bool use_rmw = MODE == 0 && a.rmw && fo > 0;
                uint32_t lb0[NLW] = {}, lb1[NLW] = {};
                auto low_byte = [&](const uint32_t* lb, uint32_t j) __attribute__((always_inline)) -> uint32_t {
                    uint32_t w = lb[0];
#pragma unroll
                    for (int k = 1; k < NLW; ++k) w = (j >> 2) == (uint32_t)k ? lb[k] : w; // (selects, not a dynamic index: the words stay in registers)
                    return (w >> (8u * (j & 3u))) & 255u;
                };
                // sum of the candidate's bounds in the lower lists at positions <= upto (added from position 0 up)
                auto low_rest = [&](const uint32_t* lb, uint32_t upto) __attribute__((always_inline)) -> float {
                    float r = 0.f;
#pragma unroll
                    for (int j = 0; j < NL; ++j)
                        if ((uint32_t)j <= upto && (uint32_t)j < fo) r = r + __uint_as_float(cx.m(slot_at((uint32_t)j), M_RSCALE)) * (float)low_byte(lb, (uint32_t)j);
                    return r;
                };
                if (use_rmw) {
                    uint32_t e0[NL] = {}, e1[NL] = {};
#pragma unroll
                    for (int j = 0; j < NL; ++j) { // (all gathers are issued before the first is consumed)
                        if ((uint32_t)j < fo) {
                            const uint32_t x = slot_at((uint32_t)j);
                            const uint8_t* tab = a.rmw + 64ull * cx.m(x, M_RBASE);
                            const uint32_t sh = cx.m(x, M_RSHIFT);
                            e0[j] = v0 ? (uint32_t)tab[c0 >> sh] : 0u;
                            e1[j] = v1 ? (uint32_t)tab[c1 >> sh] : 0u;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        lb0[j >> 2] |= e0[j] << (8 * (j & 3));
                        lb1[j >> 2] |= e1[j] << (8 * (j & 3));
                    }
                }
                // The owner's term score is bounded by its freq alone (doc_term_weight falls with norm_len, so the
                // collection's shortest document bounds it) and by its block's max weight: most postings of a list have
                // small freqs and fall below the threshold here -- before their norm_len is gathered or another list
                // is probed for them.
                if (!cx.m(e, M_FDEC)) cx.decode_freqs(e);
                float pb0, pb1;
                {
                    const float we = qw(e), me = cbw(e);
                    const float fb0 = we * doc_term_weight((Lfreqs + 128u * e)[lane], a.min_norm_len);
                    const float fb1 = we * doc_term_weight((Lfreqs + 128u * e)[lane + 64], a.min_norm_len);
                    pb0 = (fb0 < me ? fb0 : me) + (use_rmw ? low_rest(lb0, fo - 1) : ub_low);
                    pb1 = (fb1 < me ? fb1 : me) + (use_rmw ? low_rest(lb1, fo - 1) : ub_low);
                }
                for (uint32_t p2 = p + 1; p2 < nt; ++p2) {
                    const uint32_t x = slot_at(p2);
                    if (!((live >> x) & 1u)) continue;
                    PT_BEGIN(cx);
                    uint32_t q0, q1;
                    const bool f0 = member_bsearch((Ldocs + 128u * x), c0, v0, q0);
                    const bool f1 = member_bsearch((Ldocs + 128u * x), c1, v1, q1);
                    const float mx = cbw(x);
                    if (f0) { pp0.set(x, q0); atomicOr(&L.dupw[x][q0 >> 5], 1u << (q0 & 31u)); fm0 |= 1u << x; pb0 += mx; }
                    if (f1) { pp1.set(x, q1); atomicOr(&L.dupw[x][q1 >> 5], 1u << (q1 & 31u)); fm1 |= 1u << x; pb1 += mx; }
                    PT_END(cx, PH_MEMBER);
                }
                wave_sync();
                // max-score bound first: nothing below it is gathered, scored or looked up in the lower lists
                bool s0 = v0 && tk.would_enter(pb0 * BOUND_SLACK), s1 = v1 && tk.would_enter(pb1 * BOUND_SLACK);
                uint64_t b0 = ballot(s0), b1 = ballot(s1);
                if (!(b0 | b1)) continue;
                const uint32_t ns = (uint32_t)(__builtin_popcountll(b0) + __builtin_popcountll(b1));
                cx.s_bytes += 4ull * ns;
                cx.s_scored += ns;
                unsigned long long a0 = 0, a1 = 0; // fixed-point sums (FxScale); sc0 / sc1 are their float values
                float sc0 = 0.f, sc1 = 0.f, nl0, nl1;
                {
                    PT_BEGIN(cx);
                    nl0 = s0 ? a.norm_lens[c0] : 0.f;
                    nl1 = s1 ? a.norm_lens[c1] : 0.f;
                    const float w = qw(e);
                    if (s0) a0 = fx.of(w * doc_term_weight((Lfreqs + 128u * e)[lane], nl0));
                    if (s1) a1 = fx.of(w * doc_term_weight((Lfreqs + 128u * e)[lane + 64], nl1));
                    PT_END(cx, PH_SCORE);
                }
                for (uint32_t p2 = p + 1; p2 < nt; ++p2) {
                    const uint32_t x = slot_at(p2);
                    const bool h0 = s0 && ((fm0 >> x) & 1u), h1 = s1 && ((fm1 >> x) & 1u);
                    if (!(ballot(h0) | ballot(h1))) continue;
                    if (!cx.m(x, M_FDEC)) cx.decode_freqs(x);
                    const float w = qw(x);
                    if (h0) a0 += fx.of(w * doc_term_weight((Lfreqs + 128u * x)[pp0.get(x)], nl0));
                    if (h1) a1 += fx.of(w * doc_term_weight((Lfreqs + 128u * x)[pp1.get(x)], nl1));
                }
                // the lower lists, highest bound first; a candidate stops as soon as it cannot enter (queries.hpp:553-564).
                // A list is first only positioned: its block is decoded if some candidate could still enter with the
                // block's best weight on top of its score.
#ifdef DS2I_PHASE_TIMING
                const unsigned long long pt_probe0 = __builtin_readcyclecounter();
#endif
                for (uint32_t p2 = fo; p2-- > 0;) {
                    const float ubp = ubw(p2), lowb = p2 ? ubw(p2 - 1) : 0.f;
                    sc0 = fx.value(a0);
                    sc1 = fx.value(a1);
                    float ubp0 = ubp, ubp1 = ubp, lowb0 = lowb, lowb1 = lowb, own0 = __builtin_inff(), own1 = own0;
                    if (use_rmw) { // per candidate: its own bounds in the lists at positions <= p2, < p2, and in list p2 itself
                        ubp0 = low_rest(lb0, p2);
                        ubp1 = low_rest(lb1, p2);
                        lowb0 = p2 ? low_rest(lb0, p2 - 1) : 0.f;
                        lowb1 = p2 ? low_rest(lb1, p2 - 1) : 0.f;
                        const float sc = __uint_as_float(cx.m(slot_at(p2), M_RSCALE));
                        own0 = sc * (float)low_byte(lb0, p2);
                        own1 = sc * (float)low_byte(lb1, p2);
                    }
                    s0 = s0 && tk.would_enter((sc0 + ubp0) * BOUND_SLACK);
                    s1 = s1 && tk.would_enter((sc1 + ubp1) * BOUND_SLACK);
                    bool r0 = s0, r1 = s1; // still to be looked up in list x
                    if (use_rmw) { // (a zero byte: no posting of the list in the candidate's doc-id range)
                        r0 = r0 && low_byte(lb0, p2) != 0u;
                        r1 = r1 && low_byte(lb1, p2) != 0u;
                    }
                    if (!(ballot(s0) | ballot(s1))) break;
                    if (!(ballot(r0) | ballot(r1))) continue;
                    const uint32_t x = slot_at(p2);
                    for (;;) {
                        const uint64_t rb0 = ballot(r0), rb1 = ballot(r1);
                        if (!(rb0 | rb1)) break;
                        const uint32_t amin = rb0 ? bcast(c0, (uint32_t)__builtin_ctzll(rb0)) : bcast(c1, (uint32_t)__builtin_ctzll(rb1));
                        if (!position(x, amin, true, false, 0.f, 0u, false)) break; // nothing >= amin in list x
                        const uint32_t bm = cx.m(x, M_BMAX);
                        const bool w0 = r0 && c0 <= bm, w1 = r1 && c1 <= bm;
                        bool t0 = w0, t1 = w1;
                        if constexpr (BLOCKMAX) {
                            const float cbx = cbw(x);
                            t0 = w0 && tk.would_enter((sc0 + ((own0 < cbx ? own0 : cbx) + lowb0)) * BOUND_SLACK);
                            t1 = w1 && tk.would_enter((sc1 + ((own1 < cbx ? own1 : cbx) + lowb1)) * BOUND_SLACK);
                        }
                        if (ballot(t0) | ballot(t1)) {
                            ensure_docs(x);
                            uint32_t q0, q1;
                            const bool f0 = member_bsearch((Ldocs + 128u * x), c0, t0, q0);
                            const bool f1 = member_bsearch((Ldocs + 128u * x), c1, t1, q1);
                            if (ballot(f0) | ballot(f1)) {
                                if (!cx.m(x, M_FDEC)) cx.decode_freqs(x);
                                const float w = qw(x);
                                if (f0) a0 += fx.of(w * doc_term_weight((Lfreqs + 128u * x)[q0], nl0));
                                if (f1) a1 += fx.of(w * doc_term_weight((Lfreqs + 128u * x)[q1], nl1));
                            }
                        }
                        s0 = s0 && !(w0 && !t0); // cannot enter even with this block's best posting
                        s1 = s1 && !(w1 && !t1);
                        r0 = r0 && !w0;
                        r1 = r1 && !w1;
                    }
                }
#ifdef DS2I_PHASE_TIMING
                cx.s_phase[PH_PROBE] += __builtin_readcyclecounter() - pt_probe0;
                const unsigned long long pt_ins0 = __builtin_readcyclecounter();
#endif
                sc0 = fx.value(a0);
                sc1 = fx.value(a1);
                bool inserted = false;
                for (int half = 0; half < 2; ++half) {
                    const bool al = half ? s1 : s0;
                    const float sc = half ? sc1 : sc0;
                    uint64_t todo = ballot(al && tk.would_enter(sc));
                    while (todo) {
                        const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                        todo &= todo - 1;
                        const float v = __uint_as_float(bcast(__float_as_uint(sc), src));
                        if (tk.insert(v)) {
                            inserted = true;
                            if (shared_floor && lane == 0) sh.add(v);
                        }
                    }
                }
                if (inserted) update_non_ess(); // queries.hpp:568-574
#ifdef DS2I_PHASE_TIMING
                cx.s_phase[PH_INSERT] += __builtin_readcyclecounter() - pt_ins0;
#endif
            }
            if (hi == 0xFFFFFFFFu) break;
            lo = hi + 1;
        }
        if (whole) {
            if (lane == 0) { a.out_count[q] = MODE == 0 ? tk.n : count; if (a.out_freq_sum) a.out_freq_sum[q] = fsum; }
            if (MODE == 0) store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
        } else {
            if (lane == 0) { a.unit_count[uid] = MODE == 0 ? tk.n : count; a.unit_freq_sum[uid] = fsum; }
            if (MODE == 0) store_topk(a.unit_topk, a.unit_topk_len, a.k, uid, tk);
        }
        if (STATS && a.unit_clock && lane == 0) { a.unit_clock[2ull * uid] = t_unit; a.unit_clock[2ull * uid + 1] = wall_clock64(); }
#ifdef DS2I_PHASE_TIMING
        cx.s_phase[PH_TOTAL] += __builtin_readcyclecounter() - pt_unit0;
#endif
    }
    cx.flush_stats(a.stats);
}

// ------------------------------------------------------------------ top-k of the union as streams (wand / maxscore / ranked_or)
// The three operators return the k best scores of the union of the query's lists (queries.hpp:200-319, 404-476, 478-591);
// what differs in the reference is only how they avoid scoring everything. Here: the lists of a query are ordered by
// decreasing max score, and a document BELONGS to the first list of that order that contains it. A unit streams a block
// range of one list e (its "driver") exactly like ranked_and streams its shortest list -- table window in registers,
// next block prefetched -- and for every posting of the block gathers its range-table byte in every other list:
//   * the lists AFTER e (lower max score) are optional: their bytes bound what they can add, a zero byte says the document
//     is not in that list, so only the few candidates whose bound can enter the heap are ever looked up there;
//   * the lists BEFORE e are exclusions: a candidate found in one of them belongs to that list's units and is dropped
//     (a zero byte settles that without a lookup).
// A document owned by list e occurs in no list of higher max score, so it scores at most S_e = the sum of the max scores
// from e down: once the threshold passes S_e the units of list e -- and of every later list -- end at once. That is
// MaxScore's essential / non-essential split (queries.hpp:529-574), evaluated per unit; no list is walked in lock step
// with another, no window is cut at block boundaries, and every unit is an independent stream for the dispatcher.
// Every document's score is computed by exactly one unit, as the float32 sum of its term scores in the fixed order
// driver, optional lists by decreasing max score: wand == maxscore == ranked_or bit for bit, run after run.
#ifndef DS2I_UT_WAVES2
#define DS2I_UT_WAVES2 6
#endif
constexpr int UT_WAVES(int tmax) { return tmax <= 2 ? DS2I_UT_WAVES2 : tmax <= 4 ? 5 : tmax <= 8 ? 3 : 1; }
template <int TMAX, bool META_IN_LDS, bool WITH_S16>
struct LdsUnionTopk : Lds<TMAX, META_IN_LDS, false, WITH_S16, (TMAX > 2) ? 2 : TMAX> {};

template <int TMAX, int CODEC_T, bool STATS = true>
__global__ void __launch_bounds__(64, UT_WAVES(TMAX)) k_union_topk(BatchArgs a) {
    constexpr bool REG = TMAX <= 4;
    typedef typename std::conditional<REG, MetaReg<TMAX>, MetaLds>::type META;
    __shared__ LdsUnionTopk<TMAX, !REG, CODEC_T != CODEC_PEF && CODEC_T != CODEC_OPTPFOR> L;
    const uint32_t lane = lane_id();
    constexpr bool SHARE_F = TMAX > 2; // one freqs buffer for the driver, one shared by the lists that are looked up
    CtxT<CODEC_T, META, STATS, SHARE_F> cx = make_ctx<CODEC_T, META, STATS, SHARE_F>(L, a);
    cx.want_freqs = cx.side(); // (the driver's freqs with its doc-ids: every posting gets its own bound before any gather)
    const float* const bmw = a.bmw;
    const uint8_t* const rmw = a.rmw;
    constexpr int NW = (TMAX + 3) / 4; // a candidate's bytes, four lists to a dword (byte i = list slot i; slot 0 unused)
    for (uint32_t tkt = blockIdx.x; tkt < a.nslice; tkt += gridDim.x) {
        const uint32_t uid = a.order[tkt];
        const unsigned long long t_unit = (STATS && a.unit_clock) ? wall_clock64() : 0ull;
        const Unit u = a.units[uid];
        const uint32_t vq = u.q;
        const uint32_t q = uniform(a.vq_info[3u * vq]), nexcl = uniform(a.vq_info[3u * vq + 1u]);
        const float s_all = __uint_as_float(uniform(a.vq_info[3u * vq + 2u]));
        const bool whole = u.nparts == 1;
        const uint32_t t0 = a.q_off[vq], nt = a.q_off[vq + 1] - t0;
        TopK tk;
        tk.init(a.k);
        auto finish_unit = [&]() __attribute__((always_inline)) {
            if (whole) {
                if (lane == 0) a.out_count[q] = tk.n;
                store_topk(a.out_topk, a.out_topk_len, a.k, q, tk);
            } else {
                if (lane == 0) { a.unit_count[uid] = tk.n; a.unit_freq_sum[uid] = 0; }
                store_topk(a.unit_topk, a.unit_topk_len, a.k, uid, tk);
            }
            if (STATS && a.unit_clock && lane == 0) { a.unit_clock[2ull * uid] = t_unit; a.unit_clock[2ull * uid + 1] = wall_clock64(); }
        };
        if (nt == 0 || nt > (uint32_t)TMAX) { finish_unit(); continue; }
        auto bind_one = [&](auto ic) __attribute__((always_inline)) { const uint32_t i = ic; cx.bind(i, a.qterms[t0 + i]); return true; };
        DS2I_LIST_LOOP(0, bind_one)
        cx.s_bytes += 4ull * nt;
        // ---- floors: all lower bounds of the final k-th score of the union
        if (a.seed_topk && a.seed_len[q] >= a.k) { // the ranked_and pass over (a sub-query of) the same query, relaxed for re-association
            const float kth = a.seed_topk[(size_t)q * a.k + a.k - 1];
            tk.floor = __uint_as_float(uniform(__float_as_uint(kth * (1.0f - 1.0e-5f))));
        }
        {   // some term has k blocks whose best posting alone reaches floor1 (host, from the upload-time block weights)
            const float f1 = __uint_as_float(uniform(__float_as_uint(a.qterms[t0].floor1))) * (1.0f - 1.0e-5f);
            if (f1 > tk.floor) tk.floor = f1;
        }
        const bool shared_floor = !whole && a.q_hist;
        ScoreHist sh;
        sh.init(shared_floor ? a.q_hist : nullptr, shared_floor ? a.q_hist_slot[q] : 0u, shared_floor ? s_all : 0.f, 1.0f - 1.0f / 1048576.0f);
        auto adopt_floor = [&]() __attribute__((always_inline)) {
            const float f = sh.floor(tk.k);
            if (f > tk.floor) tk.floor = f;
        };
        if (shared_floor) adopt_floor();
        const float qw0 = __uint_as_float(cx.m(0, M_QW));
        const bool two_trips = TMAX > 2 && a.ut_first && nexcl + a.ut_first + 1u < nt; // (optional lists are left for a second trip)
        // S_e: a document owned by the driver is in no list of higher max score
        const float s_e = __uint_as_float(uniform(__float_as_uint(a.qterms[t0].max_bmw + a.qterms[t0].suf_bmw)));
        if (!tk.would_enter(s_e * BOUND_SLACK)) { finish_unit(); continue; }
        const float* const w0tab = bmw + cx.m(0, M_PBASE);
        // a candidate's bytes: byte (i & 3) of word (i >> 2) = its range-table entry in list slot i
        auto byte_of = [&](const uint32_t* pk, uint32_t i) __attribute__((always_inline)) -> uint32_t {
            uint32_t w = pk[0];
#pragma unroll
            for (int k2 = 1; k2 < NW; ++k2) w = (i >> 2) == (uint32_t)k2 ? pk[k2] : w;
            return (w >> (8u * (i & 3u))) & 255u;
        };
        // what the optional lists after slot `after` can add to the candidate (slots nexcl+1 .. nt-1 are the optional ones)
        auto rest_of = [&](const uint32_t* pk, uint32_t after) __attribute__((always_inline)) -> float {
            float r = 0.f;
            if constexpr (REG) { // (register-resident list state: the slot must be a compile-time constant)
                auto add_one = [&](auto jc) __attribute__((always_inline)) {
                    constexpr uint32_t j = decltype(jc)::value;
                    if (j < nt && j > nexcl && j > after) r = r + __uint_as_float(cx.m(j, M_RSCALE)) * (float)byte_of(pk, j);
                };
                static_loop_down<TMAX, 1>(add_one);
            } else {
                for (uint32_t j = nt; j-- > 1;) {
                    if (j <= nexcl || j <= after) break;
                    r = r + __uint_as_float(cx.m(j, M_RSCALE)) * (float)byte_of(pk, j);
                }
            }
            return r;
        };
        // what the optional lists can add to any document at most (their list maxima, summed like rest_of sums their bytes)
        uint32_t pk_ff[NW];
#pragma unroll
        for (int k2 = 0; k2 < NW; ++k2) pk_ff[k2] = 0xFFFFFFFFu;
        const float opt_all = rest_of(pk_ff, 0);
        // ---- the driver as a stream (see k_conjunctive): table window in registers, next block requested ahead
        const bool pstream = cx.is_pef();
        const uint2* const tab0 = pstream ? nullptr : cx.skip + cx.m(0, M_PBASE);
        const uint32_t* const cmax0 = pstream ? (const uint32_t*)cx.ptr(0, M_MAXS_LO) : nullptr;
        const uint32_t* const ent0 = pstream ? (const uint32_t*)cx.ptr(0, M_END_LO) : nullptr;
        const uint8_t* data0 = nullptr;
        if (!pstream) {
            const uint32_t nb0 = cx.m(0, M_NB);
            data0 = cx.ptr(0, M_MAXS_LO) + 4ull * nb0 + 4ull * (nb0 - 1);
        }
        uint32_t s_first = 0, pf_blk = 0xFFFFFFFFu, pf_d0 = 0, pf_d1 = 0, pf_x = 0;
        uint2 s_e2 = make_uint2(0xFFFFFFFFu, 0u);
        float s_w = 0.f, s_rb = 0.f;
        auto s_fill = [&](uint32_t first) __attribute__((always_inline)) {
            s_first = first;
            const uint32_t idx = first + lane;
            s_e2 = make_uint2(0xFFFFFFFFu, 0u);
            s_w = 0.f;
            if (idx < u.blk_end) {
                if (pstream) s_e2.x = cmax0[idx]; else s_e2 = tab0[idx];
                s_w = w0tab[idx];
            }
            // the optional lists' largest entry over the block's own doc-id span (<= 16 bytes of the level that is wide enough)
            const uint32_t prev_max = (uint32_t)__shfl_up((int)s_e2.x, 1);
            const uint32_t base = (lane == 0) ? 0u : prev_max + 1u, top = s_e2.x;
            const bool row = idx < u.blk_end && (lane > 0 || idx == 0) && top != 0xFFFFFFFFu && base <= top;
            float acc = 0.f;
            auto span_max = [&](uint32_t sh, uint32_t rbase, float scale) __attribute__((always_inline)) {
                const RmwLevels g(a.num_docs, sh);
                const uint8_t* tb = rmw + 64ull * rbase;
                uint32_t best = 255u;
                {   // branch-free (see k_conjunctive): a lane without a row reads entry 0 and discards it
                    const uint32_t b2 = row ? base : 0u, t2 = row ? top : 0u;
                    uint32_t lsh = sh, lvl = 0;
                    while (lvl < 2 && (t2 >> lsh) - (b2 >> lsh) >= 16u) { lsh += 6; ++lvl; }
                    const uint32_t lo2 = b2 >> lsh, hi2 = t2 >> lsh;
                    const bool fits = hi2 - lo2 < 16u;
                    const uint32_t m = max_of_bytes16(tb + g.off[lvl] + (fits ? lo2 : 0u), fits ? hi2 - lo2 + 1u : 1u);
                    if (row && fits) best = m;
                }
                acc = acc + scale * (float)best;
            };
            if constexpr (REG) {
                auto one = [&](auto jc) __attribute__((always_inline)) {
                    constexpr uint32_t j = decltype(jc)::value;
                    if (j < nt && j > nexcl) span_max(cx.m(j, M_RSHIFT), cx.m(j, M_RBASE), __uint_as_float(cx.m(j, M_RSCALE)));
                };
                static_loop_down<TMAX, 1>(one);
            } else {
                for (uint32_t j = nt; j-- > 1;) {
                    if (j <= nexcl) break;
                    span_max(cx.m(j, M_RSHIFT), cx.m(j, M_RBASE), __uint_as_float(cx.m(j, M_RSCALE)));
                }
            }
            s_rb = acc;
        };
        auto s_live = [&](uint32_t from) __attribute__((always_inline)) -> uint64_t {
            const uint32_t idx = s_first + lane;
            bool ok = idx >= from && idx < u.blk_end && (lane > 0 || idx == 0);
            ok = ok && tk.would_enter((qw0 * s_w + s_rb) * BOUND_SLACK);
            return ballot(ok);
        };
        auto s_next = [&](uint32_t from) __attribute__((always_inline)) -> uint32_t {
            for (;;) {
                if (from >= u.blk_end) return u.blk_end;
                const uint64_t hit = s_live(from);
                if (hit) return s_first + (uint32_t)__builtin_ctzll(hit);
                if (s_first + 64 >= u.blk_end) return u.blk_end;
                s_fill(s_first + 63);
                from = from > s_first + 1 ? from : s_first + 1;
            }
        };
        s_fill(u.blk_begin ? u.blk_begin - 1 : 0);
        uint32_t from = u.blk_begin, floor_tick = 1;
        for (;;) {
            ++cx.s_rounds;
            if (from >= u.blk_end) break;
            if (shared_floor && (floor_tick++ & (DS2I_FLOOR_EVERY - 1)) == 0) {
                adopt_floor();
                if (!tk.would_enter(s_e * BOUND_SLACK)) break; // the driver became non-essential
            }
            const uint32_t blk = s_next(from);
            if (blk >= u.blk_end) break;
            from = blk + 1;
            const uint32_t f = blk - s_first, fp = f ? f - 1 : 0;
            const float wblk = qw0 * __uint_as_float(bcast(__float_as_uint(s_w), f));
            const bool staged = pf_blk == blk;
            if (pstream) {
                cx.decode_docs_pef(0, blk, staged ? &pf_d0 : nullptr);
            } else {
                typename decltype(cx)::BlockInfo bi;
                bi.bmax = bcast(s_e2.x, f);
                bi.next_ep = bcast(s_e2.y, f);
                bi.base = blk ? bcast(s_e2.x, fp) + 1u : 0u;
                bi.ep = blk ? bcast(s_e2.y, fp) : 0u;
                if (staged) {
                    const uint8_t* p = data0 + bi.ep;
                    cx.win.gbase = (const uint8_t*)((uintptr_t)p & ~(uintptr_t)3);
                    cx.win.nbytes = 512;
                    cx.win.st[lane] = pf_d0;
                    cx.win.st[lane + 64] = pf_d1;
                    if (cx.side()) { cx.exc[lane] = pf_x; cx.slot_blk = cx.m(0, M_PBASE) + blk; } // (its side slot came with them)
                    wave_sync();
                }
                cx.decode_docs(0, blk, &bi, staged);
            }
            {   // request what the block that is next as things stand will need first
                const uint64_t nx = s_live(blk + 1);
                pf_blk = 0xFFFFFFFFu;
                if (nx) {
                    const uint32_t fn = (uint32_t)__builtin_ctzll(nx);
                    if (pstream) {
                        const uint32_t nb2 = s_first + fn, cm = bcast(s_e2.x, fn);
                        pf_d0 = lane == PC_WORDS ? cm : 0u;
                        if (lane < PC_WORDS) pf_d0 = ent0[(size_t)nb2 * PC_WORDS + lane];
                    } else {
                        const uint32_t* g = (const uint32_t*)((uintptr_t)(data0 + bcast(s_e2.y, fn - 1)) & ~(uintptr_t)3);
                        pf_d0 = g[lane];
                        pf_d1 = g[lane + 64];
                        if (cx.side()) pf_x = cx.xslots[(size_t)XSLOT_DW * (cx.m(0, M_PBASE) + s_first + fn) + lane];
                    }
                    pf_blk = s_first + fn;
                }
            }
            const uint32_t c0 = L.docs[0][lane], c1 = L.docs[0][lane + 64];
            bool al0 = c0 != 0xFFFFFFFFu, al1 = c1 != 0xFFFFFFFFu;
            // block_optpfor through the side slots: the block's freqs came with its doc-ids, so every posting has a bound of its
            // OWN term score (doc_term_weight falls with norm_len: the collection's shortest document bounds it from the freq
            // alone). Only the postings that could enter the heap with that bound + the optional lists' maxima ask the other
            // lists' tables at all -- a gather is a cache line per posting and list, and this operator's were 56 GB per batch --
            // and the tests below use the posting's own bound where they used the block's weight.
            float wb0 = wblk, wb1 = wblk;
            if (cx.side() && cx.m(0, M_FDEC)) {
                const float o0 = qw0 * doc_term_weight(L.freqs[0][lane], a.min_norm_len), o1 = qw0 * doc_term_weight(L.freqs[0][lane + 64], a.min_norm_len);
                wb0 = o0 < wblk ? o0 : wblk;
                wb1 = o1 < wblk ? o1 : wblk;
                al0 = al0 && tk.would_enter((wb0 + opt_all) * BOUND_SLACK);
                al1 = al1 && tk.would_enter((wb1 + opt_all) * BOUND_SLACK);
                if (!(ballot(al0) | ballot(al1))) continue;
            }
            // ---- the candidates' bytes in every other list: one gather per list, all issued before the first is consumed
            uint32_t pk0[NW] = {}, pk1[NW] = {};
            {
                // Two trips when the driver has two or more optional lists: first the exclusion lists and the optional list
                // of highest max score (slots 1 .. split), then -- only for the candidates that can still enter the heap with
                // that list's byte and the LIST maxima of the ones after it -- the rest. The later lists are the long,
                // low-scoring ones: their tables are the ones in which every candidate hits a line of its own.
                uint32_t e0[TMAX] = {}, e1[TMAX] = {};
                const uint32_t split = two_trips ? nexcl + a.ut_first : (uint32_t)TMAX;
                float f0 = 0.f, f1 = 0.f, suf_split = 0.f;
                auto load_first = [&](auto ic) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(ic)::value;
                    if (i > split) return true;
                    const uint8_t* tab = rmw + 64ull * cx.m(i, M_RBASE);
                    const uint32_t sh = cx.m(i, M_RSHIFT);
                    e0[i] = al0 ? (uint32_t)tab[c0 >> sh] : 0u;
                    e1[i] = al1 ? (uint32_t)tab[c1 >> sh] : 0u;
                    if (i > nexcl) { // an optional list of the first trip: what its byte says it can add
                        const float sc = __uint_as_float(cx.m(i, M_RSCALE));
                        f0 = f0 + sc * (float)e0[i];
                        f1 = f1 + sc * (float)e1[i];
                        if (i == split) suf_split = __uint_as_float(cx.m(i, M_SUF)); // + the list maxima of the ones after it
                    }
                    return true;
                };
                static_list_loop<1, TMAX>(nt, load_first);
                if (split + 1u < nt) {
                    al0 = al0 && tk.would_enter((wb0 + f0 + suf_split) * BOUND_SLACK);
                    al1 = al1 && tk.would_enter((wb1 + f1 + suf_split) * BOUND_SLACK);
                    auto load_rest = [&](auto ic) __attribute__((always_inline)) {
                        constexpr uint32_t i = decltype(ic)::value;
                        if (i <= split) return true;
                        const uint8_t* tab = rmw + 64ull * cx.m(i, M_RBASE);
                        const uint32_t sh = cx.m(i, M_RSHIFT);
                        e0[i] = al0 ? (uint32_t)tab[c0 >> sh] : 0u;
                        e1[i] = al1 ? (uint32_t)tab[c1 >> sh] : 0u;
                        return true;
                    };
                    static_list_loop<1, TMAX>(nt, load_rest);
                }
                auto pack_one = [&](auto ic) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(ic)::value;
                    pk0[i >> 2] |= e0[i] << (8u * (i & 3u));
                    pk1[i >> 2] |= e1[i] << (8u * (i & 3u));
                    return true;
                };
                static_list_loop<1, TMAX>(nt, pack_one);
            }
            float r0 = rest_of(pk0, 0), r1 = rest_of(pk1, 0);
            al0 = al0 && tk.would_enter((wb0 + r0) * BOUND_SLACK);
            al1 = al1 && tk.would_enter((wb1 + r1) * BOUND_SLACK);
            if (!(ballot(al0) | ballot(al1))) continue; // nobody of this block can enter: its freqs stay undecoded
            if (a.rmh) {
                // ---- membership hints (BatchArgs::rmh; block_optpfor indexes): a non-zero byte says SOME posting of list i lies in
                // the candidate's range; where the range holds exactly one posting the hint says at which offset. A candidate
                // elsewhere in that range is not in list i: its byte is cleared -- no lookup there (the most expensive thing this
                // kernel does: a block search and a block decode per list), nothing added to its bound by that list, and for an
                // exclusion list the verdict "not theirs". One more byte per candidate and list, for the survivors of the test above.
                const long long hd = (long long)(a.rmh - a.rmw);
                auto clear_byte = [&](uint32_t* pk, uint32_t i) __attribute__((always_inline)) {
#pragma unroll
                    for (int k2 = 0; k2 < NW; ++k2)
                        if ((i >> 2) == (uint32_t)k2) pk[k2] &= ~(255u << (8u * (i & 3u)));
                };
                auto hint_chunk = [&](auto i0c) __attribute__((always_inline)) { // lists i0 .. i0+3: their loads first, then the tests
                    uint32_t hv0[4], hv1[4];
                    auto ld = [&](auto kc) __attribute__((always_inline)) {
                        constexpr uint32_t k2 = decltype(kc)::value;
                        const uint32_t i = (uint32_t)i0c + k2;
                        hv0[k2] = hv1[k2] = 255u;
                        if (i < nt) {
                            const uint8_t* ht = rmw + 64ull * cx.m(i, M_RBASE) + hd;
                            const uint32_t sh = cx.m(i, M_RSHIFT);
                            if (al0 && sh != 0u && byte_of(pk0, i) != 0u) hv0[k2] = (uint32_t)ht[c0 >> sh]; // (one doc-id per entry: nothing to add)
                            if (al1 && sh != 0u && byte_of(pk1, i) != 0u) hv1[k2] = (uint32_t)ht[c1 >> sh];
                        }
                    };
                    auto ts = [&](auto kc) __attribute__((always_inline)) {
                        constexpr uint32_t k2 = decltype(kc)::value;
                        const uint32_t i = (uint32_t)i0c + k2;
                        if (i < nt) {
                            const uint32_t sh = cx.m(i, M_RSHIFT);
                            if (!((hv0[k2] == 255u) | (hv0[k2] == rmh_code(c0, sh)))) clear_byte(pk0, i);
                            if (!((hv1[k2] == 255u) | (hv1[k2] == rmh_code(c1, sh)))) clear_byte(pk1, i);
                        }
                    };
                    if constexpr (REG) { // (register-resident list state: slots must be compile-time constants)
                        constexpr uint32_t I0 = decltype(i0c)::value;
                        auto ld_c = [&](auto kc) __attribute__((always_inline)) { ld(kc); return true; };
                        (void)ld_c;
                        auto one_ld = [&](auto ic) __attribute__((always_inline)) {
                            constexpr uint32_t i = decltype(ic)::value;
                            const uint8_t* ht = rmw + 64ull * cx.m(i, M_RBASE) + hd;
                            const uint32_t sh = cx.m(i, M_RSHIFT);
                            hv0[i - I0] = (al0 && sh != 0u && byte_of(pk0, i) != 0u) ? (uint32_t)ht[c0 >> sh] : 255u;
                            hv1[i - I0] = (al1 && sh != 0u && byte_of(pk1, i) != 0u) ? (uint32_t)ht[c1 >> sh] : 255u;
                            return true;
                        };
                        static_list_loop<I0, (I0 + 4 < TMAX ? I0 + 4 : TMAX)>(nt, one_ld);
                        auto one_ts = [&](auto ic) __attribute__((always_inline)) {
                            constexpr uint32_t i = decltype(ic)::value;
                            const uint32_t sh = cx.m(i, M_RSHIFT);
                            if (!((hv0[i - I0] == 255u) | (hv0[i - I0] == rmh_code(c0, sh)))) clear_byte(pk0, i);
                            if (!((hv1[i - I0] == 255u) | (hv1[i - I0] == rmh_code(c1, sh)))) clear_byte(pk1, i);
                            return true;
                        };
                        static_list_loop<I0, (I0 + 4 < TMAX ? I0 + 4 : TMAX)>(nt, one_ts);
                    } else {
                        ld(std::integral_constant<uint32_t, 0>{}); ld(std::integral_constant<uint32_t, 1>{});
                        ld(std::integral_constant<uint32_t, 2>{}); ld(std::integral_constant<uint32_t, 3>{});
                        ts(std::integral_constant<uint32_t, 0>{}); ts(std::integral_constant<uint32_t, 1>{});
                        ts(std::integral_constant<uint32_t, 2>{}); ts(std::integral_constant<uint32_t, 3>{});
                    }
                };
                if constexpr (REG) {
                    hint_chunk(std::integral_constant<uint32_t, 1>{});
                } else {
                    for (uint32_t i0 = 1; i0 < nt; i0 += 4) hint_chunk(i0);
                }
                r0 = rest_of(pk0, 0);
                r1 = rest_of(pk1, 0);
                al0 = al0 && tk.would_enter((wb0 + r0) * BOUND_SLACK);
                al1 = al1 && tk.would_enter((wb1 + r1) * BOUND_SLACK);
                if (!(ballot(al0) | ballot(al1))) continue;
            }
            // ---- the driver's own term score: freq-only bound first, then the norm_len gather
            if (!cx.m(0, M_FDEC)) cx.decode_freqs(0);
            const uint32_t f0 = L.freqs[0][lane], f1 = L.freqs[0][lane + 64];
            al0 = al0 && tk.would_enter((qw0 * doc_term_weight(f0, a.min_norm_len) + r0) * BOUND_SLACK);
            al1 = al1 && tk.would_enter((qw0 * doc_term_weight(f1, a.min_norm_len) + r1) * BOUND_SLACK);
            const float nl0 = al0 ? a.norm_lens[c0] : 0.f, nl1 = al1 ? a.norm_lens[c1] : 0.f;
            float pa0 = al0 ? qw0 * doc_term_weight(f0, nl0) : 0.f, pa1 = al1 ? qw0 * doc_term_weight(f1, nl1) : 0.f;
            {
                const uint32_t nv = (uint32_t)(__builtin_popcountll(ballot(al0)) + __builtin_popcountll(ballot(al1)));
                cx.s_bytes += 4ull * nv;
                cx.s_scored += nv;
            }
            al0 = al0 && tk.would_enter((pa0 + r0) * BOUND_SLACK);
            al1 = al1 && tk.would_enter((pa1 + r1) * BOUND_SLACK);
            // ---- the other lists, one after the other: exclusions first (slots 1 .. nexcl), then the optional lists by
            // decreasing max score; a list is consulted only for the candidates whose byte there is not zero
            auto resolve_list = [&](auto ic) __attribute__((always_inline)) -> bool {
                const uint32_t i = ic;
                if (!(ballot(al0) | ballot(al1))) return false;
                const bool excl = i <= nexcl;
                bool n0 = al0 && byte_of(pk0, i) != 0u, n1 = al1 && byte_of(pk1, i) != 0u; // still to be looked up in list i
                const float qw = __uint_as_float(cx.m(i, M_QW));
                for (;;) {
                    const uint64_t b0 = ballot(n0), b1 = ballot(n1);
                    if (!(b0 | b1)) break;
                    const uint32_t amin = b0 ? bcast(c0, (uint32_t)__builtin_ctzll(b0)) : bcast(c1, (uint32_t)__builtin_ctzll(b1));
                    if (cx.m(i, M_CUR) == 0xFFFFFFFFu || amin > cx.m(i, M_BMAX)) {
                        const uint32_t cur = cx.m(i, M_CUR);
                        uint32_t blk2, nbmax = 0;
                        float wdummy = 0.f;
                        typename decltype(cx)::BlockInfo bi;
                        const bool tabbed = !cx.is_pef() && cx.skip;
                        if (tabbed) blk2 = cx.find_block_info(i, cur + 1, amin, bi, nullptr, wdummy);
                        else blk2 = cx.find_block(i, cur + 1, amin, nbmax, nullptr, wdummy);
                        cx.s_bm_examined += 1;
                        cx.s_bytes += 4;
                        if (blk2 >= cx.m(i, M_NB)) break; // the list has nothing at or after amin: nobody left is in it
                        cx.decode_docs(i, blk2, tabbed ? &bi : nullptr);
                    }
                    const uint32_t bm = cx.m(i, M_BMAX);
                    const bool w0 = n0 && c0 <= bm, w1 = n1 && c1 <= bm;
                    uint32_t p0 = 0, p1 = 0;
                    const bool m0 = member_bsearch(L.docs[i], c0, w0, p0), m1 = member_bsearch(L.docs[i], c1, w1, p1);
                    if (excl) { // found in a list of higher max score: the document is that list's
                        al0 = al0 && !m0;
                        al1 = al1 && !m1;
                    } else if (ballot(m0) | ballot(m1)) {
                        if (!cx.freqs_ready(i)) cx.decode_freqs(i);
                        const uint32_t* fr = cx.F(i);
                        if (m0) pa0 = pa0 + qw * doc_term_weight(fr[p0], nl0);
                        if (m1) pa1 = pa1 + qw * doc_term_weight(fr[p1], nl1);
                    }
                    n0 = n0 && !w0;
                    n1 = n1 && !w1;
                }
                if (!excl) { // who cannot reach the heap any more is not looked up in the lists still to come
                    al0 = al0 && tk.would_enter((pa0 + rest_of(pk0, i)) * BOUND_SLACK);
                    al1 = al1 && tk.would_enter((pa1 + rest_of(pk1, i)) * BOUND_SLACK);
                }
                return true;
            };
            DS2I_LIST_LOOP(1, resolve_list)
            // ---- whoever is still alive has its complete score
            for (int half = 0; half < 2; ++half) {
                const bool al = half ? al1 : al0;
                const float sc = half ? pa1 : pa0;
                uint64_t todo = ballot(al && tk.would_enter(sc));
                while (todo) {
                    const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const float v = __uint_as_float(bcast(__float_as_uint(sc), src));
                    if (tk.insert(v) && shared_floor && lane == 0) sh.add(v);
                }
            }
        }
        finish_unit();
    }
    cx.flush_stats(a.stats);
}

// ------------------------------------------------------------------ or_query as a stream
// or_query<with_freqs> (queries.hpp:88-131) returns the size of the union of the query's lists (and touches every freq).
// There is nothing to prune and nothing to rank, so the lists need not be walked in lock step at all: a unit owns a doc-id
// range; it takes the range 32 Ki doc-ids at a time, and for every list decodes the blocks that reach into the current
// piece -- each block once, front to back, whatever the other lists do -- setting one bit per posting in a 4 KiB bitmap
// in LDS (ds_or, no global atomics). The piece's popcount is its part of the union. No list is ever searched for another
// list's documents, no window is cut at a block boundary, and the number of lists is not a template parameter (<= 64
// per query; longer queries keep the one-document-per-step kernel). A block that straddles the end of a piece is decoded
// again for the next piece (one extra decode per list and piece); pieces without any block are never visited, because the
// next piece starts at the first doc-id any list can still hold.
constexpr uint32_t UNION_PIECE = 32768u;   // doc-ids per bitmap (4 KiB)
constexpr uint32_t UNION_MAX_LISTS = 64u;
struct LdsUnion : Lds<1, false> { // (one list slot at a time, its enumerator state in registers: MetaReg<1>)
    uint32_t bits[UNION_PIECE / 32];
    uint32_t nextblk[UNION_MAX_LISTS];  // first block of list i not yet fully behind the stream
    uint32_t nextbase[UNION_MAX_LISTS]; // smallest doc-id that block can still contribute
};
template <bool WITH_FREQS, int CODEC_T, bool STATS = true>
__global__ void __launch_bounds__(64, 5) k_union(BatchArgs a) {
    __shared__ LdsUnion L;
    const uint32_t lane = lane_id();
    CtxT<CODEC_T, MetaReg<1>, STATS> cx = make_ctx<CODEC_T, MetaReg<1>, STATS>(L, a);
    cx.want_freqs = WITH_FREQS; // or_freq reads every freq it passes
    for (uint32_t tkt = blockIdx.x; tkt < a.nslice; tkt += gridDim.x) {
        const uint32_t uid = a.order[tkt];
        const unsigned long long t_unit = (STATS && a.unit_clock) ? wall_clock64() : 0ull;
        const Unit u = a.units[uid];
        const uint32_t q = u.q;
        const bool whole = u.nparts == 1;
        const uint32_t unit_lo = whole ? 0u : u.blk_begin, unit_hi = whole ? a.num_docs : u.blk_end;
        const uint32_t t0 = a.q_off[q], nt = a.q_off[q + 1] - t0;
        unsigned long long count = 0, fsum = 0;
        if (nt && nt <= UNION_MAX_LISTS && unit_lo < unit_hi) {
            // position every list on its first block that reaches into the unit
            for (uint32_t i = 0; i < nt; ++i) {
                if (!WITH_FREQS && a.rmw_bitmaps && RmwLevels::has_bitmap(a.qterms[t0 + i].n, a.num_docs)) { // (served from its bitmap: no block to find)
                    if (lane == 0) { L.nextblk[i] = 0u; L.nextbase[i] = unit_lo; }
                    continue;
                }
                cx.bind(0, a.qterms[t0 + i]);
                uint32_t bmax = 0;
                float w;
                const uint32_t nb = cx.m(0, M_NB);
                const uint32_t blk = unit_lo ? cx.find_block(0, 0, unit_lo, bmax, nullptr, w) : 0u;
                cx.s_bm_examined += 1;
                cx.s_bytes += 4;
                if (lane == 0) {
                    L.nextblk[i] = blk < nb ? blk : 0xFFFFFFFFu;
                    L.nextbase[i] = blk < nb ? unit_lo : 0xFFFFFFFFu;
                }
            }
            wave_sync();
            for (;;) {
                // the piece starts at the first doc-id some list can still hold
                uint32_t mine = 0xFFFFFFFFu;
                for (uint32_t i = lane; i < nt; i += 64) mine = L.nextbase[i] < mine ? L.nextbase[i] : mine;
                const uint32_t lo = bcast(wave_incl_min_scan(mine), 63);
                if (lo >= unit_hi) break;
                const uint32_t hi = unit_hi - lo > UNION_PIECE ? lo + UNION_PIECE : unit_hi; // [lo, hi)
                ++cx.s_rounds;
#pragma unroll
                for (uint32_t k = 0; k < UNION_PIECE / 32 / 64; ++k) L.bits[k * 64 + lane] = 0u;
                wave_sync();
                for (uint32_t i = 0; i < nt; ++i) {
                    if (uniform(L.nextbase[i]) >= hi) continue;
                    // a dense list (>= one document in 64) has its exact bitmap behind its range table: its part of the
                    // piece is 1024 words to OR in, not a hundred blocks to decode
                    const QTerm& qt = a.qterms[t0 + i];
                    const bool from_bitmap = a.rmw_bitmaps && RmwLevels::has_bitmap(qt.n, a.num_docs);
                    if (from_bitmap) {
                        const uint32_t* bm = (const uint32_t*)(a.rmw + 64ull * qt.rmw_off64 + RmwLevels(a.num_docs, qt.rmw_shift).bytes());
                        const uint32_t w0 = lo >> 5, shft = lo & 31u, nbits = hi - lo;
#pragma unroll
                        for (uint32_t k = 0; k < UNION_PIECE / 32 / 64; ++k) {
                            const uint32_t idx = k * 64 + lane;
                            if (32u * idx < nbits) {
                                uint32_t v = __builtin_amdgcn_alignbit(bm[w0 + idx + 1], bm[w0 + idx], shft); // bit j = doc-id lo + 32 idx + j
                                const uint32_t left = nbits - 32u * idx;
                                if (left < 32u) v &= (1u << left) - 1u;
                                L.bits[idx] |= v;
                            }
                        }
                        if constexpr (!WITH_FREQS) {
                            if (lane == 0) L.nextbase[i] = hi < unit_hi ? hi : 0xFFFFFFFFu;
                            wave_sync();
                            continue;
                        }
                        // or_query<true> still reads every freq (queries.hpp:118-120): the list's blocks are walked below, but a
                        // block that lies wholly inside the piece has nothing left to say about doc-ids -- its docs part is
                        // stepped over by its header and only its freqs are decoded (OptPFor full blocks)
                        wave_sync();
                    }
                    cx.bind(0, a.qterms[t0 + i]);
                    const uint32_t nb = cx.m(0, M_NB);
                    uint32_t b = uniform(L.nextblk[i]), base = hi;
                    // block indexes with the skip table: the list's table rows for the next 63 blocks sit in registers (lane j =
                    // row wfirst + j, lane 0 the row before the first block served), and the bytes of block b + 1 are requested
                    // while block b is decoded -- the stream costs no dependent round trip per block
                    const bool tabbed = !cx.is_pef() && cx.skip;
                    const uint2* const tab = tabbed ? cx.skip + cx.m(0, M_PBASE) : nullptr;
                    const uint8_t* const data = tabbed ? cx.ptr(0, M_MAXS_LO) + 4ull * nb + 4ull * (nb - 1) : nullptr;
                    uint32_t wfirst = 0, pf_blk = 0xFFFFFFFFu, pf0 = 0, pf1 = 0, pfx = 0; // (pfx: the block's exception side slot, with its bytes)
                    uint2 we = make_uint2(0xFFFFFFFFu, 0u);
                    auto wfill = [&](uint32_t first) __attribute__((always_inline)) {
                        wfirst = first;
                        we = make_uint2(0xFFFFFFFFu, 0u);
                        if (first + lane < nb) we = tab[first + lane];
                    };
                    if (tabbed && b < nb) wfill(b ? b - 1 : 0);
                    while (b < nb) {
                        bool freqs_only = false;
                        if (tabbed) {
                            if (b - wfirst > 63u) wfill(b - 1);
                            const uint32_t f = b - wfirst, fp = f ? f - 1 : 0;
                            typename decltype(cx)::BlockInfo bi;
                            bi.bmax = bcast(we.x, f);
                            bi.next_ep = bcast(we.y, f);
                            bi.base = b ? bcast(we.x, fp) + 1u : 0u;
                            bi.ep = b ? bcast(we.y, fp) : 0u;
                            if (bi.base >= hi) { base = bi.base; break; } // the block starts behind the piece: not decoded yet
                            const bool staged = pf_blk == b;
                            const uint8_t* const pblk = data + bi.ep;
                            if (staged) {
                                cx.win.gbase = (const uint8_t*)((uintptr_t)pblk & ~(uintptr_t)3);
                                cx.win.nbytes = 512;
                                cx.win.st[lane] = pf0;
                                cx.win.st[lane + 64] = pf1;
                                if (cx.side()) { cx.exc[lane] = pfx; cx.slot_blk = cx.m(0, M_PBASE) + b; }
                                wave_sync();
                            }
                            if constexpr (WITH_FREQS && CODEC_T == CODEC_OPTPFOR)
                                freqs_only = from_bitmap && bi.base >= lo && bi.bmax < hi && (b + 1u) * 128u <= cx.m(0, M_N);
                            if (freqs_only) {
                                if (!staged) {
                                    uint32_t hint = bi.next_ep - bi.ep;
                                    if (hint == 0 || hint > STAGE_DW * 4 - 4) hint = STAGE_DW * 4 - 4;
                                    cx.win.load(pblk, hint);
                                }
                                const uint32_t hdr = uniform(cx.win.rd32(pblk)); // OptPFor header: b | exceptions | Simple16 words
                                const uint32_t hb = hdr >> 26, hew = hdr & 0xFFFFu;
                                const uint32_t docs_bytes = hb >= 32 ? 4u * (1u + 128u) : 4u * (1u + hew + 4u * hb);
                                const uint64_t fo = (uint64_t)(pblk + docs_bytes - cx.arena);
                                cx.setm(0, M_CUR, b);
                                cx.setm(0, M_SIZE, 128u);
                                cx.setm(0, M_BMAX, bi.bmax);
                                cx.setm(0, M_FREQ_LO, (uint32_t)fo);
                                cx.setm(0, M_FREQ_HI, (uint32_t)(fo >> 32));
                                cx.setm(0, M_FDEC, 0);
                                cx.setm(0, M_DDEC, 0);
                                cx.s_bytes += 8; // endpoint + header
                            } else {
                                cx.decode_docs(0, b, &bi, staged);
                            }
                            pf_blk = 0xFFFFFFFFu;
                            if (b + 1 < nb && bi.bmax + 1u < hi) { // the next block reaches into this piece too: request its bytes now
                                const uint32_t* g = (const uint32_t*)((uintptr_t)(data + bi.next_ep) & ~(uintptr_t)3);
                                pf0 = g[lane];
                                pf1 = g[lane + 64];
                                if (cx.side()) pfx = cx.xslots[(size_t)XSLOT_DW * (cx.m(0, M_PBASE) + b + 1u) + lane];
                                pf_blk = b + 1;
                            }
                        } else {
                            cx.decode_docs(0, b);
                        }
                        if (freqs_only) { // (every posting of the block is inside the piece, and its bits are set already)
                            cx.decode_freqs(0);
                            unsigned long long fs = (unsigned long long)L.freqs[0][lane] + L.freqs[0][lane + 64];
                            for (int o = 32; o; o >>= 1) fs += __shfl_xor(fs, o);
                            fsum += fs;
                        } else {
                            const uint32_t d0 = L.docs[0][lane], d1 = L.docs[0][lane + 64];
                            const bool in0 = d0 >= lo && d0 < hi, in1 = d1 >= lo && d1 < hi; // (the padding doc-id 0xFFFFFFFF is never inside)
                            if (!from_bitmap) {
                                if (in0) atomicOr(&L.bits[(d0 - lo) >> 5], 1u << ((d0 - lo) & 31u));
                                if (in1) atomicOr(&L.bits[(d1 - lo) >> 5], 1u << ((d1 - lo) & 31u));
                            }
                            if constexpr (WITH_FREQS) { // or_query<true> reads the freq of every posting it passes (queries.hpp:118-120)
                                if (ballot(in0) | ballot(in1)) {
                                    if (!cx.m(0, M_FDEC)) cx.decode_freqs(0); // (side slots: decode_docs delivered them already)
                                    unsigned long long fs = (unsigned long long)(in0 ? L.freqs[0][lane] : 0u) + (in1 ? L.freqs[0][lane + 64] : 0u);
                                    for (int o = 32; o; o >>= 1) fs += __shfl_xor(fs, o);
                                    fsum += fs;
                                }
                            }
                        }
                        const uint32_t bmax = cx.m(0, M_BMAX);
                        if (bmax >= hi) { base = hi; break; }       // reaches past the piece: taken up again by the next one
                        ++b;
                        base = bmax + 1u;
                        if (base >= hi) break;
                    }
                    if (lane == 0) {
                        L.nextblk[i] = b;
                        L.nextbase[i] = b < nb ? base : 0xFFFFFFFFu;
                    }
                    wave_sync();
                }
                uint32_t pc = 0;
#pragma unroll
                for (uint32_t k = 0; k < UNION_PIECE / 32 / 64; ++k) pc += (uint32_t)__builtin_popcount(L.bits[k * 64 + lane]);
                pc = bcast(wave_incl_scan(pc), 63);
                count += pc;
                wave_sync();
            }
        }
        if (whole) {
            if (lane == 0) { a.out_count[q] = count; if (a.out_freq_sum) a.out_freq_sum[q] = fsum; }
        } else {
            if (lane == 0) { a.unit_count[uid] = count; a.unit_freq_sum[uid] = fsum; }
        }
        if (STATS && a.unit_clock && lane == 0) { a.unit_clock[2ull * uid] = t_unit; a.unit_clock[2ull * uid + 1] = wall_clock64(); }
    }
    cx.flush_stats(a.stats);
}

// ------------------------------------------------------------------ list decode
// One wave per 128-posting block of ONE list; writes absolute doc-ids and freqs.
__global__ void __launch_bounds__(64) k_decode_list(DecodeArgs a) {
    __shared__ Lds<1> L;
    BatchArgs ba{};
    ba.arena = a.arena;
    ba.bits0 = a.bits0;
    ba.bits1 = a.bits1;
    ba.codec = a.codec;
    ba.num_docs = a.num_docs;
    Ctx cx = make_ctx<-1, MetaLds>(L, ba);
    const uint32_t lane = lane_id();
    cx.bind(0, a.term);
    const uint32_t nb = cx.m(0, M_NB);
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        cx.decode_docs(0, b);
        cx.decode_freqs(0);
        const uint32_t sz = cx.m(0, M_SIZE);
        const size_t gpos = cx.m(0, M_GPOS);
        for (uint32_t i = lane; i < sz; i += 64) {
            a.out_docs[gpos + i] = L.docs[0][i];
            a.out_freqs[gpos + i] = L.freqs[0][i];
        }
        wave_sync();
    }
    cx.flush_stats(a.stats);
}

// The same through the exception side slots + tail table (block_optpfor with BatchArgs::xslots): the decoder of the stream
// kernels (optpfor_decode_side), so that every list-decode test of a block_optpfor index exercises it and the tables.
__global__ void __launch_bounds__(64) k_decode_list_side(DecodeArgs a) {
    __shared__ uint32_t st[STAGE_DW];
    __shared__ uint32_t xs[XSLOT_DW];
    const uint32_t lane = lane_id();
    const QTerm t = a.term;
    const uint32_t n = t.n, nb = (n + 127u) >> 7;
    const uint32_t vl = 1u + (n >= (1u << 7)) + (n >= (1u << 14)) + (n >= (1u << 21)) + (n >= (1u << 28));
    const uint8_t* const data = a.arena + t.list_off + vl + 4ull * nb + 4ull * (nb - 1);
    const uint2* const tab = (const uint2*)a.skip + t.blk_base;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t sz = ((b + 1) * 128u <= n) ? 128u : (n & 127u);
        const uint32_t base = b ? tab[b - 1].x + 1u : 0u, ep = b ? tab[b - 1].y : 0u;
        uint32_t v0, v1, f0, f1;
        if (sz == 128u) {
            const uint32_t* const g = (const uint32_t*)(data + ep);
            const uint32_t* const gx = a.xslots + (size_t)XSLOT_DW * (t.blk_base + b);
            st[lane] = g[lane];
            st[lane + 64] = g[lane + 64];
            xs[lane] = gx[lane];
            wave_sync();
            const SlotHead h = optpfor_slot_head(xs);
            if (h.flag == 0u) {
                uint32_t cd, cf;
                optpfor_decode_pair(st, xs, h, v0, v1, f0, f1, cd, cf);
            } else {
                uint32_t nd = 0;
                const uint32_t cons = optpfor_decode_side(st, STAGE_DW, xs, data + ep, a.xovf, 0u, 0u, v0, v1, &nd);
                const uint32_t skip_dw = cons >> 2;
                optpfor_decode_side(st + skip_dw, skip_dw < STAGE_DW ? STAGE_DW - skip_dw : 0u, xs, data + ep + cons, a.xovf, 1u, nd, f0, f1);
            }
        } else {
            const uint32_t* const tl = a.tails + t.aux1;
            v0 = lane < sz ? tl[lane] : 0u;
            v1 = lane + 64 < sz ? tl[lane + 64] : 0u;
            f0 = lane < sz ? tl[sz + lane] : 0u;
            f1 = lane + 64 < sz ? tl[sz + lane + 64] : 0u;
        }
        const uint32_t g0 = (lane < sz) ? v0 + 1u : 0u, g1 = (lane + 64 < sz) ? v1 + 1u : 0u;
        const uint32_t i0 = wave_incl_scan(g0);
        const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
        const size_t gpos = (size_t)b * 128u;
        if (lane < sz) { a.out_docs[gpos + lane] = base + i0 - 1u; a.out_freqs[gpos + lane] = f0 + 1u; }
        if (lane + 64 < sz) { a.out_docs[gpos + lane + 64] = base + i1 - 1u; a.out_freqs[gpos + lane + 64] = f1 + 1u; }
        wave_sync();
    }
}

// ------------------------------------------------------------------ upload-time block-max weights
// bmw[block] = max over the block's postings of bm25::doc_term_weight(freq, norm_len[doc]) -- the block-level analogue
// of wand_data's max_term_weight (wand_data.hpp:40-52), with the scoring code's own float32 arithmetic. One wave per
// item = <=64 consecutive blocks of one list: lane j keeps the weight of block blk_begin + j, one coalesced store.
// Range-table entries (BatchArgs::rmw): 0 = no posting in the doc-id range; otherwise 1 + floor(255 * w / list max) capped
// at 255, so that entry * (list max / 255) >= w for every posting of the range (the cap meets w <= list max; the float
// rounding of the two products is far inside the pruning bound's BOUND_SLACK).
DS2I_DEV uint32_t rmw_quantise(float w, float inv) {
    const uint32_t q = (uint32_t)(w * inv) + 1u;
    return q > 255u ? 255u : q;
}
// entry = max(entry, q): bytes have no atomic max, so the containing dword is replaced by compare-and-swap
DS2I_DEV void rmw_raise(uint8_t* tab, uint32_t entry, uint32_t q) {
    unsigned int* word = (unsigned int*)(tab + (entry & ~3u));
    const uint32_t sh = 8u * (entry & 3u);
    unsigned int old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (((old >> sh) & 255u) < q) {
        const unsigned int want = (old & ~(255u << sh)) | (q << sh);
        if (__hip_atomic_compare_exchange_strong(word, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
}

// membership hint of a range: first posting -> its code, any further posting -> 255 (bytes have no atomics: CAS on the dword)
DS2I_DEV void rmh_mark(uint8_t* tab, uint32_t entry, uint32_t code) {
    unsigned int* word = (unsigned int*)(tab + (entry & ~3u));
    const uint32_t sh = 8u * (entry & 3u);
    unsigned int old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const uint32_t cur = (old >> sh) & 255u;
        if (cur == 255u) break;
        const unsigned int want = (old & ~(255u << sh)) | ((cur ? 255u : code) << sh);
        if (__hip_atomic_compare_exchange_strong(word, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
}

__global__ void __launch_bounds__(64) k_block_max_weights(BmwArgs a) {
    __shared__ Lds<1> L;
    BatchArgs ba{};
    ba.arena = a.arena;
    ba.bits0 = a.bits0;
    ba.bits1 = a.bits1;
    ba.codec = a.codec;
    ba.num_docs = a.num_docs;
    Ctx cx = make_ctx<-1, MetaLds>(L, ba);
    const uint32_t lane = lane_id();
    for (uint32_t item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const BmwItem it = a.items[item];
        const QTerm t = a.lists[it.list];
        if (a.rmw && a.rmw_level) { // coarser levels of the range table: entry e = max of the 64 entries below it
            const RmwLevels g(a.num_docs, t.rmw_shift);
            const uint8_t* src = a.rmw + 64ull * t.rmw_off64 + g.off[a.rmw_level - 1];
            uint8_t* dst = a.rmw + 64ull * t.rmw_off64 + g.off[a.rmw_level];
            const uint32_t end = it.blk_begin + 4096u < g.e[a.rmw_level] ? it.blk_begin + 4096u : g.e[a.rmw_level];
            for (uint32_t e = it.blk_begin + lane; e < end; e += 64) { // (every level is zero-padded to 64 bytes: no tail case)
                const uint4* p = (const uint4*)(src + 64ull * e);
                uint32_t m = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 v = p[k];
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t x = w[j];
                        uint32_t b = x & 255u, c = (x >> 8) & 255u, d = (x >> 16) & 255u, f = x >> 24;
                        b = b > c ? b : c;
                        d = d > f ? d : f;
                        b = b > d ? b : d;
                        m = m > b ? m : b;
                    }
                }
                dst[e] = (uint8_t)m;
            }
            continue;
        }
        cx.bind(0, t);
        const uint32_t nb = cx.m(0, M_NB);
        const uint32_t end = it.blk_begin + 64u < nb ? it.blk_begin + 64u : nb;
        float mine = 0.f;
        uint8_t* const rtab = a.rmw ? a.rmw + 64ull * t.rmw_off64 : nullptr;
        uint8_t* const htab = (a.rmw && a.rmh) ? a.rmh + 64ull * t.rmw_off64 : nullptr;
        const float rinv = t.max_weight > 0.f ? 255.0f / t.max_weight : 0.f; // second pass: max_weight = the list's largest weight
        unsigned int* const bm = (a.rmw && a.bitmaps && RmwLevels::has_bitmap(t.n, a.num_docs))
                                     ? (unsigned int*)(a.rmw + 64ull * t.rmw_off64 + RmwLevels(a.num_docs, t.rmw_shift).bytes()) : nullptr;
        for (uint32_t b = it.blk_begin; b < end; ++b) {
            cx.decode_docs(0, b);
            cx.decode_freqs(0);
            const uint32_t sz = cx.m(0, M_SIZE);
            float w = 0.f;
            if (lane < sz) w = doc_term_weight(L.freqs[0][lane], a.norm_lens[L.docs[0][lane]]);
            if (a.rmw && lane < sz) rmw_raise(rtab, L.docs[0][lane] >> t.rmw_shift, rmw_quantise(w, rinv));
            if (htab) {
                if (lane < sz) rmh_mark(htab, L.docs[0][lane] >> t.rmw_shift, rmh_code(L.docs[0][lane], t.rmw_shift));
                if (lane + 64 < sz) rmh_mark(htab, L.docs[0][lane + 64] >> t.rmw_shift, rmh_code(L.docs[0][lane + 64], t.rmw_shift));
            }
            if (bm) { // dense list: its exact bitmap
                if (lane < sz) atomicOr(bm + (L.docs[0][lane] >> 5), 1u << (L.docs[0][lane] & 31u));
                if (lane + 64 < sz) atomicOr(bm + (L.docs[0][lane + 64] >> 5), 1u << (L.docs[0][lane + 64] & 31u));
            }
            if (lane + 64 < sz) {
                const float w1 = doc_term_weight(L.freqs[0][lane + 64], a.norm_lens[L.docs[0][lane + 64]]);
                if (a.rmw) rmw_raise(rtab, L.docs[0][lane + 64] >> t.rmw_shift, rmw_quantise(w1, rinv));
                w = w1 > w ? w1 : w;
            }
            if (a.rmw) { wave_sync(); continue; } // second pass: bmw[] and the list maxima are final already
            for (int o = 32; o; o >>= 1) {
                const float x = __shfl_xor(w, o);
                w = x > w ? x : w;
            }
            if (lane == b - it.blk_begin) mine = w;
            wave_sync();
        }
        if (a.rmw) continue;
        if (it.blk_begin + lane < end) a.bmw[t.blk_base + it.blk_begin + lane] = mine;
        float lm = mine;
        for (int o = 32; o; o >>= 1) {
            const float x = __shfl_xor(lm, o);
            lm = x > lm ? x : lm;
        }
        if (lane == 0) atomicMax(a.list_bmw + it.list, __float_as_uint(lm)); // weights >= 0: bit patterns order like values
    }
}

// ------------------------------------------------------------------ upload-time exception side slots + tail table
// block_optpfor: every full block's OptPFor exceptions, re-stated from its two Simple16 streams as position masks + values
// ready to be OR-ed in (layout: device_codecs.hpp, optpfor_decode_side), and every list's partial last block (interpolative
// on disk) as plain gaps-1 / freqs-1. Both come out of the general decoders, i.e. they hold exactly what a query-time
// decode of the on-disk bytes would produce; the image itself is left as it is. One wave per item = <=64 consecutive
// blocks of one list.
__global__ void __launch_bounds__(64) k_build_side_tables(SideArgs a) {
    __shared__ Lds<1> L;
    BatchArgs ba{};
    ba.arena = a.arena;
    ba.codec = CODEC_OPTPFOR;
    ba.num_docs = a.num_docs;
    ba.skip = a.skip;
    Ctx cx = make_ctx<-1, MetaLds>(L, ba); // (the general decoders: CODEC_OPTPFOR as a template argument means "through the side tables")
    const uint32_t lane = lane_id();
    uint32_t bad = 0;
    for (uint32_t item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const BmwItem it = a.items[item];
        const QTerm t = a.lists[it.list];
        cx.bind(0, t);
        const uint32_t nb = cx.m(0, M_NB);
        const uint32_t end = it.blk_begin + 64u < nb ? it.blk_begin + 64u : nb;
        const uint2* const tab = (const uint2*)a.skip + t.blk_base;
        const uint8_t* const data = cx.ptr(0, M_MAXS_LO) + 4ull * nb + 4ull * (nb - 1);
        for (uint32_t b = it.blk_begin; b < end; ++b) {
            cx.decode_docs(0, b);
            cx.decode_freqs(0);
            const uint32_t sz = cx.m(0, M_SIZE);
            const uint32_t base = b ? tab[b - 1].x + 1u : 0u, ep = b ? tab[b - 1].y : 0u;
            // gaps - 1 / freqs - 1 as the block decoders deliver them (value i in lane i & 63, slot i >> 6)
            const uint32_t d0 = L.docs[0][lane], d1 = L.docs[0][lane + 64];
            const uint32_t v0 = lane < sz ? (lane ? d0 - L.docs[0][lane - 1] - 1u : d0 - base) : 0u;
            const uint32_t v1 = lane + 64 < sz ? d1 - L.docs[0][lane + 63] - 1u : 0u;
            const uint32_t f0 = lane < sz ? L.freqs[0][lane] - 1u : 0u, f1 = lane + 64 < sz ? L.freqs[0][lane + 64] - 1u : 0u;
            if (sz < 128u) { // the list's partial last block
                uint32_t* const dst = a.tails + t.aux1; // entry: sz gaps-1, sz freqs-1, bytes of the docs part, bytes of the freqs part
                if (lane < sz) { dst[lane] = v0; dst[sz + lane] = f0; }
                if (lane + 64 < sz) { dst[lane + 64] = v1; dst[sz + lane + 64] = f1; }
                const unsigned long long fo = ((unsigned long long)cx.m(0, M_FREQ_HI) << 32) | cx.m(0, M_FREQ_LO);
                if (lane == 0) {
                    dst[2u * sz] = (uint32_t)(fo - (unsigned long long)(data + (b ? tab[b - 1].y : 0u) - a.arena));
                    dst[2u * sz + 1] = (uint32_t)(t.list_end - fo);
                }
                wave_sync();
                continue;
            }
            const uint8_t* const pd = data + ep;
            const uint8_t* const pf = cx.ptr(0, M_FREQ_LO);
            const uint32_t hd = uniform(ld32(pd)), hf = uniform(ld32(pf));
            const uint32_t bd = hd >> 26, bf = hf >> 26;
            const uint32_t keepd = bd < 32u ? ~((1u << bd) - 1u) : 0u, keepf = bf < 32u ? ~((1u << bf) - 1u) : 0u;
            const uint32_t a0 = v0 & keepd, a1 = v1 & keepd, g0 = f0 & keepf, g1 = f1 & keepf;
            const uint64_t md0 = ballot(a0 != 0u), md1 = ballot(a1 != 0u), mf0 = ballot(g0 != 0u), mf1 = ballot(g1 != 0u);
            const uint32_t nd = (uint32_t)(__builtin_popcountll(md0) + __builtin_popcountll(md1));
            const uint32_t nf = (uint32_t)(__builtin_popcountll(mf0) + __builtin_popcountll(mf1));
            if ((bd < 32u && nd != ((hd >> 16) & 0x3FFu)) || (bf < 32u && nf != ((hf >> 16) & 0x3FFu))) ++bad; // (corrupt image)
            uint32_t* const slot = a.xslots + (size_t)XSLOT_DW * (t.blk_base + b);
            uint32_t* dst = slot + XSLOT_ADDS;
            uint32_t ovf = 0;
            bool ok = true;
            // the common case a wave decodes without a branch: neither part raw, both (and the dword a lane may read past the freqs
            // part) inside the 128 dwords staged from the block's start, every add in the slot
            const uint32_t tot_d = 1u + (hd & 0xFFFFu) + 4u * bd, tot_f = 1u + (hf & 0xFFFFu) + 4u * bf;
            const bool common = bd < 32u && bf < 32u && tot_d + tot_f + 1u <= STAGE_DW && nd + nf <= XSLOT_CAP;
            if (nd + nf > XSLOT_CAP) {
                unsigned long long off = 0;
                if (lane == 0) off = atomicAdd(a.xovf_cursor, (unsigned long long)(nd + nf));
                off = ((unsigned long long)bcast((uint32_t)(off >> 32), 0) << 32) | bcast((uint32_t)off, 0);
                ok = off + nd + nf <= a.xovf_cap && off + nd + nf < 0xFFFFFFFFull; // (otherwise the host re-runs the pass with the room the cursor asks for)
                dst = a.xovf + off;
                ovf = (uint32_t)off + 1u;
            }
            const uint32_t words[8] = {(uint32_t)md0, (uint32_t)(md0 >> 32), (uint32_t)md1, (uint32_t)(md1 >> 32),
                                       (uint32_t)mf0, (uint32_t)(mf0 >> 32), (uint32_t)mf1, (uint32_t)(mf1 >> 32)};
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) mine = lane == (uint32_t)i ? words[i] : mine;
            mine = lane == XSLOT_HDR ? hd : lane == XSLOT_HDR + 1 ? hf : lane == XSLOT_FLAG ? (common ? 0u : (XSLOT_SLOW | (ok ? ovf : 0u))) : mine;
            if (lane <= XSLOT_FLAG) slot[lane] = mine;
            if (ok) {
                const uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(md0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)md0, 0u));
                const uint32_t r1 = (uint32_t)__builtin_popcountll(md0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(md1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)md1, 0u));
                const uint32_t s0 = nd + __builtin_amdgcn_mbcnt_hi((uint32_t)(mf0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf0, 0u));
                const uint32_t s1 = nd + (uint32_t)__builtin_popcountll(mf0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mf1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf1, 0u));
                if (a0) dst[r0] = a0;
                if (a1) dst[r1] = a1;
                if (g0) dst[s0] = g0;
                if (g1) dst[s1] = g1;
            }
            wave_sync();
        }
    }
    if (bad && lane == 0) atomicAdd(a.bad, bad);
}

// top[list][0..63] = the 64 largest bmw values of the list, descending, padded with 0 (one wave per list)
__global__ void __launch_bounds__(64) k_list_top_bmw(const float* bmw, const QTerm* lists, uint32_t nlists, float* top) {
    const uint32_t lane = lane_id();
    for (uint32_t l = blockIdx.x; l < nlists; l += gridDim.x) {
        const QTerm t = lists[l];
        const uint32_t nb = t.nblocks;
        const float* w = bmw + t.blk_base;
        TopK tk;
        tk.init(64);
        for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
            const float v = b0 + lane < nb ? w[b0 + lane] : -1.f;
            uint64_t todo = ballot(b0 + lane < nb && tk.would_enter(v));
            while (todo) {
                const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                tk.insert(__uint_as_float(bcast(__float_as_uint(v), src)));
            }
        }
        top[(size_t)l * 64 + lane] = lane < tk.n ? tk.v : 0.f;
    }
}

// ------------------------------------------------------------------ primitive self-test
__global__ void __launch_bounds__(64) k_selftest(const uint32_t* in, uint32_t* out) {
    const uint32_t lane = lane_id();
    uint32_t x = in[blockIdx.x * 64 + lane];
    out[blockIdx.x * 64 + lane] = wave_incl_scan(x);
}

__global__ void __launch_bounds__(64) k_selftest_bm25(const uint32_t* freqs, const float* norm_lens, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * 64 + lane_id();
    if (i < n) out[i] = doc_term_weight(freqs[i], norm_lens[i]);
}

// wand / maxscore / ranked_or of a ONE-term query are exactly its ranked_and result: copy it from the seed pass
struct CopySeedArgs {
    const uint32_t* queries;
    uint32_t n, k;
    const float* seed_topk;
    const uint32_t* seed_len;
    const unsigned long long* seed_count;
    float* out_topk;
    uint32_t* out_len;
    unsigned long long* out_count;
};
__global__ void __launch_bounds__(64) k_copy_seed(CopySeedArgs a) {
    const uint32_t lane = lane_id();
    for (uint32_t w = blockIdx.x; w < a.n; w += gridDim.x) {
        const uint32_t q = a.queries[w];
        if (lane < a.k) a.out_topk[(size_t)q * a.k + lane] = a.seed_topk[(size_t)q * a.k + lane];
        if (lane == 0) { a.out_len[q] = a.seed_len[q]; a.out_count[q] = a.seed_count[q]; }
    }
}

// FETCH_SIZE calibration (MI355X_MICROARCH.md §HBM): streams `ndw` dwords of the arena with the same
// access shape as Window::load (one dword per lane, 64 consecutive lanes) and folds them into a checksum.
__global__ void __launch_bounds__(64) k_calib_read(const uint32_t* base, unsigned long long ndw, uint32_t* out) {
    uint32_t acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 64 + lane_id(); i < ndw;
         i += (unsigned long long)gridDim.x * 64)
        acc ^= base[i];
    for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o);
    if (lane_id() == 0 && acc == 0x12345678u) out[0] = acc; // keep the loads alive
}

} // namespace

// ------------------------------------------------------------------ launchers (called from capi.cpp)
// the block_optpfor specialisations decode through the upload-time side tables; an index uploaded without them runs the
// runtime-codec instantiations
static inline bool optpfor_side(const BatchArgs& a) { return a.codec == CODEC_OPTPFOR && a.xslots != nullptr && a.tails != nullptr; }

namespace ds2i_launch {

struct Batch {
    BatchArgs a;
};

// The file is compiled once per list-count class (-DDS2I_TU_TMAX=2|4|8|16: only launch_t<TMAX> and the kernels it
// instantiates) and once without the macro (everything else): five translation units that build.py compiles in
// parallel -- the kernel templates are by far the slowest part of the build.
// DS2I_TU_TMAX == 0: the translation unit of the "long" class (k_daat_long, every operator in reference order)
#if defined(DS2I_TU_TMAX) && DS2I_TU_TMAX == 0
hipError_t launch_long(int op, const BatchArgs& a, unsigned grid, hipStream_t s) {
    dim3 g(grid), b(64);
    if (a.k > 64) { // top-k beyond one score per lane: 16 scores per lane (k <= 1024)
        typedef TopKBig<16> BIG;
        switch (op & 0xFF) {
        case OP_RANKED_AND: hipLaunchKernelGGL((k_daat_long<OP_RANKED_AND, BIG>), g, b, 0, s, a); break;
        case OP_WAND: hipLaunchKernelGGL((k_daat_long<OP_WAND, BIG>), g, b, 0, s, a); break;
        case OP_MAXSCORE: hipLaunchKernelGGL((k_daat_long<OP_MAXSCORE, BIG>), g, b, 0, s, a); break;
        case OP_RANKED_OR: hipLaunchKernelGGL((k_daat_long<OP_RANKED_OR, BIG>), g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (op & 0xFF) {
    case OP_AND: hipLaunchKernelGGL((k_daat_long<OP_AND>), g, b, 0, s, a); break;
    case OP_AND_FREQ: hipLaunchKernelGGL((k_daat_long<OP_AND_FREQ>), g, b, 0, s, a); break;
    case OP_OR: hipLaunchKernelGGL((k_daat_long<OP_OR>), g, b, 0, s, a); break;
    case OP_OR_FREQ: hipLaunchKernelGGL((k_daat_long<OP_OR_FREQ>), g, b, 0, s, a); break;
    case OP_RANKED_AND: hipLaunchKernelGGL((k_daat_long<OP_RANKED_AND>), g, b, 0, s, a); break;
    case OP_WAND: hipLaunchKernelGGL((k_daat_long<OP_WAND>), g, b, 0, s, a); break;
    case OP_MAXSCORE: hipLaunchKernelGGL((k_daat_long<OP_MAXSCORE>), g, b, 0, s, a); break;
    case OP_RANKED_OR: hipLaunchKernelGGL((k_daat_long<OP_RANKED_OR>), g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#else
hipError_t launch_long(int op, const BatchArgs& a, unsigned grid, hipStream_t s);
#endif

template <int TMAX>
hipError_t launch_t(int op, const BatchArgs& a, unsigned grid, hipStream_t s)
#if defined(DS2I_TU_TMAX) && DS2I_TU_TMAX > 0
{
    dim3 g(grid), b(64);
    const size_t dyn = 1024u * (size_t)a.dyn_lists; // union kernels: docs + freqs of dyn_lists list slots
    const size_t dyn_docs = 512u * (size_t)a.dyn_lists; // or_query never reads a freq: docs only
    switch (op) {
    // the conjunctive kernels are specialised for block_optpfor (the benchmark codec), the freq_index family and
    // block_mixed (configs[4]; its three block types stay a run-time switch, QMX drops out); block_varint /
    // block_interpolative / block_qmx go through the runtime-dispatch instantiation (CODEC_T = -1)
    case OP_AND:
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_PEF, false>), g, b, 0, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_PEF>), g, b, 0, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_MIXED>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, -1>), g, b, 0, s, a);
        break;
    case OP_AND_FREQ:
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_PEF, false>), g, b, 0, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_PEF>), g, b, 0, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_MIXED>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, -1>), g, b, 0, s, a);
        break;
    case OP_RANKED_AND:
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF && !a.stats) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_PEF, false>), g, b, 0, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_PEF>), g, b, 0, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_MIXED>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, -1>), g, b, 0, s, a);
        break;
    // the ranked disjunctive operators get the same two codec specialisations (BASELINE configs[3] runs them on
    // block_optpfor); or / or_freq and the reference-order conjunctions stay on the runtime-dispatch instantiation
    case OP_OR: hipLaunchKernelGGL((k_disjunctive<TMAX, -1, true, 1>), g, b, dyn_docs, s, a); break;
    case OP_OR_FREQ: hipLaunchKernelGGL((k_disjunctive<TMAX, -1, true, 2>), g, b, dyn, s, a); break;
    case 0x100 | OP_OR: hipLaunchKernelGGL((k_daat<OP_OR, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_OR_FREQ: hipLaunchKernelGGL((k_daat<OP_OR_FREQ, TMAX>), g, b, 0, s, a); break;
    // wand / maxscore / ranked_or: the block-synchronous disjunctive kernel (identical results by definition)
    case OP_WAND:
    case OP_MAXSCORE:
    case OP_RANKED_OR:
        if (a.vq_info) { // the streaming form (units = (query, driving list, block range)); needs the range tables
            if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_union_topk<TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
            else if (optpfor_side(a)) hipLaunchKernelGGL((k_union_topk<TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
            else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_union_topk<TMAX, CODEC_PEF>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union_topk<TMAX, -1>), g, b, 0, s, a);
            break;
        }
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_OPTPFOR, false>), g, b, dyn, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_OPTPFOR>), g, b, dyn, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_PEF>), g, b, dyn, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_MIXED>), g, b, dyn, s, a);
        else hipLaunchKernelGGL((k_disjunctive<TMAX, -1>), g, b, dyn, s, a);
        break;
    // reference-order (one document per step) traversals of the same operators: op | OP_REFERENCE_ORDER
    case 0x100 | OP_WAND: hipLaunchKernelGGL((k_daat<OP_WAND, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_MAXSCORE: hipLaunchKernelGGL((k_daat<OP_MAXSCORE, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_RANKED_OR: hipLaunchKernelGGL((k_daat<OP_RANKED_OR, TMAX>), g, b, 0, s, a); break;
    // reference-order (one candidate per step) conjunctive traversal: op | OP_REFERENCE_ORDER
    case 0x100 | OP_AND: hipLaunchKernelGGL((k_daat<OP_AND, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_AND_FREQ: hipLaunchKernelGGL((k_daat<OP_AND_FREQ, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_RANKED_AND: hipLaunchKernelGGL((k_daat<OP_RANKED_AND, TMAX>), g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template hipError_t launch_t<DS2I_TU_TMAX>(int, const BatchArgs&, unsigned, hipStream_t);
#else
;
extern template hipError_t launch_t<2>(int, const BatchArgs&, unsigned, hipStream_t);
extern template hipError_t launch_t<4>(int, const BatchArgs&, unsigned, hipStream_t);
extern template hipError_t launch_t<8>(int, const BatchArgs&, unsigned, hipStream_t);
extern template hipError_t launch_t<16>(int, const BatchArgs&, unsigned, hipStream_t);
#endif

} // namespace ds2i_launch

#if !defined(DS2I_TU_TMAX)
extern "C" {

// tmax_class: 0 -> TMAX 2, 1 -> TMAX 4, 2 -> TMAX 8, 3 -> TMAX 16 (LDS footprint per wave grows with TMAX),
// 4 -> more than 16 terms (state in global scratch)
hipError_t ds2i_launch_batch(int op, int tmax_class, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    if ((op == OP_OR || op == OP_OR_FREQ) && a.dyn_lists == 0xFFFFFFFFu) { // or_query as a stream: one kernel for every list count
        const dim3 g(grid), b(64);
        const bool f = op == OP_OR_FREQ;
        if (optpfor_side(a) && !a.stats) {
            if (f) hipLaunchKernelGGL((k_union<true, CODEC_OPTPFOR, false>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union<false, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        } else if (optpfor_side(a)) {
            if (f) hipLaunchKernelGGL((k_union<true, CODEC_OPTPFOR>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union<false, CODEC_OPTPFOR>), g, b, 0, s, a);
        } else {
            if (f) hipLaunchKernelGGL((k_union<true, -1>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union<false, -1>), g, b, 0, s, a);
        }
        return hipGetLastError();
    }
    switch (tmax_class) {
    case 0: return ds2i_launch::launch_t<2>(op, a, grid, s);
    case 1: return ds2i_launch::launch_t<4>(op, a, grid, s);
    case 2: return ds2i_launch::launch_t<8>(op, a, grid, s);
    case 3: return ds2i_launch::launch_t<16>(op, a, grid, s);
    default: return ds2i_launch::launch_long(op, a, grid, s);
    }
}

uint32_t ds2i_meta_words(void) { return (uint32_t)M_WORDS; }

hipError_t ds2i_launch_block_max_weights(const void* args, unsigned grid, hipStream_t s) {
    const BmwArgs& a = *(const BmwArgs*)args;
    hipLaunchKernelGGL(k_block_max_weights, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_build_side_tables(const void* args, unsigned grid, hipStream_t s) {
    const SideArgs& a = *(const SideArgs*)args;
    hipLaunchKernelGGL(k_build_side_tables, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t ds2i_launch_list_top_bmw(const float* bmw, const void* lists, uint32_t nlists, float* out, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL(k_list_top_bmw, dim3(grid), dim3(64), 0, s, bmw, (const QTerm*)lists, nlists, out);
    return hipGetLastError();
}

hipError_t ds2i_launch_merge(const void* args, unsigned grid, hipStream_t s) {
    const MergeArgs& a = *(const MergeArgs*)args;
    hipLaunchKernelGGL(k_merge, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_decode_list_side(const void* args, unsigned grid, hipStream_t s) {
    const DecodeArgs& a = *(const DecodeArgs*)args;
    hipLaunchKernelGGL(k_decode_list_side, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t ds2i_launch_decode_list(const void* args, unsigned grid, hipStream_t s) {
    const DecodeArgs& a = *(const DecodeArgs*)args;
    hipLaunchKernelGGL(k_decode_list, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_selftest(const uint32_t* in, uint32_t* out, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest, dim3(blocks), dim3(64), 0, s, in, out);
    return hipGetLastError();
}

hipError_t ds2i_launch_selftest_bm25(const uint32_t* freqs, const float* norm_lens, float* out, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_bm25, dim3((n + 63) / 64), dim3(64), 0, s, freqs, norm_lens, out, n);
    return hipGetLastError();
}

hipError_t ds2i_launch_copy_seed(const uint32_t* queries, uint32_t n, uint32_t k, const float* seed_topk, const uint32_t* seed_len,
                                 const unsigned long long* seed_count, float* out_topk, uint32_t* out_len,
                                 unsigned long long* out_count, hipStream_t s) {
    CopySeedArgs a{queries, n, k, seed_topk, seed_len, seed_count, out_topk, out_len, out_count};
    hipLaunchKernelGGL(k_copy_seed, dim3(n < 1024 ? n : 1024), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_calib_read(const uint32_t* base, unsigned long long ndw, uint32_t* out, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL(k_calib_read, dim3(grid), dim3(64), 0, s, base, ndw, out);
    return hipGetLastError();
}

size_t ds2i_sizeof_batch_args() { return sizeof(BatchArgs); }
size_t ds2i_sizeof_decode_args() { return sizeof(DecodeArgs); }
}
#endif // !DS2I_TU_TMAX
