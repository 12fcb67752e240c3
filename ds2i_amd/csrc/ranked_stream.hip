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
#include "stream_common.hpp"

using namespace ds2i_dev;
using namespace ds2i_dev::stream;

namespace {

#ifndef DS2I_RS_OCC2
#define DS2I_RS_OCC2 6
#endif
#ifndef DS2I_RS_OCC4
#define DS2I_RS_OCC4 6 // (round 5, on the leaner kernel: 5 -> 6 waves per SIMD for 3 / 4 lists +3.5 % end to end; 6 -> 7 / 8 for 2 lists: nothing)
#endif
// 5..8 lists: what fits -- one more decoded block (1 KB of LDS) and nine more parked scalars per list:
// 7 936 .. 11 008 bytes of LDS per wave
#define RS_WAVES(NT) ((NT) <= 2 ? DS2I_RS_OCC2 : (NT) <= 4 ? DS2I_RS_OCC4 : (NT) <= 6 ? 4 : (NT) <= 8 ? 3 : 2)

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

// stage C as ONE loop body over the lists (run-time list index) from this capacity on; below it the body is unrolled per list
#ifndef RS_ONE_BODY
#define RS_ONE_BODY 4
#endif
#ifndef RS_HINT_FIRST
#define RS_HINT_FIRST(nt) ((nt) > 2)
#endif
// AND = true: and_query (queries.hpp:35-86, count only) through the same pipeline -- no scores, no heap: a candidate of list 0 is
// a result iff every other list holds it. The membership hints settle that exactly wherever a range holds one posting and is at
// most 254 doc-ids wide (rmh_code is injective there): such a candidate is counted without list j ever being searched or decoded;
// only candidates in ranges with several postings (hint 255), in wider ranges, or of an upload without hints are probed.
// FREQS (with AND): and_query<with_freqs> -- the freq of every list is touched for every document of the intersection (queries.hpp:62-84;
// here: summed into the query's checksum). The hints still rule out every candidate that is in no intersection without a search; what
// passes them is looked up in every list (a member's freq is in that list's block), list 0's freq came with its doc-id.
// NK > 1 (ranked_and with k > 64; topk_queue has no limit, queries.hpp:152-197): NK scores per lane (TopKBig<NK>: k <= 64 NK) at the
// price of NK registers and NK times the work per heap insert -- fewer waves per SIMD. These instantiations also take one-term
// queries (nt = 1 in a capacity-4 launch: no list 1, every test on it passes), which otherwise keep their class kernel.
#define RS_WAVES_K(NT, NK) ((NK) == 1 ? RS_WAVES(NT) : (RS_WAVES(NT) < ((NK) <= 4 ? 4 : 3) ? RS_WAVES(NT) : ((NK) <= 4 ? 4 : 3)))
template <int NT, bool STATS, bool AND = false, bool FREQS = false, int NK = 1>
__global__ void __launch_bounds__(64, RS_WAVES_K(NT, NK)) k_ranked_stream(BatchArgs a_unused) {
    static_assert(AND || !FREQS, "FREQS is a variant of AND");
    static_assert(NK == 1 || !AND, "the big heap is ranked_and's");
    static_assert(NT >= 2 && NT <= 16, "list capacities 2..16");
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
        // NT is the list CAPACITY of the instantiation (2, 4, 6, 8); the query has nt <= NT lists (UnitRec::pad; two lists: always 2).
        // A list slot j >= nt does not exist: its bytes are never loaded, count as zero, and no test looks at them.
        const uint32_t nt = NT == 2 ? 2u : uniform(u.pad);
        const bool has1 = NT == 2 || nt > 1u; // (a one-term query in a capacity-4 launch, NK > 1 only: there is no list 1)
        const QTerm* const qt = rs_uniform_ptr(a->qterms + uniform(u.qt_off)); // nt terms
        typename std::conditional<NK == 1, TopK, TopKBig<NK>>::type tk;
        tk.init(a->k);
        unsigned long long and_count = 0; // (AND: results of this unit)
        unsigned long long and_fsum = 0;  // (FREQS: this lane's share of the unit's freq checksum)
        // (AND, batches prepared with want_matches: the doc-ids of the intersection, in order, into the unit's segment of the match
        // buffer -- 128 slots per block of list 0 from the query's offset, compacted by ds2i_hip_batch_fetch_matches)
        uint32_t* and_out = nullptr;
        if constexpr (AND) {
            uint32_t* const om = a->out_matches;
            if (om) and_out = om + rs_uniform64(a->match_off[q]) + 128ull * blk_begin;
        }
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
            if ((uint32_t)j < nt) {
                rt[j] = a->rmw + 64ull * uniform(qt[j].rmw_off64);
                rsh[j] = uniform(qt[j].rmw_shift);
                rsc[j] = rs_uniformf(qt[j].rmw_scale);
            } else {
                rt[j] = a->rmw;
                rsh[j] = 31u;
                rsc[j] = 0.f;
            }
        };
        rs_for<1, NT>(bind_one);
        float rscv = 0.f; // lane j = rsc[j] (read by stage C's one-body list loop)
        if constexpr (NT >= RS_ONE_BODY) {
            auto spread = [&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; rscv = lane == (uint32_t)j ? rsc[j] : rscv; };
            rs_for<1, NT>(spread);
        }
        // the collection's shortest document: a posting of list 0 with freq f scores at most qw0 * doc_term_weight(f, min_nl) there
        const float min_nl = a->min_norm_len;
        const long long hdelta = a->rmh ? (long long)(a->rmh - a->rmw) : 0ll; // hint of an entry = the byte at the same offset of the parallel buffer
        // Three and four lists: the byte fetched ahead for every candidate is list 1's HINT, not its weight. A weight byte lets a
        // candidate through whenever its range holds any posting (one candidate in 4..6, each then costing a line per further
        // list); the hint also settles the ranges with a single posting, so that the further lists are asked about a few per cent
        // of the candidates only -- first their hints, then, for what is left, every list's weight for the threshold test.
        // (Two lists: the weight stays first, its threshold test removes more than the hint does.)
        const bool hint_first = (AND || RS_HINT_FIRST(NT)) && hdelta != 0; // (AND: the hint is the answer, the weight says nothing it needs)
        const uint8_t* const gt1 = hint_first ? rt[1] + hdelta : rt[1];
        // block of list j whose doc-ids and freqs are in L.dj[j-1] / L.fj[j-1] (cur = ~0: none) and its block_max. Only stage C
        // touches them: they live in the lanes of one VGPR (v_readlane / v_writelane at a constant lane).
        // Together with the list's geometry (postings, first row in the per-block tables, arena offset, query weight): read once
        // per unit, here, so that stage C starts with its first memory request instead of a dependent read of the QTerm.
        // (nine lanes per list: one VGPR up to 8 lists, three for 16)
        constexpr int NCOLD = ((NT - 1) * 9 + 63) / 64;
        uint32_t cold[NCOLD];
#pragma unroll
        for (int i = 0; i < NCOLD; ++i) cold[i] = 0xFFFFFFFFu;
        auto cold_get = [&](uint32_t l) __attribute__((always_inline)) -> uint32_t { // l wave-uniform (a constant wherever the list index is one)
            uint32_t v = cold[0];
            if constexpr (NCOLD > 1) { if (l >= 64u) v = cold[1]; }
            if constexpr (NCOLD > 2) { if (l >= 128u) v = cold[2]; }
            return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(l & 63u));
        };
        auto cold_set_at = [&](uint32_t l, uint32_t v) __attribute__((always_inline)) { // l known at run time only
            if constexpr (NCOLD == 1) rs_writelane_at(cold[0], v, l);
            else {
                if (l < 64u) rs_writelane_at(cold[0], v, l);
                else if (NCOLD == 2 || l < 128u) rs_writelane_at(cold[1], v, l & 63u);
                else rs_writelane_at(cold[NCOLD - 1], v, l & 63u);
            }
        };
        enum { C_CUR = 0, C_BMAX = 1, C_N = 2, C_BB = 3, C_LOLO = 4, C_LOHI = 5, C_QW = 6, C_TLLO = 7, C_TLHI = 8, C_PER = 9 };
#define cget(l) cold_get((uint32_t)(l))
#define cset(l, v) rs_writelane<((l) & 63)>(cold[(l) >> 6], (v))
        {
            auto park = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int CB = (j - 1) * C_PER;
                if ((uint32_t)j >= nt) return;
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
        const bool shared_floor = AND ? false : (!whole && q_hist);
        ScoreHist sh;
        sh.init(shared_floor ? q_hist : nullptr, shared_floor ? uniform(u.hist_slot) : 0u, shared_floor ? rs_uniformf(qt[0].max_bmw + qt[0].suf_bmw) : 0.f,
                1.0f - 1.0f / 1048576.0f);
        // can a score enter the heap: s >= floor && (heap not full || s > k-th score) (TopK::would_enter), branch-free on two
        // wave-uniform values that are refreshed whenever the heap or the floor changes
        float e_floor = tk.floor, e_gt = -__builtin_inff();
        auto refresh = [&]() __attribute__((always_inline)) { e_floor = tk.floor; e_gt = tk.n < tk.k ? -__builtin_inff() : tk.thr; };
        auto enters = [&](float s) __attribute__((always_inline)) -> bool {
            if constexpr (AND) return true; // (no threshold: every document of the intersection is a result)
            else return (s >= e_floor) & (s > e_gt);
        };
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
        // (kept per row as well: the other lists' part of that bound -- what they can add to any posting of the block at most. The
        // candidates' first test uses it where the lists' maxima would let more of them through to list 1's table)
        uint32_t s_first = 0, s_valid = 0;
        uint2 s_e = make_uint2(0xFFFFFFFFu, 0u);
        float s_ub = -1.f, s_rest = 0.f;

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
                if ((uint32_t)j >= nt) return;
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
            rs_settle_vm(); // (once per 63 blocks: no compiler-visible load stays "possibly pending" on the hot path, stream_common.hpp)
            s_rest = acc;
            if constexpr (AND) s_ub = (dead || !row) ? -1.0f : 0.0f; // (a live row: some posting of every other list lies in the block's span)
            else s_ub = (dead || !row) ? -1.0f : (qw0 * s_w + acc) * BOUND_SLACK; // (scores are >= 0: -1 never enters)
        };
        // a block of list 0 on its way through the stages
        struct Blk { uint32_t blk, base, ep; float rest; };
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
                    if constexpr (!AND) o.rest = __uint_as_float(bcast(__float_as_uint(s_rest), f));
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
        using GP = typename std::conditional<(NT > 9), unsigned __int128, typename std::conditional<(NT > 5), unsigned long long, uint32_t>::type>::type;
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
                if constexpr (!AND) {
                    boA0 = qw0 * rs_dtw_bound(fv0 + 1u, min_nl);
                    boA1 = qw0 * rs_dtw_bound(fv1 + 1u, min_nl);
                }
                if constexpr (!AND || FREQS) {
                    L.stage[bufA][lane] = fv0 + 1u; // (same lanes write and read: no fence needed before stage C's read an iteration later)
                    L.stage[bufA][lane + 64] = fv1 + 1u;
                }
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
                // the byte fetched ahead: list 1's weight (2 lists) or hint (3, 4 lists); without a list 1: "several postings", which passes every test
                const uint32_t x0 = has1 ? L.gb[0][lane] : 255u, x1 = has1 ? L.gb[1][lane] : 255u;
                GP gP0 = x0, gP1 = x1;
                // (the threshold only rises: a candidate alive now was alive when the gathers were issued, so its byte is there)
                bool ok0 = (dB0 != 0xFFFFFFFFu) & enters((boB0 + B.rest) * BOUND_SLACK) & (x0 != 0u);
                bool ok1 = (dB1 != 0xFFFFFFFFu) & enters((boB1 + B.rest) * BOUND_SLACK) & (x1 != 0u);
                // AND: bit j = list j has to be searched for this candidate (its hint did not settle it: several postings in the range,
                // a range wider than rmh_code is injective over, or no hints at all)
                uint32_t need0 = 0, need1 = 0;
                if (hint_first) {
                    ok0 = ok0 & ((x0 == 255u) | (x0 == rmh_code(dB0, rsh[1])));
                    ok1 = ok1 & ((x1 == 255u) | (x1 == rmh_code(dB1, rsh[1])));
                    gP0 = gP1 = 0u;
                    if constexpr (AND) {
                        need0 = ((x0 == 255u) | (rsh[1] > 7u)) ? 2u : 0u;
                        need1 = ((x1 == 255u) | (rsh[1] > 7u)) ? 2u : 0u;
                    }
                    LC(PH_C_SURV1, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                    if constexpr (NT > 2) {
                        if (ballot(ok0) | ballot(ok1)) { // the further lists' hints, all requested before any is tested
                            uint32_t h0[NT] = {}, h1[NT] = {};
                            auto hload = [&](auto jc) __attribute__((always_inline)) {
                                constexpr int j = decltype(jc)::value;
                                const uint8_t* const ht = rt[j] + hdelta;
                                h0[j] = (ok0 & ((uint32_t)j < nt)) ? (uint32_t)ht[dB0 >> rsh[j]] : 0u;
                                h1[j] = (ok1 & ((uint32_t)j < nt)) ? (uint32_t)ht[dB1 >> rsh[j]] : 0u;
                                LC(PH_MEMBER, lines_of(ht + (dB0 >> rsh[j]), ok0, 1u) + lines_of(ht + (dB1 >> rsh[j]), ok1, 1u));
                            };
                            rs_for<2, NT>(hload);
                            auto htest = [&](auto jc) __attribute__((always_inline)) {
                                constexpr int j = decltype(jc)::value;
                                if ((uint32_t)j >= nt) return;
                                ok0 = ok0 & (h0[j] != 0u) & ((h0[j] == 255u) | (h0[j] == rmh_code(dB0, rsh[j])));
                                ok1 = ok1 & (h1[j] != 0u) & ((h1[j] == 255u) | (h1[j] == rmh_code(dB1, rsh[j])));
                                if constexpr (AND) {
                                    need0 |= ((h0[j] == 255u) | (rsh[j] > 7u)) ? (1u << j) : 0u;
                                    need1 |= ((h1[j] == 255u) | (rsh[j] > 7u)) ? (1u << j) : 0u;
                                }
                            };
                            rs_for<2, NT>(htest);
                            rs_settle_vm();
                        }
                    }
                    if (!AND && (ballot(ok0) | ballot(ok1))) { // every list's weight byte for what is left (AND: nothing to weigh)
                        auto wload = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            gP0 |= (GP)((ok0 & ((uint32_t)j < nt)) ? (uint32_t)rt[j][dB0 >> rsh[j]] : 0u) << (8 * (j - 1));
                            gP1 |= (GP)((ok1 & ((uint32_t)j < nt)) ? (uint32_t)rt[j][dB1 >> rsh[j]] : 0u) << (8 * (j - 1));
                            LC(PH_FREQS, lines_of(rt[j] + (dB0 >> rsh[j]), ok0, 1u) + lines_of(rt[j] + (dB1 >> rsh[j]), ok1, 1u));
                        };
                        rs_for<1, NT>(wload);
                        rs_settle_vm();
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
                            if ((uint32_t)j >= nt) return;
                            gP0 |= (GP)(uint32_t)rt[j][(ok0 ? dB0 : 0u) >> rsh[j]] << (8 * (j - 1));
                            gP1 |= (GP)(uint32_t)rt[j][(ok1 ? dB1 : 0u) >> rsh[j]] << (8 * (j - 1));
                        };
                        rs_for<2, NT>(load_one);
                        auto test_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            if ((uint32_t)j >= nt) return;
                            ok0 = ok0 & (gbyte(gP0, j) != 0u);
                            ok1 = ok1 & (gbyte(gP1, j) != 0u);
                        };
                        rs_for<2, NT>(test_one);
                        rs_settle_vm();
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
                if constexpr (AND) {
                    if (!hint_first) { // an upload without hints: a non-zero byte is the answer only where an entry is one doc-id
                        auto nm = [&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; if (rsh[j] != 0u && (uint32_t)j < nt) need0 |= 1u << j; };
                        rs_for<1, NT>(nm);
                        need1 = need0;
                    }
                    if constexpr (FREQS) need0 = need1 = (1u << nt) - 2u; // (a member's freq has to be fetched from every list)
                }
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
                        const bool hashint = rsh[j] != 0u && (uint32_t)j < nt; // (one doc-id per entry: the weight byte was the answer)
                        const uint32_t h0 = (ok0 & hashint) ? (uint32_t)ht[dB0 >> rsh[j]] : 255u, h1 = (ok1 & hashint) ? (uint32_t)ht[dB1 >> rsh[j]] : 255u;
                        LC(PH_MEMBER, lines_of(ht + (dB0 >> rsh[j]), ok0 & hashint, 1u) + lines_of(ht + (dB1 >> rsh[j]), ok1 & hashint, 1u));
                        ok0 = ok0 & ((h0 == 255u) | (h0 == rmh_code(dB0, rsh[j])));
                        ok1 = ok1 & ((h1 == 255u) | (h1 == rmh_code(dB1, rsh[j])));
                    };
                    rs_for<1, NT>(hint_one);
                    rs_settle_vm();
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
                    // (AND: nothing is scored -- no norm_lens, no freqs -- and list 1 is searched only if somebody needs it: no rows ahead)
                    Rows rows1{make_uint2(0xFFFFFFFFu, 0u), 0.f};
                    float nl0 = 1.f, nl1 = 1.f, pa0 = 0.f, pa1 = 0.f;
                    uint32_t fs0 = 0, fs1 = 0; // (FREQS: the candidate's freqs so far, list 0's first)
                    if constexpr (FREQS) { fs0 = L.stage[bufB][lane]; fs1 = L.stage[bufB][lane + 64]; }
                    if constexpr (!AND) {
                        if (has1) rows1 = rows_load((const uint2*)rs_args()->skip + cget(C_BB), rs_args()->bmw + cget(C_BB), (cget(C_N) + 127u) >> 7, cget(C_CUR) + 1u);
                        nl0 = ok0 ? norm_lens[dB0] : 1.f;
                        nl1 = ok1 ? norm_lens[dB1] : 1.f;
                        LC(PH_SCORE, lines_of(norm_lens + dB0, ok0, 4u) + lines_of(norm_lens + dB1, ok1, 4u));
                        const uint32_t fB0 = L.stage[bufB][lane], fB1 = L.stage[bufB][lane + 64];
                        pa0 = qw0 * doc_term_weight(fB0, nl0);
                        pa1 = qw0 * doc_term_weight(fB1, nl1);
                        {
                            const uint32_t nv = (uint32_t)(__builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                            s_scored += nv;
                            s_bytes += 4ull * nv;
                        }
                        ok0 = ok0 & enters((pa0 + r0) * BOUND_SLACK);
                        ok1 = ok1 & enters((pa1 + r1) * BOUND_SLACK);
                    }
                    // lists 1 .. NT-1 in order: a candidate moves on only while partial score + what the later lists can add to IT
                    // can still enter the heap
                    // (up to four lists the body is unrolled per list, the lanes of `cold` are compile-time constants; beyond that it is ONE
                    // loop body over the lists: unrolled, the 8-list kernel was 80 KB of code -- more than the instruction cache two CUs
                    // share -- and ran its 77 queries in the time the 5-list kernel ran 338)
                    auto probe = [&](auto jc) __attribute__((always_inline)) {
                        constexpr bool RT = std::is_same<decltype(jc), uint32_t>::value; // run-time list index
                        const uint32_t j = (uint32_t)jc;
                        if (j >= nt) return;
                        const uint32_t CB = (j - 1u) * C_PER; // this list's lanes of `cold`
                        float rsc_j = 0.f; // rsc[j]: from lane j of rscv where j is a run-time index (an array indexed at run time goes to scratch)
                        if constexpr (RT) rsc_j = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rscv), (int)j));
                        else rsc_j = rsc[decltype(jc)::value];
                        // (AND: only the candidates list j's hint left open are searched; the others are members already)
                        uint64_t todo0 = AND ? ballot(ok0 & (((need0 >> j) & 1u) != 0u)) : ballot(ok0), todo1 = AND ? ballot(ok1 & (((need1 >> j) & 1u) != 0u)) : ballot(ok1);
                        if (!(todo0 | todo1)) return;
                        const uint32_t nj = cget(CB + C_N), nbj = (nj + 127u) >> 7, bbj = cget(CB + C_BB);
                        const uint32_t vlj = 1u + (nj >= (1u << 7)) + (nj >= (1u << 14)) + (nj >= (1u << 21)) + (nj >= (1u << 28));
                        const uint8_t* const dataj = arena + (((unsigned long long)cget(CB + C_LOHI) << 32) | cget(CB + C_LOLO)) + vlj + 4ull * nbj + 4ull * (nbj - 1);
                        const uint2* const tabj = (const uint2*)rs_args()->skip + bbj;
                        const float* const wtabj = rs_args()->bmw + bbj;
                        const float qwj = __uint_as_float(cget(CB + C_QW));
                        bool rows_fresh = !AND && j == 1; // (rows1 was loaded for from = list 1's current block + 1)
                        const float rj0 = rest_of(gP0, j), rj1 = rest_of(gP1, j); // the lists after j
                        const float bj0 = rsc_j * (float)gbyte(gP0, j), bj1 = rsc_j * (float)gbyte(gP1, j);
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
                                if constexpr (RT) {
                                    cold_set_at(CB + C_CUR, curj);
                                    cold_set_at(CB + C_BMAX, bmj);
                                } else {
                                    cset((decltype(jc)::value - 1) * C_PER + C_CUR, curj);
                                    cset((decltype(jc)::value - 1) * C_PER + C_BMAX, bmj);
                                }
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
                            if constexpr (AND) {
                                mem0 = mem0 | m0;
                                mem1 = mem1 | m1;
                                if constexpr (FREQS) {
                                    if (m0) fs0 += fj[q0];
                                    if (m1) fs1 += fj[q1];
                                }
                            } else {
                                if (m0) { pa0 = pa0 + qwj * doc_term_weight(fj[q0], nl0); mem0 = true; }
                                if (m1) { pa1 = pa1 + qwj * doc_term_weight(fj[q1], nl1); mem1 = true; }
                            }
                        }
                        // members whose score can still enter go on to the next list (a candidate the loop left unsettled -- list j
                        // ended below it -- is not a member)
                        if constexpr (AND) {
                            ok0 = ok0 & (mem0 | (((need0 >> j) & 1u) == 0u));
                            ok1 = ok1 & (mem1 | (((need1 >> j) & 1u) == 0u));
                        } else {
                            ok0 = ok0 & mem0 & enters((pa0 + rj0) * BOUND_SLACK);
                            ok1 = ok1 & mem1 & enters((pa1 + rj1) * BOUND_SLACK);
                        }
                    };
                    if constexpr (NT >= RS_ONE_BODY) {
#pragma nounroll
                        for (uint32_t j = 1; j < nt; ++j) probe(j);
                    }
                    else rs_for<1, NT>(probe);
                    if constexpr (AND) {
                        const uint64_t m0 = ballot(ok0), m1 = ballot(ok1);
                        if (and_out) { // (value i of the block sits in lane i & 63, slot i >> 6: slot 0 first keeps the doc-ids ascending)
                            const uint64_t below = (1ull << lane) - 1ull;
                            const unsigned long long at0 = and_count + (unsigned long long)__builtin_popcountll(m0 & below);
                            const unsigned long long at1 = and_count + (unsigned long long)__builtin_popcountll(m0) + (unsigned long long)__builtin_popcountll(m1 & below);
                            if (ok0) and_out[at0] = dB0;
                            if (ok1) and_out[at1] = dB1;
                        }
                        and_count += (uint32_t)(__builtin_popcountll(m0) + __builtin_popcountll(m1)); // documents of the intersection
                    }
                    if constexpr (FREQS) and_fsum += (unsigned long long)(ok0 ? fs0 : 0u) + (unsigned long long)(ok1 ? fs1 : 0u);
                    // pa0 / pa1 are complete scores of documents of the intersection now
                    uint32_t inserted = 0;
                    for (int half = 0; !AND && half < 2; ++half) {
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
                    rs_settle_vm(); // (stage C is over: its loads are settled for the compiler too)
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
                const bool al0 = (dB0 != 0xFFFFFFFFu) & enters((boB0 + B.rest) * BOUND_SLACK), al1 = (dB1 != 0xFFFFFFFFu) & enters((boB1 + B.rest) * BOUND_SLACK);
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
        if constexpr (AND) { // and_query returns the size of the intersection (queries.hpp:85); the parts of a split query are summed by k_merge
            if constexpr (FREQS) for (int o = 32; o; o >>= 1) and_fsum += __shfl_xor(and_fsum, o);
            if (lane == 0) {
                if (whole) { r->out_count[q] = and_count; if (r->out_freq_sum) r->out_freq_sum[q] = and_fsum; }
                else { r->unit_count[uid] = and_count; r->unit_freq_sum[uid] = and_fsum; }
            }
        } else if (whole) {
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
#ifdef DS2I_RS_BIGK_TU
// ranked_and with 64 < k <= 1024 (compiled as a translation unit of its own: -DDS2I_RS_BIGK_TU, ds2i_amd/build.py): k <= 256 keeps four
// scores per lane, beyond that sixteen; cap as below, one-term queries ride in the capacity-4 launch
hipError_t ds2i_launch_ranked_stream_bigk(int cap, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    const bool st = a.stats != nullptr;
#define DS2I_RSK_CASE(N) case N: \
        if (a.k <= 256) { if (st) hipLaunchKernelGGL((k_ranked_stream<N, true, false, false, 4>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<N, false, false, false, 4>), g, b, 0, s, a); } \
        else { if (st) hipLaunchKernelGGL((k_ranked_stream<N, true, false, false, 16>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<N, false, false, false, 16>), g, b, 0, s, a); } \
        break;
    switch (cap) {
    DS2I_RSK_CASE(2) DS2I_RSK_CASE(4) DS2I_RSK_CASE(6) DS2I_RSK_CASE(8) DS2I_RSK_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef DS2I_RSK_CASE
    return hipGetLastError();
}
#else
// cap = list capacity of the launch (2, 4, 6, 8, 16): every query of it has cap - 1 or cap (16: 9 .. 16) distinct terms (UnitRec::pad = the count; the
// planner's DS2I_STREAM_NT_MAX caps it); the caller has checked that the index is block_optpfor with skip table, block weights, range
// tables and side slots, and that k <= 64
hipError_t ds2i_launch_ranked_stream(int cap, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    const bool st = a.stats != nullptr;
#define DS2I_RS_CASE(N) case N: if (st) hipLaunchKernelGGL((k_ranked_stream<N, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<N, false>), g, b, 0, s, a); break;
    switch (cap) {
    DS2I_RS_CASE(2) DS2I_RS_CASE(4) DS2I_RS_CASE(6) DS2I_RS_CASE(8) DS2I_RS_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef DS2I_RS_CASE
    return hipGetLastError();
}
#endif
#ifndef DS2I_RS_BIGK_TU
// and_query (counts; with_freqs: counts + the freq checksum) through the same pipeline (k_ranked_stream<cap, ., AND = true, FREQS>); same preconditions
hipError_t ds2i_launch_and_rstream(int cap, int with_freqs, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    const bool st = a.stats != nullptr;
#define DS2I_AND_CASE(N) case N: \
        if (with_freqs) { if (st) hipLaunchKernelGGL((k_ranked_stream<N, true, true, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<N, false, true, true>), g, b, 0, s, a); } \
        else { if (st) hipLaunchKernelGGL((k_ranked_stream<N, true, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_ranked_stream<N, false, true>), g, b, 0, s, a); } \
        break;
    switch (cap) {
    DS2I_AND_CASE(2) DS2I_AND_CASE(4) DS2I_AND_CASE(6) DS2I_AND_CASE(8) DS2I_AND_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef DS2I_AND_CASE
    return hipGetLastError();
}
#endif
}
