// wand / maxscore / ranked_or on a block_optpfor index with the upload-time tables, as a software-pipelined stream
// (gfx950 / CDNA4, wave64; one wavefront per work unit, wave-uniform control flow, no MFMA: integer work).
//
// The three operators return the k best scores of the UNION of the query's lists (reference queries.hpp:200-319 wand,
// 404-476 ranked_or, 478-591 maxscore); what differs in the reference is only how they avoid scoring everything. The
// decomposition is k_union_topk's (kernels.hip), which stays the kernel of every other case (other codecs, no tables, 9+
// lists): the lists of a query are ordered by decreasing max score and a document BELONGS to the first list of that order
// that holds it. A unit streams a block range of one list -- its driver, slot 0 of a "virtual query" -- and meets two kinds
// of other lists:
//   * slots 1 .. nexcl: the lists of higher max score are EXCLUSIONS -- a candidate found there is that list's;
//   * slots nexcl+1 .. NT-1: the lists of lower max score are OPTIONAL -- a member adds its term score.
// A document owned by list e occurs in no list of higher max score, so it scores at most S_e = the max scores from e down:
// once the threshold passes S_e the units of list e end at once (MaxScore's essential / non-essential split,
// queries.hpp:529-574, evaluated per unit). Every document's score is computed by exactly one unit, as the float32 sum of its
// term scores in the fixed order driver, optional lists by decreasing max score: wand == maxscore == ranked_or bit for bit.
//
// What this file changes is how a unit is EXECUTED -- k_ranked_stream's pipeline (ranked_stream.hip) instead of one block
// per wave with compiler-waited gathers:
//     stage N (block i+2)  chosen from the table window (block weight + the optional lists' span maxima against the heap
//                          threshold), its bytes and its exception side slot requested (LDS-DMA)
//     stage A (block i+1)  docs AND freqs decoded in one branch-free pass; every posting gets a bound of its own term score
//                          (its freq at the collection's shortest document, capped by the block's weight)
//     stage B (block i)    list 1's byte of every candidate still alive -- fetched an iteration ahead by LDS-DMA: the HINT where
//                          list 1 is an exclusion, the WEIGHT where it is optional -- is consumed; the further lists' bytes are
//                          requested for the survivors only. An exclusion list's hint settles membership exactly wherever a range
//                          holds one posting and is at most 128 doc-ids wide: such a candidate is dropped or cleared without
//                          list j ever being searched or decoded
//     stage C (block i)    only if somebody survived: norm_len, exact driver score, then slots 1 .. NT-1 in order (exclusions:
//                          locate block -> decode -> membership; optional: locate -> block-weight test -> decode -> membership
//                          -> score), heap insert
// Every pruning test is a true upper bound of the float32 score the scoring code would compute (device_score.hpp,
// BOUND_SLACK) and topk_queue::insert is strict, so the parts of a query end with the same multiset of scores whatever
// was pruned (tests/test_gpu.py: wand == maxscore == ranked_or, <= 1e-5 against the oracle's reference-order traversal).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_enum.hpp"
#include "device_score.hpp"
#include "stream_common.hpp"

using namespace ds2i_dev;
using namespace ds2i_dev::stream;

namespace {

// NT = list CAPACITY of an instantiation (2, 4, 6, 8, 16); a unit's virtual query has nt <= NT lists (UnitRec::pad). One launch per
// capacity: 2 | 3-4 | 5-6 | 7-8 | 9-16 lists -- five launch groups of a batch, each with one tail.
// 2 lists: 6 waves per SIMD; 3..4: 5; beyond: what the LDS of one decoded block per list leaves (16 lists: 20 KB per wave)
#define US_WAVES(NT) ((NT) <= 2 ? 6 : (NT) <= 4 ? 5 : (NT) <= 6 ? 4 : (NT) <= 8 ? 3 : 2)

template <int NT>
struct LdsUS {
    uint32_t stage[3][STAGE_DW]; // driver: bytes of the blocks in stage B/C, in stage A and on their way in (LDS-DMA)
    uint32_t xs[3][XSLOT_DW];    // their exception side slots (LDS-DMA, with the bytes)
    uint32_t gb[2][64];          // list 1's range-table byte of every posting of the block in stage B (LDS-DMA, one dword per lane)
    uint32_t stb[STAGE_DW];      // stage C: bytes of the block of list j being decoded
    uint32_t xsb[XSLOT_DW];      // its side slot
    uint32_t dj[NT - 1][128];    // lists 1 .. NT-1: doc-ids of their current block
    uint32_t fj[NT - 1][128];    // and its freqs
    uint32_t fw[64];             // the query's shared floor word, as fetched an iteration ago (LDS-DMA: one copy per lane)
    uint32_t lst[NT - 1][16];    // lists 1 .. NT-1: what stage C needs of them (LS_*), read where a candidate gets that far
};
// (stage C is ONE loop body over the lists, whatever NT is: unrolled per list the 8-list kernel was 80 KB of code -- more than the
// instruction cache two CUs share -- and ran 63 queries in the time the 5-list kernel ran 334)
enum { LS_CUR = 0, LS_BMAX, LS_N, LS_BB, LS_LOLO, LS_LOHI, LS_QW, LS_TLLO, LS_TLHI, LS_RSC };

// NK > 1 (k > 64; topk_queue has no limit, queries.hpp:152-197): NK scores per lane (TopKBig<NK>: k <= 64 NK), fewer waves per SIMD
#define US_WAVES_K(NT, NK) ((NK) == 1 ? US_WAVES(NT) : (US_WAVES(NT) < ((NK) <= 4 ? 4 : 3) ? US_WAVES(NT) : ((NK) <= 4 ? 4 : 3)))
template <int NT, bool STATS, int NK = 1>
__global__ void __launch_bounds__(64, US_WAVES_K(NT, NK)) k_union_stream(BatchArgs a_unused) {
    static_assert(NT >= 2 && NT <= 16, "list capacities 2..16");
    __shared__ LdsUS<NT> L;
    const uint32_t lane = lane_id();
    typename std::conditional<STATS, uint32_t, NullCounter>::type s_docs_blocks, s_freqs_blocks, s_bm_examined, s_scored, s_rounds;
    typename std::conditional<STATS, unsigned long long, NullCounter>::type s_bytes;
    s_docs_blocks = s_freqs_blocks = s_bm_examined = s_scored = s_rounds = 0;
    s_bytes = 0;
#ifdef DS2I_US_PHASE
    // diagnostic build (-DDS2I_US_PHASE, instrumented runs): shader cycles of a wave by where it spends them and a few event counts,
    // reported through Stats::phase_cycles (profiles/probes/us_phase_probe.py). PT(slot) closes the interval since the previous PT.
    unsigned long long pt[PH_COUNT] = {};
    unsigned long long pt_prev = __builtin_readcyclecounter();
#define PT(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); pt[slot] += t_ - pt_prev; pt_prev = t_; } while (0)
#define EV(slot, n) pt[slot] += (n)
#else
#define PT(slot) ((void)0)
#define EV(slot, n) ((void)0)
#endif
    const uint32_t nslice = rs_args()->nslice;
    for (uint32_t tkt = blockIdx.x; tkt < nslice; tkt += gridDim.x) {
        KArgs a = rs_args(); // (fields read below stay live for the unit; the cold ones are re-read at their use site)
        const UnitRec u = a->urec[tkt]; // (one 32-byte record: the unit, its virtual query's terms, its real query, its histogram)
        const uint32_t uid = uniform(u.uid);
        if constexpr (STATS) { // diagnostic (DS2I_UNIT_CLOCK=1): when the unit started / ended
            unsigned long long* const clk = a->unit_clock;
            if (clk && lane == 0) clk[2ull * uid] = wall_clock64();
        }
        const uint32_t q = uniform(u.q), blk_begin = uniform(u.blk_begin), blk_end = uniform(u.blk_end);
        const uint32_t nexcl = uniform(u.pad) & 255u; // slots 1 .. nexcl are exclusions, the rest optional
        const uint32_t nt = NT == 2 ? 2u : uniform(u.pad) >> 8; // lists of this virtual query (2 .. NT)
        const bool whole = uniform(u.nparts) == 1u;
        const QTerm* const qt = rs_uniform_ptr(a->qterms + uniform(u.qt_off)); // nt terms
        typename std::conditional<NK == 1, TopK, TopKBig<NK>>::type tk;
        tk.init(a->k);
        // ---- list 0: the driver
        const uint32_t n0 = uniform(qt[0].n), nb0 = (n0 + 127u) >> 7;
        const uint32_t vl0 = 1u + (n0 >= (1u << 7)) + (n0 >= (1u << 14)) + (n0 >= (1u << 21)) + (n0 >= (1u << 28));
        const uint32_t bb0 = uniform(qt[0].blk_base);
        const uint8_t* const data0 = a->arena + rs_uniform64(qt[0].list_off) + vl0 + 4ull * nb0 + 4ull * (nb0 - 1);
        const float qw0 = rs_uniformf(qt[0].q_weight);
        const uint32_t* const xs0 = a->xslots + (size_t)XSLOT_DW * bb0; // the driver's side slots
        // ---- lists 1 .. NT-1: range table (hot), the rest of the QTerm is parked for stage C
        const uint8_t* rt[NT];
        uint32_t rsh[NT];
        float rsc[NT];
        auto bind_one = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if ((uint32_t)j < nt) {
                rt[j] = a->rmw + 64ull * uniform(qt[j].rmw_off64);
                rsh[j] = uniform(qt[j].rmw_shift);
                rsc[j] = rs_uniformf(qt[j].rmw_scale);
            } else { // no such list: its byte is never loaded and counts as zero
                rt[j] = a->rmw;
                rsh[j] = 31u;
                rsc[j] = 0.f;
            }
        };
        rs_for<1, NT>(bind_one);
        // range-table weight bytes of a lane's two candidates, packed: byte j - 1 = list j (v_cvt_f32_ubyteN unpacks for free); one
        // dword holds lists 1..4, six and more lists take a pair, ten and more a quad. The bytes of EXCLUSION lists stay zero in these
        // words, so that the sums below run over every list and add exactly what the optional ones can add.
        using GP = typename std::conditional<(NT > 9), unsigned __int128, typename std::conditional<(NT > 5), unsigned long long, uint32_t>::type>::type;
        auto gbyte = [](GP g, uint32_t j) __attribute__((always_inline)) -> uint32_t { return (uint32_t)(g >> (8u * (j - 1u))) & 255u; };
        // what the lists after list `after` can add to a candidate, from their bytes (summed from the last list down, so that the
        // value for `after` is a prefix of the same chain whatever `after` is)
        auto rest_of = [&](GP g, uint32_t after) __attribute__((always_inline)) -> float {
            float r = 0.f;
            auto add_one = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if ((uint32_t)j > after) r = r + rsc[j] * (float)gbyte(g, (uint32_t)j);
            };
            rs_for_down<NT, 1>(add_one);
            return r;
        };
        GP g_ff = 0; // byte 255 for every optional list: their list maxima
        {
            auto mark = [&](auto jc) __attribute__((always_inline)) { constexpr int j = decltype(jc)::value; if ((uint32_t)j > nexcl && (uint32_t)j < nt) g_ff |= (GP)255u << (8 * (j - 1)); };
            rs_for<1, NT>(mark);
        }
        const float min_nl = a->min_norm_len;
        const long long hdelta = a->rmh ? (long long)(a->rmh - a->rmw) : 0ll; // hint of an entry = the byte at the same offset of the parallel buffer
        // the byte fetched ahead for every candidate: list 1's HINT where list 1 is an exclusion (the hint answers "is it theirs"),
        // its WEIGHT where it is optional (the weight bounds what it can add; the hint is asked for the survivors only)
        const bool ex1 = nexcl >= 1u;
        const uint8_t* const gt1 = (ex1 && hdelta != 0) ? rt[1] + hdelta : rt[1];
        // block of list j whose doc-ids and freqs are in L.dj[j-1] / L.fj[j-1] (cur = ~0: none; ~0 - 1: the list is exhausted) and its
        // block_max, together with the list's geometry: L.lst[j-1], written once per unit, read by stage C only
        constexpr uint32_t EXHAUSTED = 0xFFFFFFFEu;
        {
            auto park = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                if ((uint32_t)j < nt && lane == 0) {
                    const unsigned long long lo = qt[j].list_off, tl = qt[j].aux1;
                    uint32_t* const ls = L.lst[j - 1];
                    ls[LS_CUR] = 0xFFFFFFFFu;
                    ls[LS_BMAX] = 0u;
                    ls[LS_N] = qt[j].n;
                    ls[LS_BB] = qt[j].blk_base;
                    ls[LS_LOLO] = (uint32_t)lo;
                    ls[LS_LOHI] = (uint32_t)(lo >> 32);
                    ls[LS_QW] = __float_as_uint(qt[j].q_weight);
                    ls[LS_TLLO] = (uint32_t)tl;
                    ls[LS_TLHI] = (uint32_t)(tl >> 32);
                    ls[LS_RSC] = __float_as_uint(qt[j].rmw_scale);
                }
            };
            rs_for<1, NT>(park);
            wave_sync();
        }
        // ---- floors: all lower bounds of the final k-th score of the union
        {   // some term has k blocks whose best posting alone reaches floor1 (host, from the upload-time block weights)
            const float f1 = rs_uniformf(qt[0].floor1) * (1.0f - 1.0e-5f);
            if (f1 > tk.floor) tk.floor = f1;
            const float* const seed = a->seed_topk; // the ranked_and pass over (a sub-query of) the same query, relaxed for re-association
            if (seed && uniform(a->seed_len[q]) >= tk.k) {
                const float kth = rs_uniformf(seed[(size_t)q * tk.k + tk.k - 1]) * (1.0f - 1.0e-5f);
                if (kth > tk.floor) tk.floor = kth;
            }
        }
        // the parts of a query (every driving list's units) share a score histogram; its scale = the query's score bound, which the
        // planner leaves in the driver's max_weight (QTerm; the union kernels do not read that field otherwise)
        unsigned int* const q_hist = a->q_hist;
        const bool shared_floor = !whole && q_hist && a->q_floor;
        ScoreHist sh;
        sh.init(shared_floor ? q_hist : nullptr, shared_floor ? uniform(u.hist_slot) : 0u, shared_floor ? rs_uniformf(qt[0].max_weight) : 0.f, 1.0f - 1.0f / 1048576.0f);
        // can a score enter the heap: s >= floor && (heap not full || s > k-th score) (TopK::would_enter), branch-free on two
        // wave-uniform values that are refreshed whenever the heap or the floor changes
        float e_floor = tk.floor, e_gt = -__builtin_inff();
        auto refresh = [&]() __attribute__((always_inline)) { e_floor = tk.floor; e_gt = tk.n < tk.k ? -__builtin_inff() : tk.thr; };
        auto enters = [&](float s) __attribute__((always_inline)) -> bool { return (s >= e_floor) & (s > e_gt); };
        // the floor the histogram implies is published in one word per split query (BatchArgs::q_floor): whoever puts a score into
        // its heap re-reads the histogram and raises the word; everybody else gets the word with every block, an iteration ahead
        unsigned int* const fwp = shared_floor ? (unsigned int*)rs_uniform_ptr(a->q_floor + uniform(u.hist_slot)) : nullptr;
        auto adopt_word = [&](uint32_t bits) __attribute__((always_inline)) {
            const float f = __uint_as_float(bits);
            if (bits != 0u && f > tk.floor) { tk.floor = f; refresh(); }
        };
        if (shared_floor) adopt_word(uniform(__hip_atomic_load(fwp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        // S_e: a document owned by the driver is in no list of higher max score
        const float s_e = rs_uniformf(qt[0].max_bmw + qt[0].suf_bmw) * BOUND_SLACK;
        // ---- the 64-row window of the driver's table (lane j: row s_first + j; lane 0 is the row before the first block the window
        // can serve, unless that is block 0): the block's weight and what a document of that block can score at most = block weight +
        // for each OPTIONAL list the largest range-table entry over the block's own doc-id span; -1 = no such row
        // ... kept per row as well: the optional lists' part of that bound (what any posting of the block can gain at most -- the
        // candidates' first test uses it where a list maximum would let nearly everybody through: 73-114 of 128 candidates of a 3-8-list
        // block asked every table, 1-2 survived their bytes) and one bit per list: some posting of list j lies in the block's doc-id
        // span at all (a clear bit: nobody of this block is in list j -- its table is not asked, for an exclusion list the verdict is in)
        uint32_t s_first = 0, s_valid = 0, s_mask = 0;
        GP s_g = 0; // ... and, three and more lists: the OPTIONAL lists' span maxima themselves, byte j - 1 = list j (0: exclusion list / nobody there)
        uint2 s_e2 = make_uint2(0xFFFFFFFFu, 0u);
        float s_ub = -1.f, s_w = 0.f, s_rest = 0.f;
        auto s_fill = [&](uint32_t first) __attribute__((always_inline)) {
            s_first = first;
            const uint32_t idx = first + lane;
            s_e2 = make_uint2(0xFFFFFFFFu, 0u);
            s_w = 0.f;
            {   // (once per 63 blocks: the table pointers are re-derived here rather than carried through the loop)
                const uint2* const tab0 = (const uint2*)rs_args()->skip + bb0;
                const float* const w0tab = rs_args()->bmw + bb0;
                if (idx < blk_end) { s_e2 = tab0[idx]; s_w = w0tab[idx]; }
            }
            const uint32_t prev_max = (uint32_t)__shfl_up((int)s_e2.x, 1);
            const uint32_t base = (lane == 0) ? 0u : prev_max + 1u, top = s_e2.x;
            const bool row = idx < blk_end && (lane > 0 || idx == 0) && top != 0xFFFFFFFFu && base <= top;
            const uint32_t b2 = row ? base : 0u, t2 = row ? top : 0u; // branch-free: a lane without a row reads entry 0 and discards it
            float acc = 0.f;
            uint32_t mask = 0;
            GP gmax = 0;
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
                const uint32_t best = (row && fits) ? m : 255u; // (255 = the list maximum)
                mask |= best != 0u ? 1u << j : 0u;
                if ((uint32_t)j > nexcl) { // (an exclusion list adds nothing)
                    acc = acc + rsc[j] * (float)best;
                    if constexpr (NT > 2) gmax |= (GP)best << (8 * (j - 1));
                }
            };
            rs_for_down<NT, 1>(one_list);
            s_rest = acc;
            s_mask = mask;
            s_g = gmax;
            s_ub = row ? (qw0 * s_w + acc) * BOUND_SLACK : -1.0f; // (scores are >= 0: -1 never enters)
            rs_settle_vm(); // (once per 63 blocks: no compiler-visible load stays "possibly pending" on the hot path, stream_common.hpp)
        };
        // a block of the driver on its way through the stages (w = its block weight x the driver's query weight)
        struct Blk { uint32_t blk, base, ep, mask; float w, rest; GP g; };
        auto bcast_gp = [](GP v, uint32_t src) __attribute__((always_inline)) -> GP {
            GP r = 0;
            for (int i = 0; i < (int)(sizeof(GP) / 4); ++i) r |= (GP)bcast((uint32_t)(v >> (32 * i)), src) << (32 * i);
            return r;
        };
        auto select = [&](uint32_t from, Blk& o) __attribute__((always_inline)) -> uint32_t { // first block >= from worth a visit
            for (;;) {
                if (from >= blk_end) return 0u;
                const uint32_t idx = s_first + lane;
                const uint64_t hit = ballot((idx >= from) & (s_ub >= 0.f) & enters(s_ub));
                if (__builtin_expect(hit != 0, 1)) {
                    const uint32_t f = (uint32_t)__builtin_ctzll(hit), fp = f ? f - 1 : 0;
                    o.blk = s_first + f;
                    o.base = o.blk ? bcast(s_e2.x, fp) + 1u : 0u;
                    o.ep = o.blk ? bcast(s_e2.y, fp) : 0u;
                    o.w = qw0 * __uint_as_float(bcast(__float_as_uint(s_w), f));
                    o.rest = __uint_as_float(bcast(__float_as_uint(s_rest), f));
                    o.mask = bcast(s_mask, f);
                    if constexpr (NT > 2) o.g = bcast_gp(s_g, f);
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
        uint32_t haveA = 0, haveB = 0, from = blk_begin;
        uint32_t dA0 = 0xFFFFFFFFu, dA1 = 0xFFFFFFFFu, dB0 = 0xFFFFFFFFu, dB1 = 0xFFFFFFFFu; // doc-ids (value lane, lane + 64)
        // (a block's freqs are needed once more only if stage C scores it: they wait in the block's staging buffer, whose bytes
        // are dead once decoded, instead of in four registers)
        float boA0 = 0.f, boA1 = 0.f, boB0 = 0.f, boB1 = 0.f; // bound of the posting's own term score
        const uint32_t st_base = rs_lds_offset(&L.stage[0][0]), gb_base = rs_lds_offset(&L.gb[0][0]), xs_base = rs_lds_offset(&L.xs[0][0]);
        const uint32_t fw_base = rs_lds_offset(&L.fw[0]);
        const uint32_t voff = lane * 4u;
        uint32_t bufB = 0, bufA = 1, bufN = 2;
        if (!enters(s_e)) from = blk_end; // the driver is non-essential from the start
        PT(PH_UNIT);
        for (;;) {
            // ---------------- stage N: the next block worth a visit as things stand now, its bytes and side slot requested
            // A's bytes (and the floor word) were requested an iteration ago; the only loads issued after them are B's two gathers
            if (haveA) {
                ++s_rounds;
                if (haveB) rs_wait_vm<2>(); else rs_wait_vm<0>();
                PT(PH_PREFETCH);
                if (shared_floor) adopt_word(uniform(L.fw[0]));
            }
            Blk N{};
            if (!enters(s_e)) from = blk_end; // the threshold passed S_e: nothing this driver still owns can enter
            const uint32_t haveN = select(from, N);
            if (!haveN) from = blk_end; // (the threshold only rises: what is not worth a visit now never will be)
            if (haveN) {
                from = N.blk + 1u;
                const uint8_t* const g = rs_uniform_ptr(data0 + N.ep); // (full blocks of a block_optpfor list are dword aligned; a partial last block is not read from here)
                rs_prefetch_blk((const uint8_t*)((uintptr_t)g & ~(uintptr_t)3), st_base + bufN * (STAGE_DW * 4u), rs_uniform_ptr(xs0 + (size_t)XSLOT_DW * N.blk),
                                xs_base + bufN * (XSLOT_DW * 4u), voff);
#ifdef DS2I_US_FW_SC1
                if (shared_floor) rs_fetch_word(fwp, fw_base);
#else
                if (shared_floor) rs_fetch_word_cached(fwp, fw_base);
#endif
            }
            PT(PH_STREAM);
            if (haveA) {
                // ---------------- stage A: docs and freqs of block A; every posting gets a bound of its OWN term score
                const uint32_t szA = ((A.blk + 1u) * 128u <= n0) ? 128u : (n0 & 127u);
                uint32_t v0, v1, fv0, fv1, consA, consF;
                if (__builtin_expect(szA == 128u, 1)) rs_decode_full(L.stage[bufA], L.xs[bufA], data0 + A.ep, rs_args()->xovf, v0, v1, fv0, fv1, consA, consF);
                else rs_tail(rs_args()->tails, rs_uniform64(qt[0].aux1), szA, v0, v1, fv0, fv1, consA, consF);
                PT(PH_PROLOG);
                const uint32_t g0 = (lane < szA) ? v0 + 1u : 0u, g1 = (lane + 64 < szA) ? v1 + 1u : 0u;
                const uint32_t i0 = wave_incl_scan(g0);
                const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
                dA0 = (lane < szA) ? A.base + i0 - 1u : 0xFFFFFFFFu;
                dA1 = (lane + 64 < szA) ? A.base + i1 - 1u : 0xFFFFFFFFu;
                const float o0 = qw0 * rs_dtw_bound(fv0 + 1u, min_nl), o1 = qw0 * rs_dtw_bound(fv1 + 1u, min_nl);
                boA0 = o0 < A.w ? o0 : A.w; // (the block's weight is the maximum of its postings' exact term weights)
                boA1 = o1 < A.w ? o1 : A.w;
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
                EV(PH_C_GBLOCKS, 1);
                const uint32_t x0 = L.gb[0][lane], x1 = L.gb[1][lane];
                // (the threshold only rises: a candidate alive now was alive when the gathers were issued, so its byte is there)
                bool ok0 = (dB0 != 0xFFFFFFFFu) & enters((boB0 + B.rest) * BOUND_SLACK);
                bool ok1 = (dB1 != 0xFFFFFFFFu) & enters((boB1 + B.rest) * BOUND_SLACK);
                EV(PH_C_ALIVE, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                GP gP0 = 0, gP1 = 0;          // weight bytes of the optional lists
                uint32_t need0 = 0, need1 = 0; // bit j: exclusion list j has to be searched for this candidate
                // what an exclusion list's byte says about a candidate (hint: 0 = no posting in its range, 255 = several, else the code
                // of the ONE posting -- at most 128 doc-ids wide the code is proof; without hints the weight byte: 0 = no posting)
                auto excl_byte = [&](uint32_t x, uint32_t d, uint32_t sh, bool& ok, uint32_t& need, uint32_t bit) __attribute__((always_inline)) {
                    if (hdelta != 0) {
                        const bool same = x == rmh_code(d, sh), sure = sh <= 7u;
                        need |= ((x == 255u) | (same & !sure)) ? bit : 0u;
                        ok = ok & !(same & sure);
                    } else {
                        const bool exact = sh == 0u;
                        need |= ((x != 0u) & !exact) ? bit : 0u;
                        ok = ok & !((x != 0u) & exact);
                    }
                };
                // Three and more lists: a candidate's word starts from the block's own span maxima (what list j can add to ANY posting of
                // this block) and every byte is replaced by the candidate's own as it arrives; the bound is re-tested after list 1 and
                // after every round of further lists, and a round asks only those still alive. (Round 5/6 asked every further list about
                // every candidate list 1 let through, against list MAXIMA: 37 GB of the wand step's 54 GB of lines went to the 245
                // queries of five and more lists, one 128-byte line per candidate and list.) Every intermediate bound is >= the final
                // one term by term (same summation chain, larger bytes), so nobody the final test admits is dropped on the way.
                if constexpr (NT > 2) { gP0 = B.g; gP1 = B.g; }
                const bool in1 = (B.mask & 2u) != 0u; // (list 1 has a posting in this block's span: its bytes were fetched)
                if (!in1) {
                    // nobody of this block is in list 1: not excluded by it, nothing added by it (its byte of B.g is zero)
                } else if (ex1) {
                    excl_byte(x0, dB0, rsh[1], ok0, need0, 2u);
                    excl_byte(x1, dB1, rsh[1], ok1, need1, 2u);
                } else if constexpr (NT > 2) {
                    gP0 = (gP0 & ~(GP)255u) | (GP)x0;
                    gP1 = (gP1 & ~(GP)255u) | (GP)x1;
                    ok0 = ok0 & enters((boB0 + rest_of(gP0, 0)) * BOUND_SLACK);
                    ok1 = ok1 & enters((boB1 + rest_of(gP1, 0)) * BOUND_SLACK);
                } else {
                    gP0 = x0;
                    gP1 = x1;
                }
                if constexpr (NT > 2) {
                    auto round = [&](auto ja_c, auto jb_c) __attribute__((always_inline)) { // lists [JA, JB): requested together, then tested
                        constexpr int JA = decltype(ja_c)::value, JB = decltype(jb_c)::value;
                        if ((uint32_t)JA >= nt) return;
                        if (!(ballot(ok0) | ballot(ok1))) return;
                        uint32_t y0[JB] = {}, y1[JB] = {};
                        auto load_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            const uint8_t* const tj = ((uint32_t)j <= nexcl && hdelta != 0) ? rt[j] + hdelta : rt[j];
                            const bool inj = ((B.mask >> j) & 1u) != 0u; // (a list without a posting in the block's span is not asked: its bytes are zero)
                            y0[j] = (ok0 & inj) ? (uint32_t)tj[dB0 >> rsh[j]] : 0u;
                            y1[j] = (ok1 & inj) ? (uint32_t)tj[dB1 >> rsh[j]] : 0u;
                        };
                        rs_for<JA, JB>(load_one);
                        auto use_one = [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            if ((uint32_t)j <= nexcl) {
                                excl_byte(y0[j], dB0, rsh[j], ok0, need0, 1u << j);
                                excl_byte(y1[j], dB1, rsh[j], ok1, need1, 1u << j);
                            } else {
                                const GP keep = ~((GP)255u << (8 * (j - 1)));
                                gP0 = (gP0 & keep) | ((GP)y0[j] << (8 * (j - 1)));
                                gP1 = (gP1 & keep) | ((GP)y1[j] << (8 * (j - 1)));
                            }
                        };
                        rs_for<JA, JB>(use_one);
                        rs_settle_vm();
                        ok0 = ok0 & enters((boB0 + rest_of(gP0, 0)) * BOUND_SLACK);
                        ok1 = ok1 & enters((boB1 + rest_of(gP1, 0)) * BOUND_SLACK);
                    };
                    using std::integral_constant;
                    round(integral_constant<int, 2>{}, integral_constant<int, (NT < 4 ? NT : 4)>{});
                    if constexpr (NT > 4) round(integral_constant<int, 4>{}, integral_constant<int, (NT < 6 ? NT : 6)>{});
                    if constexpr (NT > 6) round(integral_constant<int, 6>{}, integral_constant<int, (NT < 8 ? NT : 8)>{});
                    if constexpr (NT > 8) round(integral_constant<int, 8>{}, integral_constant<int, (NT < 12 ? NT : 12)>{});
                    if constexpr (NT > 12) round(integral_constant<int, 12>{}, integral_constant<int, NT>{});
                }
                PT(PH_MEMBER);
                float r0 = rest_of(gP0, 0), r1 = rest_of(gP1, 0);
                ok0 = ok0 & enters((boB0 + r0) * BOUND_SLACK);
                ok1 = ok1 & enters((boB1 + r1) * BOUND_SLACK);
                EV(PH_C_SURV1, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                if (hdelta != 0 && (ballot(ok0) | ballot(ok1)) != 0 && nexcl + 1u < nt) {
                    // membership hints of the OPTIONAL lists: a weight byte only says that SOME posting of list j lies in the candidate's
                    // range; where that range holds exactly one posting its hint says which. A candidate at another offset is not in the
                    // list: its byte is cleared -- no lookup there, nothing added to its bound by that list.
                    uint32_t h0[NT] = {}, h1[NT] = {};
                    auto hload = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        const uint8_t* const ht = rt[j] + hdelta;
                        const bool ask = (uint32_t)j > nexcl && rsh[j] != 0u; // (one doc-id per entry: the weight byte was the answer)
                        h0[j] = (ok0 & ask & (gbyte(gP0, j) != 0u)) ? (uint32_t)ht[dB0 >> rsh[j]] : 255u;
                        h1[j] = (ok1 & ask & (gbyte(gP1, j) != 0u)) ? (uint32_t)ht[dB1 >> rsh[j]] : 255u;
                    };
                    rs_for<1, NT>(hload);
                    auto htest = [&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        const GP keep = ~((GP)255u << (8 * (j - 1)));
                        if (!((h0[j] == 255u) | (h0[j] == rmh_code(dB0, rsh[j])))) gP0 &= keep;
                        if (!((h1[j] == 255u) | (h1[j] == rmh_code(dB1, rsh[j])))) gP1 &= keep;
                    };
                    rs_for<1, NT>(htest);
                    rs_settle_vm();
                    r0 = rest_of(gP0, 0);
                    r1 = rest_of(gP1, 0);
                    ok0 = ok0 & enters((boB0 + r0) * BOUND_SLACK);
                    ok1 = ok1 & enters((boB1 + r1) * BOUND_SLACK);
                }
                PT(PH_FREQS);
                EV(PH_C_SURV2, __builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                if (__builtin_expect((ballot(ok0) | ballot(ok1)) != 0, 0)) {
                    EV(PH_C_LIVEROUNDS, 1);
                    // ---------------- stage C: somebody of block B may enter the heap: norm_len, exact driver score
                    const float* const norm_lens = rs_args()->norm_lens;
                    const uint8_t* const arena = rs_args()->arena;
                    const float nl0 = ok0 ? norm_lens[dB0] : 1.f, nl1 = ok1 ? norm_lens[dB1] : 1.f;
                    const uint32_t fB0 = L.stage[bufB][lane], fB1 = L.stage[bufB][lane + 64];
                    float pa0 = qw0 * doc_term_weight(fB0, nl0), pa1 = qw0 * doc_term_weight(fB1, nl1);
                    {
                        const uint32_t nv = (uint32_t)(__builtin_popcountll(ballot(ok0)) + __builtin_popcountll(ballot(ok1)));
                        s_scored += nv;
                        s_bytes += 4ull * nv;
                    }
                    ok0 = ok0 & enters((pa0 + r0) * BOUND_SLACK);
                    ok1 = ok1 & enters((pa1 + r1) * BOUND_SLACK);
                    PT(PH_SCORE);
                    // slots 1 .. NT-1 in order: exclusions first, then the optional lists by decreasing max score. A list is consulted
                    // only for the candidates its byte left open; a candidate moves on only while partial score + what the later lists
                    // can add to IT can still enter the heap
#pragma nounroll
                    for (uint32_t j = 1; j < nt; ++j) {
                        if (!(ballot(ok0) | ballot(ok1))) break;
                        uint32_t* const ls = L.lst[j - 1];
                        const bool ex = j <= nexcl;
                        const bool w0 = ok0 & (ex ? ((need0 >> j) & 1u) != 0u : gbyte(gP0, j) != 0u);
                        const bool w1 = ok1 & (ex ? ((need1 >> j) & 1u) != 0u : gbyte(gP1, j) != 0u);
                        uint64_t todo0 = ballot(w0), todo1 = ballot(w1);
                        if (!(todo0 | todo1)) continue;
                        const float rj0 = rest_of(gP0, j), rj1 = rest_of(gP1, j); // the lists after j
                        uint32_t curj = uniform(ls[LS_CUR]), bmj = uniform(ls[LS_BMAX]);
                        if (curj != EXHAUSTED) {
                            const uint32_t nj = uniform(ls[LS_N]), nbj = (nj + 127u) >> 7, bbj = uniform(ls[LS_BB]);
                            const uint32_t vlj = 1u + (nj >= (1u << 7)) + (nj >= (1u << 14)) + (nj >= (1u << 21)) + (nj >= (1u << 28));
                            const uint8_t* const dataj = arena + (((unsigned long long)uniform(ls[LS_LOHI]) << 32) | uniform(ls[LS_LOLO])) + vlj + 4ull * nbj + 4ull * (nbj - 1);
                            const uint2* const tabj = (const uint2*)rs_args()->skip + bbj;
                            const float* const wtabj = rs_args()->bmw + bbj;
                            const float qwj = __uint_as_float(uniform(ls[LS_QW])), rscj = __uint_as_float(uniform(ls[LS_RSC]));
                            const float bj0 = rscj * (float)gbyte(gP0, j), bj1 = rscj * (float)gbyte(gP1, j);
                            uint32_t* const dj = L.dj[j - 1];
                            uint32_t* const fj = L.fj[j - 1];
                            while (todo0 | todo1) {
                                const uint32_t amin = todo0 ? bcast(dB0, (uint32_t)__builtin_ctzll(todo0)) : bcast(dB1, (uint32_t)__builtin_ctzll(todo1));
                                if (curj == 0xFFFFFFFFu || amin > bmj) {
                                    Found fb;
                                    const bool found = find_block_rows(tabj, wtabj, nbj, curj + 1u, amin, fb, rows_load(tabj, wtabj, nbj, curj + 1u));
                                    EV(PH_C_VISIT, 1);
                                    if (!found) { // list j has nothing >= amin: nobody left is in it, now or later in this unit
                                        s_bm_examined += 1;
                                        s_bytes += 4;
                                        if (lane == 0) ls[LS_CUR] = EXHAUSTED;
                                        break;
                                    }
                                    s_bm_examined += (curj == 0xFFFFFFFFu) ? 1u : fb.blk - curj;
                                    s_bytes += 4ull * ((curj == 0xFFFFFFFFu) ? 1u : fb.blk - curj);
                                    if (!ex) {
                                        // before the block is decoded: partial + min(block weight, own byte) + later lists, per candidate inside it.
                                        // Nobody can enter even as a member: they are dead either way, the block is not decoded
                                        const float cbw = qwj * fb.w;
                                        const bool in0 = ((todo0 >> lane) & 1) && dB0 <= fb.bmax, in1 = ((todo1 >> lane) & 1) && dB1 <= fb.bmax;
                                        const float t0 = bj0 < cbw ? bj0 : cbw, t1 = bj1 < cbw ? bj1 : cbw;
                                        const bool can0 = in0 & enters(((pa0 + t0) + rj0) * BOUND_SLACK);
                                        const bool can1 = in1 & enters(((pa1 + t1) + rj1) * BOUND_SLACK);
                                        if (!(ballot(can0) | ballot(can1))) {
                                            ok0 = ok0 & !in0;
                                            ok1 = ok1 & !in1;
                                            todo0 &= ~ballot(in0);
                                            todo1 &= ~ballot(in1);
                                            continue; // (the list stays where it was: the next search restarts there)
                                        }
                                    }
                                    const uint8_t* pb = dataj + fb.ep;
                                    const uint32_t szb = ((fb.blk + 1) * 128u <= nj) ? 128u : (nj & 127u);
                                    uint32_t v0, v1, f0, f1, consD, consF2;
                                    if (__builtin_expect(szb == 128u, 1)) { // (full blocks of a block_optpfor list are dword aligned)
                                        rs_stage_block((const uint32_t*)pb, rs_args()->xslots + (size_t)XSLOT_DW * (bbj + fb.blk), L.stb, L.xsb);
                                        rs_decode_full(L.stb, L.xsb, pb, rs_args()->xovf, v0, v1, f0, f1, consD, consF2);
                                    } else {
                                        rs_tail(rs_args()->tails, ((unsigned long long)uniform(ls[LS_TLHI]) << 32) | uniform(ls[LS_TLLO]), szb, v0, v1, f0, f1, consD, consF2);
                                    }
                                    const uint32_t g0 = (lane < szb) ? v0 + 1u : 0u, g1 = (lane + 64 < szb) ? v1 + 1u : 0u;
                                    const uint32_t i0 = wave_incl_scan(g0);
                                    const uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
                                    wave_sync(); // (every lane's reads of the previous block are behind us)
                                    dj[lane] = (lane < szb) ? fb.base + i0 - 1u : 0xFFFFFFFFu;
                                    dj[lane + 64] = (lane + 64 < szb) ? fb.base + i1 - 1u : 0xFFFFFFFFu;
                                    fj[lane] = f0 + 1u;
                                    fj[lane + 64] = f1 + 1u;
                                    wave_sync();
                                    curj = fb.blk;
                                    bmj = fb.bmax;
                                    if (lane == 0) { ls[LS_CUR] = curj; ls[LS_BMAX] = bmj; }
                                    ++s_docs_blocks;
                                    EV(PH_C_BDOCS, 1);
                                    s_bytes += 4 + consD;
                                    if (!ex) { ++s_freqs_blocks; s_bytes += consF2; }
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
                                            const uint64_t hh0 = ballot(e0 == c), hh1 = ballot(e1 == c);
                                            if (hh0 | hh1) {
                                                const uint32_t pp = hh0 ? (uint32_t)__builtin_ctzll(hh0) : 64u + (uint32_t)__builtin_ctzll(hh1);
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
                                if (ex) { // found in a list of higher max score: the document is that list's
                                    ok0 = ok0 & !m0;
                                    ok1 = ok1 & !m1;
                                } else { // members take list j's term score at once
                                    if (m0) pa0 = pa0 + qwj * doc_term_weight(fj[q0], nl0);
                                    if (m1) pa1 = pa1 + qwj * doc_term_weight(fj[q1], nl1);
                                }
                            }
                        }
                        if (!ex) { // who cannot reach the heap any more is not looked up in the lists still to come
                            ok0 = ok0 & enters((pa0 + rj0) * BOUND_SLACK);
                            ok1 = ok1 & enters((pa1 + rj1) * BOUND_SLACK);
                        }
                    }
                    PT(PH_PROBE);
                    // pa0 / pa1 are complete scores of documents this driver owns now
                    uint32_t inserted = 0;
                    for (int half = 0; half < 2; ++half) {
                        const float sc = half ? pa1 : pa0;
                        uint64_t todo = ballot((half ? ok1 : ok0) & enters(sc));
                        while (todo) {
                            const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                            todo &= todo - 1;
                            const float v = __uint_as_float(bcast(__float_as_uint(sc), src));
                            EV(PH_C_HEAP, 1);
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
            PT(PH_INSERT);
            // ---------------- rotate: A becomes B (its gathers are issued now and consumed an iteration later, behind the next
            // block's decode), the block whose bytes were requested becomes A
            B = A;
            haveB = haveA;
            dB0 = dA0;
            dB1 = dA1;
            boB0 = boA0;
            boB1 = boA1;
            if (haveB) {
                // only the candidates whose own bound + the optional lists' maxima can still enter the heap ask list 1's table (the
                // others read entry 0: one shared line); a block without any such candidate is done
                const bool al0 = (dB0 != 0xFFFFFFFFu) & enters((boB0 + B.rest) * BOUND_SLACK), al1 = (dB1 != 0xFFFFFFFFu) & enters((boB1 + B.rest) * BOUND_SLACK);
                haveB = (ballot(al0) | ballot(al1)) != 0 ? 1u : 0u;
                if (haveB) { // (always two loads, so that the counted waits hold: with no posting of list 1 in the block's span everybody reads entry 0)
                    const bool in1 = (B.mask & 2u) != 0u;
                    rs_gather_u8(gt1, ((al0 & in1) ? dB0 : 0u) >> rsh[1], gb_base);
                    rs_gather_u8(gt1, ((al1 & in1) ? dB1 : 0u) >> rsh[1], gb_base + 256u);
                }
            }
            PT(PH_FIND);
            A = N;
            haveA = haveN;
            const uint32_t t = bufB;
            bufB = bufA;
            bufA = bufN;
            bufN = t;
            if (!(haveA | haveB)) break;
        }
        rs_wait_vm<0>();
        PT(PH_TOTAL);
        KArgs r = rs_args();
        if constexpr (STATS) {
            unsigned long long* const clk = r->unit_clock;
            if (clk && lane == 0) clk[2ull * uid + 1] = wall_clock64();
        }
        if (whole) {
            if (lane == 0) r->out_count[q] = tk.n;
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
#ifdef DS2I_US_PHASE
        for (int i = 0; i < PH_COUNT; ++i) if (pt[i]) atomicAdd(&stats->phase_cycles[i], pt[i]);
#endif
    }
#undef PT
#undef EV
}

} // namespace

extern "C" {
#ifdef DS2I_US_BIGK_TU
// wand / maxscore / ranked_or with 64 < k <= 1024 (a translation unit of its own: -DDS2I_US_BIGK_TU, ds2i_amd/build.py): k <= 256 keeps
// four scores per lane, beyond that sixteen; cap as below
hipError_t ds2i_launch_union_stream_bigk(int cap, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    const bool st = a.stats != nullptr;
#define DS2I_USK_CASE(N) case N: \
        if (a.k <= 256) { if (st) hipLaunchKernelGGL((k_union_stream<N, true, 4>), g, b, 0, s, a); else hipLaunchKernelGGL((k_union_stream<N, false, 4>), g, b, 0, s, a); } \
        else { if (st) hipLaunchKernelGGL((k_union_stream<N, true, 16>), g, b, 0, s, a); else hipLaunchKernelGGL((k_union_stream<N, false, 16>), g, b, 0, s, a); } \
        break;
    switch (cap) {
    DS2I_USK_CASE(2) DS2I_USK_CASE(4) DS2I_USK_CASE(6) DS2I_USK_CASE(8) DS2I_USK_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef DS2I_USK_CASE
    return hipGetLastError();
}
#else
// cap = list capacity of the launch (2, 4, 6, 8, 16): every virtual query of it has cap - 1 or cap (16: 9 .. 16) lists (driver + exclusions + optional
// lists; UnitRec::pad = exclusion lists | lists << 8); the caller has checked that the index is block_optpfor with skip table, block
// weights, range tables and side slots, that k <= 64, and has filled BatchArgs::urec and BatchArgs::q_floor
hipError_t ds2i_launch_union_stream(int cap, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    const dim3 g(grid), b(64);
    const bool st = a.stats != nullptr;
#define DS2I_US_CASE(N) case N: if (st) hipLaunchKernelGGL((k_union_stream<N, true>), g, b, 0, s, a); else hipLaunchKernelGGL((k_union_stream<N, false>), g, b, 0, s, a); break;
    switch (cap) {
    DS2I_US_CASE(2) DS2I_US_CASE(4) DS2I_US_CASE(6) DS2I_US_CASE(8) DS2I_US_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef DS2I_US_CASE
    return hipGetLastError();
}
#endif
}
