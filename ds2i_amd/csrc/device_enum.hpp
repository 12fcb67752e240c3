// Wave-cooperative posting-list enumerator over a block_freq_index arena in HBM.
// One wavefront owns up to TMAX list "slots"; each slot keeps its current decoded
// block in LDS as 128 absolute doc-ids (+ 128 freqs, decoded lazily).
//
// Mirrors block_posting_list<>::document_enumerator (reference block_posting_list.hpp:84-354):
//   open()          ctor 86-103 (vbyte n, three pointers, decode block 0)
//   next()          110-122
//   next_geq()      124-146 -- block_max skip = 64-wide ballot scan instead of the linear scan,
//                   in-block search = two ballots over the LDS-resident doc-ids (O(1))
//   freq()          165-171 (lazy decode_freqs_block 321-331)
//   decode_docs()   292-319
// Exhaustion sentinel: docid == num_docs (115,130).
#pragma once
#include <type_traits>

#include "abi_structs.hpp"
#include "device_codecs.hpp"
#include "device_pef.hpp"

namespace ds2i_dev {

enum { M_MAXS_LO = 0, M_MAXS_HI, M_N, M_NB, M_CUR, M_SIZE, M_BMAX, M_POS, M_DOCID, M_FREQ_LO, M_FREQ_HI, M_FDEC,
       M_QW, M_MAXW, M_END_LO, M_END_HI, M_DBIT_LO, M_DBIT_HI, M_FBIT_LO, M_FBIT_HI, M_GPOS, M_NEXTEP, M_HINT, M_PBASE /* blocks (chunks) of all preceding lists: the list's base in the access profile, skip table, bmw[] */,
       M_CBW /* q_weight * bmw of the current block (float bits) */,
       M_SUF /* sum of the later lists' q_weight * list max bmw (float bits) */,
       M_DDEC /* docs of block M_CUR are decoded (the disjunctive kernel positions a list on a block first and decodes it only if the block can matter) */,
       M_EP, M_BASE /* table words of a positioned, not yet decoded block: start offset, first doc-id it can hold */,
       M_RBASE, M_RSHIFT, M_RSCALE /* the list's doc-id-range table (QTerm::rmw_*): offset / 64, doc-ids per entry (log2), byte -> score bound (float bits) */,
       M_WORDS }; // 32 dwords per list slot

#ifdef DS2I_PHASE_TIMING
#define PT_BEGIN(cx) const unsigned long long pt_t0_ = __builtin_readcyclecounter()
#define PT_END(cx, ph) (cx).s_phase[ph] += __builtin_readcyclecounter() - pt_t0_
#else
#define PT_BEGIN(cx) do {} while (0)
#define PT_END(cx, ph) do {} while (0)
#endif

// Where the per-list enumerator state (M_* words) lives.
//  MetaLds: in LDS, one writer lane + wave-uniform reads -- needed when the list slot is a run-time value
//           (document-at-a-time operators walk their lists through an index array);
//  MetaReg: in registers -- the conjunctive kernels with <=4 lists address every slot with a compile-time
//           constant (fully unrolled list loops), so the state stays in SGPRs/VGPRs: no LDS round trip, no
//           v_readfirstlane, and the compiler can CSE the pointer arithmetic.
struct MetaLds {
    static constexpr int NPF = 0; // prefetch slots
    static constexpr bool SKIPTAB = true; // use the interleaved skip table (find_block_info)
    uint32_t* p;
    DS2I_DEV uint32_t get(uint32_t s, int f) const { return uniform(p[s * M_WORDS + f]); }
    DS2I_DEV void set(uint32_t s, int f, uint32_t v) { if (lane_id() == 0) p[s * M_WORDS + f] = v; }
};
template <int TMAX>
struct MetaReg {
    // prefetch slot: list 0 of the <=2-list kernel (the driving list is read sequentially). Round 1 also prefetched list 1;
    // since ranked_and prunes, list 1 is probed for few candidates and rarely sequentially -- its prefetch only cost
    // three VGPRs and wasted loads (same throughput without it, and the kernel no longer needs scratch). None with 3-4
    // lists, where the extra VGPRs cost more occupancy than the prefetch wins (measured).
    static constexpr int NPF = 0; // (superseded by the list-0 stream of k_conjunctive, kernels.hip)
    // the interleaved skip table saves the table round trip of a non-sequential decode; the <=2-list kernel decodes
    // sequentially through its prefetch and cannot afford the extra state (72 SGPRs)
    static constexpr bool SKIPTAB = TMAX > 2;
    // software prefetch of each list's NEXT sequential block (issued right after a block is decoded, consumed by the
    // next decode of that list if it is indeed block+1): table words + 512 B of block bytes, per lane
    uint32_t pf_blk[NPF], pf_tab[NPF], pf_w0[NPF], pf_w1[NPF];
    uint32_t v[TMAX * M_WORDS];
    DS2I_DEV uint32_t get(uint32_t s, int f) const { return v[s * M_WORDS + f]; }
    // (through v_readfirstlane: a word the compiler takes for divergent -- assigned under a branch it could not prove uniform --
    // would live in a VGPR, and a few dozen of those push the unrolled kernels into hundreds of scratch spills; as scalars
    // the same words spill, if they must, into the lanes of a VGPR)
    DS2I_DEV void set(uint32_t s, int f, uint32_t x) { v[s * M_WORDS + f] = uniform(x); }
};

// a counter that compiles to nothing (uninstrumented kernels)
struct NullCounter {
    DS2I_DEV NullCounter& operator+=(unsigned long long) { return *this; }
    DS2I_DEV NullCounter& operator++() { return *this; }
    DS2I_DEV NullCounter& operator=(unsigned long long) { return *this; }
    DS2I_DEV operator unsigned long long() const { return 0; }
};

// SHARE_F: the lists after list 0 decode their freqs into ONE shared buffer (slot 1) instead of one each; `fowner` says
// whose block it holds. For the ranked conjunction, which uses a later list's freqs right where it decodes them: its
// residency is capped by LDS, and 512 B per list beyond the second buys workgroups (kernels.hip, LdsConj).
template <int CODEC_T, class META = MetaLds, bool STATS = true, bool SHARE_F = false>
struct CtxT {
    uint32_t fowner = 0; // SHARE_F: the list (>= 1) whose freqs are in the shared buffer (wave-uniform)
    uint32_t* docs;  // [TMAX][128]
    uint32_t* freqs; // [TMAX][128]
    META meta;       // [TMAX][M_WORDS]
    uint32_t* exc;   // [EXC_DW]
    Window win;
    const uint8_t* arena;
    const uint8_t* bits0; // opt index: docs / freqs bit vectors
    const uint8_t* bits1;
    int codec;
    uint32_t num_docs;
    unsigned int* block_profile; // 2 counters per block (docs, freqs decodes) or null; instrumented kernels only
    const uint2* skip;           // block indexes: {block_max[b], end offset of block b} per block of the index, or null
    // CODEC_T == CODEC_OPTPFOR = block_optpfor WITH the upload-time side tables (BatchArgs::xslots / tails; an upload without
    // them runs the runtime-codec instantiation): a full block is decoded through its exception side slot (device_codecs.hpp,
    // optpfor_decode_pair / optpfor_decode_side), a list's partial last block comes from the tail table -- these kernels carry
    // no Simple16 parser and no interpolative walk. The slot of the block decoded last waits in exc[0 .. 63] (slot_blk = its
    // index-wide block number, ~0 = none). want_freqs: decode_docs() delivers the freqs with the doc-ids (one pass over the
    // staged bytes) instead of leaving them to a later decode_freqs().
    const uint32_t* xslots = nullptr;
    const uint32_t* xovf = nullptr;
    const uint32_t* tails = nullptr;
    uint32_t slot_blk = 0xFFFFFFFFu;
    bool want_freqs = false;
    static constexpr bool SIDE = CODEC_T == CODEC_OPTPFOR;
    DS2I_DEV bool side() const { return SIDE; }
    DS2I_DEV const uint32_t* tail_of(uint32_t s) const { return tails + (((uint64_t)m(s, M_DBIT_HI) << 32) | m(s, M_DBIT_LO)); }
    // the slot of block gb -> exc[0 .. 63]; `have` = its dword of this lane, loaded ahead by the caller (with the block's bytes)
    // (gb, like everything that steers control flow here, is made wave-uniform explicitly: a value the compiler takes for
    // divergent turns the branch into a masked region and every enumerator word assigned under it into a VGPR)
    DS2I_DEV void stage_slot(uint32_t gb, const uint32_t* have = nullptr) {
        gb = uniform(gb);
        if (slot_blk == gb) return;
        const uint32_t v = have ? *have : xslots[(size_t)XSLOT_DW * gb + lane_id()];
        exc[lane_id()] = v;
        slot_blk = gb;
        wave_sync();
    }
    DS2I_DEV bool is_pef() const { return CODEC_T == CODEC_PEF || (CODEC_T < 0 && codec == CODEC_PEF); }
    // per-wave statistics (wave-uniform). Like the reference's block_profiler they are a compile-time option
    // (block_posting_list.hpp:316-318 `if (Profile)`): the counters live in SGPRs, and the <=2-list kernel at
    // 8 waves/SIMD has 72 of them, so the uninstrumented instantiation spills less (+3 % / +15 % queries/s at
    // GOV2-scale / configs[1]).
    typename std::conditional<STATS, uint32_t, NullCounter>::type s_docs_blocks, s_freqs_blocks, s_bm_examined, s_scored, s_rounds;
    typename std::conditional<STATS, unsigned long long, NullCounter>::type s_bytes;
    unsigned long long s_phase[PH_COUNT];

    DS2I_DEV uint32_t* D(uint32_t s) const { return docs + 128 * s; }
    DS2I_DEV uint32_t* F(uint32_t s) const { return freqs + 128 * (SHARE_F ? (s ? 1u : 0u) : s); }
    DS2I_DEV bool freqs_ready(uint32_t s) const { return m(s, M_FDEC) && (!SHARE_F || s == 0 || fowner == s); }
    DS2I_DEV uint32_t m(uint32_t s, int f) const { return meta.get(s, f); }
    DS2I_DEV void setm(uint32_t s, int f, uint32_t v) { meta.set(s, f, v); } // v must be wave-uniform
    DS2I_DEV const uint8_t* ptr(uint32_t s, int lo) const {
        return arena + (((uint64_t)m(s, lo + 1) << 32) | m(s, lo));
    }
    DS2I_DEV uint32_t docid(uint32_t s) const { return m(s, M_DOCID); }
    DS2I_DEV uint32_t size(uint32_t s) const { return m(s, M_N); }

    DS2I_DEV void init_stats() {
        s_docs_blocks = s_freqs_blocks = s_bm_examined = s_scored = s_rounds = 0;
        s_bytes = 0;
        for (int i = 0; i < PH_COUNT; ++i) s_phase[i] = 0;
    }
    DS2I_DEV void flush_stats(Stats* st) {
        if (STATS && st && lane_id() == 0) {
            atomicAdd(&st->docs_blocks, (unsigned long long)s_docs_blocks);
            atomicAdd(&st->freqs_blocks, (unsigned long long)s_freqs_blocks);
            atomicAdd(&st->block_max_examined, (unsigned long long)s_bm_examined);
            atomicAdd(&st->algorithmic_bytes, s_bytes);
            atomicAdd(&st->postings_scored, (unsigned long long)s_scored);
            atomicAdd(&st->rounds, (unsigned long long)s_rounds);
#ifdef DS2I_PHASE_TIMING
            for (int i = 0; i < PH_COUNT; ++i) atomicAdd(&st->phase_cycles[i], s_phase[i]);
#endif
        }
    }

    // ---- decode_docs_block (block_posting_list.hpp:292-319)
    // ---- opt index: chunk b of the list = <=128 elements of one docs partition (device_pef.hpp)
    // `pre`: the chunk's directory entry (lanes 0 .. PC_WORDS-1) and its cmax (lane PC_WORDS), loaded ahead by the caller
    DS2I_DEV void decode_docs_pef(uint32_t s, uint32_t b, const uint32_t* pre = nullptr) {
        const uint32_t lane = lane_id();
        const uint8_t* cmaxp = ptr(s, M_MAXS_LO);
        const uint32_t* ent = (const uint32_t*)(ptr(s, M_END_LO) + (uint64_t)b * (4 * PC_WORDS));
        uint32_t ev = 0;
        if (pre) {
            ev = *pre;
        } else {
            if (lane < PC_WORDS) ev = ent[lane];
            if (lane == PC_WORDS) ev = ((const uint32_t*)cmaxp)[b];
        }
        const uint32_t packed = bcast(ev, PC_PACKED), cnt = packed & 0xFFu;
        const uint32_t bmax = bcast(ev, PC_WORDS);
        const uint64_t bit0 = ((uint64_t)m(s, M_DBIT_HI) << 32) | m(s, M_DBIT_LO);
        uint32_t v0, v1;
        pef_decode_side<false>(bits0, bit0, (packed >> 8) & 3u, (packed >> 10) & 63u, bcast(ev, PC_D_BASE), bcast(ev, PC_D_HI),
                               bcast(ev, PC_D_HBIAS), bcast(ev, PC_D_LO), bcast(ev, PC_SPANS) & 0xFFFFu, cnt, exc, v0, v1);
        uint32_t* dst = D(s);
        const uint32_t d0 = lane < cnt ? v0 : 0xFFFFFFFFu;
        dst[lane] = d0;
        dst[lane + 64] = lane + 64 < cnt ? v1 : 0xFFFFFFFFu;
        const uint32_t gpos = bcast(ev, PC_GPOS), first = bcast(d0, 0);
        setm(s, M_CUR, b);
        setm(s, M_SIZE, cnt);
        setm(s, M_BMAX, bmax);
        setm(s, M_POS, 0);
        setm(s, M_DOCID, first < num_docs ? first : num_docs);
        setm(s, M_FDEC, 0);
        setm(s, M_DDEC, 1);
        setm(s, M_GPOS, gpos);
        wave_sync();
        ++s_docs_blocks;
        const uint32_t l = (packed >> 10) & 63u, span = bcast(ev, PC_SPANS) & 0xFFFFu;
        s_bytes += 4 + ((span + cnt * l + 7) >> 3); // cmax entry + the chunk's high and low bits
    }
    DS2I_DEV void decode_freqs_pef(uint32_t s) {
        const uint32_t lane = lane_id();
        const uint32_t b = m(s, M_CUR);
        const uint32_t* ent = (const uint32_t*)(ptr(s, M_END_LO) + (uint64_t)b * (4 * PC_WORDS));
        uint32_t ev = 0;
        if (lane < PC_WORDS) ev = ent[lane];
        const uint32_t packed = bcast(ev, PC_PACKED), cnt = packed & 0xFFu;
        const uint64_t bit0 = ((uint64_t)m(s, M_FBIT_HI) << 32) | m(s, M_FBIT_LO);
        uint32_t s0, s1;
        pef_decode_side<true>(bits1, bit0, (packed >> 16) & 3u, (packed >> 18) & 63u, bcast(ev, PC_F_BASE), bcast(ev, PC_F_HI),
                              bcast(ev, PC_F_HBIAS), bcast(ev, PC_F_LO), bcast(ev, PC_SPANS) >> 16, cnt, exc, s0, s1);
        // freq_i = S_i - S_{i-1} (positive_sequence.hpp:48-66); S_{-1} of the chunk comes from the directory
        uint32_t p0 = __shfl_up(s0, 1), p1 = __shfl_up(s1, 1);
        const uint32_t fprev = bcast(ev, PC_F_PREV), s0_last = bcast(s0, 63);
        if (lane == 0) { p0 = fprev; p1 = s0_last; }
        uint32_t* dst = F(s);
        dst[lane] = s0 - p0;
        dst[lane + 64] = s1 - p1;
        setm(s, M_FDEC, 1);
        if (SHARE_F && s) fowner = s;
        wave_sync();
        ++s_freqs_blocks;
        s_bytes += ((bcast(ev, PC_SPANS) >> 16) + cnt * ((packed >> 18) & 63u) + 7) >> 3;
    }

    // table words of one block: byte offset of its start / of the next block's start (relative to the blocks area),
    // its block_max and the first doc-id it can hold
    struct BlockInfo { uint32_t ep, next_ep, bmax, base; };

    // `pre`: the block's table words, already known to the caller; `staged`: its bytes are in the staging window too
    DS2I_DEV void decode_docs(uint32_t s, uint32_t b, const BlockInfo* pre = nullptr, bool staged = false) {
        PT_BEGIN(*this);
        if (is_pef()) {
            decode_docs_pef(s, b);
            PT_END(*this, PH_DOCS);
            return;
        }
        const uint32_t lane = lane_id();
        const uint8_t* maxs = ptr(s, M_MAXS_LO);
        const uint32_t n = m(s, M_N), nb = m(s, M_NB);
        const uint8_t* endpoints = maxs + 4ull * nb;
        const uint8_t* data = endpoints + 4ull * (nb - 1);
        const uint8_t* lend = ptr(s, M_END_LO);
        const uint32_t cur = m(s, M_CUR);
        const uint32_t sz = SIDE ? uniform(((b + 1) * 128u <= n) ? 128u : (n & 127u)) : (((b + 1) * 128u <= n) ? 128u : (n & 127u));
        uint32_t ep = 0, bmax = 0, base = 0, next_ep = 0;
        const uint8_t* p = nullptr;
        bool have = false;
        // (side slots: the block's slot is requested here, together with the table words / block bytes below)
        const uint32_t gblk = SIDE ? uniform(m(s, M_PBASE) + b) : 0u;
        bool fetch_slot = false;
        uint32_t slot_dw = 0;
        if constexpr (SIDE) {
            fetch_slot = sz == 128u && slot_blk != gblk;
            if (fetch_slot) slot_dw = xslots[(size_t)XSLOT_DW * gblk + lane];
        }
        if constexpr (META::NPF > 0) {
            if (s < (uint32_t)META::NPF && cur != 0xFFFFFFFFu && meta.pf_blk[s < (uint32_t)META::NPF ? s : 0] == b) { // the bytes are already in registers
                have = true;
                ep = m(s, M_NEXTEP);
                base = m(s, M_BMAX) + 1u;
                p = data + ep;
                win.gbase = (const uint8_t*)((uintptr_t)p & ~(uintptr_t)3);
                win.nbytes = 512;
                win.st[lane] = meta.pf_w0[s];
                win.st[lane + 64] = meta.pf_w1[s];
                wave_sync();
                bmax = bcast(meta.pf_tab[s], 1);
                next_ep = bcast(meta.pf_tab[s], 3);
            }
        }
        if (!have && pre) { // the caller's find_block_info() already fetched the table words with its probe
            have = true;
            ep = pre->ep;
            bmax = pre->bmax;
            base = pre->base;
            next_ep = pre->next_ep;
            uint32_t hint = next_ep - ep + 8u; // (+ the dword a lane may read past the last value)
            if (hint == 8u || hint > STAGE_DW * 4 - 4) hint = STAGE_DW * 4 - 4;
            p = data + ep;
            if (!staged) win.load(p, hint);
        }
        if (!have) {
        // Table words come from ONE unconditional load with a per-lane address (lane 0: endpoint[b-1], 1: block_max[b],
        // 2: block_max[b-1], 3: endpoint[b]); lanes whose word does not exist read block_max[b] and are overridden.
        const uint8_t* a1 = maxs + 4ull * b;
        const uint8_t* taddr = a1;
        if (lane == 0 && b) taddr = endpoints + 4ull * (b - 1);
        if (lane == 2 && b) taddr = maxs + 4ull * (b - 1);
        if (lane == 3 && b + 1 < nb) taddr = endpoints + 4ull * b;
        uint32_t hv = ld32(taddr);
        if (lane == 0 && !b) hv = 0u;
        if (lane == 2) hv = b ? hv + 1u : 0u;
        if (lane == 3 && !(b + 1 < nb)) hv = (uint32_t)(lend - data);
        if (cur != 0xFFFFFFFFu && b == cur + 1) {
            // sequential access: the block starts at the endpoint read with the previous block and its base is the
            // previous block_max + 1, so the table words and the block bytes are fetched in ONE round trip (window
            // size guessed from the previous block)
            ep = m(s, M_NEXTEP);
            base = m(s, M_BMAX) + 1u;
            p = data + ep;
            uint32_t guess = m(s, M_HINT) + 32u;
            if (guess > STAGE_DW * 4 - 4) guess = STAGE_DW * 4 - 4;
            win.load(p, guess);
            bmax = bcast(hv, 1);
            next_ep = bcast(hv, 3);
        } else {
            ep = bcast(hv, 0);
            bmax = bcast(hv, 1);
            base = bcast(hv, 2);
            next_ep = bcast(hv, 3);
            uint32_t hint = next_ep - ep + 8u; // docs+freqs bytes of this block when endpoints are monotone (+ the dword a lane may read past the last value)
            if (hint == 8u || hint > STAGE_DW * 4 - 4) hint = STAGE_DW * 4 - 4;
            p = data + ep;
            win.load(p, hint);
        }
        }
        uint32_t blk_bytes = next_ep - ep;
        if (blk_bytes > STAGE_DW * 4) blk_bytes = STAGE_DW * 4;
        uint32_t v0, v1;
        uint32_t* dst = D(s);
        uint32_t consumed;
        const bool freqs_too = SIDE && want_freqs;
        if constexpr (SIDE) {
            uint32_t f0 = 0, f1 = 0, cf = 0;
            if (__builtin_expect(sz == 128u, 1)) { // (full blocks of a block_optpfor list are dword aligned in the arena)
                stage_slot(gblk, fetch_slot ? &slot_dw : nullptr);
                const SlotHead h = optpfor_slot_head(exc);
                const uint32_t off = uniform((uint32_t)(p - win.gbase));
                const uint32_t need = 4u * (2u + (h.hd & 0xFFFFu) + 4u * (h.hd >> 26) + 1u + (h.hf & 0xFFFFu) + 4u * (h.hf >> 26));
                const uint32_t in_win = uniform((p >= win.gbase && off < win.nbytes) ? 1u : 0u);
                const uint32_t fast = uniform((h.flag == 0u && in_win && off + need <= win.nbytes) ? 1u : 0u);
                if (__builtin_expect(fast != 0u, 1)) {
                    if (want_freqs) optpfor_decode_pair<true, true>(win.st + (off >> 2), exc, h, v0, v1, f0, f1, consumed, cf);
                    else optpfor_decode_pair<true, false>(win.st + (off >> 2), exc, h, v0, v1, f0, f1, consumed, cf);
                } else {
                    uint32_t nd = 0;
                    consumed = optpfor_decode_side(win.st + (in_win ? off >> 2 : 0u), in_win ? (win.nbytes - off) >> 2 : 0u, exc, p, xovf, 0u, 0u, v0, v1, &nd);
                    if (want_freqs) {
                        const uint32_t offf = off + consumed;
                        const uint32_t in_winf = uniform((in_win && offf < win.nbytes) ? 1u : 0u);
                        cf = optpfor_decode_side(win.st + (in_winf ? offf >> 2 : 0u), in_winf ? (win.nbytes - offf) >> 2 : 0u, exc, p + consumed, xovf, 1u, nd, f0, f1);
                    }
                }
            } else { // the list's partial last block: plain values in the tail table (entry: sz gaps-1, sz freqs-1, bytes of the two parts)
                const uint32_t* const t = tail_of(s);
                v0 = lane < sz ? t[lane] : 0u;
                v1 = lane + 64 < sz ? t[lane + 64] : 0u;
                consumed = t[2u * sz];
                if (want_freqs) {
                    f0 = lane < sz ? t[sz + lane] : 0u;
                    f1 = lane + 64 < sz ? t[sz + lane + 64] : 0u;
                    cf = t[2u * sz + 1u];
                }
            }
            consumed = uniform(consumed);
            if (freqs_too) {
                uint32_t* fd = F(s);
                fd[lane] = f0 + 1u;
                fd[lane + 64] = f1 + 1u;
                ++s_freqs_blocks;
                s_bytes += uniform(cf);
            }
        } else {
            consumed = decode_block<CODEC_T>(codec, win, p, bmax - base - (sz - 1), sz, dst, exc, v0, v1);
        }
        uint32_t g0 = (lane < sz) ? v0 + 1u : 0u;
        uint32_t g1 = (lane + 64 < sz) ? v1 + 1u : 0u;
        uint32_t i0 = wave_incl_scan(g0);
        uint32_t i1 = wave_incl_scan(g1) + bcast(i0, 63);
        uint32_t d0 = (lane < sz) ? base + i0 - 1u : 0xFFFFFFFFu;
        uint32_t d1 = (lane + 64 < sz) ? base + i1 - 1u : 0xFFFFFFFFu;
        dst[lane] = d0;
        dst[lane + 64] = d1;
        const uint64_t fo = (uint64_t)(p + consumed - arena);
        const uint32_t first = bcast(d0, 0);
        setm(s, M_CUR, b);
        setm(s, M_SIZE, sz);
        setm(s, M_BMAX, bmax);
        setm(s, M_POS, 0);
        setm(s, M_DOCID, first < num_docs ? first : num_docs); // num_docs doubles as the unit's doc-id limit
        setm(s, M_FREQ_LO, (uint32_t)fo);
        setm(s, M_FREQ_HI, (uint32_t)(fo >> 32));
        setm(s, M_FDEC, freqs_too ? 1u : 0u);
        if (SHARE_F && s && freqs_too) fowner = s;
        setm(s, M_DDEC, 1);
        setm(s, M_GPOS, b * 128u);
        setm(s, M_NEXTEP, next_ep);
        setm(s, M_HINT, blk_bytes);
        if constexpr (META::NPF > 0) {
          if (s < (uint32_t)META::NPF) {
            meta.pf_blk[s] = 0xFFFFFFFFu;
            if (b + 1 < nb) { // speculate that this list's next access is block b+1
                const uint32_t nb1 = b + 1;
                const uint8_t* taddr = maxs + 4ull * nb1;
                if (lane == 3 && nb1 + 1 < nb) taddr = endpoints + 4ull * nb1;
                uint32_t hv = ld32(taddr);
                if (lane == 3 && !(nb1 + 1 < nb)) hv = (uint32_t)(lend - data);
                const uint32_t* g = (const uint32_t*)((uintptr_t)(data + next_ep) & ~(uintptr_t)3);
                meta.pf_tab[s] = hv;
                meta.pf_w0[s] = g[lane];
                meta.pf_w1[s] = g[lane + 64];
                meta.pf_blk[s] = nb1;
            }
          }
        }
        wave_sync();
        ++s_docs_blocks;
        if (STATS && block_profile && lane == 0) atomicAdd(block_profile + 2ull * (m(s, M_PBASE) + b), 1u);
        s_bytes += 4 + consumed; // endpoint + docs part
        PT_END(*this, PH_DOCS);
    }

    // ---- decode_freqs_block (block_posting_list.hpp:321-331)
    DS2I_DEV void decode_freqs(uint32_t s) {
        PT_BEGIN(*this);
        if (is_pef()) {
            decode_freqs_pef(s);
            PT_END(*this, PH_FREQS);
            return;
        }
        const uint32_t lane = lane_id();
        const uint8_t* p = ptr(s, M_FREQ_LO);
        const uint32_t sz = m(s, M_SIZE);
        if (!win.covers(p, 64)) {
            uint64_t room = (uint64_t)(ptr(s, M_END_LO) - p);
            win.load(p, room < 256 ? (uint32_t)room + 8 : 256u);
        }
        uint32_t v0, v1;
        uint32_t* dst = F(s);
        uint32_t consumed;
        if constexpr (SIDE) {
            if (__builtin_expect(uniform(sz) == 128u, 1)) {
                stage_slot(m(s, M_PBASE) + m(s, M_CUR));
                const uint32_t off = uniform((uint32_t)(p - win.gbase));
                const uint32_t in_win = uniform((p >= win.gbase && off < win.nbytes) ? 1u : 0u);
                consumed = optpfor_decode_side(win.st + (in_win ? off >> 2 : 0u), in_win ? (win.nbytes - off) >> 2 : 0u, exc, p, xovf, 1u, (uniform(exc[XSLOT_HDR]) >> 16) & 0x3FFu, v0, v1);
            } else {
                const uint32_t* const t = tail_of(s);
                v0 = lane < sz ? t[sz + lane] : 0u;
                v1 = lane + 64 < sz ? t[sz + lane + 64] : 0u;
                consumed = t[2u * sz + 1u];
            }
            consumed = uniform(consumed);
        } else {
            consumed = decode_block<CODEC_T>(codec, win, p, 0xFFFFFFFFu, sz, dst, exc, v0, v1);
        }
        dst[lane] = v0 + 1u;
        dst[lane + 64] = v1 + 1u;
        setm(s, M_FDEC, 1);
        if (SHARE_F && s) fowner = s;
        wave_sync();
        ++s_freqs_blocks;
        if (STATS && block_profile && lane == 0) atomicAdd(block_profile + 2ull * (m(s, M_PBASE) + m(s, M_CUR)) + 1, 1u);
        s_bytes += consumed;
        PT_END(*this, PH_FREQS);
    }

    // ---- ctor (block_posting_list.hpp:86-103). bind() only records the list geometry; open()
    // additionally decodes block 0 like the reference constructor does.
    DS2I_DEV void bind(uint32_t s, const QTerm& t) {
        if (is_pef()) { // freq_index::operator[] (freq_index.hpp:192-214); header fields were parsed at upload
            setm(s, M_MAXS_LO, (uint32_t)t.list_off); // cmax[] of the chunk directory
            setm(s, M_MAXS_HI, (uint32_t)(t.list_off >> 32));
            setm(s, M_N, t.n);
            setm(s, M_NB, t.term); // chunks
            setm(s, M_QW, __float_as_uint(t.q_weight));
            setm(s, M_MAXW, __float_as_uint(t.max_weight));
            setm(s, M_END_LO, (uint32_t)t.list_end); // chunk entries
            setm(s, M_END_HI, (uint32_t)(t.list_end >> 32));
            setm(s, M_DBIT_LO, (uint32_t)t.aux0);
            setm(s, M_DBIT_HI, (uint32_t)(t.aux0 >> 32));
            setm(s, M_FBIT_LO, (uint32_t)t.aux1);
            setm(s, M_FBIT_HI, (uint32_t)(t.aux1 >> 32));
            setm(s, M_CUR, 0xFFFFFFFFu);
            setm(s, M_BMAX, 0);
            setm(s, M_FDEC, 0);
            setm(s, M_PBASE, t.blk_base);
            setm(s, M_CBW, 0);
            setm(s, M_DDEC, 0);
            setm(s, M_SUF, __float_as_uint(t.suf_bmw));
            setm(s, M_RBASE, t.rmw_off64);
            setm(s, M_RSHIFT, t.rmw_shift);
            setm(s, M_RSCALE, __float_as_uint(t.rmw_scale));
            wave_sync();
            s_bytes += 16 + 8; // two collection offsets + gamma(occurrences), n
            return;
        }
        const uint32_t n = t.n;
        const uint32_t vl = 1u + (n >= (1u << 7)) + (n >= (1u << 14)) + (n >= (1u << 21)) + (n >= (1u << 28));
        const uint64_t maxs = t.list_off + vl;
        setm(s, M_MAXS_LO, (uint32_t)maxs);
        setm(s, M_MAXS_HI, (uint32_t)(maxs >> 32));
        setm(s, M_N, n);
        setm(s, M_NB, (n + 127u) >> 7);
        setm(s, M_QW, __float_as_uint(t.q_weight));
        setm(s, M_MAXW, __float_as_uint(t.max_weight));
        setm(s, M_END_LO, (uint32_t)t.list_end);
        setm(s, M_END_HI, (uint32_t)(t.list_end >> 32));
        setm(s, M_CUR, 0xFFFFFFFFu); // no block decoded yet
        setm(s, M_BMAX, 0);
        setm(s, M_FDEC, 0);
        setm(s, M_PBASE, t.blk_base);
        setm(s, M_CBW, 0);
        setm(s, M_DDEC, 0);
        setm(s, M_SUF, __float_as_uint(t.suf_bmw));
        setm(s, M_RBASE, t.rmw_off64);
        setm(s, M_RSHIFT, t.rmw_shift);
        setm(s, M_RSCALE, __float_as_uint(t.rmw_scale));
        if constexpr (SIDE) { // (block indexes do not use the freq_index bit offsets: the list's entry of the tail table rides there)
            setm(s, M_DBIT_LO, (uint32_t)t.aux1);
            setm(s, M_DBIT_HI, (uint32_t)(t.aux1 >> 32));
        }
        if constexpr (META::NPF > 0) { if (s < (uint32_t)META::NPF) meta.pf_blk[s] = 0xFFFFFFFFu; }
        wave_sync();
        s_bytes += vl + 8; // vbyte(n) + list offset
    }
    DS2I_DEV void open(uint32_t s, const QTerm& t) {
        bind(s, t);
        s_bytes += 4; // block_max[0]
        ++s_bm_examined;
        decode_docs(s, 0);
    }

    // first block >= from whose block_max >= lb, or nb if none. The reference scans block_max
    // linearly (block_posting_list.hpp:134-137); here one wave probes 64 entries at a time: first the
    // 64 entries right after the current block (short skips), then a 64-ary search over the rest.
    // `bmax` receives block_max of the returned block (valid iff the result < nb).
    // With a weight table `wtab` (bmw[] of this list) the block's max weight rides along: its load is issued together
    // with the block_max probe, so the caller's bound test costs no extra round trip. `w` is valid iff result < nb.
    DS2I_DEV uint32_t find_block(uint32_t s, uint32_t from, uint32_t lb) {
        uint32_t bmax;
        float w;
        return find_block(s, from, lb, bmax, nullptr, w);
    }
    DS2I_DEV uint32_t find_block(uint32_t s, uint32_t from, uint32_t lb, uint32_t& bmax, const float* wtab, float& w) {
        const uint8_t* maxs = ptr(s, M_MAXS_LO);
        const uint32_t nb = m(s, M_NB);
        const uint32_t lane = lane_id();
        bmax = 0;
        w = 0.f;
        if (from >= nb) return nb;
        {
            uint32_t idx = from + lane;
            uint32_t v = (idx < nb) ? ld32(maxs + 4ull * idx) : 0xFFFFFFFFu;
            float wv = 0.f;
            if (wtab && idx < nb) wv = wtab[idx];
            uint64_t hit = ballot(v >= lb);
            if (hit) {
                const uint32_t f = (uint32_t)__builtin_ctzll(hit);
                uint32_t blk = from + f;
                bmax = bcast(v, f);
                w = __uint_as_float(bcast(__float_as_uint(wv), f));
                return blk < nb ? blk : nb;
            }
        }
        uint32_t lo = from + 64, hi = nb; // answer in [lo, hi) or none
        if (lo >= hi) return nb;
        while (hi - lo > 64) {
            const uint32_t stride = (hi - lo + 63) / 64;
            uint32_t idx = lo + (lane + 1) * stride - 1;
            if (idx >= hi) idx = hi - 1;
            uint32_t v = ld32(maxs + 4ull * idx);
            uint64_t hit = ballot(v >= lb);
            if (!hit) return nb; // even block hi-1 (probed by the last lanes) is below lb
            uint32_t f = (uint32_t)__builtin_ctzll(hit);
            uint32_t nhi = lo + (f + 1) * stride;
            hi = nhi < hi ? nhi : hi;
            lo = lo + f * stride;
        }
        uint32_t idx = lo + lane;
        uint32_t v = (idx < hi) ? ld32(maxs + 4ull * idx) : 0xFFFFFFFFu;
        float wv = 0.f;
        if (wtab && idx < hi) wv = wtab[idx];
        uint64_t hit = ballot(v >= lb);
        if (!hit) return nb;
        const uint32_t f = (uint32_t)__builtin_ctzll(hit);
        uint32_t blk = lo + f;
        bmax = bcast(v, f);
        w = __uint_as_float(bcast(__float_as_uint(wv), f));
        return blk < hi ? blk : nb;
    }

    // find_block over the interleaved skip table: the probe that locates the block also returns its table words (the
    // entry of the block before it rides in the neighbouring lane), so the decode that follows needs no table load.
    // Same result as find_block(); `info` is valid iff the returned block < nb.
    DS2I_DEV uint32_t find_block_info(uint32_t s, uint32_t from, uint32_t lb, BlockInfo& info) {
        float w;
        return find_block_info(s, from, lb, info, nullptr, w);
    }
    DS2I_DEV uint32_t find_block_info(uint32_t s, uint32_t from, uint32_t lb, BlockInfo& info, const float* wtab, float& w) {
        const uint32_t nb = m(s, M_NB);
        const uint32_t lane = lane_id();
        w = 0.f;
        if (from >= nb) return nb;
        const uint2* tab = skip + m(s, M_PBASE);
        float wv = 0.f;
        auto finish = [&](uint2 e, uint32_t first_idx, uint64_t hit) -> uint32_t { // lane j holds entry first_idx + j
            const uint32_t f = (uint32_t)__builtin_ctzll(hit);
            const uint32_t blk = first_idx + f;
            w = __uint_as_float(bcast(__float_as_uint(wv), f));
            info.bmax = bcast(e.x, f);
            info.next_ep = bcast(e.y, f);
            const uint32_t pf = f ? f - 1 : 0;
            const uint32_t pmax = bcast(e.x, pf), pend = bcast(e.y, pf);
            info.base = blk ? pmax + 1u : 0u;
            info.ep = blk ? pend : 0u;
            return blk;
        };
        {   // entries from-1 .. from+62 (lane 0 = the block before `from`, never a candidate itself)
            const uint32_t first = from ? from - 1 : 0;
            const uint32_t idx = first + lane;
            uint2 e = make_uint2(0xFFFFFFFFu, 0u);
            if (idx < nb) e = tab[idx];
            if (wtab && idx < nb) wv = wtab[idx];
            uint64_t hit = ballot(idx >= from && idx < nb && e.x >= lb);
            if (hit) return finish(e, first, hit);
            if (first + 64 >= nb) return nb;
        }
        uint32_t lo = (from ? from - 1 : 0) + 64, hi = nb; // answer in [lo, hi) or none
        while (hi - lo > 63) {
            const uint32_t stride = (hi - lo + 63) / 64;
            uint32_t idx = lo + (lane + 1) * stride - 1;
            if (idx >= hi) idx = hi - 1;
            const uint32_t v = tab[idx].x;
            uint64_t hit = ballot(v >= lb);
            if (!hit) return nb;
            const uint32_t f = (uint32_t)__builtin_ctzll(hit);
            const uint32_t nhi = lo + (f + 1) * stride;
            hi = nhi < hi ? nhi : hi;
            lo = lo + f * stride;
        }
        // <= 63 candidates left: probe lo-1 .. (lo >= 64 here, so lo-1 exists)
        const uint32_t first = lo - 1;
        const uint32_t idx = first + lane;
        uint2 e = make_uint2(0xFFFFFFFFu, 0u);
        if (idx < hi) e = tab[idx];
        if (wtab && idx < hi) wv = wtab[idx];
        uint64_t hit = ballot(idx >= lo && idx < hi && e.x >= lb);
        if (!hit) return nb;
        return finish(e, first, hit);
    }

    // find_block_info for a list that moves to its NEXT blocks (from = current + 1, so every entry's block_max is >= lb
    // anyway): the first block >= from that `stop(block_max, bmw)` accepts; the blocks it rejects are skipped without being
    // decoded, 63 per probe. The caller's predicate says "this block could hold a result, or it ends the range the bound
    // was computed for". Returns nb when every remaining block is rejected.
    template <class STOP>
    DS2I_DEV uint32_t find_block_where(uint32_t s, uint32_t from, uint32_t lb, BlockInfo& info, const float* wtab, float& w, STOP stop) {
        const uint32_t nb = m(s, M_NB);
        const uint32_t lane = lane_id();
        w = 0.f;
        if (from >= nb) return nb;
        const uint2* tab = skip + m(s, M_PBASE);
        uint32_t first = from ? from - 1 : 0; // lane 0 holds the entry before the first candidate
        for (;;) {
            const uint32_t idx = first + lane;
            uint2 e = make_uint2(0xFFFFFFFFu, 0u);
            float wv = 0.f;
            if (idx < nb) { e = tab[idx]; wv = wtab[idx]; }
            const uint64_t hit = ballot(idx >= from && idx < nb && e.x >= lb && stop(e.x, wv));
            if (hit) {
                const uint32_t f = (uint32_t)__builtin_ctzll(hit);
                const uint32_t blk = first + f;
                w = __uint_as_float(bcast(__float_as_uint(wv), f));
                info.bmax = bcast(e.x, f);
                info.next_ep = bcast(e.y, f);
                const uint32_t pf = f ? f - 1 : 0;
                const uint32_t pmax = bcast(e.x, pf), pend = bcast(e.y, pf);
                info.base = blk ? pmax + 1u : 0u;
                info.ep = blk ? pend : 0u;
                return blk;
            }
            if (first + 64 >= nb) return nb;
            first += 63;
            from = first + 1;
        }
    }

    // ---- next_geq (block_posting_list.hpp:124-146)
    DS2I_DEV void next_geq(uint32_t s, uint32_t lb) {
        if (m(s, M_CUR) == 0xFFFFFFFFu || lb > m(s, M_BMAX)) {
            const uint32_t cur = m(s, M_CUR);
            uint32_t blk = find_block(s, cur + 1, lb);
            if (blk >= m(s, M_NB)) {
                s_bm_examined += 1; // block_max(nb-1) test
                s_bytes += 4;
                setm(s, M_DOCID, num_docs);
                wave_sync();
                return;
            }
            s_bm_examined += (cur == 0xFFFFFFFFu) ? 1u : blk - cur;
            s_bytes += 4ull * ((cur == 0xFFFFFFFFu) ? 1u : blk - cur);
            decode_docs(s, blk);
        }
        const uint32_t lane = lane_id();
        const uint32_t* d = D(s);
        uint32_t d0 = d[lane], d1 = d[lane + 64];
        uint64_t m0 = ballot(d0 >= lb), m1 = ballot(d1 >= lb);
        uint32_t idx, val;
        if (m0) {
            idx = (uint32_t)__builtin_ctzll(m0);
            val = bcast(d0, idx);
        } else {
            uint32_t l = (uint32_t)__builtin_ctzll(m1);
            idx = 64 + l;
            val = bcast(d1, l);
        }
        setm(s, M_POS, idx);
        setm(s, M_DOCID, val < num_docs ? val : num_docs);
        wave_sync();
    }

    // ---- next (block_posting_list.hpp:110-122)
    DS2I_DEV void next(uint32_t s) {
        uint32_t pos = m(s, M_POS) + 1;
        if (pos == m(s, M_SIZE)) {
            uint32_t cur = m(s, M_CUR);
            if (cur + 1 == m(s, M_NB)) {
                setm(s, M_POS, pos);
                setm(s, M_DOCID, num_docs);
                wave_sync();
                return;
            }
            ++s_bm_examined;
            s_bytes += 4;
            decode_docs(s, cur + 1);
        } else {
            uint32_t v = uniform(D(s)[pos]);
            setm(s, M_POS, pos);
            setm(s, M_DOCID, v < num_docs ? v : num_docs);
            wave_sync();
        }
    }

    // ---- freq (block_posting_list.hpp:165-171)
    DS2I_DEV uint32_t freq(uint32_t s) {
        if (!freqs_ready(s)) decode_freqs(s);
        return uniform(F(s)[m(s, M_POS)]);
    }
};

typedef CtxT<-1, MetaLds> Ctx;

// bm25::doc_term_weight (bm25.hpp:11-15). Compiled with -ffp-contract=off so the
// float32 operation order matches the reference exactly.
DS2I_DEV float doc_term_weight(uint32_t freq, float norm_len) {
    const float b = 0.5f, k1 = 1.2f;
    float f = (float)freq;
    return f / (f + k1 * (1.0f - b + b * norm_len)); // IEEE division: a v_rcp_f32 shortcut measured no gain
}

// ---- top-k scores (topk_queue, queries.hpp:152-197), k <= 64: lane j keeps the j-th
// largest score. insert() enters iff size<k or score > min (strict), like the reference.
struct TopK {
    float v;      // per lane
    uint32_t n;   // uniform
    uint32_t k;
    float floor;  // scores below it can never be in the final top-k (seeded lower bound); -inf = none
    // The k-th best score (-inf until k scores are held), kept as a wave-uniform copy that insert() refreshes.
    // would_enter() is called on per-lane values, often behind a short-circuit (`alive && would_enter(..)`), i.e. in
    // DIVERGENT code: reading lane k-1 of `v` there is undefined when that lane is inactive (after a spill the register is
    // reloaded for the active lanes only -- the runtime-codec kernels lost results to exactly that), so no cross-lane
    // operation may sit on that path. It also saves a v_readlane per test.
    float thr;
    DS2I_DEV void init(uint32_t k_) { v = -__builtin_inff(); n = 0; k = k_; floor = -__builtin_inff(); thr = -__builtin_inff(); }
    DS2I_DEV float threshold() const { return thr; }
    DS2I_DEV bool would_enter(float s) const { return s >= floor && (n < k || s > thr); }
    DS2I_DEV bool insert(float s) { // s wave-uniform; wave-uniform control flow only
        if (!would_enter(s)) return false;
        const uint32_t lane = lane_id();
        uint64_t ge = ballot(lane < n && v >= s);
        uint32_t p = (uint32_t)__builtin_popcountll(ge);
        float up = __shfl_up(v, 1);
        v = (lane < p) ? v : (lane == p) ? s : up;
        if (n < k) ++n;
        if (lane >= k) v = -__builtin_inff();
        thr = __uint_as_float(bcast(__float_as_uint(v), k - 1));
        return true;
    }
};

// ---- top-k beyond 64 scores (k <= 64 * NK): score i of the descending order lives in register i / 64 of lane i % 64.
// Same contract as TopK; used by the one-document-per-step kernels when a caller asks for k > 64 (the reference's
// topk_queue has no limit, queries.hpp:152-197).
template <int NK>
struct TopKBig {
    float v[NK];
    uint32_t n, k;
    float floor;
    DS2I_DEV void init(uint32_t k_) {
#pragma unroll
        for (int r = 0; r < NK; ++r) v[r] = -__builtin_inff();
        n = 0;
        k = k_;
        floor = -__builtin_inff();
        thr = -__builtin_inff();
    }
    float thr; // wave-uniform copy of the k-th best score, refreshed by insert() (see TopK::thr)
    DS2I_DEV float kth() const {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < NK; ++r)
            if ((uint32_t)r == ((k - 1) >> 6)) t = __uint_as_float(bcast(__float_as_uint(v[r]), (k - 1) & 63u));
        return t;
    }
    DS2I_DEV float threshold() const { return thr; }
    DS2I_DEV bool would_enter(float s) const { return s >= floor && (n < k || s > thr); }
    DS2I_DEV bool insert(float s) { // s wave-uniform
        if (!would_enter(s)) return false;
        const uint32_t lane = lane_id();
        uint32_t p = 0; // scores >= s keep their place
#pragma unroll
        for (int r = 0; r < NK; ++r) p += (uint32_t)__builtin_popcountll(ballot((uint32_t)r * 64u + lane < n && v[r] >= s));
#pragma unroll
        for (int r = NK - 1; r >= 0; --r) { // descending: register r - 1 still holds its old values when r takes its carry
            const float carry = r ? __uint_as_float(bcast(__float_as_uint(v[r > 0 ? r - 1 : 0]), 63)) : 0.f;
            float up = __shfl_up(v[r], 1);
            if (lane == 0) up = carry;
            const uint32_t i = (uint32_t)r * 64u + lane;
            float nv = (i < p) ? v[r] : (i == p) ? s : up;
            if (i >= k) nv = -__builtin_inff();
            v[r] = nv;
        }
        if (n < k) ++n;
        thr = kth();
        return true;
    }
};

} // namespace ds2i_dev
