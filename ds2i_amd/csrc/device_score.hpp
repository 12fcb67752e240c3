// Pieces shared by the ranked kernels (kernels.hip, ranked_stream.hip): the score histogram the parts of a split query
// share, the slack of the pruning bounds, and the span maximum of a range-table level.
#pragma once
#include "device_enum.hpp"

namespace ds2i_dev {

// Shared score histogram of the parts of a split ranked query: 256 buckets over [0, the query's score bound]. Every
// score that enters a part's heap is counted; a part's floor is the lower edge of the highest bucket with >= k documents
// at or above it -- a lower bound of the whole query's k-th score that tightens with every part's progress, not just
// with the best single part. `relax` widens the edge for operators whose parts may add a document's term scores in
// different orders (the disjunctive kernel); ranked_and's parts add them in the same order, bit for bit.
struct ScoreHist {
    unsigned int* h; // 256 counters of this query, or null
    float scale, inv, relax;
    DS2I_DEV void init(unsigned int* base, uint32_t q, float score_bound, float relax_) {
        h = base ? base + 256u * q : nullptr;
        scale = score_bound * (1.0f / 256.0f);
        inv = score_bound > 0.f ? 256.0f / score_bound : 0.f;
        relax = relax_;
    }
    // lower bound of the k-th score of the union, or -inf (relaxed agent-scope loads: the counters are updated by
    // other CUs' atomics and must not come from a stale L1 line)
    struct Snapshot { uint32_t x, y, z, w; }; // lane l: buckets 252-4l .. 255-4l (highest scores in lane 0)
    DS2I_DEV Snapshot load() const {
        const unsigned int* hp = h + 252u - 4u * lane_id();
        Snapshot s;
        s.x = __hip_atomic_load(hp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.y = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.z = __hip_atomic_load(hp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.w = __hip_atomic_load(hp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return s;
    }
    DS2I_DEV float floor(uint32_t k) const { return floor(load(), k); }
    // the floor a snapshot implies. Counters only grow and any floor ever valid stays valid, so a snapshot may be as old
    // as the caller likes: the disjunctive kernel loads one per round and resolves it a round later, which takes the
    // load's latency off the round's critical path
    DS2I_DEV float floor(const Snapshot& sn, uint32_t k) const {
        const uint32_t x = sn.x, y = sn.y, z = sn.z, w = sn.w;
        const uint32_t mine = x + y + z + w;
        const uint32_t incl = wave_incl_scan(mine);
        const uint64_t full = ballot(incl >= k);
        if (!full) return -__builtin_inff();
        const uint32_t fl = (uint32_t)__builtin_ctzll(full);
        const uint32_t before = bcast(incl - mine, fl), bw = bcast(w, fl), bz = bcast(z, fl), by = bcast(y, fl);
        uint32_t bucket = 255u - 4u * fl, c = before + bw; // inside lane fl the buckets from the top are w, z, y, x
        if (c < k) { --bucket; c += bz; if (c < k) { --bucket; c += by; if (c < k) --bucket; } }
        return (float)bucket * scale * relax;
    }
    DS2I_DEV void add(float v) const { // one lane
        uint32_t b = (uint32_t)(v * inv);
        b = b > 255u ? 255u : b;
        __hip_atomic_fetch_add(h + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// Slack of the ranked_and pruning bound. A document's score is the float32 sum of its term scores in list order
// (queries.hpp:372-380); the bound adds, in a different association, the current blocks' weights of the lists already
// positioned and one precomputed suffix sum for the lists still to come. Both evaluate the same real sum of <= 17
// non-negative terms, so they differ by at most ~3 * 17 * 2^-24 relative; 2^-17 covers that with margin and costs no
// pruning power. Everything else about the bound is exact: float multiplication and addition are monotone, and bmw[]
// holds the maxima of the very doc_term_weight values the scoring code computes.
static constexpr float BOUND_SLACK = 1.0f + 1.0f / 131072.0f;

// largest of the first `cnt` (1..16) bytes at lp. All 16 bytes are read with one unconditional (unaligned) load -- up to 15
// of them beyond the span, still inside the table area (every table is padded, the area ends with 64 spare bytes) -- so
// the lanes of a window have their loads in flight together; written as 16 predicated byte loads the compiler emitted
// 16 dependent round trips.
DS2I_DEV uint32_t max_of_bytes16(const uint8_t* lp, uint32_t cnt) {
    uint32_t w[4];
    __builtin_memcpy(w, lp, 16);
    uint32_t m = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        uint32_t v = (w[k >> 2] >> (8u * (k & 3u))) & 255u;
        v = k < cnt ? v : 0u;
        m = m > v ? m : v;
    }
    return m;
}


} // namespace ds2i_dev
