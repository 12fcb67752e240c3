// Deterministic synthetic collection + query generator (SURVEY.md §8(d) "Concrete
// synthetic inputs"). Host-side product utility: bench.py, tests and the CLI build
// their indexes from it. Every list is generated from (seed, term) alone, so the
// collection never has to be materialised (GOV2-scale = ~1 B postings) and the test
// oracle can regenerate any list independently.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ds2i_host {

struct synth_params {
    uint64_t seed = 0;
    uint32_t num_docs;
    uint32_t num_terms;
    double zipf_exp;       // list length of rank r = max(min_len, top_df_frac*N * r^-zipf_exp)
    double top_df_frac;
    uint32_t min_len;
    uint32_t clustered_every; // every k-th list alternates 8x / (1/8)x density segments; 0 = never
    // Correlated terms (0 = lists are independent thinnings of the doc-id space): documents come in runs of 4096 doc-ids,
    // each run belongs to one of `topics` topics, every term has a home topic, and a term is `topic_boost` times as likely
    // in a document of its home topic as its collection-wide rate says (the other documents are thinned so that the list
    // length stays what the Zipf law gives). Two terms of one topic then co-occur far above chance:
    // P(d in t2 | d in t1) ~ (boost^2 / topics) x df2 / N for boost << topics.
    uint32_t topics = 0;
    uint32_t topic_boost = 0;
};

inline uint32_t synth_home_topic(synth_params const& p, uint32_t term) {
    uint64_t x = p.seed ^ (0xA24BAED4963EE407ull * (uint64_t(term) + 1));
    x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29;
    return (uint32_t)(x % p.topics);
}
inline uint32_t synth_run_topic(synth_params const& p, uint64_t run) {
    uint64_t x = (p.seed * 0x2545F4914F6CDD1Dull) ^ (0xD1342543DE82EF95ull * (run + 1));
    x ^= x >> 32; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 30;
    return (uint32_t)(x % p.topics);
}

struct xoshiro256ss {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit xoshiro256ss(uint64_t seed) { for (auto& v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    // uniform in (0,1]
    double unit() { return double((next() >> 11) + 1) * (1.0 / 9007199254740992.0); }
};

inline uint64_t synth_target_len(synth_params const& p, uint32_t term) {
    double v = p.top_df_frac * double(p.num_docs) * std::pow(double(term) + 1.0, -p.zipf_exp);
    uint64_t n = v < 1.0 ? 1 : (uint64_t)v;
    n = std::max<uint64_t>(n, p.min_len);
    return std::min<uint64_t>(n, p.num_docs);
}

// Generates list `term`; returns its length (>=1). docs strictly increasing in [0,N).
inline uint64_t synth_list(synth_params const& p, uint32_t term, std::vector<uint32_t>& docs,
                           std::vector<uint32_t>& freqs) {
    docs.clear();
    freqs.clear();
    const uint64_t N = p.num_docs;
    const uint64_t target = synth_target_len(p, term);
    xoshiro256ss rng(p.seed ^ (0xD6E8FEB86659FD93ull * (uint64_t(term) + 1)));
    const double q0 = double(target) / double(N);
    const bool topical = p.topics > 1 && p.topic_boost > 1;
    const bool clustered = !topical && p.clustered_every && (term % p.clustered_every) == p.clustered_every - 1;
    const uint32_t home = topical ? synth_home_topic(p, term) : 0;
    const double f_in = topical ? 1.0 / double(p.topics) : 0.0;
    const double q_in = topical ? std::min(0.9, q0 * double(p.topic_boost)) : 0.0;
    const double q_out = topical ? std::max(0.0, (q0 - f_in * q_in) / (1.0 - f_in)) : 0.0;
    docs.reserve(target + target / 8 + 16);
    freqs.reserve(target + target / 8 + 16);
    uint64_t pos = 0; // next candidate docid
    bool dense = (rng.next() & 1) != 0;
    while (pos < N) {
        uint64_t seg_end = N;
        double q = q0;
        if (topical) { // runs of 4096 doc-ids: the term's home topic at q_in, every other run at q_out
            const uint64_t run = pos >> 12;
            seg_end = std::min<uint64_t>(N, (run + 1) << 12);
            q = synth_run_topic(p, run) == home ? q_in : q_out;
            if (q <= 0.0) { pos = seg_end; continue; }
        } else if (clustered) {
            uint64_t seg = uint64_t(1) << (12 + (rng.next() % 5));
            seg_end = std::min<uint64_t>(N, pos + seg);
            q = dense ? std::min(0.95, q0 * 8.0) : q0 / 8.0;
            dense = !dense;
        }
        if (q >= 1.0) q = 0.999999;
        const double inv = 1.0 / std::log1p(-q);
        while (true) {
            uint64_t skip = (uint64_t)(std::log(rng.unit()) * inv); // geometric(q) - 1
            if (skip >= seg_end - pos) { pos = seg_end; break; }
            pos += skip;
            docs.push_back((uint32_t)pos);
            uint32_t r = (uint32_t)rng.next() | 0x80000000u;
            freqs.push_back(1u + (uint32_t)__builtin_ctz(r));
            ++pos;
            if (pos >= seg_end) break;
        }
    }
    if (docs.empty()) {
        docs.push_back((uint32_t)(rng.next() % N));
        freqs.push_back(1);
    }
    return docs.size();
}

// Document lengths with the empirical SHAPE of the reference's test collection (test_collection.sizes
// has min 1, median 413, mean 1770, max 61081 -- a heavy tail an exponential does not have, and the shortest documents
// are what the freq-only score bound hangs on): the 257 values below are that file's 0/256 .. 256/256 quantiles; a
// document's length is read off the piecewise-linear inverse CDF at a 32-bit hash of its id (integer arithmetic only).
inline void synth_doc_sizes(synth_params const& p, std::vector<uint32_t>& sizes) {
    static const uint32_t quantile[257] = {
        1, 9, 15, 19, 21, 23, 25, 28, 30, 32, 35, 39, 44, 48, 54, 56, 58, 60, 64, 68, 72, 75, 79, 82, 84, 87, 89, 92,
        96, 99, 102, 105, 107, 109, 111, 113, 116, 118, 121, 124, 126, 129, 131, 133, 135, 137, 140, 142, 144, 147,
        149, 152, 154, 156, 157, 159, 160, 162, 163, 165, 166, 169, 170, 172, 173, 175, 178, 180, 183, 186, 188, 190,
        192, 195, 199, 202, 205, 207, 210, 213, 216, 219, 222, 226, 229, 232, 236, 239, 242, 245, 249, 251, 255, 260,
        264, 269, 272, 276, 280, 284, 289, 292, 296, 301, 305, 309, 312, 315, 320, 324, 329, 333, 337, 341, 345, 351,
        355, 360, 365, 369, 374, 379, 384, 388, 394, 399, 404, 408, 413, 416, 421, 425, 430, 435, 440, 444, 450, 454,
        460, 464, 469, 474, 478, 483, 490, 497, 503, 509, 516, 520, 527, 534, 540, 548, 556, 564, 572, 580, 588, 597,
        605, 613, 623, 633, 644, 655, 667, 677, 693, 708, 726, 742, 760, 779, 800, 816, 834, 855, 872, 895, 908, 928,
        948, 974, 1004, 1036, 1062, 1089, 1116, 1146, 1190, 1223, 1260, 1289, 1331, 1367, 1410, 1455, 1488, 1534,
        1590, 1639, 1696, 1734, 1785, 1855, 1923, 1986, 2057, 2113, 2183, 2249, 2294, 2353, 2414, 2472, 2523, 2607,
        2680, 2762, 2857, 2955, 3069, 3172, 3310, 3451, 3617, 3767, 3877, 3992, 4200, 4341, 4500, 4662, 4829, 5006,
        5135, 5252, 5473, 5756, 6019, 6385, 6742, 7093, 7721, 8436, 9138, 10034, 11163, 12038, 13706, 16292, 18246,
        22246, 28537, 39754, 61081};
    sizes.resize(p.num_docs);
    xoshiro256ss rng(p.seed ^ 0x5125CAFEull);
    for (uint32_t d = 0; d < p.num_docs; ++d) {
        const uint32_t u = (uint32_t)(rng.next() >> 32);
        const uint32_t seg = u >> 24, frac = u & 0xFFFFFFu;
        const uint64_t lo = quantile[seg], hi = quantile[seg + 1];
        sizes[d] = (uint32_t)std::max<uint64_t>(1, lo + (((hi - lo) * frac) >> 24));
    }
}

// Query log shaped like test/test_data/queries: length histogram of the real file,
// term rank log-uniform in [1,V], 1% of queries repeat a term.
inline void synth_queries(uint64_t seed, uint32_t num_terms, uint32_t nq, std::vector<uint32_t>& terms,
                          std::vector<uint32_t>& offsets) {
    static const uint32_t hist[11] = {45, 168, 124, 74, 41, 22, 13, 9, 2, 1, 1}; // lengths 1..11, /500
    xoshiro256ss rng(seed);
    terms.clear();
    offsets.assign(1, 0);
    for (uint32_t q = 0; q < nq; ++q) {
        uint32_t x = (uint32_t)(rng.next() % 500), len = 1, acc = 0;
        for (uint32_t l = 0; l < 11; ++l) { acc += hist[l]; if (x < acc) { len = l + 1; break; } }
        size_t begin = terms.size();
        for (uint32_t i = 0; i < len; ++i) {
            double u = rng.unit();
            double r = std::pow(double(num_terms), 1.0 - u); // in [1,V)
            uint32_t t = (uint32_t)r;
            if (t < 1) t = 1;
            if (t > num_terms) t = num_terms;
            terms.push_back(t - 1);
        }
        if (len > 1 && (rng.next() % 100) == 0) terms.back() = terms[begin];
        offsets.push_back((uint32_t)terms.size());
    }
}

// The same query log with `same_topic_pct` percent of the multi-term queries drawn from ONE topic (all terms share the home
// topic of the first): the queries whose lists are correlated.
inline void synth_queries_topical(synth_params const& p, uint64_t seed, uint32_t nq, uint32_t same_topic_pct, std::vector<uint32_t>& terms,
                                  std::vector<uint32_t>& offsets) {
    synth_queries(seed, p.num_terms, nq, terms, offsets);
    if (p.topics < 2) return;
    xoshiro256ss rng(seed ^ 0x70C1CA1ull);
    for (uint32_t q = 0; q < nq; ++q) {
        const uint32_t b = offsets[q], e = offsets[q + 1];
        if (e - b < 2 || rng.next() % 100 >= same_topic_pct) continue;
        const uint32_t home = synth_home_topic(p, terms[b]);
        for (uint32_t i = b + 1; i < e; ++i) { // same rank distribution (log-uniform), restricted to the topic
            for (int tries = 0; tries < 100000; ++tries) {
                const double r = std::pow(double(p.num_terms), 1.0 - rng.unit());
                uint32_t t = (uint32_t)r;
                t = t < 1 ? 1 : t > p.num_terms ? p.num_terms : t;
                if (synth_home_topic(p, t - 1) == home) { terms[i] = t - 1; break; }
            }
        }
    }
}

} // namespace ds2i_host
