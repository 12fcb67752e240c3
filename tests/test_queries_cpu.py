"""Query functors of the oracle vs the codec-independent brute-force oracle (numpy) on a seeded collection.

Mirrors reference test/test_ranked_queries.cpp:10-75 (wand / maxscore top-10 == ranked_or within 0.1 %;
we hold 1e-5) and extends it to and / or / ranked_and, for every block codec.
"""
import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from helpers import Collection, brute_and, brute_or, brute_ranked, queries_for, small_params

RTOL = 1e-5


@pytest.fixture(scope="module")
def coll(built_lib):
    return Collection(small_params(num_docs=20000, num_terms=300))


@pytest.fixture(scope="module")
def queries(coll):
    return queries_for(coll, 150) + [[], [5], [5, 5], [7, 3, 7, 3], [0, 1, 2]]


@pytest.mark.parametrize("codec", list(d.BLOCK_CODECS))
def test_and_or_match_brute_force(coll, queries, codec):
    idx = o.Index(codec, coll.index_image(codec), coll.wand_image())
    for q in queries:
        r = idx.query("and", q, want_matches=True)
        exp = brute_and(coll, q)
        assert r["count"] == len(exp) and np.array_equal(r["matches"], exp)
        assert idx.query("or", q)["count"] == len(brute_or(coll, q))
        rf = idx.query("and_freq", q)
        fs = sum(int(coll.lists[t][1][np.searchsorted(coll.lists[t][0], exp)].sum()) for t in set(q)) if len(exp) else 0
        assert rf["count"] == len(exp) and rf["freq_sum"] == fs


@pytest.mark.parametrize("codec", ["block_optpfor", "block_mixed"])
def test_ranked_match_brute_force(coll, queries, codec):
    idx = o.Index(codec, coll.index_image(codec), coll.wand_image())
    for q in queries:
        exp_and = brute_ranked(coll, q, 10, True, "size")
        got = idx.query("ranked_and", q)
        assert got["count"] == len(exp_and)
        np.testing.assert_allclose(got["topk"], exp_and, rtol=RTOL)
        exp_or = brute_ranked(coll, q, 10, False, "term")
        for op in ("ranked_or", "wand", "maxscore"):
            got = idx.query(op, q)
            assert got["count"] == len(exp_or), (op, q)
            np.testing.assert_allclose(got["topk"], exp_or, rtol=RTOL, err_msg=str((op, q)))


def test_wand_data_layout(coll):
    """wand image = u64 N | float norm_lens | u64 V | float max_term_weight (wand_data.hpp:71-78)."""
    w = coll.wand_image()
    N = int(np.frombuffer(w[:8], dtype=np.uint64)[0])
    assert N == coll.num_docs
    nl = np.frombuffer(w[8:8 + 4 * N], dtype=np.float32)
    assert np.array_equal(nl, coll.norm_lens)
    V = int(np.frombuffer(w[8 + 4 * N:16 + 4 * N], dtype=np.uint64)[0])
    assert V == len(coll.lists)


def test_profile_counts_reference_traversal(coll, queries):
    idx = o.Index("block_optpfor", coll.index_image("block_optpfor"), coll.wand_image())
    r = idx.query("ranked_and", [0, 1], profile=True)
    p = r["profile"]
    assert p["docs_blocks"] >= 2 and p["block_max_examined"] >= p["docs_blocks"]
    assert p["algorithmic_bytes"] > 0 and p["postings_scored"] == len(brute_and(coll, [0, 1]))


@pytest.mark.parametrize("codec", ["block_optpfor", "opt"])
def test_threaded_oracle_batch_equals_sequential(coll, queries, codec):
    """query_batch_mt (the oracle on every host core: what the GPU tests compare a FULL 4096-query batch against at 25 M
    docs) gives the single-thread answers, and its per-query doc-id checksum is the checksum of the brute-force list."""
    idx = o.Index(codec, coll.index_image(codec), coll.wand_image())
    for op in ("and", "ranked_and", "wand", "maxscore"):
        a = idx.query_batch(op, queries, k=10)
        b = idx.query_batch_mt(op, queries, k=10, threads=4, match_hash=(op == "and"))
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), op
        if op == "and":
            for i, q in enumerate(queries):
                m = brute_and(coll, q).astype(np.uint64)
                assert (m * (2 * np.arange(len(m), dtype=np.uint64) + 1)).sum(dtype=np.uint64) == b[4][i]


def test_correlated_collection_is_correlated_and_answered_exactly(built_lib):
    """The robustness workload of bench.py (`--workload gov2c`: topical documents, a term far more likely in the documents of
    its home topic; queries drawn from one topic) at toy size: its same-topic term pairs really co-occur more than
    independent lists of the same lengths would -- that is what it is for -- the generator is deterministic, and the
    oracle answers its queries like the brute force does (the GPU parity tests then compare against that oracle)."""
    p = d.SynthParams(seed=0xD5210007, num_docs=40000, num_terms=256, zipf_exp=0.6, top_df_frac=0.2, min_len=64, clustered_every=4,
                      topics=8, topic_boost=32)
    coll = Collection(p)
    again = d.synth_list(p, 17)
    assert np.array_equal(again[0], coll.lists[17][0]) and np.array_equal(again[1], coll.lists[17][1])
    same = d.synth_queries_topical(p, 0x51E21, 200, same_topic_pct=100)
    assert same == d.synth_queries_topical(p, 0x51E21, 200, same_topic_pct=100)
    got = exp = 0.0
    for q in same:
        ts = sorted(set(q))
        for i in range(len(ts)):
            for j in range(i + 1, len(ts)):
                a, b = coll.lists[ts[i]][0], coll.lists[ts[j]][0]
                got += len(np.intersect1d(a, b, assume_unique=True))
                exp += len(a) * len(b) / coll.num_docs
    assert exp > 0 and got > 1.5 * exp, (got, exp)  # same-topic pairs: well above what independence predicts
    p0 = d.SynthParams(seed=p.seed, num_docs=p.num_docs, num_terms=p.num_terms, zipf_exp=p.zipf_exp, top_df_frac=p.top_df_frac,
                       min_len=p.min_len, clustered_every=p.clustered_every)
    flat = Collection(p0)
    g0 = e0 = 0.0
    for q in same[:60]:
        ts = sorted(set(q))
        for i in range(len(ts)):
            for j in range(i + 1, len(ts)):
                a, b = flat.lists[ts[i]][0], flat.lists[ts[j]][0]
                g0 += len(np.intersect1d(a, b, assume_unique=True))
                e0 += len(a) * len(b) / flat.num_docs
    assert e0 > 0 and g0 < 1.6 * e0 and got / exp > 1.3 * (g0 / e0), (g0, e0, got, exp)  # the same pairs without topics: far fewer co-occurrences
    queries = d.synth_queries_topical(p, 0x51E22, 120, same_topic_pct=50)
    idx = o.Index("block_optpfor", coll.index_image("block_optpfor"), coll.wand_image())
    for q in queries:
        r = idx.query("and", q, want_matches=True)
        e = brute_and(coll, q)
        assert r["count"] == len(e) and np.array_equal(r["matches"], e)
        ra = idx.query("ranked_and", q)
        ea = brute_ranked(coll, q, 10, True, "size")
        assert ra["count"] == len(ea)
        np.testing.assert_allclose(ra["topk"], ea, rtol=RTOL)
        eo = brute_ranked(coll, q, 10, False, "term")
        for op in ("wand", "maxscore"):
            ro = idx.query(op, q)
            assert ro["count"] == len(eo), (op, q)
            np.testing.assert_allclose(ro["topk"], eo, rtol=RTOL, err_msg=str((op, q)))
