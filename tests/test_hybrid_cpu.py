"""block_mixed space/time optimiser on the CPU (ds2i_hybrid_*, host_hybrid.hpp): the build-side counterpart of
reference optimal_hybrid_index.cpp + mixed_block::compute_space_time. The oracle decodes every optimised index."""
import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from helpers import Collection, brute_and, brute_ranked, queries_for, small_params

RTOL = 1e-5


@pytest.fixture(scope="module")
def coll(built_lib):
    return Collection(small_params(num_docs=30000, num_terms=120))


def _builder(coll, access=None, model=None):
    hb = d.HybridBuilder(coll.num_docs, model)
    base = 0
    for docs, freqs in coll.lists:
        nb = (len(docs) + 127) // 128
        hb.add_posting_list(docs, freqs, None if access is None else access[base:base + nb])
        base += nb
    return hb


def test_budget_is_respected_and_monotone(coll):
    hb = _builder(coll)
    lo, hi = hb.analyse()
    assert lo < hi
    prev_space, prev_time = None, None
    for frac in (0.0, 0.25, 0.5, 0.75, 1.0):
        budget = int(lo + frac * (hi - lo))
        img, info = hb.freeze(budget)
        assert info["space"] <= budget
        if prev_space is not None:
            assert info["space"] >= prev_space and info["model_time"] <= prev_time + 1e-3
        prev_space, prev_time = info["space"], info["model_time"]
        # the image is a valid block_mixed index holding the same postings
        idx = o.Index("block_mixed", img)
        for t in range(0, len(coll.lists), 7):
            dd, ff = idx.enumerate(t)
            assert np.array_equal(dd, coll.lists[t][0]) and np.array_equal(ff, coll.lists[t][1])
    # unlimited budget == the fastest point of every block; smallest budget == the smallest
    img_fast, fast = hb.freeze(None)
    assert fast["space"] == hi
    img_small, small = hb.freeze(lo)
    assert small["space"] == lo and len(img_small) < len(img_fast)
    with pytest.raises(d.Ds2iError):
        hb.freeze(lo - 1)


def test_gpu_model_prefers_pfor_and_access_counts_move_the_choice(coll):
    """With the MI355X model OptPFor is the fastest decoder: the unlimited-budget index has no varint block and keeps
    interpolative only where its tree has (almost) no live node. A CPU-like model (cheap varint) flips that. Blocks
    with high access counts get the fast encoding first when space is scarce."""
    hb = _builder(coll)
    _, fast = hb.freeze(None)
    assert fast["type_counts"]["docs"][1] == 0 and fast["type_counts"]["freqs"][1] == 0
    assert fast["type_counts"]["docs"][0] > 0
    cpu_like = d.HybridModel.default()
    cpu_like.varint = 50.0
    _, f2 = _builder(coll, model=cpu_like).freeze(None)
    assert f2["type_counts"]["docs"][1] > 0
    # skewed access: only list 0's blocks are hot
    nb_all = sum((len(dd) + 127) // 128 for dd, _ in coll.lists)
    nb0 = (len(coll.lists[0][0]) + 127) // 128
    access = np.zeros((nb_all, 2), np.uint32)
    access[:nb0] = 1000
    hb_hot = _builder(coll, access=access)
    lo, hi = hb_hot.analyse()
    img, info = hb_hot.freeze(int(lo + 0.1 * (hi - lo)))
    idx_small = o.Index("block_mixed", _builder(coll).freeze(lo)[0])
    idx_hot = o.Index("block_mixed", img)
    # list 0 grew (it was given the fast encodings), a cold list of similar size did not grow as much
    grow0 = (idx_hot.list_offset(1) - idx_hot.list_offset(0)) - (idx_small.list_offset(1) - idx_small.list_offset(0))
    assert grow0 > 0


def test_queries_on_optimised_index_match_brute_force(coll):
    hb = _builder(coll)
    lo, hi = hb.analyse()
    img, _ = hb.freeze(int(lo + 0.3 * (hi - lo)))
    idx = o.Index("block_mixed", img, coll.wand_image())
    for q in queries_for(coll, 60) + [[], [3], [3, 3]]:
        r = idx.query("and", q, want_matches=True)
        exp = brute_and(coll, q)
        assert r["count"] == len(exp) and np.array_equal(r["matches"], exp)
        np.testing.assert_allclose(idx.query("ranked_and", q)["topk"], brute_ranked(coll, q, 10, True, "size"), rtol=RTOL)


def test_synth_build_hybrid_equals_the_raw_collection(built_lib):
    """the streaming form (lists regenerated per pass) encodes exactly the synthetic collection"""
    p = small_params(num_docs=20000, num_terms=64)
    img, wand, postings, tc = d.synth_build_hybrid(p, budget_frac=0.5)
    img_ref, wand_ref, postings_ref = d.synth_build(p, "block_mixed")
    assert postings == postings_ref and bytes(wand) == bytes(wand_ref)
    idx = o.Index("block_mixed", img)
    for t in range(0, p.num_terms, 5):
        docs, freqs = d.synth_list(p, t)
        dd, ff = idx.enumerate(t)
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs)
    assert sum(tc["docs"]) > 0
    lo = d.synth_build_hybrid(p, budget_frac=0.0)[0]
    hi = d.synth_build_hybrid(p, budget_frac=1.0)[0]
    assert len(lo) <= len(img) <= len(hi)
