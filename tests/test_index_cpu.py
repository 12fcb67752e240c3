"""Index container + enumerator contract on the CPU (oracle restatement over product-built images).

Mirrors reference test/test_block_posting_list.cpp:13-108 (sequential next() docid+freq, next_geq of every
element / beyond last / universe) and test/test_block_freq_index.cpp:13-68 (build -> freeze -> map ->
enumerate; docid()==num_docs after the last posting). has_next_geq<> is always false in the reference
(SURVEY.md §4 caveat), so next_geq is covered directly here.
"""
import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from helpers import Collection, small_params

CODECS = list(d.BLOCK_CODECS)


@pytest.fixture(scope="module")
def coll(built_lib):
    return Collection(small_params(num_docs=20000, num_terms=120))


@pytest.mark.parametrize("codec", CODECS)
def test_freeze_map_enumerate(coll, codec):
    img = coll.index_image(codec)
    idx = o.Index(codec, img)
    assert idx.size() == len(coll.lists) and idx.num_docs() == coll.num_docs
    for t, (docs, freqs) in enumerate(coll.lists):
        assert idx.list_size(t) == len(docs)
        dd, ff = idx.enumerate(t)  # also asserts docid()==num_docs after the last next()
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), (codec, t)


@pytest.mark.parametrize("codec", ["block_optpfor", "block_qmx", "block_mixed"])
def test_next_geq(coll, codec):
    idx = o.Index(codec, coll.index_image(codec))
    N = coll.num_docs
    rng = np.random.default_rng(5)
    for t in range(0, len(coll.lists), 7):
        docs, freqs = coll.lists[t]
        # every element, in order
        got, gf = idx.next_geq(t, docs)
        assert np.array_equal(got, docs) and np.array_equal(gf, freqs)
        # successor semantics for arbitrary non-decreasing probes incl. gaps, last+1 and the universe
        probes = np.sort(np.concatenate([rng.integers(0, N, 50), docs[:: max(1, len(docs) // 20)] + 1, [docs[-1] + 1, N]]))
        probes = np.minimum(probes, N).astype(np.uint32)
        got, _ = idx.next_geq(t, probes)
        pos = np.searchsorted(docs, probes)
        expect = np.where(pos < len(docs), docs[np.minimum(pos, len(docs) - 1)], N)
        assert np.array_equal(got, expect.astype(np.uint32)), (codec, t)


def test_endpoints_elias_fano(coll):
    """block_freq_index::operator[] finds list i through EF move(i) over m_endpoints (block_freq_index.hpp:85-94)."""
    img = coll.index_image("block_optpfor")
    idx = o.Index("block_optpfor", img)
    off = 0
    for t, (docs, freqs) in enumerate(coll.lists):
        assert idx.list_offset(t) == off
        off += len(d.encode_posting_list("block_optpfor", docs, freqs))


def test_many_lists_cross_ef_sampling(built_lib):
    """> 2^8 lists so that EF pointers1 (log_sampling1 = 8) are exercised by move()."""
    rng = np.random.default_rng(9)
    lists = []
    for t in range(700):
        n = int(rng.integers(1, 40))
        docs = np.sort(rng.choice(5000, size=n, replace=False)).astype(np.uint32)
        lists.append((docs, rng.integers(1, 5, n).astype(np.uint32)))
    img = d.build_index("block_varint", 5000, lists)
    idx = o.Index("block_varint", img)
    for t in (0, 1, 255, 256, 257, 511, 512, 699):
        dd, ff = idx.enumerate(t)
        assert np.array_equal(dd, lists[t][0]) and np.array_equal(ff, lists[t][1])


def test_builder_rejects_empty_list(built_lib):
    with pytest.raises(d.Ds2iError):
        d.build_index("block_optpfor", 10, [(np.zeros(0, np.uint32), np.zeros(0, np.uint32))])
