"""The Elias-Fano freq_index family on the CPU -- "opt" (partitioned, the C3 configuration), "ef", "single" and
"uniform" (index_types.hpp:18-32): product writer (host_pef.hpp) x oracle enumerators.

Mirrors reference test/test_freq_index.cpp (build -> freeze -> map -> enumerate), test_generic_sequence.hpp:28-164
(move/next and the next_geq spec the reference never runs -- SURVEY.md §4 caveat: successor for every gap
position, beyond-last, beyond-universe) and test_partitioned_sequence.cpp (singletons, short lists in big universes,
single- and multi-partition lists), plus every query functor against the brute-force oracle.
"""
import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from helpers import Collection, brute_and, brute_or, brute_ranked, queries_for, small_params

RTOL = 1e-5


@pytest.fixture(scope="module")
def coll(built_lib):
    return Collection(small_params(num_docs=20000, num_terms=200))


KINDS = list(d.FREQ_INDEX_KINDS)


@pytest.fixture(scope="module", params=KINDS)
def kind(request):
    return request.param


@pytest.fixture(scope="module")
def idx(coll, kind):
    return o.Index(kind, coll.index_image(kind), coll.wand_image())


def test_freeze_map_enumerate(coll, idx):
    assert idx.size() == len(coll.lists) and idx.num_docs() == coll.num_docs
    for t, (docs, freqs) in enumerate(coll.lists):
        assert idx.list_size(t) == len(docs)
        dd, ff = idx.enumerate(t)  # next(): position()==i, docid()==num_docs after the last
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t


def test_next_geq_spec(coll, idx):
    N = coll.num_docs
    rng = np.random.default_rng(11)
    for t in range(0, len(coll.lists), 5):
        docs, freqs = coll.lists[t]
        got, gf = idx.next_geq(t, docs)
        assert np.array_equal(got, docs) and np.array_equal(gf, freqs)
        probes = np.sort(np.concatenate([rng.integers(0, N, 80), docs[:: max(1, len(docs) // 25)] + 1,
                                         docs[:: max(1, len(docs) // 7)] + (1 << 9), [docs[-1] + 1, N]]))
        probes = np.minimum(probes, N).astype(np.uint32)
        got, gf = idx.next_geq(t, probes)
        pos = np.searchsorted(docs, probes)
        ok = pos < len(docs)
        exp = np.where(ok, docs[np.minimum(pos, len(docs) - 1)], N).astype(np.uint32)
        assert np.array_equal(got, exp), t
        assert np.array_equal(gf[ok], freqs[pos[ok]])


def test_move_random_access(coll, idx):
    rng = np.random.default_rng(12)
    for t in range(0, len(coll.lists), 9):
        docs, freqs = coll.lists[t]
        ps = np.sort(rng.integers(0, len(docs), min(60, len(docs)))).astype(np.uint32)
        dd, ff = idx.move(t, ps)
        assert np.array_equal(dd, docs[ps]) and np.array_equal(ff, freqs[ps])


@pytest.mark.parametrize("kind", KINDS)
def test_shapes_partitions(built_lib, kind):
    """singletons, tiny lists in a huge universe, dense runs (all-ones / bitmap partitions), long multi-partition lists"""
    N = 1 << 22
    rng = np.random.default_rng(13)
    lists = [(np.array([5], np.uint32), np.array([3], np.uint32)),
             (np.array([N - 1], np.uint32), np.array([1], np.uint32)),
             (np.array([0, N - 1], np.uint32), np.array([1, 2], np.uint32)),
             (np.arange(1000, 1000 + 5000, dtype=np.uint32), np.ones(5000, np.uint32)),                 # all ones
             (np.sort(rng.choice(3 * 4096, 4096, replace=False) + 777).astype(np.uint32), rng.integers(1, 9, 4096).astype(np.uint32)),  # bitmap-ish
             (np.sort(rng.choice(N, 30000, replace=False)).astype(np.uint32), rng.integers(1, 300, 30000).astype(np.uint32)),
             (np.concatenate([np.arange(100, 2100), np.sort(rng.choice(N - 10000, 3000, replace=False)) + 10000]).astype(np.uint32),
              rng.integers(1, 4, 5000).astype(np.uint32))]
    img = d.build_index(kind, N, lists)
    idx = o.Index(kind, img)
    for t, (docs, freqs) in enumerate(lists):
        dd, ff = idx.enumerate(t)
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t
        got, _ = idx.next_geq(t, np.array([0, docs[0], docs[-1], min(N, int(docs[-1]) + 1), N], np.uint32))
        assert list(got) == [docs[0], docs[0], docs[-1], N, N]


def test_queries_match_brute_force(coll, idx, kind):
    queries = queries_for(coll, 120 if kind == "opt" else 40) + [[], [5], [5, 5], [7, 3, 7, 3]]
    for q in queries:
        r = idx.query("and", q, want_matches=True)
        exp = brute_and(coll, q)
        assert r["count"] == len(exp) and np.array_equal(r["matches"], exp)
        assert idx.query("or", q)["count"] == len(brute_or(coll, q))
        got = idx.query("ranked_and", q)
        np.testing.assert_allclose(got["topk"], brute_ranked(coll, q, 10, True, "size"), rtol=RTOL)
        exp_or = brute_ranked(coll, q, 10, False, "term")
        for op in ("ranked_or", "wand", "maxscore"):
            got = idx.query(op, q)
            assert got["count"] == len(exp_or)
            np.testing.assert_allclose(got["topk"], exp_or, rtol=RTOL)


def test_opt_is_smaller_than_block_indexes(coll):
    assert len(coll.index_image("opt")) < len(coll.index_image("block_optpfor"))
    # the optimal partitioning never loses to the fixed one or to no partitioning (partitioned_sequence.hpp:36-44)
    assert len(coll.index_image("opt")) <= len(coll.index_image("uniform"))
    assert len(coll.index_image("opt")) <= len(coll.index_image("single"))


def test_layouts_differ_only_in_the_sequences(coll):
    """same header, same collection framing: the four images share params | num_docs | #lists (freq_index.hpp:234-243)"""
    heads = {k: bytes(coll.index_image(k)[:5 + 8 + 8]) for k in KINDS}
    assert len(set(heads.values())) == 1
    assert len({len(coll.index_image(k)) for k in KINDS}) > 1


def _emulate_chunk_side(bits, bit0, typ, l, base, hi, hbias, lo, count, freq):
    """Python restatement of device_pef.hpp::pef_decode_side from the directory fields (bits = python int)."""
    if typ == 2:
        return [base + j for j in range(count)]
    out, pos = [], bit0 + hi
    for j in range(count):
        while not (bits >> pos) & 1:
            pos += 1
        hp = pos - bit0
        if typ == 1:
            out.append(base + (hp - hbias))
        else:
            low = (bits >> (bit0 + lo + j * l)) & ((1 << l) - 1) if l else 0
            out.append(base + (((hp - hbias - j) << l) | low) + (j if freq else 0))
        pos += 1
    return out


def test_upload_chunk_directory(coll, kind):
    """The directory built at GPU upload (host_pef.hpp::build_dir) decodes back to the raw lists with the device's
    formulas: chunks never cross a docs or freqs partition, cmax[] is each chunk's last doc-id."""
    img = coll.index_image(kind)
    for t in list(range(0, len(coll.lists), 11)) + [0, 1]:
        docs, freqs = coll.lists[t]
        cmax, chunks, (n, dbit0, fbit0, doff, foff) = d.opt_list_directory(img, t, kind)
        assert n == len(docs) and len(cmax) == len(chunks)
        dbits = int.from_bytes(img[doff:], "little")  # the bit vector words start at this byte; later bytes are harmless
        fbits = int.from_bytes(img[foff:], "little")
        pos = 0
        for b, c in enumerate(chunks):
            gpos, packed = int(c[0]), int(c[1])
            cnt = packed & 0xFF
            assert gpos == pos and 1 <= cnt <= 128
            dv = _emulate_chunk_side(dbits, dbit0, (packed >> 8) & 3, (packed >> 10) & 63, int(c[2]), int(c[3]), int(c[4]), int(c[5]), cnt, False)
            assert dv == docs[pos:pos + cnt].tolist(), (t, b)
            assert cmax[b] == docs[pos + cnt - 1]
            sv = _emulate_chunk_side(fbits, fbit0, (packed >> 16) & 3, (packed >> 18) & 63, int(c[6]), int(c[7]), int(c[8]), int(c[9]), cnt, True)
            prev = int(c[10])
            fv = [sv[0] - prev] + [sv[i] - sv[i - 1] for i in range(1, cnt)]
            assert fv == freqs[pos:pos + cnt].tolist(), (t, b)
            dspan, fspan = int(c[11]) & 0xFFFF, int(c[11]) >> 16
            if (packed >> 8) & 3 != 2:
                assert dspan >= cnt
            pos += cnt
        assert pos == n
