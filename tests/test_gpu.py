"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle.

Bit-exact for doc-id lists, counts and decoded postings; BM25 top-k within 1e-5 relative (north_star).
"""
import os

import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from helpers import Collection, brute_and, brute_ranked, mixed_block_type_counts, queries_for, small_params

pytestmark = pytest.mark.gpu
RTOL = 1e-5
CODECS = list(d.CODECS)
ALL_OPS = ["and", "and_freq", "or", "or_freq", "ranked_and", "wand", "maxscore", "ranked_or"]


@pytest.fixture(scope="module")
def coll(built_lib):
    return Collection(small_params(num_docs=20000, num_terms=300))


@pytest.fixture(scope="module")
def queries(coll):
    return queries_for(coll, 300) + [[], [5], [5, 5], [7, 3, 7, 3], [0, 1, 2], [0], [299, 298, 297, 296, 295, 294]]


@pytest.fixture(scope="module")
def images(coll):
    return {c: coll.index_image(c) for c in CODECS}, coll.wand_image()


def test_wave_scan_primitive(built_lib):
    rng = np.random.default_rng(0)
    x = rng.integers(0, 1 << 20, size=(64, 64), dtype=np.uint64).astype(np.uint32)
    out = np.zeros_like(x)
    rc = built_lib.ds2i_hip_selftest_scan(0, x.ctypes.data, out.ctypes.data, 64)
    assert rc == 0, built_lib.ds2i_hip_last_error()
    assert np.array_equal(out, np.cumsum(x, axis=1, dtype=np.uint32))


def test_device_bm25_equals_reference_fixture(built_lib):
    """The kernels' doc_term_weight against outputs of the reference's own bm25.hpp (tests/golden/bm25_reference.json,
    generated from /root/reference/bm25.hpp compiled as-is): bit-exact float32."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bm25_reference.json")))
    dtw = np.array(g["doc_term_weight"], dtype=np.uint64)
    dtw = dtw[dtw[:, 0] < (1 << 32)]
    f = np.ascontiguousarray(dtw[:, 0].astype(np.uint32))
    nl = np.ascontiguousarray(dtw[:, 1].astype(np.uint32).view(np.float32))
    out = np.zeros(len(f), dtype=np.float32)
    rc = built_lib.ds2i_hip_selftest_bm25(0, f.ctypes.data, nl.ctypes.data, out.ctypes.data, len(f))
    assert rc == 0, built_lib.ds2i_hip_last_error()
    assert np.array_equal(out.view(np.uint32), dtw[:, 2].astype(np.uint32))


def test_device_qmx_decodes_reference_encoder_bytes(built_lib):
    """The HIP QMX decoder fed with byte streams written by the REFERENCE's own encoder (tests/golden/
    qmx_reference_blocks.json, from /root/reference/qmx_codec.hpp compiled as-is): posting lists are assembled by hand
    from those bytes (block_posting_list.hpp:13-53 layout), checked to be what the index image holds, and decoded on
    the GPU."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qmx_reference_blocks.json")))
    cases = [c for c in g["cases"] if c["mag"] <= 20]
    lists, raw = [], []
    for j in range(0, len(cases) - 1, 2):
        gaps, fr = np.asarray(cases[j]["values"], np.uint64), np.asarray(cases[j + 1]["values"], np.uint64)
        docs = (np.cumsum(gaps + 1) - 1).astype(np.uint32)          # docs block codes gap - 1
        freqs = (fr + 1).astype(np.uint32)                            # freqs block codes freq - 1
        lists.append((docs, freqs))
        blk = lambda c: d.encode_vbyte(len(c["hex"]) // 2) + bytes.fromhex(c["hex"])  # qmx_block: vbyte(len) | stream
        raw.append(d.encode_vbyte(128) + int(docs[-1]).to_bytes(4, "little") + blk(cases[j]) + blk(cases[j + 1]))
    N = int(max(int(dd[-1]) for dd, _ in lists)) + 1
    img = d.build_index("block_qmx", N, lists)
    for r in raw:  # the image's list bytes ARE the reference encoder's bytes
        assert r in img
    gidx = d.Index("block_qmx", img)
    for t, (docs, freqs) in enumerate(lists):
        dd, ff = gidx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t


@pytest.mark.parametrize("codec", CODECS)
def test_decode_every_list(coll, images, codec):
    idx = d.Index(codec, images[0][codec])
    assert idx.size() == len(coll.lists) and idx.num_docs() == coll.num_docs
    for t, (docs, freqs) in enumerate(coll.lists):
        assert idx.list_size(t) == len(docs)
        dd, ff = idx[t]
        assert np.array_equal(dd, docs), (codec, t)
        assert np.array_equal(ff, freqs), (codec, t)


@pytest.fixture(autouse=True)
def _freq_layouts_native(monkeypatch):
    """opt / ef / single / uniform images are transcoded to block_optpfor at upload by default (like block_mixed). The tests
    of this module are about the partitioned-sequence kernels themselves, so they keep the image (DS2I_PEF_NATIVE=1, read by
    ds2i_hip_index_open); test_freq_layouts_are_transcoded_at_upload and the 25 M-doc configs[2] test drop the switch."""
    monkeypatch.setenv("DS2I_PEF_NATIVE", "1")


@pytest.mark.parametrize("kind", list(d.FREQ_INDEX_KINDS))
def test_freq_layouts_are_transcoded_at_upload(coll, queries, images, kind, monkeypatch):
    """the default upload of an opt / ef / single / uniform image: its lists are decoded by the partitioned-sequence kernels
    once, the device holds a block_optpfor index with side tables (ds2i_hip_index_get_info says so), every list it
    decodes is the image's list and every operator answers like the oracle reading the ORIGINAL image"""
    monkeypatch.delenv("DS2I_PEF_NATIVE")
    idx = d.Index(kind, images[0][kind], images[1])
    info = idx.info()
    assert info["transcoded_from"] == d.CODECS[kind] and info["has_side_tables"] and info["has_range_tables"]
    for t, (docs, freqs) in enumerate(coll.lists):
        dd, ff = idx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t
    oidx = o.Index(kind, images[0][kind], images[1])
    for op in ALL_OPS:
        _check_against_oracle(idx, oidx, op, queries)
    monkeypatch.setenv("DS2I_PEF_NATIVE", "1")
    assert d.Index(kind, images[0][kind], images[1]).info()["transcoded_from"] == -1


@pytest.fixture
def mixed_native(monkeypatch):
    """block_mixed images are transcoded to block_optpfor at upload by default; DS2I_MIXED_NATIVE=1 (read by
    ds2i_hip_index_open) keeps the image and runs the mixed-codec kernels."""
    monkeypatch.setenv("DS2I_MIXED_NATIVE", "1")


def test_block_mixed_native_decoders_and_kernels(coll, queries, images, mixed_native):
    """The mixed-codec device path itself (mixed_block.hpp:198-217 by type byte; k_ranked_stream_mixed and the CODEC_MIXED
    instantiations): every list decoded == the raw lists, every operator == the oracle. (Without the switch a block_mixed
    upload is transcoded and these kernels only run inside the upload's decode pass.)"""
    idx = d.Index("block_mixed", images[0]["block_mixed"], images[1])
    for t, (docs, freqs) in enumerate(coll.lists):
        dd, ff = idx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t
    oidx = o.Index("block_mixed", images[0]["block_mixed"], images[1])
    for op in ALL_OPS:
        _check_against_oracle(idx, oidx, op, queries)


def test_block_mixed_is_transcoded_at_upload(coll, images):
    """the default upload of a block_mixed image: the device holds a block_optpfor index with its side tables (reported by
    ds2i_hip_index_get_info), and every list it decodes is the mixed image's list"""
    idx = d.Index("block_mixed", images[0]["block_mixed"], images[1])
    info = idx.info()
    assert info["has_side_tables"] and info["side_table_bytes"] > 0
    for t, (docs, freqs) in enumerate(coll.lists):
        dd, ff = idx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t


def test_block_optpfor_without_side_tables(coll, queries, images, monkeypatch):
    """DS2I_NO_XSLOTS=1 (read by ds2i_hip_index_open): no exception side slots, the kernels parse the on-disk OptPFor bytes
    (Simple16 streams, interpolative tails: block_codecs.hpp:143-214, 46-79) -- the runtime-codec instantiations and
    k_conjunctive instead of k_ranked_stream. Every list and every operator must still equal the oracle."""
    monkeypatch.setenv("DS2I_NO_XSLOTS", "1")
    idx = d.Index("block_optpfor", images[0]["block_optpfor"], images[1])
    monkeypatch.delenv("DS2I_NO_XSLOTS")
    info = idx.info()
    assert not info["has_side_tables"] and info["side_table_bytes"] == 0
    for t, (docs, freqs) in enumerate(coll.lists):
        dd, ff = idx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t
    oidx = o.Index("block_optpfor", images[0]["block_optpfor"], images[1])
    for op in ALL_OPS:
        _check_against_oracle(idx, oidx, op, queries)


@pytest.mark.parametrize("budget", ["half", "3x"])
def test_table_budget(coll, queries, images, budget, monkeypatch):
    """DS2I_TABLE_BUDGET (read by ds2i_hip_index_open): the upload drops whole structures (hints, side slots) or halves the range
    tables' granularity until the resident bytes fit; ds2i_hip_index_get_info says what was built; every list and every
    operator still equals the oracle on whatever kernels that leaves."""
    img, wand = images[0]["block_optpfor"], images[1]
    full = d.Index("block_optpfor", img, wand)
    fb, finfo = full.device_bytes(), full.info()
    full.close()
    assert finfo["table_budget_bytes"] == 0 and finfo["has_membership_hints"] and finfo["has_side_tables"] and finfo["range_table_entries_per_posting"] == 4
    want = fb // 2 if budget == "half" else 3 * len(img)
    monkeypatch.setenv("DS2I_TABLE_BUDGET", str(want) if budget == "half" else "3x")
    idx = d.Index("block_optpfor", img, wand)
    monkeypatch.delenv("DS2I_TABLE_BUDGET")
    info = idx.info()
    assert info["table_budget_bytes"] == want
    assert idx.device_bytes() <= want or not info["has_range_tables"]   # (nothing left to drop: the budget is below the bare index)
    assert idx.device_bytes() < fb
    assert (info["range_table_entries_per_posting"], info["has_membership_hints"], info["has_side_tables"]) != (4, 1, 1)
    for t, (docs, freqs) in enumerate(coll.lists):
        dd, ff = idx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t
    oidx = o.Index("block_optpfor", img, wand)
    for op in ALL_OPS:
        _check_against_oracle(idx, oidx, op, queries)


def test_decode_list_through_both_decoders(coll, images, monkeypatch):
    """ds2i_hip_decode_list on an index WITH side slots: the slot decoder (default) and, with DS2I_DECODE_GENERAL=1, the
    decoder of the on-disk bytes -- the same postings both ways"""
    for general in (False, True):
        if general:
            monkeypatch.setenv("DS2I_DECODE_GENERAL", "1")
        idx = d.Index("block_optpfor", images[0]["block_optpfor"], images[1])  # (the knobs are read by the upload)
        assert idx.info()["has_side_tables"]
        for t, (docs, freqs) in enumerate(coll.lists):
            dd, ff = idx[t]
            assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), (general, t)


def _check_against_oracle(gidx, oidx, op, queries, k=10, reference_order=False):
    b = d.Batch(gidx, op, queries, k=k, want_matches=op in ("and", "and_freq"), reference_order=reference_order)
    st = b.run()
    count, topk, tlen, fsum = b.fetch()
    ocount, otopk, otlen, ofsum, _ = oidx.query_batch(op, queries, k=k)
    assert np.array_equal(count, ocount), op
    if op in ("and_freq", "or_freq"):
        assert np.array_equal(fsum, ofsum), op
    if op in ("ranked_and", "wand", "maxscore", "ranked_or"):
        assert np.array_equal(tlen, otlen), op
        for i in range(len(queries)):
            np.testing.assert_allclose(topk[i, :tlen[i]], otopk[i, :otlen[i]], rtol=RTOL, err_msg=str((op, queries[i])))
            assert np.all(np.isneginf(topk[i, tlen[i]:]))
    if op in ("and", "and_freq"):
        got = b.fetch_matches(count)
        for i, q in enumerate(queries):
            exp = oidx.query("and", q, want_matches=True)["matches"]
            assert np.array_equal(got[i], exp), (op, q)
    b.close()
    return st


def _scale_properties(gidx, oidx, queries, nsample=48, k=10):
    """Size-independent properties of a full batch + the oracle on a spread sample (the oracle needs seconds per
    query at this scale)."""
    and_count, _, _, _ = gidx.query_batch("and", queries)
    rc, rtopk, rlen, _ = gidx.query_batch("ranked_and", queries, k=k)
    assert np.array_equal(rlen, np.minimum(and_count, k).astype(np.uint32)) and np.array_equal(rc, rlen)
    for i in range(len(queries)):
        assert np.all(np.diff(rtopk[i, :rlen[i]]) <= 0)
        assert np.all(np.isneginf(rtopk[i, rlen[i]:]))
    rc2, rtopk2, rlen2, _ = gidx.query_batch("ranked_and", queries, k=k)  # idempotent, bit for bit
    assert np.array_equal(rtopk, rtopk2) and np.array_equal(rlen, rlen2)
    # a pipelined submission of the same batch in three pieces gives the same answers
    pipe = d.Pipeline(gidx, depth=3)
    cut = [0, len(queries) // 3, 2 * len(queries) // 3, len(queries)]
    tickets = [pipe.submit("ranked_and", queries[cut[j]:cut[j + 1]], k=k) for j in range(3)]
    parts = [pipe.wait(t) for t in tickets]
    pipe.close()
    assert np.array_equal(np.concatenate([p[1] for p in parts]), rtopk)
    step = max(1, len(queries) // nsample)
    idxs = list(range(0, len(queries), step))[:nsample]
    sample = [queries[i] for i in idxs]
    oc, otk, otl, _, _ = oidx.query_batch("ranked_and", sample, k=k)
    assert np.array_equal(rlen[idxs], otl)
    f = np.isfinite(otk)
    np.testing.assert_allclose(rtopk[idxs][f], otk[f], rtol=RTOL)
    oac, _, _, _, _ = oidx.query_batch("and", sample)
    assert np.array_equal(and_count[idxs], oac)
    _and_match_lists_equal_oracle(gidx, oidx, sample)
    return rtopk, rlen


def _full_batch_equals_oracle(gidx, oidx, queries, k=10, union_n=1024, chunk=512):
    """The WHOLE batch against the oracle, whatever the scale: the oracle answers it on every host core the process may use
    (oracle.Index.query_batch_mt: one query_ctx per thread over the immutable index, profile_queries.cpp:21-39 style).
    ranked_and: top-k of every query; and: counts and doc-id LISTS of every query (bit-exact, compared through an
    order-sensitive 64-bit checksum per query -- the lists of a 4096-query batch at 25 M docs are gigabytes);
    wand / maxscore: the first union_n queries. gidx: one device index or several uploads of the same image (the oracle
    answers once)."""
    gidxs = gidx if isinstance(gidx, (list, tuple)) else [gidx]
    oc, otk, otl, _, _ = oidx.query_batch_mt("ranked_and", queries, k=k)
    f = np.isfinite(otk)
    for g in gidxs:
        rc, rtopk, rlen, _ = g.query_batch("ranked_and", queries, k=k)
        assert np.array_equal(rlen, otl) and np.array_equal(rc, oc)
        assert np.array_equal(np.isfinite(rtopk), f)
        np.testing.assert_allclose(rtopk[f], otk[f], rtol=RTOL)
    oac, _, _, _, ohash = oidx.query_batch_mt("and", queries, match_hash=True)
    for g in gidxs:
        for lo in range(0, len(queries), chunk):  # (the doc-id buffer of a batch is sized for the shortest lists' full lengths)
            part = queries[lo:lo + chunk]
            b = d.Batch(g, "and", part, want_matches=True)
            b.run()
            count = b.fetch()[0]
            got = b.fetch_matches(count)
            b.close()
            assert np.array_equal(count, oac[lo:lo + chunk])
            for i, m in enumerate(got):
                m64 = m.astype(np.uint64)
                h = (m64 * (2 * np.arange(len(m64), dtype=np.uint64) + 1)).sum(dtype=np.uint64)
                assert h == ohash[lo + i], (lo + i, part[i])
    for op in ("wand", "maxscore"):
        sub = queries[:union_n]
        _, ot, ol, _, _ = oidx.query_batch_mt(op, sub, k=k)
        f = np.isfinite(ot)
        for g in gidxs:
            _, gt, gl, _ = g.query_batch(op, sub, k=k)
            assert np.array_equal(gl, ol), op
            assert np.array_equal(np.isfinite(gt), f), op
            np.testing.assert_allclose(gt[f], ot[f], rtol=RTOL, err_msg=op)


def _and_match_lists_equal_oracle(gidx, oidx, sample):
    """north_star: and_query doc-id LISTS bit-exact (not only their lengths), at whatever scale the index has"""
    b = d.Batch(gidx, "and", sample, want_matches=True)
    b.run()
    count = b.fetch()[0]
    got = b.fetch_matches(count)
    b.close()
    for i, q in enumerate(sample):
        exp = oidx.query("and", q, want_matches=True)["matches"]
        assert len(got[i]) == len(exp) and np.array_equal(got[i], exp), q


def _union_topk_equals_oracle(gidx, oidx, queries, nsample=32, k=10, ops=("wand", "maxscore")):
    """wand / maxscore against the ORACLE (not against each other) on a spread sample of the batch"""
    step = max(1, len(queries) // nsample)
    idxs = list(range(0, len(queries), step))[:nsample]
    sample = [queries[i] for i in idxs]
    assert len(sample) >= min(32, len(queries)) and sum(len(set(q)) > 2 for q in sample) >= 4
    for op in ops:
        _, gt, gl, _ = gidx.query_batch(op, sample, k=k)
        _, ot, ol, _, _ = oidx.query_batch(op, sample, k=k)
        assert np.array_equal(gl, ol), op
        f = np.isfinite(ot)
        assert np.array_equal(np.isfinite(gt), f), op
        np.testing.assert_allclose(gt[f], ot[f], rtol=RTOL, err_msg=op)


@pytest.mark.parametrize("codec", CODECS)
def test_pruning_tables_are_exact_maxima(coll, images, codec):
    """The upload-time tables ranked_and prunes with, against numpy on the raw lists, for every index kind: bmw[b] is
    exactly the largest bm25::doc_term_weight of block b (bit for bit: it is computed with the scoring arithmetic), and
    every byte of the doc-id-range table is 0 iff its range holds no posting of the list, else an upper bound (<= 1/255
    of the list maximum above) of the largest weight in the range."""
    from helpers import doc_term_weight
    gidx = d.Index(codec, images[0][codec], images[1])
    for t in list(range(0, 40)) + list(range(40, len(coll.lists), 7)):
        docs, freqs = coll.lists[t]
        w = doc_term_weight(freqs, coll.norm_lens[docs])
        bw = gidx.block_weights(t)
        if codec in d.BLOCK_CODECS:
            nb = (len(docs) + 127) // 128
            assert len(bw) == nb
            exp = np.array([w[b * 128:(b + 1) * 128].max() for b in range(nb)], dtype=np.float32)
            assert np.array_equal(bw, exp), (codec, t)
        else:  # chunks of <= 128 postings cut at partition boundaries: the maxima of consecutive runs, in order
            assert len(bw) >= (len(docs) + 127) // 128 and np.float32(bw.max()) == np.float32(w.max()), (codec, t)
        tab, sh, mx = gidx.range_table(t)
        assert len(tab) == (coll.num_docs >> sh) + 1 and np.float32(mx) == np.float32(w.max())
        rmax = np.zeros(len(tab), dtype=np.float32)
        np.maximum.at(rmax, docs >> sh, w)
        occupied = np.zeros(len(tab), dtype=bool)
        occupied[docs >> sh] = True
        assert np.array_equal(tab != 0, occupied), (codec, t)
        bound = tab.astype(np.float32) * np.float32(mx / 255.0)
        assert np.all(bound[occupied] * np.float32(1 + 2 ** -17) >= rmax[occupied]), (codec, t)
        assert np.all(bound[occupied] <= rmax[occupied] + np.float32(mx) * np.float32(1.01 / 255.0)), (codec, t)
        assert 4 * len(docs) <= len(tab) or sh == 0   # DS2I_RMW_G = 4 entries per posting at least (or one per doc-id)
        bm, _, _ = gidx.range_table(t, 0)  # dense lists (>= one document in 64) carry their exact bitmap
        if 64 * len(docs) >= coll.num_docs:
            bits = np.unpackbits(bm, bitorder="little")[:coll.num_docs]
            assert np.array_equal(np.flatnonzero(bits).astype(np.uint32), docs), (codec, t)
        else:
            assert len(bm) == 0
        prev = tab
        for level in (2, 3):  # the coarser levels: maxima of 64 entries of the level below
            up, shl, _ = gidx.range_table(t, level)
            assert shl == sh + 6 * (level - 1) and len(up) == (len(prev) + 63) // 64
            padded = np.zeros(len(up) * 64, dtype=np.uint8)
            padded[:len(prev)] = prev
            assert np.array_equal(up, padded.reshape(-1, 64).max(axis=1)), (codec, t, level)
            prev = up
        hints, hsh, _ = gidx.range_table(t, 4)  # membership hints (block_optpfor): 0 empty, 255 several postings, else 1 + offset % 254
        if True:  # (every index kind carries them since the ranked / and / wand kernels of all kinds consult them)
            assert hsh == sh and len(hints) == len(tab)
            cnt = np.bincount(docs >> sh, minlength=len(tab))
            exp = np.zeros(len(tab), dtype=np.uint8)
            exp[cnt > 1] = 255
            single = cnt[docs >> sh] == 1
            exp[(docs >> sh)[single]] = (1 + (docs[single] & ((1 << sh) - 1)) % 254).astype(np.uint8)
            assert np.array_equal(hints, exp), (codec, t)


def test_block_mixed_image_holds_all_three_block_types(images):
    """The block_mixed index every test of this module runs on dispatches to all three decoders: > 5 % of its full
    blocks are OptPFor, > 5 % VarInt-G8IU, > 5 % interpolative (docs parts; the freqs parts mix too)."""
    tc = mixed_block_type_counts(images[0]["block_mixed"], o)
    nd, nf = sum(tc["docs"]), sum(tc["freqs"])
    assert nd > 500
    assert all(c > 0.05 * nd for c in tc["docs"]), tc
    assert sum(c > 0.05 * nf for c in tc["freqs"]) >= 2, tc


@pytest.mark.parametrize("codec", CODECS)
@pytest.mark.parametrize("op", ALL_OPS)
def test_query_ops_match_oracle(coll, queries, images, codec, op):
    gidx = d.Index(codec, images[0][codec], images[1])
    oidx = o.Index(codec, images[0][codec], images[1])
    _check_against_oracle(gidx, oidx, op, queries)


@pytest.mark.parametrize("codec", ["block_optpfor", "block_mixed", "opt"])
@pytest.mark.parametrize("op", ["wand", "maxscore", "ranked_or", "or", "or_freq"])
def test_reference_order_disjunctive_traversals(coll, queries, images, codec, op):
    """wand / maxscore / ranked_or also exist as the reference's one-document-per-step traversals (k_daat,
    DS2I_OP_REFERENCE_ORDER); the default is the block-synchronous kernel. Both must equal the oracle."""
    gidx = d.Index(codec, images[0][codec], images[1])
    oidx = o.Index(codec, images[0][codec], images[1])
    _check_against_oracle(gidx, oidx, op, queries, reference_order=True)
    if op not in ("or", "or_freq"):
        for k in (1, 3, 64):
            _check_against_oracle(gidx, oidx, op, queries[:40], k=k)


@pytest.mark.parametrize("codec", ["block_optpfor", "opt", "block_qmx"])
@pytest.mark.parametrize("op", ["and", "and_freq", "ranked_and", "wand"])
def test_uninstrumented_kernels_give_the_same_results(coll, queries, images, codec, op):
    """ds2i_hip_batch_set_instrumented(0): kernels compiled without the statistics counters (block_optpfor / opt
    conjunctive operators; a no-op elsewhere) return exactly what the instrumented ones return."""
    gidx = d.Index(codec, images[0][codec], images[1])
    b = d.Batch(gidx, op, queries, k=10)
    st = b.run()
    ref = b.fetch()
    b.set_instrumented(False)
    st2 = b.run()
    got = b.fetch()
    ranked = op in ("ranked_and", "wand")
    for i, (x, y) in enumerate(zip(ref, got)):  # count, topk, topk_len, freq_sum (top-k only defined for ranked ops)
        # (wand: the parts of a split query race for the shared floor, which decides in what order a document's term
        # scores are met -- they are summed in fixed point, so the bits do not depend on it)
        if (ranked and i < 3) or (not ranked and i in (0, 3)):
            assert np.array_equal(x, y), i
    assert st2.kernel_ms > 0
    b.set_instrumented(True)
    st3 = b.run()
    if op not in ("wand", "ranked_and"):  # the parts of a split ranked query race for the shared floor: same results, varying work
        assert st3.docs_blocks_decoded == st.docs_blocks_decoded and st3.algorithmic_bytes == st.algorithmic_bytes
    b.close()


@pytest.mark.parametrize("op", ["and", "and_freq", "ranked_and"])
def test_reference_order_kernel_and_algorithmic_bytes(coll, queries, images, op):
    """The one-candidate-per-step GPU traversal decodes exactly the blocks the reference decodes."""
    codec = "block_optpfor"
    gidx = d.Index(codec, images[0][codec], images[1])
    oidx = o.Index(codec, images[0][codec], images[1])
    st = _check_against_oracle(gidx, oidx, op, queries, reference_order=True)
    _, _, _, _, prof = oidx.query_batch(op, queries, profile=True)
    assert st.docs_blocks_decoded == prof["docs_blocks"]
    assert st.freqs_blocks_decoded == prof["freqs_blocks"]
    assert st.block_max_examined == prof["block_max_examined"]
    assert st.algorithmic_bytes == prof["algorithmic_bytes"]
    # block-synchronous kernel: never many more docs blocks than the reference decodes, and -- with the doc-id-range tables
    # ruling candidates out before another list is touched (and ranked_and's score bounds on top) -- usually fewer
    b = d.Batch(gidx, op, queries)
    st2 = b.run()
    assert 0 < st2.docs_blocks_decoded <= prof["docs_blocks"] * 1.5 + 16


def test_topk_other_k(coll, queries, images):
    codec = "block_optpfor"
    gidx = d.Index(codec, images[0][codec], images[1])
    oidx = o.Index(codec, images[0][codec], images[1])
    for k in (1, 3, 64):
        for op in ("ranked_and", "ranked_or", "wand"):
            _check_against_oracle(gidx, oidx, op, queries[:80], k=k)


def test_topk_beyond_64(coll, queries, images):
    """k up to DS2I_HIP_MAX_K_LONG: the reference's topk_queue has no limit (queries.hpp:152-197). Beyond 64 ranked_and on block_optpfor
    stays on the stream kernels (4 / 16 scores per lane; the fuzz of it: test_ranked_and_through_the_stream_pipeline); everything else
    runs the one-document-per-step kernels with a 16-scores-per-lane heap."""
    sub = queries[:60] + [[], [5], [0, 1, 2], list(range(20))]
    for codec in ("block_optpfor", "opt"):
        gidx = d.Index(codec, images[0][codec], images[1])
        oidx = o.Index(codec, images[0][codec], images[1])
        for k in (65, 100, 200, 1024):
            for op in ("ranked_and", "ranked_or", "wand", "maxscore"):
                _check_against_oracle(gidx, oidx, op, sub, k=k)
    with pytest.raises(d.Ds2iError):
        gidx.query_batch("ranked_and", sub, k=1025)


def test_brute_force_agreement(coll, queries, images):
    """Codec-independent check: the GPU results equal sorted-array intersection + BM25 in numpy."""
    gidx = d.Index("block_varint", images[0]["block_varint"], images[1])
    count, _, _, _ = gidx.query_batch("and", queries)
    rcount, topk, tlen, _ = gidx.query_batch("ranked_and", queries)
    for i, q in enumerate(queries):
        assert count[i] == len(brute_and(coll, q))
        exp = brute_ranked(coll, q, 10, True)
        assert rcount[i] == len(exp)
        np.testing.assert_allclose(topk[i, :tlen[i]], exp, rtol=RTOL)


def test_error_behaviour(coll, images):
    gidx = d.Index("block_optpfor", images[0]["block_optpfor"])  # no wand data
    with pytest.raises(d.Ds2iError) as e:
        gidx.query_batch("and", [[coll.p.num_terms]])
    assert e.value.code == -3  # DS2I_ETERM: the reference only asserts (block_freq_index.hpp:87)
    with pytest.raises(d.Ds2iError) as e:
        gidx.query_batch("ranked_and", [[1, 2]])
    assert e.value.code == -5  # ranked op without wand data (queries.cpp:108-116)
    count, _, _, _ = gidx.query_batch("and", [list(range(17)) * 70])  # 1190 terms, 17 distinct: the long path answers
    assert count[0] == len(brute_and(coll, list(range(17))))
    # DS2I_HIP_MAX_TERMS_LONG distinct terms is the only length limit
    many = [(np.array([7], np.uint32), np.array([1], np.uint32))] * 1100
    tiny = d.Index("block_varint", d.build_index("block_varint", 100, many))
    count, _, _, _ = tiny.query_batch("and", [list(range(1024)), [3]])
    assert list(count) == [1, 1]
    with pytest.raises(d.Ds2iError) as e:
        tiny.query_batch("and", [[3], list(range(1025))])
    assert e.value.code == -6
    with pytest.raises(d.Ds2iError):
        d.Index("block_optpfor", b"\x00" * 10)
    count, _, _, _ = gidx.query_batch("and", [])
    assert len(count) == 0


def test_disjunctive_scores_are_order_independent(coll, queries, images):
    """wand, maxscore and ranked_or return the top-k of the same union; the block-synchronous kernel sums a document's
    term scores in fixed point, so the three agree BIT FOR BIT, run after run, whatever the split into parts, the
    pruning threshold's history or the number of batches in flight did to the order the terms were met in."""
    gidx = d.Index("block_optpfor", images[0]["block_optpfor"], images[1])
    ref = None
    for rep in range(3):
        for op in ("wand", "maxscore", "ranked_or"):
            _, topk, tlen, _ = gidx.query_batch(op, queries, k=10)
            if ref is None:
                ref = (topk.copy(), tlen.copy())
            assert np.array_equal(tlen, ref[1]) and np.array_equal(topk, ref[0]), (rep, op)
    pipe = d.Pipeline(gidx, depth=3)
    tickets = [pipe.submit(op, queries, k=10) for op in ("maxscore", "wand", "ranked_or")]
    for t in tickets:
        _, topk, tlen = pipe.wait(t)
        assert np.array_equal(topk, ref[0])
    pipe.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_ranked_and_pruning_fuzz_bit_identical(built_lib, seed):
    """ranked_and prunes with bounds that are claimed to be exact: over random collections (sizes, densities, clustering,
    doc-length spread), random and adversarial queries and several k, the top-k must equal the oracle's BIT FOR BIT -- not
    merely within the 1e-5 the north star allows -- for the block and the Elias-Fano layouts, one-shot and pipelined."""
    rng = np.random.default_rng(1000 + seed)
    nd = int(rng.integers(3000, 120000))
    nt = int(rng.integers(20, 200))
    p = d.SynthParams(seed=0xF00D0000 + seed, num_docs=nd, num_terms=nt, zipf_exp=float(rng.uniform(0.3, 1.0)),
                      top_df_frac=float(rng.uniform(0.2, 0.9)), min_len=int(rng.integers(1, 400)), clustered_every=int(rng.integers(0, 5)))
    lists = [d.synth_list(p, t) for t in range(nt)]
    sizes = d.synth_doc_sizes(p)
    if seed % 2 == 0:  # extreme doc-length spread: tiny and huge norm_lens next to each other
        sizes = np.where(rng.random(nd) < 0.1, 1, sizes).astype(np.uint32)
    wand = d.build_wand(sizes, lists)
    qs = d.synth_queries(0xABC0 + seed, nt, 300)
    qs += [[int(t)] for t in rng.integers(0, nt, 20)] + [[0, 1], [0, 1, 2], [nt - 1, 0], list(range(min(nt, 6)))]
    qs += [[int(x) for x in rng.integers(0, min(nt, 12), int(rng.integers(2, 5)))] for _ in range(80)]  # dense lists: big intersections
    # (the runtime-codec kernel instantiation -- block_varint / block_interpolative / block_qmx -- spills the most registers:
    # round 3's divergent-readlane bug only showed there)
    for codec in ("block_optpfor", "opt", "block_mixed", ("block_varint", "block_qmx", "block_interpolative")[seed % 3]):
        img = d.build_index(codec, nd, lists)
        gidx = d.Index(codec, img, wand)
        oidx = o.Index(codec, img, wand)
        # and / or: counts (and the doc-id lists of a sample) -- range tables, dense-list bitmaps, the streaming union
        for op in ("and", "and_freq", "or", "or_freq"):
            oc, _, _, ofs, _ = oidx.query_batch(op, qs)
            b = d.Batch(gidx, op, qs, want_matches=op == "and")
            b.run()
            gc, _, _, gfs = b.fetch()
            assert np.array_equal(gc, oc), (codec, op, np.argwhere(gc != oc)[:3])
            if op.endswith("freq"):
                assert np.array_equal(gfs, ofs), (codec, op)
            if op == "and":
                got = b.fetch_matches(gc)
                for i in range(0, len(qs), 7):
                    assert np.array_equal(got[i], oidx.query("and", qs[i], want_matches=True)["matches"]), (codec, qs[i])
            b.close()
        pipe = d.Pipeline(gidx, depth=2)
        for k in (1, 2, 10, 64):
            oc, otopk, otlen, _, _ = oidx.query_batch("ranked_and", qs, k=k)
            gc, gtopk, gtlen, _ = gidx.query_batch("ranked_and", qs, k=k)
            assert np.array_equal(gc, oc) and np.array_equal(gtlen, otlen), (codec, k)
            assert np.array_equal(gtopk, otopk), (codec, k, np.argwhere(gtopk != otopk)[:3])
            t = pipe.submit("ranked_and", qs, k=k)
            _, ptopk, _ = pipe.wait(t)
            assert np.array_equal(ptopk, otopk), (codec, k)
            # the union operators: block-max pruned on the GPU; within 1e-5 of the oracle and bit-identical to each other
            oc, otopk, otlen, _, _ = oidx.query_batch("ranked_or", qs, k=k)
            ref = None
            for op in ("wand", "maxscore", "ranked_or"):
                gc, gtopk, gtlen, _ = gidx.query_batch(op, qs, k=k)
                assert np.array_equal(gtlen, otlen), (codec, k, op, np.argwhere(gtlen != otlen)[:3])
                assert np.allclose(gtopk, otopk, rtol=1e-5, atol=0), (codec, k, op, np.argwhere(~np.isclose(gtopk, otopk, rtol=1e-5, atol=0))[:3])
                ref = gtopk if ref is None else ref
                assert np.array_equal(gtopk, ref), (codec, k, op)
        pipe.close()


def test_long_queries_more_than_16_terms(coll, queries, images):
    """The reference's functors take any number of terms (queries.hpp:35-86). Queries with more than 16 distinct terms
    run the one-document-per-step traversal with their enumerator state in global memory -- and the rest of the batch
    is answered as usual."""
    rng = np.random.default_rng(5)
    T = coll.p.num_terms
    dense = list(range(0, 24))                      # the 24 longest lists: a non-empty 24-term conjunction is plausible
    longq = [dense, dense[:17], list(rng.choice(T, 40, replace=False)), list(rng.choice(T, 100, replace=False)) + [3, 3],
             list(range(17)) * 3]
    mixed = queries[:40] + longq + queries[40:60]
    for codec in ("block_optpfor", "opt", "block_mixed"):
        gidx = d.Index(codec, images[0][codec], images[1])
        oidx = o.Index(codec, images[0][codec], images[1])
        for op in ALL_OPS:
            _check_against_oracle(gidx, oidx, op, mixed)
        _check_against_oracle(gidx, oidx, "ranked_or", longq, k=64)
        _check_against_oracle(gidx, oidx, "and", longq, reference_order=True)


def test_pipeline_matches_one_shot_and_reuses_slots(coll, images):
    """ds2i_hip_pipeline_*: distinct batches submitted back to back (three in flight) give exactly the one-shot
    answers; a fourth submit without collecting is refused; slots are reused across operators and batch sizes."""
    gidx = d.Index("block_optpfor", images[0]["block_optpfor"], images[1])
    batches = [queries_for(coll, n, seed=0x51E21 + i) for i, n in enumerate((300, 17, 1, 640, 300, 0, 45))]
    pipe = d.Pipeline(gidx, depth=3)
    for op in ("ranked_and", "and", "wand", "or_freq"):
        expect = [gidx.query_batch(op, b, k=10)[:3] for b in batches]
        tickets, got = [], []
        for i, b in enumerate(batches):
            if len(tickets) == 3:
                got.append(pipe.wait(tickets.pop(0)))
            tickets.append(pipe.submit(op, b, k=10))
        with pytest.raises(d.Ds2iError) as e:
            pipe.submit(op, batches[0], k=10)
        assert e.value.code == -8  # DS2I_EBUSY
        while tickets:
            got.append(pipe.wait(tickets.pop(0)))
        for (c, t, l), (ec, et, el) in zip(got, expect):
            assert np.array_equal(c, ec) and np.array_equal(l, el)
            if op in ("ranked_and", "wand"):
                assert np.array_equal(t, et)  # bit for bit, wand included (order-independent fixed-point sums)
    with pytest.raises(d.Ds2iError):
        pipe.wait(12345)
    pipe.close()


def test_no_device_memory_leak(coll, queries, images):
    """Every prepare / free cycle returns its device memory (round-1 leak: eight buffers per batch were never freed)."""
    import torch
    gidx = d.Index("block_optpfor", images[0]["block_optpfor"], images[1])
    def cycle(n):
        for _ in range(n):
            for op in ("ranked_and", "wand"):
                b = d.Batch(gidx, op, queries, k=10)
                b.run()
                b.close()
            gidx.query_batch("maxscore", queries[:50], k=10)
    cycle(3)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    cycle(40)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0, free1)


def test_ranked_and_block_max_pruning_prunes(built_lib):
    """ranked_and with the upload-time block-max weights scores a fraction of the postings an exhaustive traversal
    scores (results are checked against the oracle by every ranked test; here: it really prunes)."""
    p = d.SynthParams(seed=0xD52100BB, num_docs=400000, num_terms=64, zipf_exp=0.6, top_df_frac=0.5, min_len=20000,
                      clustered_every=4)
    img, wand, postings = d.synth_build(p, "block_optpfor")
    gidx = d.Index("block_optpfor", img, wand)
    oidx = o.Index("block_optpfor", img, wand)
    qs = [[t] for t in range(0, 64, 3)] + [[a, a + 1] for a in range(0, 62, 5)] + [[1, 2, 3], [5, 9, 20, 33]]
    st = _check_against_oracle(gidx, oidx, "ranked_and", qs, k=10)
    ac, _, _, _ = gidx.query_batch("and", qs)
    assert st.postings_scored < 0.5 * int(ac.sum()), (st.postings_scored, int(ac.sum()))
    for k in (1, 64):
        _check_against_oracle(gidx, oidx, "ranked_and", qs, k=k)


@pytest.mark.parametrize("codec", ["block_optpfor", "block_mixed", "opt"])
def test_correlated_collection_every_operator_equals_oracle(built_lib, codec):
    """bench.py's robustness workload (`--workload gov2c`) at a size the oracle answers in seconds: topical documents (a
    term 32 times as likely in the documents of its home topic), half of the multi-term queries drawn from one topic --
    intersections many times larger than independent lists give, i.e. the case the range tables prune least. Every
    operator equals the oracle (counts, doc-id lists, top-k), through the stream kernels (block_optpfor / block_mixed: 2-4
    terms with every upload-time table) and the class kernels (opt; 1 and 5+ terms)."""
    p = d.SynthParams(seed=0xD5210007, num_docs=1_000_000, num_terms=512, zipf_exp=0.6, top_df_frac=0.25, min_len=2000, clustered_every=4,
                      topics=16, topic_boost=32)
    img, wand, postings = d.synth_build(p, codec)
    gidx = d.Index(codec, img, wand)
    oidx = o.Index(codec, img, wand)
    qs = d.synth_queries_topical(p, 0x51E23, 384, same_topic_pct=50) + [[], [3], [3, 3], [9, 4, 9]]
    for op in ("ranked_and", "and", "and_freq", "wand", "maxscore", "ranked_or", "or", "or_freq"):
        _check_against_oracle(gidx, oidx, op, qs, k=10)
    _check_against_oracle(gidx, oidx, "ranked_and", qs, k=64)


def test_query_op_concept(coll, images):
    """queries.cpp-style use: op(index, terms) -> uint64, ranked ops expose topk()."""
    gidx = d.Index("block_qmx", images[0]["block_qmx"], images[1])
    q = [3, 40]
    assert d.and_query()(gidx, q) == len(brute_and(coll, q))
    op = d.ranked_and_query(None, 10)
    n = op(gidx, q)
    exp = brute_ranked(coll, q, 10, True)
    assert n == len(exp)
    np.testing.assert_allclose(op.topk(), exp, rtol=RTOL)


def test_adversarial_blocks(built_lib):
    """Decoder edge paths: 32-bit raw OptPFor blocks, blocks larger than the LDS staging window, ~50 % exceptions
    (more than 64 Simple16 words), wide QMX classes, huge freqs, a 2^31 universe -- every codec, decode + queries."""
    N = (1 << 31) - 1
    rng = np.random.default_rng(99)
    def mk(n, gaps):
        d = np.cumsum(gaps.astype(np.int64)) - 1
        d = d[d < N][:n]
        return d.astype(np.uint32)
    n = 128 * 9 + 57
    lists = []
    # huge gaps (b ~ 20..28) and huge freqs (b = 32 raw blocks)
    d1 = mk(n, rng.integers(1, 1 << 20, n))
    f1 = rng.integers(1, (1 << 31) - 2, len(d1)).astype(np.uint32)
    f1[(len(d1) // 128) * 128:] = rng.integers(1, 1 << 20, len(d1) % 128)  # the interpolative tail codes u32 prefix sums
    lists.append((d1, f1))
    # alternating tiny / large values: about half of each block are exceptions with wide payloads
    g = np.where(np.arange(n) % 2 == 0, 1, rng.integers(1 << 14, 1 << 19, n))
    d2 = mk(n, g)
    f2 = np.where(np.arange(len(d2)) % 2 == 0, 1, rng.integers(1 << 16, 1 << 24, len(d2))).astype(np.uint32)
    lists.append((d2, f2))
    # a few giant outliers in otherwise dense data
    g = np.ones(n, dtype=np.int64); g[::37] = 1 << 24
    d3 = mk(n, g)
    f3 = np.ones(len(d3), np.uint32); f3[::41] = (1 << 27) - 1
    lists.append((d3, f3))
    # dense run sharing doc-ids with the lists above (so intersections are non-empty)
    d4 = np.unique(np.concatenate([d1[::3], d2[::2], d3[::5], np.arange(5000, 5000 + 700, dtype=np.uint32)])).astype(np.uint32)
    lists.append((d4, rng.integers(1, 300, len(d4)).astype(np.uint32)))
    wand = d.build_wand(np.full(8, 10, np.uint32), [(np.array([0], np.uint32), np.array([1], np.uint32))])  # placeholder sizes
    queries = [[0, 3], [1, 3], [2, 3], [0, 1], [0, 1, 2, 3], [3], [0], [1], [2]]
    all_lists = lists
    for codec in CODECS:
        lists = all_lists
        if codec == "block_interpolative":  # codes u32 prefix sums of every block: keep block sums below 2^32
            lists = [(dd, (ff % (1 << 24)) + 1) for dd, ff in all_lists]
        img = d.build_index(codec, N, lists)
        gidx = d.Index(codec, img)
        oidx = o.Index(codec, img)
        for t, (docs, freqs) in enumerate(lists):
            dd, ff = gidx[t]
            assert np.array_equal(dd, docs), (codec, t)
            assert np.array_equal(ff, freqs), (codec, t)
        for op in ("and", "and_freq", "or", "or_freq"):
            _check_against_oracle(gidx, oidx, op, queries)
        for op in ("and", "and_freq"):
            _check_against_oracle(gidx, oidx, op, queries, reference_order=True)


@pytest.mark.parametrize("kind", list(d.FREQ_INDEX_KINDS))
def test_opt_index_partition_shapes(built_lib, kind):
    """opt / ef / single / uniform index on the GPU: singletons, all-ones runs, bitmap partitions, long multi-partition lists, tiny lists in a
    big universe (the shapes of test_partitioned_sequence.cpp) -- decode + next_geq-driven intersections."""
    N = 1 << 22
    rng = np.random.default_rng(13)
    lists = [(np.array([5], np.uint32), np.array([3], np.uint32)),
             (np.array([N - 1], np.uint32), np.array([1], np.uint32)),
             (np.array([0, N - 1], np.uint32), np.array([1, 2], np.uint32)),
             (np.arange(1000, 1000 + 5000, dtype=np.uint32), np.ones(5000, np.uint32)),
             (np.sort(rng.choice(3 * 4096, 4096, replace=False) + 777).astype(np.uint32), rng.integers(1, 9, 4096).astype(np.uint32)),
             (np.sort(rng.choice(N, 30000, replace=False)).astype(np.uint32), rng.integers(1, 300, 30000).astype(np.uint32)),
             (np.concatenate([np.arange(100, 2100), np.sort(rng.choice(N - 10000, 3000, replace=False)) + 10000]).astype(np.uint32),
              rng.integers(1, 4, 5000).astype(np.uint32))]
    img = d.build_index(kind, N, lists)
    wand = d.build_wand(np.full(N, 100, np.uint32), lists)
    gidx = d.Index(kind, img, wand)
    oidx = o.Index(kind, img, wand)
    for t, (docs, freqs) in enumerate(lists):
        dd, ff = gidx[t]
        assert np.array_equal(dd, docs) and np.array_equal(ff, freqs), t
    queries = [[a, b] for a in range(7) for b in range(a, 7)] + [[3, 4, 5], [3, 4, 6], [4, 5, 6], [2, 5], [0, 1]]
    for op in ALL_OPS:
        _check_against_oracle(gidx, oidx, op, queries)


@pytest.mark.parametrize("kind", list(d.FREQ_INDEX_KINDS))
def test_full_size_c2_opt_index(built_lib, kind):
    """BASELINE configs[2] shape at configs[1] scale: opt (PEF) index -- and its ef / single / uniform siblings --
    ranked_and, 4096-query batch, every query vs oracle."""
    p = d.SynthParams(seed=0xD5210002, num_docs=1000000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128,
                      clustered_every=4)
    img, wand, postings = d.synth_build(p, kind)
    queries = d.synth_queries(0x51E21, p.num_terms, 4096 if kind == "opt" else 1024)
    gidx = d.Index(kind, img, wand)
    oidx = o.Index(kind, img, wand)
    for op in ("and", "ranked_and"):
        count, topk, tlen, _ = gidx.query_batch(op, queries)
        oc, otopk, otlen, _, _ = oidx.query_batch(op, queries)
        assert np.array_equal(count, oc)
        if op == "ranked_and":
            assert np.array_equal(tlen, otlen)
            f = np.isfinite(otopk)
            np.testing.assert_allclose(topk[f], otopk[f], rtol=RTOL)
    sub = queries[:256]
    _, t_or, l_or, _ = gidx.query_batch("ranked_or", sub)
    _, o_or, ol_or, _, _ = oidx.query_batch("ranked_or", sub)
    assert np.array_equal(l_or, ol_or)
    f = np.isfinite(o_or)
    np.testing.assert_allclose(t_or[f], o_or[f], rtol=RTOL)
    for op in ("wand", "maxscore"):
        _, t2, l2, _ = gidx.query_batch(op, sub)
        assert np.array_equal(l_or, l2)
        assert np.array_equal(t2, t_or)  # wand == maxscore == ranked_or, bit for bit


def test_full_size_c2_properties(built_lib):
    """BASELINE configs[1]: 1M docs Zipf, block_optpfor, 4096-query batch. Parity on every query against the
    oracle (it finishes in seconds) plus size-independent properties."""
    p = d.SynthParams(seed=0xD5210002, num_docs=1000000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128,
                      clustered_every=4)
    img, wand, postings = d.synth_build(p, "block_optpfor")
    assert postings > 20_000_000
    queries = d.synth_queries(0x51E21, p.num_terms, 4096)
    gidx = d.Index("block_optpfor", img, wand)
    oidx = o.Index("block_optpfor", img, wand)
    acount, _, _, _ = gidx.query_batch("and", queries)
    rcount, topk, tlen, st = gidx.query_batch("ranked_and", queries)
    oc, otopk, otlen, _, _ = oidx.query_batch("ranked_and", queries)
    oac, _, _, _, _ = oidx.query_batch("and", queries)
    assert np.array_equal(acount, oac)
    _and_match_lists_equal_oracle(gidx, oidx, queries)            # every doc-id list of the batch, bit-exact
    _union_topk_equals_oracle(gidx, oidx, queries, nsample=256)
    assert np.array_equal(rcount, oc) and np.array_equal(tlen, otlen)
    assert np.array_equal(rcount, np.minimum(acount, 10))          # ranked_and keeps min(k, |AND|) scores
    with np.errstate(invalid="ignore"):  # -inf padding minus -inf
        assert np.all(np.diff(topk, axis=1)[np.isfinite(topk[:, 1:])] <= 0)  # descending
    finite = np.isfinite(otopk)
    np.testing.assert_allclose(topk[finite], otopk[finite], rtol=RTOL)
    # idempotence: a second run of the same batch gives identical bits
    rcount2, topk2, _, _ = gidx.query_batch("ranked_and", queries)
    assert np.array_equal(rcount, rcount2) and np.array_equal(topk, topk2)
    # wand == maxscore == ranked_or (the reference's own ranked test, test_ranked_queries.cpp:40-60)
    sub = queries[:512]
    _, t_or, l_or, _ = gidx.query_batch("ranked_or", sub)
    for op in ("wand", "maxscore"):
        _, t2, l2, _ = gidx.query_batch(op, sub)
        assert np.array_equal(l_or, l2)
        f = np.isfinite(t_or)
        assert np.array_equal(t2, t_or)  # wand == maxscore == ranked_or, bit for bit


def test_queries_cli_and_cpp_adaptor(coll, queries, images, tmp_path):
    """ds2i_amd/tools/queries: the reference driver's argv, query-log format and stats_line keys (queries.cpp:42-60)."""
    import json
    import os
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ds2i_amd", "tools", "queries")
    if not os.path.exists(tool):
        subprocess.check_call(["make", "-C", os.path.dirname(tool), "-s"])
    idx_path, wand_path = tmp_path / "idx", tmp_path / "wand"
    idx_path.write_bytes(images[0]["block_optpfor"])
    wand_path.write_bytes(images[1])
    log = "\n".join(" ".join(str(t) for t in q) for q in queries if q) + "\n"
    r = subprocess.run([tool, "block_optpfor", "and:ranked_and:wand:bogus", str(idx_path), str(wand_path)], input=log,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [l["query"] for l in lines] == ["and", "ranked_and", "wand"]
    for l in lines:
        assert l["type"] == "block_optpfor" and l["avg"] > 0 and l["q50"] <= l["q90"] <= l["q95"] and l["qps"] > 0
    assert "Unsupported query type: bogus" in r.stderr
    r = subprocess.run([tool, "no_such_index", "and", str(idx_path)], input=log, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ERROR: Unknown type" in r.stderr  # queries.cpp:149-151
    for l in lines:
        assert l["gpus"] == 1 and l["hbm_gbps"] > 0 and l["kernel_ms"] > 0
    # in-process multi-device path behind the C++ boundary: ds2i_hip::gpu_index_set + one host thread per
    # replica. Three replicas (on however many devices the box has -- one here, so they share it) cut every batch into three
    # contiguous slices; the concatenated answers are the one-replica answers, bit for bit, and the Python ABI's.
    dumps = []
    for g in (1, 3):
        dump = tmp_path / ("dump%d" % g)
        r = subprocess.run([tool, "block_optpfor", "and:ranked_and:maxscore", str(idx_path), str(wand_path), "--gpus", str(g),
                            "--dump", str(dump)], input=log, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        assert [l["gpus"] for l in out] == [g] * 3 and ("on %d replica(s)" % g) in r.stderr
        dumps.append(dump.read_text().splitlines())
    assert dumps[0] == dumps[1] and len(dumps[0]) == 3 * sum(1 for q in queries if q)
    gidx = d.Index("block_optpfor", images[0]["block_optpfor"], images[1])
    qs = [q for q in queries if q]
    acount, _, _, _ = gidx.query_batch("and", qs)
    _, rtopk, rlen, _ = gidx.query_batch("ranked_and", qs, k=10)
    for i in range(len(qs)):
        assert dumps[1][i].split() == ["and", str(int(acount[i]))]
        got = dumps[1][len(qs) + i].split()
        assert got[0] == "ranked_and" and int(got[1]) == int(rlen[i])
        assert [int(x, 16) for x in got[2:]] == rtopk[i, :rlen[i]].view(np.uint32).tolist()


def test_gpu_encode_is_byte_identical(coll, images):
    """block_posting_list::write + OptPFor findBestB / pack as HIP kernels. The image the GPU
    encoder produces must be the host builder's image byte for byte -- tiny lists, partial (interpolative) tail blocks,
    every exception regime, raw 32-bit blocks -- and the query path must run on it."""
    img, ms = d.gpu_encode_index(coll.num_docs, coll.lists)
    assert img == images[0]["block_optpfor"]
    # exception-count sweep (0..110 exceptions per block, docs and freqs), huge values (b = 32 raw blocks), length-1 lists
    rng = np.random.default_rng(99)
    nblk = 111
    freqs = rng.integers(1, 5, 128 * nblk).astype(np.uint32)
    gaps = rng.integers(1, 5, 128 * nblk).astype(np.uint64)
    for k in range(nblk):
        pos = rng.choice(128, k, replace=False) + 128 * k
        freqs[pos] = 1 + (1 << 10) + rng.integers(0, 1 << 9, k).astype(np.uint32)
        pos = rng.choice(128, k, replace=False) + 128 * k
        gaps[pos] = 1 + (1 << 9) + rng.integers(0, 1 << 8, k)
    docs = (np.cumsum(gaps) - 1).astype(np.uint32)
    big_d = (np.cumsum(rng.integers(1, 1 << 19, 128 * 3 + 5).astype(np.uint64)) - 1).astype(np.uint32)
    big_f = rng.integers(1, (1 << 31) - 2, len(big_d)).astype(np.uint32)
    big_f[384:] = rng.integers(1, 1 << 20, len(big_d) - 384)
    lists = [(docs, freqs), (big_d, big_f), (np.array([7], np.uint32), np.array([3], np.uint32)),
             (np.arange(0, 127, dtype=np.uint32), np.ones(127, np.uint32)), (np.arange(5, 5 + 128, dtype=np.uint32), np.full(128, 9, np.uint32))]
    N = int(max(int(dd[-1]) for dd, _ in lists)) + 10
    img2, _ = d.gpu_encode_index(N, lists)
    assert img2 == d.build_index("block_optpfor", N, lists)
    # configs[1]-shaped collection (1 M docs, Zipf lengths, clustered lists): 4096 terms of it
    p = d.SynthParams(seed=0xD5210002, num_docs=1000000, num_terms=4096, zipf_exp=0.75, top_df_frac=0.5, min_len=128,
                      clustered_every=4)
    c2 = [d.synth_list(p, t) for t in range(p.num_terms)]
    img3, ms3 = d.gpu_encode_index(p.num_docs, c2)
    host_img, _, postings = d.synth_build(p, "block_optpfor")
    assert img3 == host_img and ms3 > 0
    gidx = d.Index("block_optpfor", img3)
    dd, ff = gidx[17]
    assert np.array_equal(dd, c2[17][0]) and np.array_equal(ff, c2[17][1])
    # the whole configs[1] collection (65 536 terms, 52 M postings) through the one-call form
    p = d.SynthParams(seed=0xD5210002, num_docs=1000000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128,
                      clustered_every=4)
    gi, gw, gn, info = d.synth_build_gpu(p)
    hi, hw, hn = d.synth_build(p, "block_optpfor")
    assert gn == hn and gi == hi and gw == hw and info["device_ms"] > 0
    with pytest.raises(d.Ds2iError):
        d.gpu_encode_index(10, [(np.zeros(0, np.uint32), np.zeros(0, np.uint32))])  # "List must be nonempty"


def test_cpp_adaptor_document_enumerator(coll, images, tmp_path):
    """ds2i_hip::gpu_index::operator[] -> document_enumerator (the Index concept of SURVEY.md 8b): next / next_geq / move /
    reset / position / size and the exhaustion sentinel docid() == num_docs(), through the `enumerate` tool."""
    import os
    import subprocess
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ds2i_amd", "tools")
    tool = os.path.join(tools, "enumerate")
    subprocess.check_call(["make", "-C", tools, "-s"])
    rng = np.random.default_rng(11)
    for codec in ("block_optpfor", "opt"):
        path = tmp_path / ("idx_" + codec)
        path.write_bytes(images[0][codec])
        for t in (0, 7, 150, 299):
            docs, freqs = coll.lists[t]
            out = subprocess.run([tool, codec, str(path), str(t), "next"], capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr
            lines = out.stdout.split("\n")
            assert lines[0] == "size %d num_docs %d" % (len(docs), coll.num_docs)
            got = np.array([l.split() for l in lines[1:] if l], dtype=np.uint64)
            assert np.array_equal(got[:, 0], docs) and np.array_equal(got[:, 1], freqs), (codec, t)
            probes = np.sort(rng.integers(0, coll.num_docs + 50, 12))
            out = subprocess.run([tool, codec, str(path), str(t), "next_geq"] + [str(x) for x in probes], capture_output=True, text=True, timeout=120)
            got = np.array([l.split() for l in out.stdout.split("\n")[1:] if l], dtype=np.uint64)
            pos = np.searchsorted(docs, probes)
            exp = np.where(pos < len(docs), docs[np.minimum(pos, len(docs) - 1)], coll.num_docs)
            assert np.array_equal(got[:, 0], exp) and np.array_equal(got[:, 1], pos), (codec, t)
            ps = rng.integers(0, len(docs), 5)
            out = subprocess.run([tool, codec, str(path), str(t), "move"] + [str(x) for x in ps], capture_output=True, text=True, timeout=120)
            got = np.array([l.split() for l in out.stdout.split("\n")[1:] if l], dtype=np.uint64)
            assert np.array_equal(got[:-1, 0], docs[ps]) and np.array_equal(got[:-1, 1], ps)
            assert got[-1, 0] == docs[0] and got[-1, 1] == 0  # reset()


def test_cpp_set_query_recovers_after_a_failed_batch(images, tmp_path):
    """gpu_set_query_op caches one pipeline per replica. A batch in which one ticket fails (out-of-range term id)
    used to leave that replica's other tickets submitted and un-waited, and the next batch on the same operator object
    answered DS2I_EBUSY. The `enumerate ... set_recovery` self-check runs exactly that sequence over 3 replicas on this
    device: the bad batch throws, the good batch that follows equals one replica's answer."""
    import os
    import subprocess
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ds2i_amd", "tools")
    subprocess.check_call(["make", "-C", tools, "-s"])
    path = tmp_path / "idx_block_optpfor"
    path.write_bytes(images[0]["block_optpfor"])
    out = subprocess.run([os.path.join(tools, "enumerate"), "block_optpfor", str(path), "3", "set_recovery"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert out.stdout.strip() == "threw 1 equal 1 n 1200"


def test_block_profile_and_hybrid_optimiser(coll, queries, images):
    """GPU block-access profile (ds2i_hip_batch_block_profile) -> block_mixed optimiser (ds2i_hybrid_*) -> the
    optimised index answers every operator like the oracle, and its profile-weighted model time is lower than the
    fixed-policy index's at the same size."""
    codec = "block_mixed"
    gidx = d.Index(codec, images[0][codec], images[1])
    b = d.Batch(gidx, "ranked_and", queries, k=10)
    b.enable_block_profile()
    st = b.run()
    prof = b.block_profile()
    nb_all = sum((len(dd) + 127) // 128 for dd, _ in coll.lists)
    assert prof.shape == (nb_all, 2)
    assert int(prof[:, 0].sum()) == st.docs_blocks_decoded and int(prof[:, 1].sum()) == st.freqs_blocks_decoded
    st2 = b.run()  # accumulates (a second run may decode a few blocks more or fewer: the parts of a split query race for their shared floor)
    assert int(b.block_profile()[:, 0].sum()) == st.docs_blocks_decoded + st2.docs_blocks_decoded
    b.close()
    # wand profile includes its ranked_and seed pass
    bw = d.Batch(gidx, "wand", queries, k=10)
    bw.enable_block_profile()
    bw.run()
    assert int(bw.block_profile().sum()) > 0
    bw.close()

    hb = d.HybridBuilder(coll.num_docs)
    base = 0
    for docs, freqs in coll.lists:
        nb = (len(docs) + 127) // 128
        hb.add_posting_list(docs, freqs, prof[base:base + nb])
        base += nb
    lo, hi = hb.analyse()
    img, info = hb.freeze(int(lo + 0.5 * (hi - lo)))
    assert info["space"] <= int(lo + 0.5 * (hi - lo))
    g2 = d.Index(codec, img, images[1])
    o2 = o.Index(codec, img, images[1])
    for t in range(0, len(coll.lists), 9):
        dd, ff = g2[t]
        assert np.array_equal(dd, coll.lists[t][0]) and np.array_equal(ff, coll.lists[t][1])
    for op in ALL_OPS:
        _check_against_oracle(g2, o2, op, queries)


def test_gov2_scale_properties(built_lib):
    """BASELINE metric configuration at its stated size (25 M docs, all 32 768 terms, ~1.7 B postings, block_optpfor,
    4096-query batch): size-independent properties over the whole batch + the oracle on a 48-query sample; the
    disjunctive operators on slices of the batch."""
    p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096,
                      clustered_every=4)
    img, wand, postings = d.synth_build(p, "block_optpfor")
    assert postings > 1_500_000_000
    queries = d.synth_queries(0x51E21, p.num_terms, 4096)
    gidx = d.Index("block_optpfor", img, wand)
    oidx = o.Index("block_optpfor", img, wand)
    rtopk, rlen = _scale_properties(gidx, oidx, queries)
    _union_topk_equals_oracle(gidx, oidx, queries, nsample=40)     # configs[3]: wand + maxscore vs the oracle at 25 M docs
    _full_batch_equals_oracle(gidx, oidx, queries)                 # every query of the batch, not a sample
    and_count, _, _, _ = gidx.query_batch("and", queries[:256])
    # or >= and; wand == maxscore == ranked_or (test_ranked_queries.cpp:39-57), and they dominate ranked_and
    or_count, _, _, _ = gidx.query_batch("or", queries[:256])
    assert np.all(or_count >= and_count)
    sub = queries[:1024]
    _, wt, wl, _ = gidx.query_batch("wand", sub, k=10)
    _, mt, ml, _ = gidx.query_batch("maxscore", sub, k=10)
    _, ot, ol, _ = gidx.query_batch("ranked_or", queries[:256], k=10)
    assert np.array_equal(wl, ml) and np.array_equal(wl[:256], ol)
    f = np.isfinite(wt)
    assert np.array_equal(wt, mt)  # bit for bit (fixed-point sums)
    f = np.isfinite(ot)
    np.testing.assert_allclose(wt[:256][f], ot[f], rtol=RTOL)
    both = np.minimum(wl, rlen[:1024])
    for i in range(len(sub)):
        assert np.all(wt[i, :both[i]] >= rtopk[i, :both[i]] * (1 - RTOL))


@pytest.mark.parametrize("codec", ["block_optpfor", "block_mixed"])
def test_optpfor_exception_count_sweep(built_lib, codec):
    """One block per exception count 0..110: exercises the three OptPFor exception paths of the device decoder
    (<= 32 exceptions in registers, 33..64 two fields per lane, > 64 batched through LDS), in docs gaps and freqs."""
    rng = np.random.default_rng(99)
    nblk = 111
    freqs = rng.integers(1, 5, 128 * nblk).astype(np.uint32)
    gaps = rng.integers(1, 5, 128 * nblk).astype(np.uint64)
    for k in range(nblk):
        pos = rng.choice(128, k, replace=False) + 128 * k
        freqs[pos] = 1 + (1 << 10) + rng.integers(0, 1 << 9, k).astype(np.uint32)
        pos = rng.choice(128, k, replace=False) + 128 * k
        gaps[pos] = 1 + (1 << 9) + rng.integers(0, 1 << 8, k)
    docs = (np.cumsum(gaps) - 1).astype(np.uint32)
    N = int(docs[-1]) + 10
    other = np.sort(rng.choice(N, 3000, replace=False)).astype(np.uint32)
    lists = [(docs, freqs), (other, rng.integers(1, 9, len(other)).astype(np.uint32))]
    img = d.build_index(codec, N, lists)
    if codec == "block_optpfor":  # the sweep really produces the exception counts it is meant to
        blob = d.encode_block(codec, (freqs[128 * 50:128 * 51] - 1).astype(np.uint32))
        assert ((int.from_bytes(blob[:4], "little") >> 16) & 0x3FF) == 50
    gidx = d.Index(codec, img)
    oidx = o.Index(codec, img)
    dd, ff = gidx[0]
    assert np.array_equal(dd, docs) and np.array_equal(ff, freqs)
    for op in ("and", "and_freq", "or_freq"):
        _check_against_oracle(gidx, oidx, op, [[0, 1], [0], [1, 0]])


def test_bench_contract(built_lib):
    """bench.py prints ONE JSON line with the contract's keys (driver contract + roofline + cpu_baseline)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and j["unit"] == "queries/s"
    assert j["scaling"] == "weak" and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert "workload" in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0


def _run_bench(extra, nproc=1, env_extra=None, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra or {})
    if nproc == 1:
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", str(nproc)] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("knob", ["DS2I_NO_BMW=1", "DS2I_NO_RMW=1", "DS2I_RMW_G=1", "DS2I_NO_BITMAPS=1", "DS2I_NO_RMH=1", "DS2I_UNIT_FACTOR=64",
                                  "DS2I_UT_BLOCKS=1", "DS2I_NO_RANKED_STREAM=1", "DS2I_NO_UNION_RSTREAM=1", "DS2I_NO_LIST_STREAMS=1"])
def test_alternative_paths_give_the_same_results(built_lib, knob):
    """The library reads its knobs once per process (knobs.hpp), so each alternative path -- an upload without block-max weights (no
    tables at all), without range tables (wand / maxscore through the windowed k_disjunctive), with coarse tables, without the dense
    lists' bitmaps, without membership hints; very fine work units (many parts per query); wand / maxscore / ranked_or streams cut to
    one block per unit; ranked_and / and through the class kernels instead of k_ranked_stream; wand / maxscore / ranked_or through
    k_union_topk instead of k_union_stream; or_freq / and / and_freq without the list streams -- is driven through one fuzz collection
    (every codec, k, operator; oracle-checked) in a process of its own. (Rounds 2-5 kept 25 such switches alive; the ones whose
    alternative lost twice were removed in round 6 together with what only they reached.)"""
    import os, subprocess, sys
    env = dict(os.environ)
    k, v = knob.split("=")
    env[k] = v
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "fuzz and 2]"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("extra", ["", "DS2I_UNIT_CAP=8", "DS2I_STREAM_NT_MAX=8", "DS2I_STREAM_NT_MAX=4"])
def test_ranked_and_through_the_stream_pipeline(built_lib, extra):
    """The 2..16-term queries of a ranked_and batch on block_optpfor run k_ranked_stream<cap>, cap = the list capacity 2 | 4 | 6 | 8 | 16 of
    their launch group (queries.hpp:322-401 for any number of terms). Same results, bit for bit, whole queries and -- with units
    capped at 8 blocks -- queries split into many parts that share their floor; DS2I_STREAM_NT_MAX = 8 / 4 leaves the longer queries
    to the class kernels; the probe also asserts that the stream kernel did run for every capacity (launch groups)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if extra:
        k, v = extra.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "ranked_stream_probe.py"), "1", "2", "3"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "rs_nt8_probe ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("extra", ["", "DS2I_UNIT_CAP=8", "DS2I_NO_RMH=1", "DS2I_RMW_G=1", "DS2I_NO_RANKED_STREAM=1"])
def test_and_through_the_stream_pipeline(built_lib, extra):
    """and_query (queries.hpp:35-86) counts through k_ranked_stream<n, ., AND> (the default for `and` batches that do not ask for the
    doc-id lists): candidates whose membership hints settle every other list are counted without a search or a decode of those lists;
    one-term queries are list streams (k_and_stream). Counts equal the oracle's -- whole and split queries, an upload without hints
    (every survivor probed), coarse tables (ranges too wide for the hint to be proof), and the class kernels behind DS2I_NO_RANKED_STREAM."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if extra:
        k, v = extra.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "and_stream_probe.py"), "1", "2", "3"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "and_rstream_probe ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("extra", ["", "DS2I_NO_RMH=1", "DS2I_RMW_G=1", "DS2I_UT_BLOCKS=8", "DS2I_NO_UNION_RSTREAM=1"])
def test_union_through_the_stream_pipeline(built_lib, extra):
    """wand / maxscore / ranked_or (queries.hpp:200-319, 478-591, 404-476) through k_union_stream<cap> (union_stream.hip; the default on
    block_optpfor with every table): queries of 2..16 terms over random collections, k = 10 and 37 -- top-k lengths equal the oracle's,
    scores within 1e-5, wand == maxscore == ranked_or bit for bit, one-shot == pipelined; an upload without hints (exclusion lists
    looked up), coarse tables (ranges too wide for a hint to be proof), units of 8 blocks (many parts per driving list sharing one
    floor), and k_union_topk behind DS2I_NO_UNION_RSTREAM."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if extra:
        k, v = extra.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "union_stream_probe.py"), "1", "2"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0 and "union_stream_probe ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_two_ranks_on_one_device(built_lib):
    """The N>1 path of bench.py on a single-GPU box: two ranks (gloo rendezvous, both on cuda:0), each with a full index
    replica -- weak (every rank its own batches) and strong (one stream of batches cut with query_slice, results
    gathered with gather_concat). The answers must be the 1-rank answers (profile_queries.cpp:21-39 shares one index
    between threads the same way)."""
    two = {"DS2I_BENCH_BACKEND": "gloo", "DS2I_BENCH_ONE_DEVICE": "1"}
    common = ["--workload", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    one_w = _run_bench(common)
    two_w = _run_bench(common, nproc=2, env_extra=two)
    assert one_w["n_gpus"] == 1 and two_w["n_gpus"] == 2 and two_w["scaling"] == "weak"
    assert two_w["first_batch_checksum"] == one_w["first_batch_checksum"]  # rank 0 answers the batch it answers alone
    assert two_w["config"]["batch_per_gpu"] == 4096 and two_w["value"] > 0.5 * one_w["value"]
    assert two_w["cpu_baseline"] is None and two_w["roofline"]["frac"] > 0
    one_s = _run_bench(common + ["--scaling", "strong"])
    two_s = _run_bench(common + ["--scaling", "strong"], nproc=2, env_extra=two)
    assert two_s["scaling"] == "strong" and two_s["config"]["batch_per_gpu"] == 2048
    assert two_s["first_batch_checksum"] == one_s["first_batch_checksum"]  # gathered slices == the whole batch
    assert two_s["value"] > 0


def test_gov2_scale_opt_index_configs2(built_lib):
    """BASELINE configs[2] at its stated size: GOV2-scale (25 M docs, all 32 768 terms, ~1.7 B postings) `opt`
    partitioned-Elias-Fano index, ranked_and, 4096-query batch."""
    p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096,
                      clustered_every=4)
    img, wand, postings = d.synth_build(p, "opt")
    assert postings > 1_500_000_000
    queries = d.synth_queries(0x51E21, p.num_terms, 4096)
    assert max(max(q) for q in queries if q) > 30000  # the whole vocabulary is in play
    gidx = d.Index("opt", img, wand)
    oidx = o.Index("opt", img, wand)
    rtopk, rlen = _scale_properties(gidx, oidx, queries)
    _union_topk_equals_oracle(gidx, oidx, queries, nsample=32)
    del os.environ["DS2I_PEF_NATIVE"]  # (the module's fixture restores it) -- the default upload: transcoded to block_optpfor
    tidx = d.Index("opt", img, wand)
    assert tidx.info()["transcoded_from"] == d.CODECS["opt"] and gidx.info()["transcoded_from"] == -1
    # every query of the batch, not a sample -- the partitioned-sequence kernels on the image as it is AND the default upload
    _full_batch_equals_oracle([gidx, tidx], oidx, queries, union_n=512)
    tidx.close()
    # wand == maxscore (test_ranked_queries.cpp:39-57) and they dominate ranked_and, on a slice of the batch
    sub = queries[:512]
    _, wt, wl, _ = gidx.query_batch("wand", sub, k=10)
    _, mt, ml, _ = gidx.query_batch("maxscore", sub, k=10)
    assert np.array_equal(wl, ml)
    f = np.isfinite(wt)
    assert np.array_equal(wt, mt)  # bit for bit (fixed-point sums)
    both = np.minimum(wl, rlen[:512])
    for i in range(len(sub)):
        assert np.all(wt[i, :both[i]] >= rtopk[i, :both[i]] * (1 - RTOL))


def test_clueweb_scale_block_mixed_configs4(built_lib):
    """BASELINE configs[4] at its stated size (the per-GPU work of the 8-GPU run): ClueWeb09-B-scale (50 M docs, all
    32 768 terms, ~3.5 B postings) block_mixed index, ranked_and, 4096-query batch. The image is written with SURVEY.md
    8(d)'s deterministic policy -- VarInt-G8IU where every value fits 8 bits, OptPFor elsewhere, every 16th block of a
    list interpolative -- so all three decoders of mixed_block::decode (mixed_block.hpp:198-217) carry a substantial
    share of the blocks the queries touch (asserted below); the optimiser's images are covered at small scale by
    test_block_profile_and_hybrid_optimiser. (The sharding of the batch over ranks is covered at configs[1] scale by
    test_bench_two_ranks_on_one_device.)"""
    p = d.SynthParams(seed=0xD5210005, num_docs=50_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096,
                      clustered_every=4)
    img, wand, postings = d.synth_build(p, "block_mixed")
    assert postings > 3_000_000_000
    tc = mixed_block_type_counts(img, o, max_blocks=40000)
    nd, nf = sum(tc["docs"]), sum(tc["freqs"])
    assert nd > 20000 and all(c > 0.05 * nd for c in tc["docs"]), tc       # pfor, varint, interpolative: > 5 % each
    assert sum(c > 0.05 * nf for c in tc["freqs"]) >= 2, tc
    queries = d.synth_queries(0x51E21, p.num_terms, 4096)
    oidx = o.Index("block_mixed", img, wand)
    os.environ["DS2I_MIXED_NATIVE"] = "1"  # the mixed-codec kernels on the image as it is: properties + a 32-query oracle sample
    try:
        gidx = d.Index("block_mixed", img, wand)
        _scale_properties(gidx, oidx, queries, nsample=32)
        gidx.close()
    finally:
        del os.environ["DS2I_MIXED_NATIVE"]
    gidx = d.Index("block_mixed", img, wand)  # the default upload: transcoded to block_optpfor + side tables
    # every query of the batch against the threaded oracle at this scale too -- ranked_and top-k, `and` counts
    # and doc-id lists (checksums) of all 4096, wand / maxscore on the first 256
    _full_batch_equals_oracle(gidx, oidx, queries, union_n=256)
