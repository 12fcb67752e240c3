"""Codec contract on the CPU: product encoders (host) x oracle decoders.

Mirrors reference test/test_block_codecs.cpp:9-46 (sizes {1,16,127,128} x magnitudes 1..24 x
{sum unknown, sum given}: decode(encode(v)) == v and the decoder consumes exactly the encoded bytes),
and pins vbyte / interpolative / QMX / posting-list layout to the reference's own bytes
(tests/golden/appendix_c.json; oracle/_ref for QMX).
"""
import json
import os

import numpy as np
import pytest

import ds2i_amd as d
import oracle as o

GOLD = os.path.join(os.path.dirname(__file__), "golden", "appendix_c.json")
CODECS = list(d.BLOCK_CODECS)


@pytest.mark.parametrize("codec", CODECS)
def test_block_codec_roundtrip_grid(built_lib, codec):
    rng = np.random.default_rng(12345)
    for n in (1, 16, 127, 128):
        for mag in range(1, 25):
            v = rng.integers(0, (1 << mag) + 1, size=n, dtype=np.uint64).astype(np.uint32)
            for s in (0xFFFFFFFF, int(v.sum())):
                if s > 0xFFFFFFFE:
                    continue
                enc = d.encode_block(codec, v, s)
                dec, consumed = o.decode_block(codec, enc, n, s)
                assert np.array_equal(dec, v), (codec, n, mag, s)
                assert consumed == len(enc), (codec, n, mag, s)


@pytest.mark.parametrize("codec", CODECS)
def test_block_codec_edge_values(built_lib, codec):
    cases = [np.zeros(128, np.uint32), np.ones(128, np.uint32), np.full(128, 0xFFFFFFFF >> 4, np.uint32),
             np.arange(128, dtype=np.uint32), np.array([0] * 127 + [1 << 27], np.uint32),
             np.array([(1 << 28) - 1] + [0] * 127, np.uint32), np.full(128, 1 << 20, np.uint32)]
    for v in cases:
        if int(v.astype(np.uint64).sum()) >= 0xFFFFFFFF and codec in ("block_interpolative", "block_mixed"):
            continue  # interpolative codes u32 prefix sums: the block sum must fit (it is <= the universe)
        enc = d.encode_block(codec, v, 0xFFFFFFFF)
        dec, consumed = o.decode_block(codec, enc, 128, 0xFFFFFFFF)
        assert np.array_equal(dec, v)
        assert consumed == len(enc)


def test_golden_appendix_c(built_lib):
    g = json.load(open(GOLD))
    for v, hx in g["vbyte"]:
        assert d.encode_vbyte(v).hex() == hx
        assert o.decode_vbyte(bytes.fromhex(hx)) == (v, len(hx) // 2)
    for case in g["blocks"]:
        vals = np.array(case["values"], dtype=np.uint32)
        s = case["sum"] if case["sum"] is not None else 0xFFFFFFFF
        enc = d.encode_block(case["codec"], vals, s)
        if "hex" in case:
            assert enc.hex() == case["hex"], case["name"]
        if "len" in case:
            assert len(enc) == case["len"], case["name"]
        if "prefix" in case:
            assert enc.hex().startswith(case["prefix"]) and enc.hex().endswith(case["suffix"]), case["name"]
        dec, consumed = o.decode_block(case["codec"], enc, len(vals), s)
        assert np.array_equal(dec, vals) and consumed == len(enc), case["name"]
    pl = g["posting_list"]
    assert d.encode_posting_list(pl["codec"], pl["docs"], pl["freqs"]).hex() == pl["hex"]


def test_golden_appendix_c_qmx_rows_live(built_lib):
    """The QMX rows of appendix_c.json regenerated from the reference's own encoder (oracle/_ref): the only rows of that
    fixture whose provenance is reproducible here (the others are survey-session values, see its _provenance)."""
    R = o.ref_qmx()
    if R is None:
        pytest.skip("oracle/_ref/libqmx_ref.so not built (needs /root/reference at build time)")
    g = json.load(open(GOLD))
    n = 0
    for case in g["blocks"]:
        if case["codec"] != "block_qmx" or len(case["values"]) != 128:
            continue
        v = np.ascontiguousarray(case["values"], dtype=np.uint32)
        buf = np.zeros(8192, dtype=np.uint8)
        ln = R.ref_qmx_encode(buf.ctypes.data, v.ctypes.data)
        live = (d.encode_vbyte(ln) + bytes(buf[:ln])).hex()
        if "hex" in case:
            assert live == case["hex"], case["name"]
        if "len" in case:
            assert len(live) // 2 == case["len"], case["name"]
        if "prefix" in case:
            assert live.startswith(case["prefix"]) and live.endswith(case["suffix"]), case["name"]
        n += 1
    assert n >= 3


def test_qmx_against_reference_codec(built_lib):
    """oracle/_ref = the reference's qmx_codec.hpp compiled as-is: encoder bytes and decoder output must agree."""
    R = o.ref_qmx()
    if R is None:
        pytest.skip("oracle/_ref/libqmx_ref.so not built (needs /root/reference at build time)")
    rng = np.random.default_rng(7)
    for it in range(600):
        mag = int(rng.integers(0, 33))
        kind = it % 4
        if kind == 0:
            v = rng.integers(0, 1 << mag, size=128, dtype=np.uint64)
        elif kind == 1:
            v = np.where(rng.random(128) < 0.9, rng.integers(0, 4, 128), rng.integers(0, 1 << mag, 128))
        elif kind == 2:
            v = np.repeat(rng.integers(0, 1 << mag, 32), 4) >> rng.integers(0, 8, 128)
        else:
            v = np.where(rng.random(128) < 0.7, 1, rng.integers(0, 1 << (mag // 2), 128))
        v = np.ascontiguousarray(v, dtype=np.uint32)
        buf = np.zeros(8192, dtype=np.uint8)
        ln = R.ref_qmx_encode(buf.ctypes.data, v.ctypes.data)
        mine = d.encode_block("block_qmx", v)
        val, vl = o.decode_vbyte(mine)
        assert val == ln and mine[vl:] == bytes(buf[:ln])
        ref_out = np.zeros(640, dtype=np.uint32)
        my_out = np.zeros(640, dtype=np.uint32)
        R.ref_qmx_decode(ref_out.ctypes.data, buf.ctypes.data, ln)
        o.lib().oracle_qmx_decode_stream(my_out.ctypes.data, buf.ctypes.data, ln)
        assert np.array_equal(ref_out[:128], v) and np.array_equal(my_out[:128], v)


def test_qmx_golden_blocks_from_the_reference_encoder(built_lib):
    """tests/golden/qmx_reference_blocks.json holds byte streams written by the reference's own QMX encoder
    (generated by tests/golden/make_qmx_golden.py): the product encoder reproduces them byte for byte and the
    oracle decoder reads them back -- this pin travels without /root/reference."""
    g = json.load(open(os.path.join(os.path.dirname(GOLD), "qmx_reference_blocks.json")))
    assert len(g["cases"]) == 96
    for case in g["cases"]:
        v = np.asarray(case["values"], dtype=np.uint32)
        ref = bytes.fromhex(case["hex"])
        mine = d.encode_block("block_qmx", v)
        val, vl = o.decode_vbyte(mine)
        assert val == len(ref) and mine[vl:] == ref, (case["mag"], case["kind"])
        out = np.zeros(640, dtype=np.uint32)
        buf = np.frombuffer(ref + bytes(64), dtype=np.uint8).copy()
        o.lib().oracle_qmx_decode_stream(out.ctypes.data, buf.ctypes.data, len(ref))
        assert np.array_equal(out[:128], v)
        dec, consumed = o.decode_block("block_qmx", mine, 128, 0xFFFFFFFF)
        assert np.array_equal(dec, v) and consumed == len(mine)


def test_mixed_block_types(built_lib):
    rng = np.random.default_rng(3)
    seen = set()
    for mag in (1, 3, 7, 9, 15, 20):
        for _ in range(20):
            v = rng.integers(0, 1 << mag, size=128, dtype=np.uint64).astype(np.uint32)
            if rng.random() < 0.5:
                v[rng.integers(0, 128, 3)] = 1 << 22
            enc = d.encode_block("block_mixed", v, 0xFFFFFFFF)
            seen.add(enc[0])
            dec, consumed = o.decode_block("block_mixed", enc, 128, 0xFFFFFFFF)
            assert np.array_equal(dec, v) and consumed == len(enc)
    assert {0, 1} <= seen  # both pfor and varint blocks occur (SURVEY.md 8(d) policy: values < 256 -> varint)
    v = rng.integers(0, 50, size=77, dtype=np.uint64).astype(np.uint32)
    enc = d.encode_block("block_mixed", v, int(v.sum()))
    dec, consumed = o.decode_block("block_mixed", enc, 77, int(v.sum()))
    assert np.array_equal(dec, v) and consumed == len(enc)
