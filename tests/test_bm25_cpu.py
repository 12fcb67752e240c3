"""The float32 half of the path pinned to the reference's own bm25.hpp (bm25.hpp:7-25): tests/golden/bm25_reference.json
holds outputs of the reference header compiled as-is (oracle/_ref/libbm25_ref.so, generator: make_bm25_golden.py).
Bit-exact for the oracle's restatement and for the product's host-side scorer (query weights, max_term_weight)."""
import ctypes as C
import json
import os

import numpy as np

import ds2i_amd as d
import oracle as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bm25_reference.json")


def _gold():
    g = json.load(open(GOLD))
    dtw = np.array(g["doc_term_weight"], dtype=np.uint64)
    qtw = np.array(g["query_term_weight"], dtype=np.uint64)
    return dtw, qtw


def test_oracle_bm25_equals_reference_fixture():
    dtw, qtw = _gold()
    got = o.bm25_doc_term_weight(dtw[:, 0], dtw[:, 1].astype(np.uint32).view(np.float32))
    assert np.array_equal(got.view(np.uint32), dtw[:, 2].astype(np.uint32))
    for N in np.unique(qtw[:, 2]):
        rows = qtw[qtw[:, 2] == N]
        got = o.bm25_query_term_weight(rows[:, 0], rows[:, 1], int(N))
        assert np.array_equal(got.view(np.uint32), rows[:, 3].astype(np.uint32)), int(N)


def test_product_host_bm25_equals_reference_fixture(built_lib):
    dtw, qtw = _gold()
    built_lib.ds2i_bm25_doc_term_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    built_lib.ds2i_bm25_query_term_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    f = np.ascontiguousarray(dtw[:, 0])
    nl = np.ascontiguousarray(dtw[:, 1].astype(np.uint32).view(np.float32))
    out = np.zeros(len(f), dtype=np.float32)
    assert built_lib.ds2i_bm25_doc_term_weight(f.ctypes.data, nl.ctypes.data, len(f), out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), dtw[:, 2].astype(np.uint32))
    for N in np.unique(qtw[:, 2]):
        rows = qtw[qtw[:, 2] == N]
        q, df = np.ascontiguousarray(rows[:, 0]), np.ascontiguousarray(rows[:, 1])
        out = np.zeros(len(q), dtype=np.float32)
        assert built_lib.ds2i_bm25_query_term_weight(q.ctypes.data, df.ctypes.data, int(N), len(q), out.ctypes.data) == 0
        assert np.array_equal(out.view(np.uint32), rows[:, 3].astype(np.uint32)), int(N)


def test_live_reference_bm25_when_present():
    """With /root/reference mounted the wrapper is rebuilt by oracle/Makefile: the fixture must be what it returns."""
    R = o.ref_bm25()
    if R is None:
        import pytest
        pytest.skip("oracle/_ref/libbm25_ref.so not built (no /root/reference)")
    dtw, qtw = _gold()
    for f, nlb, wb in dtw[::37]:
        nl = float(np.uint32(nlb).view(np.float32))
        assert np.float32(R.ref_bm25_doc_term_weight(int(f), nl)).view(np.uint32) == np.uint32(wb)
    for qtf, df, N, wb in qtw[::11]:
        assert np.float32(R.ref_bm25_query_term_weight(int(qtf), int(df), int(N))).view(np.uint32) == np.uint32(wb)
