"""BASELINE.json configs[0] -- the reference's own CPU-runnable case, as plumbing (no GPU):
test_collection (real .sizes and the 500 real queries; seeded lists replace the missing .docs/.freqs blobs)
-> ds2i binary collection files -> create_freq_index tool -> block_optpfor index + wand data ->
`and` / `ranked_and` through the CPU reference path (oracle) == brute force."""
import os
import subprocess

import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from helpers import K1, B, doc_term_weight, query_term_weight

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TOOL = os.path.join(ROOT, "ds2i_amd", "tools", "create_freq_index")


def read_sequences(path):
    w = np.fromfile(path, dtype=np.uint32)
    out, pos = [], 0
    while pos < len(w):
        n = int(w[pos])
        pos += 1
        if n:
            out.append(w[pos:pos + n])
            pos += n
    return out


@pytest.fixture(scope="module")
def c1(built_lib, tmp_path_factory):
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-C", os.path.dirname(TOOL), "-s"])
    sizes = read_sequences(os.path.join(HERE, "golden", "test_collection.sizes"))[0]
    N = len(sizes)
    assert N == 10000 and sizes.min() == 1 and sizes.max() == 61081
    queries = [[int(t) for t in line.split()] for line in open(os.path.join(HERE, "golden", "test_collection.queries"))]
    assert len(queries) == 500
    used = sorted({t for q in queries for t in q})
    V = max(used) + 1
    rng = np.random.default_rng(0xC1)
    lists = {}
    for t in used:  # lengths log-uniform in [1, 5000]
        n = int(np.exp(rng.uniform(0, np.log(5000))))
        docs = np.sort(rng.choice(N, size=max(1, min(n, N)), replace=False)).astype(np.uint32)
        lists[t] = (docs, (1 + rng.geometric(0.5, len(docs)).clip(max=255) - 1).astype(np.uint32).clip(min=1))
    dummy = (np.array([0], np.uint32), np.array([1], np.uint32))
    base = str(tmp_path_factory.mktemp("c1") / "test_collection")
    with open(base + ".docs", "wb") as fd, open(base + ".freqs", "wb") as ff:
        np.array([1, N], dtype=np.uint32).tofile(fd)
        for t in range(V):
            docs, freqs = lists.get(t, dummy)
            np.array([len(docs)], dtype=np.uint32).tofile(fd)
            docs.tofile(fd)
            np.array([len(freqs)], dtype=np.uint32).tofile(ff)
            freqs.tofile(ff)
    with open(base + ".sizes", "wb") as fs:
        np.array([N], dtype=np.uint32).tofile(fs)
        sizes.tofile(fs)
    idx_path, wand_path = base + ".block_optpfor", base + ".wand"
    subprocess.check_call([TOOL, "block_optpfor", base, idx_path, wand_path])
    return dict(N=N, V=V, sizes=sizes, lists=lists, dummy=dummy, queries=queries, idx=open(idx_path, "rb").read(),
                wand=open(wand_path, "rb").read())


def test_tool_image_equals_library_image(c1):
    lists = [c1["lists"].get(t, c1["dummy"]) for t in range(c1["V"])]
    assert d.build_index("block_optpfor", c1["N"], lists) == c1["idx"]
    assert d.build_wand(c1["sizes"], lists) == c1["wand"]


def test_and_queries_cpu_reference_path(c1):
    idx = o.Index("block_optpfor", c1["idx"], c1["wand"])
    assert idx.size() == c1["V"] and idx.num_docs() == c1["N"]
    lens = c1["sizes"].astype(np.float32)
    nl = (lens / np.float32(lens.astype(np.float64).sum() / c1["N"])).astype(np.float32)
    nonempty = 0
    for q in c1["queries"]:
        ts = sorted(set(q))
        exp = c1["lists"][ts[0]][0]
        for t in ts[1:]:
            exp = np.intersect1d(exp, c1["lists"][t][0], assume_unique=True)
        r = idx.query("and", q, want_matches=True)
        assert r["count"] == len(exp) and np.array_equal(r["matches"], exp)
        nonempty += len(exp) > 0
        # ranked_and: float32 BM25 in size-sorted order
        got = idx.query("ranked_and", q)
        if len(exp):
            ents = sorted(((len(c1["lists"][t][0]), t, q.count(t)) for t in ts), key=lambda e: e[0])
            score = np.zeros(len(exp), dtype=np.float32)
            for n, t, qtf in ents:
                docs, freqs = c1["lists"][t]
                w = query_term_weight(qtf, n, c1["N"])
                score = (score + w * doc_term_weight(freqs[np.searchsorted(docs, exp)], nl[exp])).astype(np.float32)
            np.testing.assert_allclose(got["topk"], np.sort(score)[::-1][:10], rtol=1e-5)
        else:
            assert got["count"] == 0
    assert nonempty > 50
