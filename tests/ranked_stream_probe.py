#!/usr/bin/env python
"""ranked_and through the stream pipeline (k_ranked_stream<cap>, cap = list capacity 2 | 4 | 6 | 8 of a launch group: ranked_stream.hip) against the
oracle, bit for bit: random collections, queries of 2..8 distinct terms (dense lists among them: non-empty intersections of many lists), k = 1 / 10 /
64, one-shot and pipelined, whole and split queries. Run as a subprocess by tests/test_gpu.py (the library's knobs are read once per process):
`[DS2I_STREAM_NT_MAX=n] [DS2I_UNIT_CAP=8] python tests/ranked_stream_probe.py [seeds]`. The oracle is the checker here, nothing else."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ds2i_amd as d  # noqa: E402
import oracle as o  # noqa: E402


def one(seed):
    rng = np.random.default_rng(7000 + seed)
    nd = int(rng.integers(20000, 400000))
    nt = int(rng.integers(30, 120))
    p = d.SynthParams(seed=0x5EED0000 + seed, num_docs=nd, num_terms=nt, zipf_exp=float(rng.uniform(0.3, 0.9)),
                      top_df_frac=float(rng.uniform(0.3, 0.9)), min_len=int(rng.integers(1, 300)), clustered_every=int(rng.integers(0, 5)))
    lists = [d.synth_list(p, t) for t in range(nt)]
    sizes = d.synth_doc_sizes(p)
    if seed % 2 == 0:
        sizes = np.where(rng.random(nd) < 0.1, 1, sizes).astype(np.uint32)
    wand = d.build_wand(sizes, lists)
    qs = []
    for n in list(range(2, 9)) + [9, 11, 13, 16]:  # every list count up to 8, a few beyond: terms anywhere in the vocabulary (mostly empty intersections, early exits) ...
        qs += [sorted(set(int(x) for x in rng.integers(0, nt, n + 2)))[:n] for _ in range(40)]
        qs += [[int(x) for x in rng.permutation(min(nt, 14 if n <= 8 else 22))[:n]] for _ in range(60 if n <= 8 else 15)]  # ... and among the densest lists (big intersections)
    qs += [list(range(8)), list(range(7, -1, -1)), [0, 1, 2, 3, 4], [nt - 1, 0, 1, 2, 3, 4], [0], [nt - 1], [3, 3], []]
    img = d.build_index("block_optpfor", nd, lists)
    gidx = d.Index("block_optpfor", img, wand)
    oidx = o.Index("block_optpfor", img, wand)
    pipe = d.Pipeline(gidx, depth=2)
    streamed = set()
    for k in (1, 10, 64, 100, 700):  # (beyond 64: the 4- and 16-scores-per-lane instantiations, which also take the one-term queries)
        oc, otopk, otlen, _, _ = oidx.query_batch("ranked_and", qs, k=k)
        b = d.Batch(gidx, "ranked_and", qs, k=k)
        b.run()
        gc, gtopk, gtlen, _ = b.fetch()
        for c in range(4):
            streamed |= set(g["lists"] for g in b.class_groups(c) if g["pipelined_stream"])
        b.close()
        assert np.array_equal(gc, oc) and np.array_equal(gtlen, otlen), (seed, k, np.argwhere(gc != oc)[:3])
        assert np.array_equal(gtopk, otopk), (seed, k, np.argwhere(gtopk != otopk)[:3])
        t = pipe.submit("ranked_and", qs, k=k)
        _, ptopk, _ = pipe.wait(t)
        assert np.array_equal(ptopk, otopk), (seed, k, "pipelined")
    pipe.close()
    # launch groups by list capacity (2 | 3-4 | 5-6 | 7-8)
    want = {2, 4, 6, 8, 16} if int(os.environ.get("DS2I_STREAM_NT_MAX", "16")) > 8 else {2, 4, 6, 8} if int(os.environ.get("DS2I_STREAM_NT_MAX", "16")) > 4 else {2, 4}
    assert streamed == want, (streamed, want)
    nonempty = int((oc > 0).sum())
    print("seed %d: %d docs, %d terms, %d queries (%d with results), stream kernels for %s lists: bit-identical to the oracle" %
          (seed, nd, nt, len(qs), nonempty, sorted(streamed)))


if __name__ == "__main__":
    for s in ([int(x) for x in sys.argv[1:]] or [1, 2, 3]):
        one(s)
    print("rs_nt8_probe ok")
