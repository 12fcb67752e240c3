#!/usr/bin/env python
"""`and` through the stream pipeline (k_ranked_stream<n, ., AND = true>, n = 2..8; one-term queries and all-dense queries: k_and_stream) against the
oracle: counts of random and adversarial conjunctions over random collections -- short and long lists paired (ranges wider than 254 doc-ids: the hint is
no proof there), dense lists (one doc-id per table entry), clustered lists (several postings per range: hint 255), whole queries and queries split into
parts, one-term queries. Run as a subprocess by tests/test_gpu.py (the library's knobs are read once per process):
`[DS2I_UNIT_CAP=8 | DS2I_NO_RMH=1 | DS2I_RMW_G=1 | DS2I_NO_RANKED_STREAM=1] python tests/and_stream_probe.py [seeds]`. The oracle is the checker here, nothing else."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ds2i_amd as d  # noqa: E402
import oracle as o  # noqa: E402


def one(seed):
    rng = np.random.default_rng(9000 + seed)
    nd = int(rng.integers(20000, 600000))
    nt = int(rng.integers(30, 160))
    p = d.SynthParams(seed=0xA2D00000 + seed, num_docs=nd, num_terms=nt, zipf_exp=float(rng.uniform(0.3, 1.1)),
                      top_df_frac=float(rng.uniform(0.2, 0.95)), min_len=int(rng.integers(1, 300)), clustered_every=int(rng.integers(0, 4)))
    lists = [d.synth_list(p, t) for t in range(nt)]
    sizes = d.synth_doc_sizes(p)
    wand = d.build_wand(sizes, lists)
    qs = []
    for n in list(range(2, 9)) + [9, 12, 16]:
        qs += [sorted(set(int(x) for x in rng.integers(0, nt, n + 2)))[:n] for _ in range(40)]   # anywhere in the vocabulary: short lists among them
        qs += [[int(x) for x in rng.permutation(min(nt, 14))[:n]] for _ in range(40)]              # the densest lists
        qs += [[int(x) for x in rng.permutation(nt)[-min(nt, 20):][:n]] for _ in range(20)]        # only short lists (wide ranges)
    qs += [[0, nt - 1], [nt - 1, nt - 2], [0, 1], list(range(8))]
    qs = [q for q in qs if len(q) >= 2]
    qs += [[0], [nt - 1], [nt // 2], [3, 3]]  # one list (a repeated term is one list)
    img = d.build_index("block_optpfor", nd, lists)
    gidx = d.Index("block_optpfor", img, wand)
    oidx = o.Index("block_optpfor", img, wand)
    streamed = set()
    for op in ("and", "and_freq"):
        oc, _, _, ofs, _ = oidx.query_batch(op, qs)
        b = d.Batch(gidx, op, qs)
        b.run()
        gc, _, _, gfs = b.fetch()
        for c in range(4):
            streamed |= set(g["lists"] for g in b.class_groups(c) if g["pipelined_stream"])
        b.close()
        assert np.array_equal(gc, oc), (seed, op, np.argwhere(gc != oc)[:5], gc[gc != oc][:5], oc[gc != oc][:5])
        if op == "and_freq":
            assert np.array_equal(gfs, ofs), (seed, "freq checksum", np.argwhere(gfs != ofs)[:5], gfs[gfs != ofs][:5], ofs[gfs != ofs][:5])
        bm = d.Batch(gidx, op, qs, want_matches=True)  # the doc-id lists through the same kernels
        bm.run()
        mc, _, _, mfs = bm.fetch()
        got = bm.fetch_matches(mc)
        bm.close()
        assert np.array_equal(mc, oc) and (op == "and" or np.array_equal(mfs, ofs)), (seed, op, "want_matches counts")
        for i in range(0, len(qs), 7):
            exp = oidx.query("and", qs[i], want_matches=True)["matches"]
            assert np.array_equal(got[i], exp), (seed, op, qs[i], len(got[i]), len(exp))
        pipe = d.Pipeline(gidx, depth=2)
        t = pipe.submit(op, qs)
        pc, _, _ = pipe.wait(t)
        pipe.close()
        assert np.array_equal(pc, oc), (seed, op, "pipelined")
    if not os.environ.get("DS2I_NO_RANKED_STREAM"):
        assert streamed, "the stream kernel did not run"
    print("seed %d: %d docs, %d terms, %d queries (%d non-empty, %d results), stream groups for %s lists: and / and_freq counts (+ freq checksums) equal the oracle's" %
          (seed, nd, nt, len(qs), int((oc > 0).sum()), int(oc.sum()), sorted(streamed)))


if __name__ == "__main__":
    for s in ([int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]):
        one(s)
    print("and_rstream_probe ok")
