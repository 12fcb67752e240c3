"""Generates tests/golden/bm25_reference.json: inputs and the float32 outputs of the REFERENCE scorer
(/root/reference/bm25.hpp compiled as-is into oracle/_ref/libbm25_ref.so by oracle/Makefile), as bit patterns.
Run in the build container (needs /root/reference); the fixture travels, the reference does not.
Grid: freqs 1..255 and a few huge ones x norm_lens spanning the collection's doc sizes (1 .. 61081 over a mean of
1770) for doc_term_weight; qtf 1..3 x df over 7 decades x the configs' num_docs for query_term_weight."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as o

o.build()
R = o.ref_bm25()
assert R is not None, "oracle/_ref/libbm25_ref.so missing (needs /root/reference)"
rng = np.random.default_rng(2025)
freqs = np.concatenate([np.arange(1, 256), [256, 1000, 65535, 1 << 20, (1 << 31) - 1]]).astype(np.uint64)
nls = np.concatenate([np.float32([1 / 1770.0, 0.01, 0.5, 1.0, 2.0, 34.5, 61081 / 1770.0]),
                      (rng.lognormal(0.0, 1.0, 40)).astype(np.float32)]).astype(np.float32)
dtw = []
for f in freqs:
    for nl in nls:
        dtw.append([int(f), int(np.float32(nl).view(np.uint32)), int(np.float32(R.ref_bm25_doc_term_weight(int(f), float(nl))).view(np.uint32))])
qtw = []
for N in (10000, 1000000, 25000000, 50000000):
    dfs = sorted(set([1, 2, 3, 127, 128, 4096, N // 2 - 1, N // 2, N // 2 + 1, N - 1, N] + [int(x) for x in np.unique(np.geomspace(1, N, 60).astype(np.int64))]))
    for df in dfs:
        for qtf in (1, 2, 3):
            qtw.append([qtf, df, N, int(np.float32(R.ref_bm25_query_term_weight(qtf, df, N)).view(np.uint32))])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bm25_reference.json")
json.dump({"source": "reference bm25.hpp via oracle/_ref/libbm25_ref.so (float32 bit patterns)",
           "doc_term_weight": dtw, "query_term_weight": qtw}, open(path, "w"), separators=(",", ":"))
print(path, len(dtw), len(qtw), os.path.getsize(path))
