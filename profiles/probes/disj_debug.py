"""Debug helper: replays one fuzz collection through the union operators and prints the queries whose result differs
from the oracle (run on the GPU box; DS2I_NO_BMW_PRUNE / DS2I_NO_SKIPTAB switch the pruning paths off)."""
import sys
import numpy as np
import ds2i_amd as d
import oracle as o

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(1000 + seed)
nd = int(rng.integers(3000, 120000))
nt = int(rng.integers(20, 200))
p = d.SynthParams(seed=0xF00D0000 + seed, num_docs=nd, num_terms=nt, zipf_exp=float(rng.uniform(0.3, 1.0)),
                  top_df_frac=float(rng.uniform(0.2, 0.9)), min_len=int(rng.integers(1, 400)), clustered_every=int(rng.integers(0, 5)))
lists = [d.synth_list(p, t) for t in range(nt)]
sizes = d.synth_doc_sizes(p)
if seed % 2 == 0:
    sizes = np.where(rng.random(nd) < 0.1, 1, sizes).astype(np.uint32)
wand = d.build_wand(sizes, lists)
qs = d.synth_queries(0xABC0 + seed, nt, 300)
qs += [[int(t)] for t in rng.integers(0, nt, 20)] + [[0, 1], [0, 1, 2], [nt - 1, 0], list(range(min(nt, 6)))]
qs += [[int(x) for x in rng.integers(0, min(nt, 12), int(rng.integers(2, 5)))] for _ in range(80)]
print("docs", nd, "terms", nt)
for codec in sys.argv[2:] or ["block_optpfor"]:
    img = d.build_index(codec, nd, lists)
    gidx = d.Index(codec, img, wand)
    oidx = o.Index(codec, img, wand)
    for k in (1, 2, 10, 64):
        oc, otopk, otlen, _, _ = oidx.query_batch("ranked_or", qs, k=k)
        for op in ("wand", "maxscore", "ranked_or"):
            gc, gtopk, gtlen, _ = gidx.query_batch(op, qs, k=k)
            bad = [i for i in range(len(qs)) if gtlen[i] != otlen[i] or not np.allclose(gtopk[i], otopk[i], rtol=1e-5, atol=0)]
            print(codec, k, op, "bad:", len(bad))
            for i in bad[:4]:
                print("   q", i, qs[i], [len(lists[t][0]) for t in qs[i]], "gpu", gtlen[i], gtopk[i][:4], "oracle", otlen[i], otopk[i][:4])
