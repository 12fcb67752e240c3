"""Solo kernel time of the query groups (by distinct terms) of the default batch (uninstrumented, nothing else running).
usage: group_ms.py [op]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
op = sys.argv[1] if len(sys.argv) > 1 else "ranked_and"
for nt in (1, 2, 3, 4, (5, 8), (9, 16)):
    lo, hi = (nt, nt) if isinstance(nt, int) else nt
    qs = [q for q in queries if lo <= len(set(q)) <= hi]
    b = d.Batch(idx, op, qs, k=10)
    b.set_instrumented(False)
    b.run(); b.run()
    ms = sorted(b.run().as_dict()["kernel_ms"] for _ in range(7))
    print("%s %s terms: %d queries, kernel %.2f ms (median of 7, min %.2f), %.2f us per query" % (op, nt, len(qs), ms[3], ms[0], 1e3 * ms[3] / max(1, len(qs))))
