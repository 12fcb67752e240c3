"""Solo kernel time of the 2 / 3 / 4-term query groups of the default batch (uninstrumented, nothing else running)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
for nt in (1, 2, 3, 4, (5, 8), (9, 16)):
    lo, hi = (nt, nt) if isinstance(nt, int) else nt
    qs = [q for q in queries if lo <= len(set(q)) <= hi]
    b = d.Batch(idx, "ranked_and", qs, k=10)
    b.set_instrumented(False)
    b.run(); b.run()
    ms = sorted(b.run().as_dict()["kernel_ms"] for _ in range(7))
    print("%s terms: %d queries, kernel %.2f ms (median of 7, min %.2f)" % (nt, len(qs), ms[3], ms[0]))
