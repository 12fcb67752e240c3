"""Which single queries are slow when run alone? (critical-path analysis of the many-list classes)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
cls_of = lambda n: 0 if n <= 2 else 1 if n <= 4 else 2 if n <= 8 else 3
for c in (1, 2):
    qs = [q for q in queries if cls_of(len(set(q))) == c]
    ts = []
    for q in qs:
        b = d.Batch(idx, "ranked_and", [q], k=10)
        b.set_instrumented(False)
        b.run()
        t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
        b.close()
    ts = np.array(ts) * 1e3
    order = np.argsort(ts)[::-1]
    print("class %d: %d queries; alone: sum %.1f ms, max %.2f ms, p99 %.2f, median %.3f" % (c, len(qs), ts.sum(), ts.max(), np.percentile(ts, 99), np.median(ts)))
    for i in order[:8]:
        print("   %.2f ms  lists %s" % (ts[i], sorted(idx.list_size(t) for t in set(qs[i]))))
