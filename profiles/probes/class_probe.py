"""Times each LDS class of the GOV2-scale ranked_and batch on its own (standalone kernel time per class)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
cls_of = lambda n: 0 if n <= 2 else 1 if n <= 4 else 2 if n <= 8 else 3
for c in range(3):
    qs = [q for q in queries if cls_of(len(set(q))) == c]
    b = d.Batch(idx, "ranked_and", qs, k=10)
    b.run(); b.run()
    t0 = time.perf_counter()
    for _ in range(5):
        st = b.run()
    dt = (time.perf_counter() - t0) / 5
    print("class %d alone: %d queries, %.2f ms/step, stats %s" % (c, len(qs), 1e3 * dt, {k: v for k, v in st.as_dict().items() if k in ("docs_blocks_decoded", "freqs_blocks_decoded", "rounds", "block_max_examined")}))
    # per-query-size split for class 1
    if c == 1:
        for nt in (3, 4):
            q2 = [q for q in qs if len(set(q)) == nt]
            b2 = d.Batch(idx, "ranked_and", q2, k=10); b2.run()
            t0 = time.perf_counter(); st = b2.run(); print("   %d-term: %d queries %.2f ms" % (nt, len(q2), 1e3 * (time.perf_counter() - t0)))
        # the slowest single queries
        import numpy as np
        ts = []
        for q in qs[:400]:
            b3 = d.Batch(idx, "ranked_and", [q], k=10)
            t0 = time.perf_counter(); b3.run(); ts.append(time.perf_counter() - t0); b3.close()
        ts = np.array(ts) * 1e3
        order = np.argsort(ts)[::-1][:5]
        print("   slowest single queries (ms):", [(round(float(ts[i]), 2), [idx.list_size(t) for t in sorted(set(qs[i]))]) for i in order])
