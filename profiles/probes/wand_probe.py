import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
for op in ("wand", "maxscore"):
    t0 = time.perf_counter(); b = d.Batch(idx, op, queries, k=10); t1 = time.perf_counter()
    if os.environ.get("PROBE_UNINSTRUMENTED"): b.set_instrumented(False)
    print(op, "prepare %.1f ms" % (1e3 * (t1 - t0)))
    for i in range(3):
        t0 = time.perf_counter(); st = b.run(); t1 = time.perf_counter()
        print(op, "run wall %.1f ms, stats.kernel_ms %.1f" % (1e3 * (t1 - t0), st.kernel_ms), [round(b.class_stats(c)[0].kernel_ms, 1) for c in range(4)], [b.class_stats(c)[1] for c in range(4)], flush=True)
    for c in range(4):
        st, nqc = b.class_stats(c)
        s = st.as_dict()
        print("   class %d: %d queries %.1f ms docs %d freqs %d bm %d scored %d rounds %d" % (c, nqc, s["kernel_ms"], s["docs_blocks_decoded"], s["freqs_blocks_decoded"], s["block_max_examined"], s["postings_scored"], s["rounds"]))
