import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ds2i_amd as d
for wl, p in (("c2", d.SynthParams(seed=0xD5210002, num_docs=1_000_000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128, clustered_every=4)),
              ("gov2", d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4))):
    img, wand, n = d.synth_build(p, "block_optpfor")
    idx = d.Index("block_optpfor", img, wand)
    queries = d.synth_queries(0x51E21, p.num_terms, 4096)
    idx.query_batch("ranked_and", queries, k=10)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); idx.query_batch("ranked_and", queries, k=10); ts.append(time.perf_counter() - t0)
    b = d.Batch(idx, "ranked_and", queries, k=10); b.set_instrumented(False); b.run()
    t0 = time.perf_counter(); b.run(); tr = time.perf_counter() - t0
    print("%s: ds2i_hip_query_batch (prepare + run + fetch, host buffers) %.2f ms; resident run %.2f ms" % (wl, 1e3 * min(ts), 1e3 * tr))
