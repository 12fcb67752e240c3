"""block_mixed: fixed per-block policy vs the profile-driven optimiser (ds2i_hybrid_*) at equal size, on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210007, num_docs=5_000_000, num_terms=8192, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
t0 = time.time()
lists = [d.synth_list(p, t) for t in range(p.num_terms)]
sizes = d.synth_doc_sizes(p)
print("lists: %d postings, %.1fs" % (sum(len(a) for a, _ in lists), time.time() - t0), flush=True)
wand = d.build_wand(sizes, lists)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)

def bench(img, codec="block_mixed", profile=False):
    idx = d.Index(codec, img, wand)
    b = d.Batch(idx, "ranked_and", queries, k=10)
    prof = None
    if profile:
        b.enable_block_profile(); b.run(); prof = b.block_profile()
    b.set_instrumented(False)
    b.run(); b.run()
    t0 = time.perf_counter()
    for _ in range(10): b.run()
    dt = (time.perf_counter() - t0) / 10
    res = b.fetch()
    b.close()
    return 4096 / dt, prof, res

t0 = time.time(); img_fixed = d.build_index("block_mixed", p.num_docs, lists); print("fixed policy build %.1fs" % (time.time() - t0), flush=True)
img_pfor = d.build_index("block_optpfor", p.num_docs, lists)
qps_fixed, prof, res_fixed = bench(img_fixed, profile=True)
qps_pfor, _, res_pfor = bench(img_pfor, "block_optpfor")
hb = d.HybridBuilder(p.num_docs)
base = 0
for docs, freqs in lists:
    nb = (len(docs) + 127) // 128
    hb.add_posting_list(docs, freqs, prof[base:base + nb]); base += nb
t0 = time.time(); lo, hi = hb.analyse(); print("analyse %.1fs: payload min %.1f MB max %.1f MB" % (time.time() - t0, lo / 1e6, hi / 1e6), flush=True)
print("fixed-policy index %.1f MB: %.0f q/s | block_optpfor %.1f MB: %.0f q/s" % (len(img_fixed) / 1e6, qps_fixed, len(img_pfor) / 1e6, qps_pfor))
for label, budget in (("smallest", lo), ("same size as fixed policy", None), ("fastest", hi)):
    if budget is None:  # bisect the payload budget so that the image size matches the fixed-policy image
        a, bb = lo, hi
        for _ in range(12):
            mid = (a + bb) // 2
            img, info = hb.freeze(mid)
            if len(img) <= len(img_fixed): a = mid
            else: bb = mid
        budget = a
    img, info = hb.freeze(budget)
    qps, _, res = bench(img)
    assert np.array_equal(res[0], res_fixed[0]) and np.array_equal(res[2], res_fixed[2])
    print("optimised (%s): %.1f MB, model time %.3g, types %s: %.0f q/s" % (label, len(img) / 1e6, info["model_time"], info["type_counts"], qps), flush=True)
