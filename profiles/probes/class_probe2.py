"""GOV2-scale ranked_and: each kernel class alone vs all classes together (resident batch), units per class."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
cls_of = lambda n: 0 if n <= 2 else 1 if n <= 4 else 2 if n <= 8 else 3
op = sys.argv[1] if len(sys.argv) > 1 else "ranked_and"
def timeit(qs, tag):
    b = d.Batch(idx, op, qs, k=10)
    b.set_instrumented(False)
    b.run(); b.run()
    t0 = time.perf_counter()
    for _ in range(5):
        b.run()
    dt = (time.perf_counter() - t0) / 5
    ms = [b.class_stats(c)[0].kernel_ms for c in range(4)]
    print("%s: %d queries, %.2f ms/step, class kernel ms %s" % (tag, len(qs), 1e3 * dt, ["%.2f" % m for m in ms]), flush=True)
    b.close()
timeit(queries, "all classes")
for c in range(4):
    timeit([q for q in queries if cls_of(len(set(q))) == c], "class %d alone" % c)
for nt in (1, 2, 3, 4, 5, 6):
    timeit([q for q in queries if len(set(q)) == nt], "%d-term queries alone" % nt)
