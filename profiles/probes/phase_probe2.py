"""Per-phase cycle split + event counts of the ranked conjunctive kernels by exact term count
(needs the diagnostic build: DS2I_BUILD_VARIANT=phase DS2I_EXTRA_CFLAGS=-DDS2I_PHASE_TIMING python ds2i_amd/build.py;
run with DS2I_LIB_VARIANT=phase). usage: phase_probe2.py [codec] [op]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
codec = sys.argv[1] if len(sys.argv) > 1 else "block_optpfor"
op = sys.argv[2] if len(sys.argv) > 2 else "ranked_and"
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
groups = [("1 term", lambda n: n == 1, 0), ("2 terms", lambda n: n == 2, 0), ("3 terms", lambda n: n == 3, 1), ("4 terms", lambda n: n == 4, 1),
          ("5-8 terms", lambda n: 5 <= n <= 8, 2)]
for name, pred, c in groups:
    qs = [q for q in queries if pred(len(set(q)))]
    b = d.Batch(idx, op, qs, k=10)
    b.run(); b.run()
    t0 = time.perf_counter()
    st = b.run()
    dt = time.perf_counter() - t0
    ph = b.phase_cycles(c)
    tot = max(1, ph["total"])
    s = st.as_dict()
    print("%s: %d queries %.2f ms docs %d freqs %d rounds %d scored %d" % (name, len(qs), 1e3 * dt, s["docs_blocks_decoded"], s["freqs_blocks_decoded"], s["rounds"], s["postings_scored"]))
    print("   cycles: " + " ".join("%s %.1f%%" % (k, 100.0 * v / tot) for k, v in ph.items() if not k.startswith("n_") and k not in ("total", "insert")))
    print("   events: dead_rounds %d " % ph["insert"] + " ".join("%s %d" % (k, v) for k, v in ph.items() if k.startswith("n_")))
    print("   cycles/visit %.0f" % (tot / max(1, ph["n_visit"])))
