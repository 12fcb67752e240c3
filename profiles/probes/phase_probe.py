"""Per-phase cycle split of the class kernels of one operator (usage: phase_probe.py [codec] [op]) (needs a -DDS2I_PHASE_TIMING build:
DS2I_EXTRA_CFLAGS=-DDS2I_PHASE_TIMING python ds2i_amd/build.py --force)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
codec = sys.argv[1] if len(sys.argv) > 1 else "block_optpfor"
op = sys.argv[2] if len(sys.argv) > 2 else "ranked_and"
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
cls_of = lambda n: 0 if n <= 2 else 1 if n <= 4 else 2 if n <= 8 else 3
names = ["total", "docs", "freqs", "find", "member", "score", "topk/floor", "prolog(incl find/docs)", "probe(incl find/docs/freqs)", "insert"]
for c in range(3):
    qs = [q for q in queries if cls_of(len(set(q))) == c]
    b = d.Batch(idx, op, qs, k=10)
    b.run(); b.run()
    t0 = time.perf_counter()
    st = b.run()
    dt = time.perf_counter() - t0
    ph = list(b.phase_cycles(c).values())
    tot = max(1, ph[0])
    s = st.as_dict()
    print("class %d: %d queries %.2f ms docs %d freqs %d rounds %d | " % (c, len(qs), 1e3 * dt, s["docs_blocks_decoded"], s["freqs_blocks_decoded"], s["rounds"]) +
          " ".join("%s %.1f%%" % (names[i], 100.0 * ph[i] / tot) for i in range(1, len(names))) +
          " | cycles/docs-decode %.0f cycles/freqs-decode %.0f cycles/round %.0f" % (ph[1] / max(1, s["docs_blocks_decoded"]), ph[2] / max(1, s["freqs_blocks_decoded"]), tot / max(1, s["rounds"])))
