"""Where a wave of k_ranked_stream<NT> spends its cycles, for the 2 / 3 / 4-term queries of the default batch (needs the
diagnostic build: DS2I_BUILD_VARIANT=phase DS2I_EXTRA_CFLAGS=-DDS2I_RS_PHASE python ds2i_amd/build.py; run with
DS2I_LIB_VARIANT=phase). Shader cycles summed over waves; the clock reads themselves cost ~10 % and are spread evenly."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
codec = sys.argv[1] if len(sys.argv) > 1 else "block_optpfor"
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
names = [("unit", "unit setup (first window fill, first select, first prefetch)"), ("floor", "shared-histogram floor"), ("stream", "select next block (window ballots, refills)"),
         ("prefetch", "WAIT: block bytes + side slot of A"), ("docs", "prefetch issue + docs decode + prefix sums"), ("freqs", "freqs decode + freq-only bounds"),
         ("topk", "WAIT: list 1's table bytes of B"), ("member", "stage B tests (+ on-demand hint / weight gathers)"), ("score", "stage C (norm_len, list j search / decode / membership, heap)"),
         ("probe", "rotate + alive test + gather issue"), ("total", "unit epilogue")]
for nt, c in ((2, 0), (3, 1), (4, 1)):
    qs = [q for q in queries if len(set(q)) == nt]
    b = d.Batch(idx, "ranked_and", qs, k=10)
    b.run()
    st = b.run()
    ph = b.phase_cycles(c)
    s = st.as_dict()
    tot = sum(ph[k] for k, _ in names)
    print("%d terms: %d queries, %.2f ms kernel (instrumented), %d docs blocks, %.2f G wave cycles = %.0f cycles per docs block" % (nt, len(qs), s["kernel_ms"], s["docs_blocks_decoded"], tot / 1e9, tot / max(1, s["docs_blocks_decoded"])))
    for k, label in names:
        print("   %-70s %8.1f M cycles %5.1f %%" % (label, ph[k] / 1e6, 100.0 * ph[k] / max(1, tot)))
