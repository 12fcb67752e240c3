"""Cache lines (128 B) requested by k_ranked_stream's gathers, by purpose, for the 2 / 3 / 4-term queries of the default
batch (needs the diagnostic build: DS2I_BUILD_VARIANT=lines DS2I_EXTRA_CFLAGS=-DDS2I_LINE_COUNT python ds2i_amd/build.py;
run with DS2I_LIB_VARIANT=lines)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
codec = sys.argv[1] if len(sys.argv) > 1 else "block_optpfor"
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
queries = d.synth_queries(0x51E21, p.num_terms, 4096)
names = {"topk": "table bytes of list 1", "freqs": "table bytes of lists 2..", "member": "hint bytes", "score": "norm_lens", "docs": "blocks of lists 1.. (docs window)",
         "probe": "freqs windows of lists 1..", "prolog": "window fills (skip rows, block weights, span maxima)", "stream": "list-0 block bytes"}
for nt, c in ((2, 0), (3, 1), (4, 1)):
    qs = [q for q in queries if len(set(q)) == nt]
    b = d.Batch(idx, "ranked_and", qs, k=10)
    b.run()
    st = b.run()
    ph = b.phase_cycles(c)
    s = st.as_dict()
    tot = sum(ph[k] for k in names) + 6 * ph["find"]
    print("%d terms: %d queries, %.2f ms kernel, algorithmic bytes %.1f MB; lines requested %.1f M = %.2f GB" % (nt, len(qs), s["kernel_ms"], s["algorithmic_bytes"] / 1e6, tot / 1e6, tot * 128 / 1e9))
    for k, label in names.items():
        print("   %-55s %10.2f M lines  %5.1f %%" % (label, ph[k] / 1e6, 100.0 * ph[k] / tot))
    print("   %-55s %10.2f M lines  %5.1f %%  (%d searches x 6)" % ("block searches in lists 1..", 6 * ph["find"] / 1e6, 600.0 * ph["find"] / tot, ph["find"]))
    print("   list-1 lines by range width: 1 doc-id per byte %.2f M, 2: %.2f M, 4: %.2f M, 8-16: %.2f M, wider: %.2f M" % tuple(ph[k] / 1e6 for k in ("total", "insert", "prefetch", "floor", "unit")))
    print("   candidates alive at the gather (own freq bound + other lists' maxima can enter) %d in %d blocks that gathered" % (ph["n_alive"], ph["n_gblocks"]))
    print("   candidates %d, after the table bytes %d, after the hints %d, blocks with a stage C %d, list-j blocks decoded %d, freqs windows %d, heap offers %d"
          % (ph["n_visit"], ph["n_surv1"], ph["n_surv2"], ph["n_liverounds"], ph["n_bdocs"], ph["n_bfreqs"], ph["n_heap"]))
