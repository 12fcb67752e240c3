"""Where a wave of k_union_stream<cap> spends its cycles, for the queries of each list count of the GOV2-scale wand batch (needs the diagnostic
build: DS2I_BUILD_VARIANT=usphase DS2I_EXTRA_CFLAGS=-DDS2I_US_PHASE python ds2i_amd/build.py; run with DS2I_LIB_VARIANT=usphase). Shader cycles
summed over waves + event counts. usage (GPU box): DS2I_LIB_VARIANT=usphase python profiles/probes/us_phase_probe.py [counts e.g. 2,3,6,8]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, ds2i_amd as d
W = bench.WORKLOADS["gov2"]
p = d.SynthParams(seed=W["seed"], num_docs=W["num_docs"], num_terms=W["num_terms"], zipf_exp=W["zipf_exp"], top_df_frac=W["top_df_frac"],
                  min_len=W["min_len"], clustered_every=W["clustered_every"])
img, wand, n = d.synth_build(p, "block_optpfor", os.cpu_count())
idx = d.Index("block_optpfor", img, wand)
queries = d.synth_queries(0x51E21 + 7919 * 3, p.num_terms, 4096)
names = [("unit", "unit setup"), ("prefetch", "WAIT: block bytes + side slot of A"), ("stream", "floor word + select next block + prefetch issue"),
         ("prolog", "stage A: decode of docs + freqs (optpfor_decode_pair / tail)"), ("docs", "stage A: prefix sums, own bounds, freqs parked"), ("topk", "WAIT: list 1's bytes of B"), ("member", "stage B: list 1 + further lists' bytes"),
         ("freqs", "stage B: bound test + optional lists' hints"), ("score", "stage C: norm_len + exact driver score"), ("probe", "stage C: lists 1.. (search / decode / membership)"),
         ("insert", "heap inserts + floor publication"), ("find", "rotate + alive test + gather issue"), ("total", "unit epilogue")]
counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,6,8").split(",")]
for nt in counts:
    c = 0 if nt <= 2 else 1 if nt <= 4 else 2 if nt <= 8 else 3
    qs = [q for q in queries if len(set(q)) == nt]
    if not qs:
        continue
    b = d.Batch(idx, "wand", qs, k=10)
    b.run()
    st = b.run()
    ph = b.phase_cycles(c)
    s = st.as_dict()
    tot = sum(ph[k] for k, _ in names)
    blocks = max(1, ph["n_gblocks"])
    print("%d terms: %d queries, %.2f ms kernel (instrumented), %d docs blocks (all lists), %d stage-B blocks, %.2f G wave cycles = %.0f cycles per stage-B block" %
          (nt, len(qs), s["kernel_ms"], s["docs_blocks_decoded"], blocks, tot / 1e9, tot / blocks))
    for k, label in names:
        print("   %-70s %8.1f M cycles %5.1f %%" % (label, ph[k] / 1e6, 100.0 * ph[k] / max(1, tot)))
    print("   per stage-B block: alive %.1f, after bytes %.1f, after hints %.1f; blocks reaching stage C %.3f; list searches %.3f, list blocks decoded %.3f, heap inserts tried %.3f" %
          (ph["n_alive"] / blocks, ph["n_surv1"] / blocks, ph["n_surv2"] / blocks, ph["n_liverounds"] / blocks, ph["n_visit"] / blocks, ph["n_bdocs"] / blocks, ph["n_heap"] / blocks))
