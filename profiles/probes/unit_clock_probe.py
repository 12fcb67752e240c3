"""Where a class kernel's time goes: runs one instrumented GOV2-scale batch with DS2I_UNIT_CLOCK=1 (the library prints,
per class, the kernel span, the sum of the unit times and the longest units) and describes the queries of those units.
usage (GPU box): python profiles/probes/unit_clock_probe.py [op] [codec]"""
import os, sys, re, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DS2I_UNIT_CLOCK", "1")
import bench, ds2i_amd as d
op = sys.argv[1] if len(sys.argv) > 1 else "wand"
codec = sys.argv[2] if len(sys.argv) > 2 else "block_optpfor"
W = bench.WORKLOADS["gov2"]
p = d.SynthParams(seed=W["seed"], num_docs=W["num_docs"], num_terms=W["num_terms"], zipf_exp=W["zipf_exp"], top_df_frac=W["top_df_frac"],
                  min_len=W["min_len"], clustered_every=W["clustered_every"])
img, wand, postings = d.synth_build(p, codec, os.cpu_count())
idx = d.Index(codec, img, wand)
qs = d.synth_queries(0x51E21 + 7919 * 3, p.num_terms, 4096)
b = d.Batch(idx, op, qs, k=10)
b.run()
st = b.run()
print("kernel_ms", st.as_dict()["kernel_ms"], flush=True)
# the library printed the longest units on stderr; describe a few heavy queries by list sizes
sizes = lambda q: sorted(idx.list_size(t) for t in set(q))
for q in (int(x) for x in os.environ.get("PROBE_QUERIES", "").split(",") if x):
    print("query", q, "terms", len(set(qs[q])), "list sizes", sizes(qs[q]))
import json
json.dump([[idx.list_size(t) for t in sorted(set(q))] for q in qs], open("gpurun_out/unit_clock_queries.json", "w"))
