"""Where the end-to-end step goes at configs[1] scale: host time inside submit() / wait() vs the step time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ds2i_amd as d
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
P = {"c2": dict(seed=0xD5210002, num_docs=1000000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128),
     "gov2": dict(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096)}[wl]
p = d.SynthParams(clustered_every=4, **P)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
flat = [d.flatten_queries(d.synth_queries(0x51E21 + 7919 * i, p.num_terms, 4096)) for i in range(43)]
for depth in (1, 2, 3, 4):
    pipe = d.Pipeline(idx, depth=depth)
    ts, tw = [], []
    tickets = []
    def reap():
        t0 = time.perf_counter(); pipe.wait(tickets.pop(0)); tw.append(time.perf_counter() - t0)
    for i in range(3):
        tickets.append(pipe.submit("ranked_and", flat[i], k=10))
        if len(tickets) == depth: reap()
    while tickets: reap()
    ts.clear(); tw.clear()
    t00 = time.perf_counter()
    for i in range(3, 43):
        if len(tickets) == depth: reap()
        t0 = time.perf_counter(); tickets.append(pipe.submit("ranked_and", flat[i], k=10)); ts.append(time.perf_counter() - t0)
    while tickets: reap()
    tot = time.perf_counter() - t00
    print("%s depth %d: %.3f ms/step; submit() mean %.3f ms, wait() mean %.3f ms" % (wl, depth, 1e3 * tot / 40, 1e3 * np.mean(ts), 1e3 * np.mean(tw)), flush=True)
    pipe.close()
