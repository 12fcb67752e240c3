"""Does the rate of a device index depend on what the process did on the device BEFORE it was uploaded? (round 5: a
block_optpfor index that was transcoded from another image ran ~18 % below the same index uploaded directly.)
Usage (GPU box, repo root): python profiles/probes/reopen_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import ds2i_amd as d

p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
t0 = time.time()
img, wand, postings = d.synth_build(p, "block_optpfor", 0)
print("built: %d postings, %.1f MB, %.1fs" % (postings, len(img) / 1e6, time.time() - t0), flush=True)
queries = d.synth_queries(0x51E21 + 7919 * 5, p.num_terms, 4096)
NCLS = 5


def rate(idx, tag, steps=20):
    b = d.Batch(idx, "ranked_and", queries, k=10)
    b.set_instrumented(False)
    for _ in range(3):
        b.run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    ms = [0.0] * NCLS
    for _ in range(steps):
        b.run()
        for c in range(NCLS):
            ms[c] += b.class_stats(c)[0].kernel_ms
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print("%-46s %8.0f q/s resident, %.3f ms/batch, class kernel ms %s" % (tag, len(queries) / dt, 1e3 * dt, [round(x / steps, 2) for x in ms]), flush=True)
    b.close()


mode = sys.argv[1] if len(sys.argv) > 1 else "all"
t0 = time.time()
a = d.Index("block_optpfor", img, wand)
print("upload %.1fs" % (time.time() - t0), flush=True)
rate(a, "A: first upload of the process")
rate(a, "A again")
a.close()
b = d.Index("block_optpfor", img, wand)
rate(b, "B: second upload, first one closed")
c = d.Index("block_optpfor", img, wand)
rate(c, "C: third upload while B is open")
rate(b, "B again (C open)")
b.close()
c.close()
if mode == "all":
    # what a transcoding upload does: a bare index + per-list decode buffers + the encoder's buffers come and go first
    img_o, wand_o, _ = d.synth_build(p, "opt", 0)
    os.environ["DS2I_PEF_NATIVE"] = "1"
    o = d.Index("opt", img_o, wand_o)
    rate(o, "opt native")
    o.close()
    del os.environ["DS2I_PEF_NATIVE"]
    e = d.Index("block_optpfor", img, wand)
    rate(e, "E: block_optpfor after an opt index came and went")
    e.close()
    t0 = time.time()
    tt = d.Index("opt", img_o, wand_o)
    print("transcoding upload %.1fs" % (time.time() - t0), flush=True)
    rate(tt, "T: opt transcoded")
    f = d.Index("block_optpfor", img, wand)
    rate(f, "F: block_optpfor uploaded after T (T open)")
    rate(tt, "T again")
