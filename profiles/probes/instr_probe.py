"""Instruction-count probe: 1-term and dense 2-term queries per op (and / and_freq / ranked_and are three kernel
instantiations, so one rocprofv3 --pmc pass attributes SQ_INSTS_* to docs decode / freqs decode / scoring / membership).
usage: instr_probe.py <1|2> [codec]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
nt = int(sys.argv[1]); codec = sys.argv[2] if len(sys.argv) > 2 else "block_optpfor"
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=4096, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
if nt == 1:
    qs = [[t] for t in range(8, 72)]
else:
    qs = [[t, t + 64] for t in range(8, 72)]
for op in ("and", "and_freq", "ranked_and"):
    b = d.Batch(idx, op, qs, k=10)
    b.run()
    st = b.run().as_dict()
    print(op, nt, {k: st[k] for k in ("kernel_ms", "docs_blocks_decoded", "freqs_blocks_decoded", "rounds", "postings_scored")}, flush=True)
