"""What one block decode costs in wave instructions: one-term `or` / `or_freq` queries through k_union (every block of the
list decoded exactly once, nothing else to do), bitmaps off. Run under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_INSTS_VMEM_RD (profiles/probes/run_pmc_cmd.sh) and divide by the printed block counts.
usage: DS2I_NO_BITMAP_USE=1 decode_cost_probe.py [codec]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DS2I_NO_BITMAP_USE", "1")
import ds2i_amd as d
codec = sys.argv[1] if len(sys.argv) > 1 else "block_optpfor"
p = d.SynthParams(seed=0xD5210004, num_docs=25_000_000, num_terms=2048, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
qs = [[t] for t in range(8, 520)]
for op in ("or", "or_freq"):
    b = d.Batch(idx, op, qs, k=10)
    b.run()
    st = b.run().as_dict()
    print(op, {k: st[k] for k in ("kernel_ms", "docs_blocks_decoded", "freqs_blocks_decoded", "rounds")}, flush=True)
