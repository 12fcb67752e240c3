"""Decode-only probe: k_decode_list over the longest lists (docs+freqs of every block). Run under rocprofv3 --pmc."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
codec = sys.argv[1] if len(sys.argv) > 1 else "block_optpfor"
p = d.SynthParams(seed=0xD5210002, num_docs=4_000_000, num_terms=4096, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4)
img, wand, n = d.synth_build(p, codec)
idx = d.Index(codec, img, wand)
blocks = 0
t0 = time.time()
for t in range(0, 64):
    docs, freqs = idx[t]
    blocks += (len(docs) + 127) // 128
print("lists 0..63: %d blocks (x2 decodes), %.3fs wall incl. copies" % (blocks, time.time() - t0))
