"""FETCH_SIZE calibration: run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; prints the known byte count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
p = d.SynthParams(seed=0xD5210002, num_docs=4_000_000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128, clustered_every=4)
img, wand, n = d.synth_build(p, "block_optpfor")
idx = d.Index("block_optpfor", img, wand)
for _ in range(3):
    b = idx.calibration_read()
print("calibration bytes per launch:", b)
