"""GPU index encoder vs the host builder at GOV2 scale: same image, time of each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ds2i_amd as d
scale = sys.argv[1] if len(sys.argv) > 1 else "gov2"
P = {"c2": dict(seed=0xD5210002, num_docs=1000000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128),
     "gov2": dict(seed=0xD5210004, num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096)}[scale]
p = d.SynthParams(clustered_every=4, **P)
t0 = time.time(); gi, gw, gn, info = d.synth_build_gpu(p); tg = time.time() - t0
t0 = time.time(); hi, hw, hn = d.synth_build(p, "block_optpfor"); th = time.time() - t0
print("%s: %d postings, image %.1f MB; GPU path %.1fs total (generate %.1fs on host threads, encode kernels %.1f ms); host builder %.1fs; identical: %s"
      % (scale, gn, len(gi) / 1e6, tg, info["generate_s"], info["device_ms"], th, gi == hi and gw == hw))
