#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_step2.sh  -- round 5: the freq_index transcode + non-slot decoder tests, a default bench, the opt lines
set -u
OUT=gpurun_out/r5_step2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "transcoded or without_side_tables or both_decoders or test_query_ops_match_oracle or fuzz_bit_identical or partition_shapes or recovers" > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY' $OUT
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("default", round(d["value"]), d["ms_per_step"], r.get("frac"), r.get("kernel"), r.get("kernel_ms"))
for k in r.get("per_kernel",[]): print("  ",k)
PY
bash profiles/probes/r5_final.sh opt
