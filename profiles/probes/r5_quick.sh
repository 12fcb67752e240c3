#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_quick.sh [tag]  -- round 5: parity subset for the side-slot decoder + a short default bench
set -u
TAG=${1:-q}
OUT=gpurun_out/r5_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "decode_every_list or adversarial or exception_count_sweep or fuzz_bit_identical or test_query_ops_match_oracle or gov2_scale_properties or uninstrumented" > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python - <<'PY' $OUT
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("default", round(d["value"]), d["ms_per_step"], r.get("frac"), r.get("kernel"), r.get("kernel_ms"))
for k in r.get("per_kernel",[]): print("  ",k)
PY
