#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_full.sh <tag>  -- the whole -m gpu suite + the default bench
set -u
TAG=${1:-full}
OUT=gpurun_out/r5_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
python bench.py --steps 60 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY' $OUT
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("default", round(d["value"]), d["ms_per_step"], r.get("frac"), r.get("kernel"), r.get("kernel_ms"), d.get("cpu_baseline",{}).get("value"))
PY
