#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r6_sweep_g.sh  -- range tables at 2 entries per posting against 4 for every operator (GOV2 scale), and the
# alternative-path + stream-pipeline tests of the reduced knob set
set -u
OUT=gpurun_out/r6_sweep
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu.py -x -q -k "alternative_paths or union_through or and_through or ranked_stream_5_to_8 or table_budget or transcoded or error_behaviour" > $OUT/pytest_knobs.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_knobs.txt
for op in ranked_and wand and and_freq or or_freq; do
  for g in 4 2; do
    DS2I_RMW_G=$g timeout 400 python bench.py --op $op --no-oracle --steps 40 --warmup 5 > $OUT/g${g}_$op.json 2> $OUT/g${g}_$op.err
    python - $OUT/g${g}_$op.json g${g}_$op <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", round(d["config"].get("device_bytes",0)/1e9,2), "GB", "ci", d.get("value_ci95",{}).get("rel"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
