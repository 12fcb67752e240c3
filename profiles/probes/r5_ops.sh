#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_ops.sh <tag> [ops...]  -- round 5: the other operators on the GOV2-scale block_optpfor index, 20 steps each
set -u
TAG=${1:-ops}; shift
OPS=${@:-and and_freq or or_freq wand}
OUT=gpurun_out/r5_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for op in $OPS; do
  python bench.py --steps 20 --warmup 3 --op $op --no-cpu-baseline --no-oracle > $OUT/bench_$op.json 2> $OUT/bench_$op.err
  python - $OUT/bench_$op.json $op <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],2), "ms/step")
PY
done
