#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_budget_sweep.sh -- what the upload-time tables buy per byte: range-table granularity (DS2I_RMW_G) x membership hints x side slots,
# default bench (GOV2 scale, ranked_and), resident bytes beside the rate
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_budget
run() { env "$@" python bench.py --no-oracle --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'q/s', round(d['ms_per_step'],3), 'ms/step', round(d['config'].get('device_bytes',0)/1e9,2), 'GB resident')"; }
{
for cfg in "A=1" "DS2I_NO_RMH=1" "DS2I_RMW_G=2" "DS2I_RMW_G=2 DS2I_NO_RMH=1" "DS2I_RMW_G=1" "DS2I_RMW_G=1 DS2I_NO_RMH=1" "DS2I_NO_XSLOTS=1" "DS2I_RMW_G=2 DS2I_NO_RMH=1 DS2I_NO_XSLOTS=1"; do
  echo "== $cfg"; run $cfg
done
} | tee gpurun_out/r5_budget/out.txt
