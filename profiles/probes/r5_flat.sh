#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_flat.sh -- equal stream priorities (the default since round 5) against DS2I_CLASS_PRIORITY=1, per operator
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_flat
run() { python bench.py --no-oracle --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)))"; }
{
for args in "--op wand --steps 30 --warmup 3" "--op and --steps 30 --warmup 3" "--op or_freq --steps 20 --warmup 3" "--batch 512 --depth 8 --steps 160 --warmup 80" "--codec opt --steps 40 --warmup 5" "--steps 40 --warmup 5"; do
  echo "== $args : equal priorities"; run $args
  echo "== $args : DS2I_CLASS_PRIORITY=1"; DS2I_CLASS_PRIORITY=1 run $args
done
} | tee gpurun_out/r5_flat/out.txt
