#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_b512.sh  -- strong-scaling proxy: the per-GPU share (512 queries) of a 4096-query batch over 8 GPUs, knob sweep
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" python bench.py --workload gov2 --batch 512 --steps 80 --warmup 8 --no-oracle $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)), d.get('step_ms_spread'))"; }
EXTRA="" run A=1
EXTRA="--depth 8" run A=1
EXTRA="--depth 16" run A=1
EXTRA="--depth 8" run DS2I_STREAM_SETS=1
EXTRA="--depth 8" run DS2I_UNIT_CAP=32
EXTRA="--depth 8" run DS2I_UNIT_CAP=16
EXTRA="--depth 8" run DS2I_UNIT_DIV_RMW=1
EXTRA="--depth 8" run DS2I_UNIT_DIV_RMW=1 DS2I_UNIT_CAP=32
EXTRA="--depth 8" run DS2I_UNIT_FACTOR=1 DS2I_UNIT_CAP=32
