#!/bin/bash
export TMPDIR=/tmp
run() { echo "== $* $EXTRA"; env "$@" python bench.py --workload gov2 --batch 512 --no-oracle $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)), d.get('step_ms_spread'))"; }
EXTRA="--depth 8 --steps 160 --warmup 80" run DS2I_UNIT_DIV_RMW=1
EXTRA="--depth 8 --steps 160 --warmup 80" run DS2I_UNIT_DIV_RMW=1 DS2I_STREAM_SETS=1
EXTRA="--depth 4 --steps 160 --warmup 80" run DS2I_UNIT_DIV_RMW=1
EXTRA="--depth 8 --steps 160 --warmup 80" run DS2I_UNIT_DIV_RMW=1 DS2I_UNIT_FACTOR=2
EXTRA="--depth 8 --steps 160 --warmup 80" run DS2I_UNIT_DIV_RMW=2
