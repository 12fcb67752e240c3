#!/bin/bash
export TMPDIR=/tmp
run() { echo "== $* $EXTRA"; env "$@" python bench.py --workload gov2 --no-oracle $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)), d.get('step_ms_spread'))"; }
EXTRA="--batch 512 --depth 8 --steps 160 --warmup 80" run A=1
EXTRA="--batch 512 --depth 8 --steps 160 --warmup 80" run DS2I_UNIT_FLOOR=512
EXTRA="--batch 512 --depth 8 --steps 160 --warmup 80" run DS2I_UNIT_FLOOR=1024
EXTRA="--batch 1024 --depth 6 --steps 120 --warmup 40" run A=1
EXTRA="--batch 1024 --depth 6 --steps 120 --warmup 40" run DS2I_UNIT_FLOOR=512
EXTRA="--batch 2048 --depth 4 --steps 80 --warmup 20" run A=1
EXTRA="--steps 40 --warmup 5" run A=1
