# bash profiles/probes/ab_depth.sh <op> <depth>...   (batches in flight in the pipelined region, 30-step runs)
export TMPDIR=/tmp
OP=$1; shift
for dep in "$@"; do
  python bench.py --workload gov2 --op $OP --steps 30 --warmup 4 --depth $dep --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $dep', '$OP', round(d['value']), round(d['ms_per_step'],2), d['step_ms_spread'])
"
done
