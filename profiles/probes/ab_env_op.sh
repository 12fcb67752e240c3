# bash profiles/probes/ab_env_op.sh <op> "ENV=.." ...   (short GOV2-scale runs of one operator under different knobs)
export TMPDIR=/tmp
OP=$1; shift
for e in "$@"; do
  env $e python bench.py --workload gov2 --op $OP --steps 10 --warmup 2 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', '$OP', round(d['value']), round(d['ms_per_step'],2), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])
"
done
