# A/B of environment knobs on another operator: r4_ab_op.sh <op> "<env assignments>" ...
export TMPDIR=/tmp
OP=$1; shift
for cfg in "$@"; do
  echo "== $OP $cfg"
  env $cfg python bench.py --workload ${WL:-gov2} --op $OP --steps ${STEPS:-30} --warmup 3 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])"
done
