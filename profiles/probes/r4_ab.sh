# A/B of environment knobs on the default bench: r4_ab.sh "<env assignments>" ...
export TMPDIR=/tmp
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --workload gov2 --steps 40 --warmup 4 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), d['step_ms_spread'], [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])"
done
