# A/B over environment knobs: bash profiles/probes/ab_env.sh "VAR=a" "VAR=b OTHER=c" ...   (default bench workload, short)
export TMPDIR=/tmp
for e in "$@"; do
  env $e python bench.py --workload gov2 --steps 10 --warmup 2 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', 'qps', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'resident', round(d.get('kernel_resident_qps',0)), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])
"
done
