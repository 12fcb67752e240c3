# A/B of library variants (profiles/tmp_libs/lib_<name>.so; "main" = the product) on the default bench, same box: r4_ablib.sh main hint8 ...
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  echo "== $v (rep $rep)"
  if [ "$v" = main ]; then unset DS2I_LIB_VARIANT; else export DS2I_LIB_VARIANT=$v; fi
  python bench.py --workload gov2 --steps ${STEPS:-40} --warmup 4 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), 'resident', round(d.get('kernel_resident_qps',0)), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])"
done
done
