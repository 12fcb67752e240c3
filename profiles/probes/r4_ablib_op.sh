# A/B of library variants for one operator on the GOV2-scale bench, same box: r4_ablib_op.sh <op> <variant> ...   ("main" = the product)
export TMPDIR=/tmp
OP=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset DS2I_LIB_VARIANT; else export DS2I_LIB_VARIANT=$v; fi
  python bench.py --workload gov2 --op $OP --steps ${STEPS:-30} --warmup 4 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$OP $v rep $rep', round(d['value']), round(d['ms_per_step'],2))"
done
done
