export TMPDIR=/tmp
OUT=gpurun_out/r04_freq; mkdir -p $OUT

for v in main base; do for op in or_freq and_freq; do
  if [ "$v" = main ]; then unset DS2I_LIB_VARIANT; else export DS2I_LIB_VARIANT=$v; fi
  python bench.py --workload gov2 --op $op --steps 20 --warmup 3 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $op', round(d['value']), round(d['ms_per_step'],2))"
done; done
