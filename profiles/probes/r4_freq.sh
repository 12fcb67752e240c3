export TMPDIR=/tmp
OUT=gpurun_out/r04_freq; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "(query_ops and (or_freq or and_freq)) or fuzz or exception_count or adversarial or long_queries" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
for v in main base; do for op in or_freq and_freq; do
  if [ "$v" = main ]; then unset DS2I_LIB_VARIANT; else export DS2I_LIB_VARIANT=$v; fi
  python bench.py --workload gov2 --op $op --steps 20 --warmup 3 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $op', round(d['value']), round(d['ms_per_step'],2))"
done; done
