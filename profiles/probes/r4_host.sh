export TMPDIR=/tmp
OUT=gpurun_out/r04_host; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "pipeline or scale_properties or bench_contract or error" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for cfg in "DS2I_PLAN_THREAD=0" "DS2I_PLAN_THREAD=1" "DS2I_PLAN_THREAD=0" "DS2I_PLAN_THREAD=1"; do
  for wl in c2 gov2; do
  env $cfg python bench.py --workload $wl --steps 60 --warmup 5 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg $wl', round(d['value']), round(d['ms_per_step'],3), 'resident', round(d['kernel_resident_qps']), 'e2e/resident', round(d['end_to_end_over_resident'],3))"
  done
done
