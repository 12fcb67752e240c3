export TMPDIR=/tmp
OUT=gpurun_out/r04_plan; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "pipeline or fuzz or error or long_queries or (query_ops and (or_freq or and_freq))" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for cfg in "DS2I_PLAN_THREADS=1" "DS2I_PLAN_THREADS=2" "DS2I_PLAN_THREADS=4" "DS2I_PLAN_THREADS=8"; do
  for wl in c2 gov2; do
  env $cfg DS2I_DEBUG_PLAN=1 python bench.py --workload $wl --steps 60 --warmup 5 --no-oracle 2>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg $wl', round(d['value']), round(d['ms_per_step'],3), 'resident', round(d['kernel_resident_qps']), 'e2e/resident', round(d['end_to_end_over_resident'],3))"
  grep "ds2i plan" $OUT/err.txt | tail -3 | awk '{print "   plan us:", $NF, $(NF-1)}' | tail -1
  done
done
for op in or_freq and_freq; do python bench.py --workload gov2 --op $op --steps 20 --warmup 3 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$op', round(d['value']), round(d['ms_per_step'],2))"; done
