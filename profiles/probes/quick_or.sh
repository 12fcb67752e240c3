export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_or}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "${2:-query_ops and (or- or or_freq) or long_queries or brute or gov2_scale_properties}" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
for op in or or_freq; do
python bench.py --workload gov2 --op $op --steps 8 --warmup 2 --no-oracle 2>$OUT/bench_$op.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$op qps', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'resident', round(d.get('kernel_resident_qps',0)), [(round(c['ms_per_launch'],2), c['docs_blocks_decoded']) for c in d['roofline']['per_class']])
"
done
