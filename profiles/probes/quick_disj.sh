export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_disj}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "${2:-query_ops or disjunctive or full_size_c2 or topk or long_queries or alternative or uninstrumented}" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
for op in wand maxscore; do
python bench.py --workload gov2 --op $op --steps 8 --warmup 2 --no-oracle 2>$OUT/bench_$op.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$op qps', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'resident', round(d.get('kernel_resident_qps',0)), [(round(c['ms_per_launch'],2), c['postings_scored'], c['docs_blocks_decoded']) for c in d['roofline']['per_class']])
"
done
