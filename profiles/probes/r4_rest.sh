export TMPDIR=/tmp
OUT=gpurun_out/r04_rest; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu.py -m gpu -x -q -k "bench_two_ranks or alternative or opt_index_configs2 or clueweb or queries_cli or (query_ops and (wand or maxscore or ranked_or)) or fuzz or union or disjunctive" > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log
for op in wand maxscore; do python bench.py --workload gov2 --op $op --steps 30 --warmup 3 --no-oracle 2>/dev/null > $OUT/bench_$op.json; python -c "
import json; d=json.loads(open('$OUT/bench_$op.json').read().strip().splitlines()[-1]); print('$op', round(d['value']), round(d['ms_per_step'],2), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])"; done
DS2I_NO_RMH_USE=1 python bench.py --workload gov2 --op wand --steps 30 --warmup 3 --no-oracle 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wand without hints', round(d['value']), round(d['ms_per_step'],2))"
