export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "query_ops and (opt or ef or single or uniform) or partition_shapes or full_size_c2_opt or fuzz and 3]" 2>&1 | tail -4
for a in "--codec opt" "" ; do python bench.py --workload gov2 $a --steps 12 --warmup 3 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', round(d['value']), round(d['ms_per_step'],2), [round(c['ms_per_launch'],2) for c in d['roofline']['per_class']])
"; done
