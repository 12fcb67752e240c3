export TMPDIR=/tmp
OUT=gpurun_out/prof_r02d; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --workload gov2 --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$KS" $OUT/kernel_stats.csv; rm -rf $OUT/kt
head -5 $OUT/kernel_stats.csv
python - <<'PY'
import json
j=json.load(open('gpurun_out/prof_r02d/bench.json')); r=j['roofline']
print(j['value'], r['kernel'], 'timed', r['kernel_ms'], 'all', r['kernel_ms_all_launches'], r['launches_all'], 'frac', r['frac'])
PY
