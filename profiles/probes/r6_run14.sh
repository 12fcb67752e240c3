set -u
OUT=gpurun_out/${1:-r6p}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --op and --steps 40 --warmup 5 --no-oracle > $OUT/bench_and.json 2> $OUT/bench_and.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$KS" $OUT/and_kernel_stats.csv; rm -rf $OUT/kt
head -14 $OUT/and_kernel_stats.csv | cut -c1-150
python - $OUT <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_and.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["step_ms_spread"])
PY
