set -u
OUT=gpurun_out/${1:-r6ae}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --op and_freq --steps 20 --warmup 3 --no-oracle --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$KS" $OUT/kernel_stats_and_freq.csv; rm -rf $OUT/kt
head -12 $OUT/kernel_stats_and_freq.csv | cut -c1-200
DS2I_UNIT_CLOCK=1 timeout 300 python bench.py --op and_freq --steps 3 --warmup 1 --no-cpu-baseline --no-oracle 2>&1 >/dev/null | grep -i "ds2i plan\|stream" | tail -6
