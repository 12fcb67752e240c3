# rocprofv3 kernel stats of a short bench run of one operator: bash profiles/probes/kt_op.sh <op> [extra bench args]
export TMPDIR=/tmp
OP=$1; shift
OUT=gpurun_out/kt_$OP; rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --workload gov2 --op $OP --steps 10 --warmup 2 --no-oracle "$@" > $OUT/bench.json 2> $OUT/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$KS" $OUT/kernel_stats.csv; rm -rf $OUT/kt
head -14 $OUT/kernel_stats.csv | cut -c1-200
