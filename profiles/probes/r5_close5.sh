#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_close5.sh  -- DS2I_STREAM_NT_MAX=8 against the default, interleaved repeats on one box (the first
# A/B's three single runs disagreed with each other); then the budget tests on the re-ordered plan and the k = 100 line
set -u
OUT=gpurun_out/r5_close5
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
run() { env "$@" timeout 100 python bench.py --no-oracle --steps 100 --warmup 5 2>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'q/s', round(d['ms_per_step'],3), 'ms/step')"; grep "^class 2" $OUT/err.txt | cut -c1-100; }
{
for cfg in "A=1" "DS2I_STREAM_NT_MAX=8" "A=2" "DS2I_STREAM_NT_MAX=8" "A=3" "DS2I_STREAM_NT_MAX=8"; do echo "== $cfg"; run $cfg; echo "t=$(( $(date +%s) - T0 ))s"; done
} | tee $OUT/ab.txt
timeout 120 python -m pytest tests/test_gpu.py -m gpu -q -k "table_budget" 2>&1 | tail -2
timeout 100 python bench.py --workload c2 --k 100 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_k100.json 2> $OUT/bench_c2_k100.err; tail -c 200 $OUT/bench_c2_k100.json; tail -2 $OUT/bench_c2_k100.err
echo "t=$(( $(date +%s) - T0 ))s"
