#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_base.sh   (round 5: baseline of the round-4 build on this round's boxes)
set -u
OUT=gpurun_out/r5_base
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
python bench.py --steps 60 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 60 --warmup 5 --batch 512 --no-cpu-baseline > $OUT/bench_b512.json 2> $OUT/bench_b512.err
python bench.py --steps 30 --warmup 3 --op and_freq --no-cpu-baseline > $OUT/bench_and_freq.json 2> $OUT/bench_and_freq.err
python bench.py --steps 30 --warmup 3 --op or_freq --no-cpu-baseline > $OUT/bench_or_freq.json 2> $OUT/bench_or_freq.err
tail -c 600 $OUT/bench_default.json; echo; cut -c1-300 $OUT/bench_b512.json; echo; cut -c1-200 $OUT/bench_and_freq.json; echo; cut -c1-200 $OUT/bench_or_freq.json
grep -ciE "^" $OUT/counters_list.txt
