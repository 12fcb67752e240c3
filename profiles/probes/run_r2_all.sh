#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/run_r2_all.sh <tag>
# All BENCH-format lines of the round: the default bench (metric config), the other configs and operators.
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 1500 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name: $(cut -c1-160 $OUT/bench_$name.json)"; }
run default_gov2
run c2 --workload c2
run gov2_opt --workload gov2 --codec opt
run gov2_wand --workload gov2 --op wand
run gov2_maxscore --workload gov2 --op maxscore
run gov2_ranked_or --workload gov2 --op ranked_or --steps 6
run gov2_and --workload gov2 --op and
run gov2_and_freq --workload gov2 --op and_freq
run gov2_or --workload gov2 --op or --steps 4 --warmup 1
run gov2_or_freq --workload gov2 --op or_freq --steps 4 --warmup 1
run cw09_mixed --workload cw09 --codec block_mixed
