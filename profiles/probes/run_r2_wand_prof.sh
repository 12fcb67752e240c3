#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/run_r2_wand_prof.sh   -- rocprofv3 kernel stats + one SQ counter pass of the wand bench
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02_wand; mkdir -p $OUT
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --workload gov2 --op wand --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp "$KS" $OUT/kernel_stats.csv; rm -rf $OUT/kt
head -8 $OUT/kernel_stats.csv
bash profiles/probes/sq_pass.sh r02_wand gov2 wand "k_disjunctive" | grep ", false, 0>" 
cp gpurun_out/sq_r02_wand/counters.txt $OUT/counters_sq.txt
