#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_final.sh <part>   (round 5 closing measurements; part = prof | pmc | ops | opt | scale | cw09 | probes)
set -u
PART=${1:-prof}
OUT=gpurun_out/r5_final
mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "frac", r.get("frac"), "step_frac", r.get("step_frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
bench() { name=$1; shift; python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; line $OUT/bench_$name.json $name; }
if [ $PART = prof ]; then
  # default bench under rocprofv3 --kernel-trace --stats, then the PMC passes (own runs)
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 60 --warmup 5 > $OUT/prof_bench.json 2> $OUT/prof_bench.err
  KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); if [ -n "$KS" ]; then cp "$KS" $OUT/kernel_stats.csv; fi; rm -rf $OUT/kt
  head -12 $OUT/kernel_stats.csv; line $OUT/prof_bench.json traced_default
  bench default_gov2 --steps 60 --warmup 5
fi
if [ $PART = pmc ]; then
  bash profiles/probes/r5_pmc.sh final ranked_and
  bash profiles/probes/r5_pmc.sh final_wand wand
fi
if [ $PART = ops ]; then
  bench c2 --workload c2 --steps 60 --warmup 5
  for op in wand maxscore ranked_or and and_freq or or_freq; do bench gov2_$op --workload gov2 --op $op --steps 30 --warmup 3; done
  bench gov2c --workload gov2c --steps 30 --warmup 3
  bench gov2c_wand --workload gov2c --op wand --steps 30 --warmup 3
fi
if [ $PART = opt ]; then
  # the freq_index layouts: default upload (transcoded to block_optpfor) and the partitioned-sequence kernels on the image as it is
  bench gov2_opt --workload gov2 --codec opt --steps 30 --warmup 3
  bench gov2_opt_wand --workload gov2 --codec opt --op wand --steps 30 --warmup 3 --no-cpu-baseline
  DS2I_PEF_NATIVE=1 python bench.py --workload gov2 --codec opt --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_gov2_opt_native.json 2> $OUT/bench_gov2_opt_native.err; line $OUT/bench_gov2_opt_native.json gov2_opt_native
  DS2I_PEF_NATIVE=1 python bench.py --workload gov2 --codec opt --op wand --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_gov2_opt_wand_native.json 2> $OUT/bench_gov2_opt_wand_native.err; line $OUT/bench_gov2_opt_wand_native.json gov2_opt_wand_native
fi
if [ $PART = scale ]; then
  bench gov2_b512 --batch 512 --depth 8 --steps 160 --warmup 80 --no-cpu-baseline
  bench gov2_b1024 --batch 1024 --depth 6 --steps 120 --warmup 40 --no-cpu-baseline
  bench gov2_b2048 --batch 2048 --depth 4 --steps 80 --warmup 20 --no-cpu-baseline
  DS2I_STREAM_SETS=1 python bench.py --batch 512 --depth 8 --steps 160 --warmup 80 --no-cpu-baseline > $OUT/bench_gov2_b512_sets.json 2> $OUT/bench_gov2_b512_sets.err; line $OUT/bench_gov2_b512_sets.json gov2_b512_stream_sets
fi
if [ $PART = cw09 ]; then
  bench cw09_optpfor --workload cw09 --codec block_optpfor --steps 30 --warmup 3
  bench cw09_b512 --workload cw09 --codec block_optpfor --batch 512 --depth 8 --steps 160 --warmup 80 --no-cpu-baseline --no-oracle
  bench cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3
  bench cw09_mixed_optimised --workload cw09 --codec block_mixed --mixed-policy optimised --steps 30 --warmup 3 --no-cpu-baseline
  DS2I_MIXED_NATIVE=1 python bench.py --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_cw09_mixed_fixed_native.json 2> $OUT/bench_cw09_mixed_fixed_native.err; line $OUT/bench_cw09_mixed_fixed_native.json cw09_mixed_fixed_native
fi
if [ $PART = probes ]; then
  DS2I_LIB_VARIANT=lines timeout 400 python profiles/probes/line_probe.py > $OUT/lines.txt 2>&1; cat $OUT/lines.txt
  DS2I_LIB_VARIANT=phase timeout 400 python profiles/probes/rs_phase_probe.py > $OUT/phase.txt 2>&1; cat $OUT/phase.txt
  profiles/probes/calib_valu > $OUT/calib_valu.txt 2>&1; cat $OUT/calib_valu.txt
fi
