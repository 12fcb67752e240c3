#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r6_final.sh <part>   -- the round's bench lines on the build that ships (copied to profiles/r06_bench_*.json)
set -u
PART=${1:-ops}
OUT=gpurun_out/r6_final
mkdir -p $OUT
export TMPDIR=/tmp
line() { python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", round(d["config"].get("device_bytes",0)/1e9,2), "GB", "frac", round(r.get("frac") or 0,3), "step_frac", round(r.get("step_frac") or 0,3), r.get("kernel"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
bench() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; line $OUT/bench_$name.json $name; }
if [ $PART = ops ]; then
  bench default_gov2 --steps 60 --warmup 5
  for op in wand maxscore ranked_or and and_freq or or_freq; do bench gov2_$op --workload gov2 --op $op --steps 30 --warmup 3; done
  bench c2 --workload c2 --steps 60 --warmup 5
  bench c2_k100 --workload c2 --k 100 --steps 10 --warmup 2 --no-cpu-baseline
  bench gov2c --workload gov2c --steps 30 --warmup 3
  bench gov2c_wand --workload gov2c --op wand --steps 30 --warmup 3
fi
if [ $PART = opt ]; then
  bench gov2_opt --workload gov2 --codec opt --steps 30 --warmup 3
  bench gov2_opt_wand --workload gov2 --codec opt --op wand --steps 30 --warmup 3 --no-cpu-baseline
  DS2I_PEF_NATIVE=1 bench gov2_opt_native --workload gov2 --codec opt --steps 30 --warmup 3 --no-cpu-baseline
  DS2I_TABLE_BUDGET=3x bench gov2_opt_budget3x --workload gov2 --codec opt --steps 30 --warmup 3 --no-cpu-baseline
  bench gov2_b256 --batch 256 --depth 8 --steps 200 --warmup 80 --no-cpu-baseline --no-oracle
  bench gov2_b512 --batch 512 --depth 8 --steps 160 --warmup 80 --no-cpu-baseline
  bench gov2_wand_b512 --op wand --batch 512 --depth 8 --steps 100 --warmup 40 --no-cpu-baseline --no-oracle
  bench gov2_wand_k100 --op wand --k 100 --steps 20 --warmup 3 --no-cpu-baseline --no-oracle
  bench gov2_k100 --k 100 --steps 30 --warmup 3 --no-cpu-baseline
  bench gov2_b1024 --batch 1024 --depth 6 --steps 120 --warmup 40 --no-cpu-baseline
  bench gov2_b2048 --batch 2048 --depth 4 --steps 80 --warmup 20 --no-cpu-baseline
fi
if [ $PART = cw09 ]; then
  bench cw09_optpfor --workload cw09 --codec block_optpfor --steps 30 --warmup 3
  bench cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3
  bench cw09_mixed_optimised --workload cw09 --codec block_mixed --mixed-policy optimised --steps 30 --warmup 3 --no-cpu-baseline
  DS2I_MIXED_NATIVE=1 bench cw09_mixed_fixed_native --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3 --no-cpu-baseline
  bench cw09_b512 --workload cw09 --codec block_optpfor --batch 512 --depth 8 --steps 160 --warmup 80 --no-cpu-baseline --no-oracle
fi
