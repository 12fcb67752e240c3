#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_close2.sh  -- the bench lines of the shipped build that the first session of
# round 5 lost with its container: freq_index layouts (default upload = transcoded), strong-scaling proxy, configs[4] family.
# Lines without an oracle leg carry no roofline / cpu_baseline (they are rates only); the two with one say so in their JSON.
set -u
OUT=gpurun_out/r5_close2
mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "frac", r.get("frac"), "step_frac", r.get("step_frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "GB", round(d["config"].get("device_bytes",0)/1e9,2))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
bench() { name=$1; shift; timeout 400 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; line $OUT/bench_$name.json $name; }
T0=$(date +%s)
bench gov2_opt --workload gov2 --codec opt --steps 30 --warmup 3
bench gov2_opt_wand --workload gov2 --codec opt --op wand --steps 30 --warmup 3 --no-oracle
grep -i "upload\|built" $OUT/bench_gov2_opt.err | head -4
echo "t=$(( $(date +%s) - T0 ))s"
bench gov2_b512 --batch 512 --depth 8 --steps 160 --warmup 80 --no-oracle
bench gov2_b1024 --batch 1024 --depth 6 --steps 120 --warmup 40 --no-oracle
bench gov2_b2048 --batch 2048 --depth 4 --steps 80 --warmup 20 --no-oracle
echo "t=$(( $(date +%s) - T0 ))s"
bench cw09_optpfor --workload cw09 --codec block_optpfor --steps 30 --warmup 3
bench cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3 --no-oracle
DS2I_MIXED_NATIVE=1 timeout 400 python bench.py --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3 --no-oracle > $OUT/bench_cw09_mixed_fixed_native.json 2> $OUT/bench_cw09_mixed_fixed_native.err; line $OUT/bench_cw09_mixed_fixed_native.json cw09_mixed_fixed_native
echo "t=$(( $(date +%s) - T0 ))s"
bench cw09_mixed_optimised --workload cw09 --codec block_mixed --mixed-policy optimised --steps 30 --warmup 3 --no-oracle
grep -i "upload\|built" $OUT/bench_cw09_mixed_fixed.err $OUT/bench_cw09_mixed_fixed_native.err | head -6
echo "t=$(( $(date +%s) - T0 ))s"
