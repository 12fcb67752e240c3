# round-3 bench lines of every operator / index kind -> gpurun_out/<tag>/*.json (copied to profiles/r03_bench_*.json)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_all}
mkdir -p $OUT
run() { name=$1; shift; python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
print("$name", "qps", round(d["value"]), "ms/step", round(d["ms_per_step"],2), d["step_ms_spread"], "frac", round(d["roofline"]["frac"],4) if d["roofline"].get("frac") else None, "cpu", d.get("cpu_baseline",{}).get("value"))
PY
}
run default_gov2
S="--steps 30 --warmup 4"
run gov2_wand --workload gov2 --op wand $S
run gov2_maxscore --workload gov2 --op maxscore $S
run gov2_ranked_or --workload gov2 --op ranked_or $S
run gov2_and --workload gov2 --op and $S
run gov2_and_freq --workload gov2 --op and_freq $S
run gov2_or --workload gov2 --op or $S
run gov2_or_freq --workload gov2 --op or_freq $S
run gov2_opt --workload gov2 --codec opt $S
run c2 --workload c2 $S
run cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed $S
run cw09_mixed_optimised --workload cw09 --codec block_mixed --mixed-policy optimised $S
run cw09_optpfor --workload cw09 --codec block_optpfor $S
