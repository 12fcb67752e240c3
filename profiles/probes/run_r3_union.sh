# bench lines of the streamed top-k-of-union operators (k_union_topk) + the issue / icache PMC pass of the default bench
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_union}
mkdir -p $OUT
run() { name=$1; shift; python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
print("$name", "qps", round(d["value"]), "ms/step", round(d["ms_per_step"],2), d["step_ms_spread"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
S="--steps 30 --warmup 4"
run gov2_wand --workload gov2 --op wand $S
run gov2_maxscore --workload gov2 --op maxscore $S
run gov2_ranked_or --workload gov2 --op ranked_or $S
run gov2_opt_wand --workload gov2 --codec opt --op wand $S --no-cpu-baseline
