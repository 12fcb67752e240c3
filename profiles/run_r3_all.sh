# round-3 bench lines of every operator / index kind (short runs; the default line is taken separately with the oracle legs)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_all}
mkdir -p $OUT
run() { name=$1; shift; python bench.py --steps 12 --warmup 3 --no-oracle "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
print("$name", "qps", round(d["value"]), "ms/step", round(d["ms_per_step"],2), [(c["kernel"][-8:], round(c["ms_per_launch"],2)) for c in d["roofline"]["per_class"]])
PY
}
run gov2_and --workload gov2 --op and
run gov2_and_freq --workload gov2 --op and_freq
run gov2_or --workload gov2 --op or
run gov2_or_freq --workload gov2 --op or_freq
run gov2_ranked_or --workload gov2 --op ranked_or
run gov2_opt --workload gov2 --codec opt
run c2 --workload c2
run cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed
run cw09_optpfor --workload cw09 --codec block_optpfor
