# round-4 bench lines of the shipped build (one JSON line each under gpurun_out/r04_all/)
export TMPDIR=/tmp
O=gpurun_out/r04_all; mkdir -p $O
run() { name=$1; shift; python bench.py "$@" > $O/$name.json 2> $O/$name.err; python -c "
import json
d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$name', round(d['value']), round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'step_frac', r.get('step_frac') and round(r['step_frac'],4), 'e2e/res', round(d['end_to_end_over_resident'],3), 'cpu', d.get('cpu_baseline') and round(d['cpu_baseline']['value'],1))
" 2>&1 | tail -1; }
run default_gov2
for op in wand maxscore ranked_or and and_freq or or_freq; do run gov2_$op --workload gov2 --op $op --steps 30 --warmup 4; done
run gov2_opt --workload gov2 --codec opt --steps 30 --warmup 4
run gov2_opt_wand --workload gov2 --codec opt --op wand --steps 30 --warmup 4 --no-oracle
run c2 --workload c2 --steps 60 --warmup 5
run gov2c --workload gov2c --steps 30 --warmup 4
run gov2c_wand --workload gov2c --op wand --steps 30 --warmup 4
run cw09_optpfor --workload cw09 --steps 30 --warmup 4
run cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 4
run cw09_mixed_optimised --workload cw09 --codec block_mixed --mixed-policy optimised --steps 30 --warmup 4
