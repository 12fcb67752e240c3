# final lines of the round for the shipped build: rocprofv3 profile of the default bench, then the default and opt lines
bash profiles/probes/run_r3_prof.sh r03_gov2 gov2 ranked_and > /dev/null 2>&1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_final
python bench.py > gpurun_out/r03_final/default_gov2.json 2> gpurun_out/r03_final/default_gov2.err
python bench.py --workload gov2 --codec opt --steps 30 --warmup 4 > gpurun_out/r03_final/gov2_opt.json 2> gpurun_out/r03_final/gov2_opt.err
python bench.py --workload gov2 --op and_freq --steps 30 --warmup 4 > gpurun_out/r03_final/gov2_and_freq.json 2> gpurun_out/r03_final/gov2_and_freq.err
for f in default_gov2 gov2_opt gov2_and_freq; do python -c "
import json
d=json.loads(open('gpurun_out/r03_final/$f.json').read().strip().splitlines()[-1])
print('$f', round(d['value']), round(d['ms_per_step'],2), d['step_ms_spread'], round(d['roofline']['frac'],4), d['roofline'].get('kernel_ms'), d['roofline'].get('kernel_ms_all_launches'))
"; done
head -4 gpurun_out/prof_r03_gov2/kernel_stats.csv
