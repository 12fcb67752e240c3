# last GPU pass of round 4 on the shipped build: whole GPU suite + smoke, the committed profile (kernel stats + PMC passes), the default and wand bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_final/tests.log 2>&1
tail -3 gpurun_out/r04_final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash profiles/probes/run_r4_prof.sh r04 gov2 ranked_and > gpurun_out/r04_final/prof.log 2>&1
grep -E "k_ranked_stream<2, false" gpurun_out/prof_r04/counters_sq.txt | cut -c60-200
O=gpurun_out/r04_all; mkdir -p $O
python bench.py > $O/default_gov2.json 2> $O/default_gov2.err; tail -c 600 $O/default_gov2.json
python bench.py --workload gov2 --op wand --steps 30 --warmup 4 > $O/gov2_wand.json 2> $O/gov2_wand.err; tail -c 300 $O/gov2_wand.json
python bench.py --workload gov2 --op maxscore --steps 30 --warmup 4 > $O/gov2_maxscore.json 2> $O/gov2_maxscore.err
python bench.py --workload gov2 --op ranked_or --steps 30 --warmup 4 > $O/gov2_ranked_or.json 2> $O/gov2_ranked_or.err
