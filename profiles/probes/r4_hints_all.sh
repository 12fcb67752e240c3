export TMPDIR=/tmp
OUT=gpurun_out/r04_hints_all; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "fuzz or query_ops or prun or full_size_c2 or brute or mixed" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
one() { name=$1; shift; python bench.py "$@" --steps 30 --warmup 4 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['value']), round(d['ms_per_step'],2), [(k['kernel'][:22], round(k['ms_per_launch'],2)) for k in d['roofline']['per_kernel']])"; }
one gov2 --workload gov2
one gov2_and --workload gov2 --op and
one gov2_and_freq --workload gov2 --op and_freq
one gov2_opt --workload gov2 --codec opt
one cw09_mixed_fixed --workload cw09 --codec block_mixed --mixed-policy fixed
one cw09_mixed_optimised --workload cw09 --codec block_mixed --mixed-policy optimised
DS2I_NO_RANKED_STREAM=1 python bench.py --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 4 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cw09_mixed_fixed class kernels only', round(d['value']), round(d['ms_per_step'],2))"
