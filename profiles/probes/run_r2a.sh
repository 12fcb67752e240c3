set -x
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -30 gpurun_out/r2a/pytest.log
timeout 600 python bench.py --workload c2 --steps 10 --warmup 2 > gpurun_out/r2a/bench_c2.json 2> gpurun_out/r2a/bench_c2.err
tail -12 gpurun_out/r2a/bench_c2.err; cat gpurun_out/r2a/bench_c2.json
timeout 900 python bench.py --workload gov2 --steps 20 --warmup 3 > gpurun_out/r2a/bench_gov2.json 2> gpurun_out/r2a/bench_gov2.err
tail -14 gpurun_out/r2a/bench_gov2.err; cat gpurun_out/r2a/bench_gov2.json
DS2I_NO_BMW_PRUNE=1 timeout 900 python bench.py --workload gov2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench_gov2_noprune.json 2> gpurun_out/r2a/bench_gov2_noprune.err
tail -8 gpurun_out/r2a/bench_gov2_noprune.err; cat gpurun_out/r2a/bench_gov2_noprune.json
