set -x
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r2b/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -25 gpurun_out/r2b/pytest.log
timeout 900 python bench.py --workload gov2 --steps 20 --warmup 3 > gpurun_out/r2b/bench_gov2.json 2> gpurun_out/r2b/bench_gov2.err
grep -E "class|upload|Traceback|Error" gpurun_out/r2b/bench_gov2.err | cut -c1-400; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2b/bench_gov2.json'))
print({k:j[k] for k in ('value','ms_per_step','kernel_resident_qps','end_to_end_over_resident')}, j['roofline']['kernel'], j['roofline']['frac'], j['cpu_baseline'] and j['cpu_baseline']['value'])
PY
timeout 600 python bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_c2.json 2> gpurun_out/r2b/bench_c2.err
grep -E "class|Traceback|Error" gpurun_out/r2b/bench_c2.err | cut -c1-300; cut -c1-400 gpurun_out/r2b/bench_c2.json
