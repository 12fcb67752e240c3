export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
DS2I_DEBUG_PLAN=1 python profiles/probes/class_probe2.py 2>&1 | grep -v amdgpu.ids | grep -v "ds2i plan" | tee gpurun_out/r2c/class_probe2.log
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1; tail -5 gpurun_out/r2c/pytest.log
timeout 900 python bench.py --workload gov2 --steps 20 --warmup 3 > gpurun_out/r2c/bench_gov2.json 2> gpurun_out/r2c/bench_gov2.err
grep -E "^class|Traceback|Error" gpurun_out/r2c/bench_gov2.err | cut -c1-330; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2c/bench_gov2.json'))
print({k:j[k] for k in ('value','ms_per_step','kernel_resident_qps','end_to_end_over_resident')}, j['roofline']['kernel'], j['roofline']['frac'], j['cpu_baseline'] and j['cpu_baseline']['value'])
PY
