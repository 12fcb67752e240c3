set -u
OUT=gpurun_out/${1:-r6ag}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "
import json;d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]);print('default', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
