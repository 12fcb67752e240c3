#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_close3.sh  -- last call of round 5: the GPU suite + smoke on the final build, the default bench
# line, a k = 100 line (the big-heap kernels, never benchmarked before), then what the upload-time tables buy per byte (DS2I_TABLE_BUDGET / the knobs
# it chooses between): rate and resident bytes of the default bench per configuration -> profiles/r05_table_budget.txt
set -u
OUT=gpurun_out/r5_close3
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
echo "t=$(( $(date +%s) - T0 ))s"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench_default_gov2.json 2> $OUT/bench_default_gov2.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_close3/bench_default_gov2.json").read().strip().splitlines()[-1])
print("default", round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms", "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
PY
echo "t=$(( $(date +%s) - T0 ))s"
timeout 200 python bench.py --workload c2 --k 100 --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_k100.json 2> $OUT/bench_c2_k100.err; tail -c 300 $OUT/bench_c2_k100.json | head -c 300; echo
echo "t=$(( $(date +%s) - T0 ))s"
run() { env "$@" timeout 200 python bench.py --no-oracle --steps 40 --warmup 5 2>$OUT/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'q/s', round(d['ms_per_step'],3), 'ms/step', round(d['config'].get('device_bytes',0)/1e9,2), 'GB resident')"; grep -h "TABLE_BUDGET" $OUT/sweep.err | head -1; }
{
for cfg in "A=1" "DS2I_NO_RMH=1" "DS2I_RMW_G=2" "DS2I_RMW_G=2 DS2I_NO_RMH=1" "DS2I_NO_XSLOTS=1" "DS2I_TABLE_BUDGET=8x" "DS2I_TABLE_BUDGET=5x" "DS2I_TABLE_BUDGET=3x"; do
  echo "== $cfg"; run $cfg
  echo "t=$(( $(date +%s) - T0 ))s" >&2
done
} | tee $OUT/table_budget.txt
echo "t=$(( $(date +%s) - T0 ))s"
