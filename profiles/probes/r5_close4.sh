#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_close4.sh  -- the GPU suite on the final build (no -x: every failure is reported), smoke,
# then the default bench with k_ranked_stream<5..8> on and off (DS2I_STREAM_NT_MAX; rates only)
set -u
OUT=gpurun_out/r5_close4
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
echo "t=$(( $(date +%s) - T0 ))s"
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
run() { env "$@" timeout 120 python bench.py --no-oracle --steps 60 --warmup 5 2>$OUT/bench_$1.err | tee $OUT/bench_$1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'q/s', round(d['ms_per_step'],3), 'ms/step')"; grep "^class 2" $OUT/bench_$1.err | cut -c1-110; }
{
for cfg in "DS2I_STREAM_NT_MAX=8" "A=1" "DS2I_STREAM_NT_MAX=6"; do echo "== $cfg"; run $cfg; echo "t=$(( $(date +%s) - T0 ))s"; done
} | tee $OUT/ab.txt
