#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_nt8.sh  -- k_ranked_stream<5..8> behind DS2I_STREAM_NT_MAX=8: parity probe (oracle, bit for bit),
# then the default bench with and without it on the same box (rates only: --no-oracle)
set -u
OUT=gpurun_out/r5_nt8
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
DS2I_STREAM_NT_MAX=8 timeout 150 python profiles/probes/rs_nt8_probe.py 1 2 > $OUT/probe.txt 2>&1; echo "probe rc=$?"; tail -4 $OUT/probe.txt
echo "t=$(( $(date +%s) - T0 ))s"
DS2I_STREAM_NT_MAX=8 DS2I_UNIT_CAP=8 timeout 100 python profiles/probes/rs_nt8_probe.py 3 > $OUT/probe_split.txt 2>&1; echo "probe(split) rc=$?"; tail -2 $OUT/probe_split.txt
echo "t=$(( $(date +%s) - T0 ))s"
run() { env "$@" timeout 120 python bench.py --no-oracle --steps 60 --warmup 5 2>$OUT/bench_$1.err | tee $OUT/bench_$1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'q/s', round(d['ms_per_step'],3), 'ms/step')"; grep "^class 2" $OUT/bench_$1.err | cut -c1-110; }
{
for cfg in "DS2I_STREAM_NT_MAX=8" "A=1" "DS2I_STREAM_NT_MAX=6"; do echo "== $cfg"; run $cfg; echo "t=$(( $(date +%s) - T0 ))s"; done
} | tee $OUT/ab.txt
