#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r6_first.sh <part>   -- the measurements round 5 left for the start of round 6
#   g2     range tables at 2 entries per posting (with hints) against 4, every operator, GOV2 scale: is 2 the better default? (ranked_and: 969 k at 19.6 GB
#          against 958 k at 29.9 GB, profiles/r05_table_budget.txt)
#   and    after `git apply profiles/probes/r6_and_rstream.patch` + build: parity probe of the AND stream in its four variants, then `and` with and without it
#   prof   kernel trace of the default bench on the build with the 5-8-term class on the stream kernel (profiles/r05_gov2/ predates it)
set -u
PART=${1:-g2}
OUT=gpurun_out/r6_first
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" > $OUT/$name.json 2> $OUT/$name.err; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", round(d["config"].get("device_bytes",0)/1e9,2), "GB")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
if [ $PART = g2 ]; then
  for op in ranked_and wand maxscore and and_freq or; do
    run g4_$op A=1 python bench.py --op $op --no-oracle --steps 40 --warmup 5
    run g2_$op DS2I_RMW_G=2 python bench.py --op $op --no-oracle --steps 40 --warmup 5
  done
fi
if [ $PART = and ]; then
  for extra in A=1 DS2I_UNIT_CAP=8 DS2I_NO_RMH=1 DS2I_RMW_G=1; do
    env DS2I_AND_RSTREAM=1 $extra timeout 300 python profiles/probes/and_rstream_probe.py 1 2 3 > $OUT/and_probe_$extra.txt 2>&1; echo "probe $extra rc=$?"; tail -2 $OUT/and_probe_$extra.txt
  done
  run and_default A=1 python bench.py --op and --steps 30 --warmup 3
  run and_rstream DS2I_AND_RSTREAM=1 python bench.py --op and --steps 30 --warmup 3
  run and_default2 A=1 python bench.py --op and --no-oracle --steps 30 --warmup 3
  run and_rstream2 DS2I_AND_RSTREAM=1 python bench.py --op and --no-oracle --steps 30 --warmup 3
fi
if [ $PART = prof ]; then
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 60 --warmup 5 > $OUT/prof_bench.json 2> $OUT/prof_bench.err
  KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); if [ -n "$KS" ]; then cp "$KS" $OUT/kernel_stats.csv; fi; rm -rf $OUT/kt
  head -16 $OUT/kernel_stats.csv
  bash profiles/probes/r5_pmc.sh final ranked_and
fi
