#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_cw09.sh <tag>  -- configs[4] collection: block_optpfor, block_mixed (fixed policy) transcoded at upload (default) and native
set -u
TAG=${1:-cw}
OUT=gpurun_out/r5_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), "q/s", round(d["ms_per_step"],2), "ms/step")
PY
}
python bench.py --workload cw09 --codec block_optpfor --steps 30 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_cw09_optpfor.json 2> $OUT/bench_cw09_optpfor.err; line $OUT/bench_cw09_optpfor.json cw09_optpfor
python bench.py --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_cw09_mixed_fixed.json 2> $OUT/bench_cw09_mixed_fixed.err; line $OUT/bench_cw09_mixed_fixed.json cw09_mixed_fixed_transcoded
DS2I_MIXED_NATIVE=1 python bench.py --workload cw09 --codec block_mixed --mixed-policy fixed --steps 30 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_cw09_mixed_fixed_native.json 2> $OUT/bench_cw09_mixed_fixed_native.err; line $OUT/bench_cw09_mixed_fixed_native.json cw09_mixed_fixed_native
grep -i "index\|upload\|build" $OUT/bench_cw09_mixed_fixed.err | head -8
