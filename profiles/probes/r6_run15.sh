set -u
OUT=gpurun_out/${1:-r6q}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/and_stream_probe.py 1 2 > $OUT/and_probe.txt 2>&1; echo "and probe rc=$?"; tail -1 $OUT/and_probe.txt
B="timeout 400 python bench.py --no-oracle --steps 40 --warmup 5"
for i in 1 2; do $B --op and > $OUT/bench_and_$i.json 2> $OUT/bench_and_$i.err; done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", d["step_ms_spread"], " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
PY
