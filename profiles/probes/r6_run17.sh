set -u
OUT=gpurun_out/${1:-r6r}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/ranked_stream_probe.py 1 2 > $OUT/ranked_probe.txt 2>&1; echo "ranked probe rc=$?"; tail -1 $OUT/ranked_probe.txt
timeout 600 python tests/union_stream_probe.py 1 2 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -1 $OUT/union_probe.txt
timeout 600 python tests/and_stream_probe.py 1 2 > $OUT/and_probe.txt 2>&1; echo "and probe rc=$?"; tail -1 $OUT/and_probe.txt
B="timeout 400 python bench.py --no-oracle --steps 40 --warmup 5"
for op in ranked_and wand and and_freq; do $B --op $op > $OUT/bench_$op.json 2> $OUT/bench_$op.err; done
$B --op ranked_and > $OUT/bench_ranked_and2.json 2> $OUT/bench_ranked_and2.err
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
