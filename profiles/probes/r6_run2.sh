set -u
OUT=gpurun_out/${1:-r6b}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/union_stream_probe.py 1 2 3 4 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -3 $OUT/union_probe.txt
for op in wand ranked_and; do
  timeout 400 python bench.py --op $op --steps 30 --warmup 3 > $OUT/bench_$op.json 2> $OUT/bench_$op.err; echo "bench $op rc=$?"
done
python - $OUT <<'PY'
import json,sys
for n in ["wand","ranked_and"]:
    try:
        d=json.loads(open(sys.argv[1]+"/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3), round(k.get("ms_alone") or 0,3))
        for k in d["roofline"].get("per_class",[]): print("   C", k["queries"], round(k["ms_per_launch"],3), "blocks", k.get("docs_blocks_decoded"), "scored", k.get("postings_scored"))
    except Exception as e: print(n, "FAILED", e)
PY
