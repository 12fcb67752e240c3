set -u
OUT=gpurun_out/${1:-r6c}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "pruning_fuzz or ranked_stream_5_to_8 or and_through or test_query_ops_match_oracle" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
DS2I_UNIT_CLOCK=1 timeout 400 python profiles/probes/unit_clock_probe.py wand > $OUT/unit_clock_wand.txt 2>&1; grep -A9 'unit clock' $OUT/unit_clock_wand.txt | head -60
for op in ranked_and and; do
  timeout 400 python bench.py --op $op --steps 30 --warmup 3 > $OUT/bench_$op.json 2> $OUT/bench_$op.err; echo "bench $op rc=$?"
done
python - $OUT <<'PY'
import json,sys
for n in ["ranked_and","and"]:
    try:
        d=json.loads(open(sys.argv[1]+"/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3), round(k.get("ms_alone") or 0,3))
        for k in d["roofline"].get("per_class",[]): print("   C", k["queries"], round(k["ms_per_launch"],3), "blocks", k.get("docs_blocks_decoded"), "scored", k.get("postings_scored"))
    except Exception as e: print(n, "FAILED", e)
PY
