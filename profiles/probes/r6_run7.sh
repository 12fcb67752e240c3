set -u
OUT=gpurun_out/${1:-r6g}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tests/union_stream_probe.py 1 2 3 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -2 $OUT/union_probe.txt
timeout 600 python tests/ranked_stream_probe.py 1 2 3 > $OUT/ranked_probe.txt 2>&1; echo "ranked probe rc=$?"; tail -2 $OUT/ranked_probe.txt
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "pruning_fuzz or test_query_ops_match_oracle or disjunctive_scores or topk_other_k" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
B="timeout 400 python bench.py --no-oracle --steps 30 --warmup 3"
$B --op wand > $OUT/bench_wand.json 2> $OUT/bench_wand.err
$B --op maxscore > $OUT/bench_maxscore.json 2> $OUT/bench_maxscore.err
$B --op ranked_and > $OUT/bench_ranked_and.json 2> $OUT/bench_ranked_and.err
$B --op ranked_and > $OUT/bench_ranked_and2.json 2> $OUT/bench_ranked_and2.err
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3))
        for k in d["roofline"].get("per_class",[]): print("   C", k["queries"], round(k["ms_per_launch"],3), "blocks", k.get("docs_blocks_decoded"), "scored", k.get("postings_scored"))
    except Exception as e: print(f, "FAILED", e)
PY
