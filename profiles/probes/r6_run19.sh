set -u
OUT=gpurun_out/${1:-r6t}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python tests/union_stream_probe.py 1 2 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -3 $OUT/union_probe.txt
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "topk_beyond_64 or topk_other_k or long_queries or pipeline_matches" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
timeout 600 python bench.py --workload c2 --k 100 --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_c2_k100.json 2> $OUT/bench_c2_k100.err
timeout 600 python bench.py --op wand --k 100 --steps 20 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_gov2_wand_k100.json 2> $OUT/bench_gov2_wand_k100.err
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
