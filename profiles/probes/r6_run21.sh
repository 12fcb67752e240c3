set -u
OUT=gpurun_out/${1:-r6v}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tests/union_stream_probe.py 1 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -2 $OUT/union_probe.txt
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "union_through or topk_beyond_64 or gov2_scale or reference_order_disj or one_term" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
for op in wand maxscore; do
timeout 600 python bench.py --op $op --steps 30 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_gov2_$op.json 2> $OUT/bench_gov2_$op.err
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "resident", round(d["kernel_resident_qps"]), " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_alone"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
