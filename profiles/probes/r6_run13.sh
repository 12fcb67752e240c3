set -u
OUT=gpurun_out/${1:-r6o}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tests/ranked_stream_probe.py 1 2 3 > $OUT/ranked_probe.txt 2>&1; echo "ranked probe rc=$?"; tail -2 $OUT/ranked_probe.txt
DS2I_UNIT_CAP=8 timeout 900 python tests/ranked_stream_probe.py 1 > $OUT/ranked_probe_cap.txt 2>&1; echo "ranked probe cap rc=$?"; tail -1 $OUT/ranked_probe_cap.txt
timeout 600 python tests/and_stream_probe.py 1 2 3 > $OUT/and_probe.txt 2>&1; echo "and probe rc=$?"; tail -2 $OUT/and_probe.txt
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "pruning_fuzz or test_query_ops_match_oracle or long_queries" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
B="timeout 400 python bench.py --no-oracle --steps 40 --warmup 5"
for op in ranked_and and and_freq; do $B --op $op > $OUT/bench_$op.json 2> $OUT/bench_$op.err; done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
