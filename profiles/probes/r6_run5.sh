set -u
OUT=gpurun_out/${1:-r6e}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/union_stream_probe.py 1 2 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -2 $OUT/union_probe.txt
timeout 600 python tests/ranked_stream_probe.py 1 2 > $OUT/ranked_probe.txt 2>&1; echo "ranked probe rc=$?"; tail -2 $OUT/ranked_probe.txt
timeout 400 python bench.py --op wand --no-oracle --steps 30 --warmup 3 > $OUT/bench_wand.json 2> $OUT/bench_wand.err
timeout 400 python bench.py --op maxscore --no-oracle --steps 30 --warmup 3 > $OUT/bench_maxscore.json 2> $OUT/bench_maxscore.err
timeout 400 python bench.py --op ranked_and --no-oracle --steps 40 --warmup 5 > $OUT/bench_ranked_and.json 2> $OUT/bench_ranked_and.err
DS2I_UNIT_CLOCK=1 timeout 400 python profiles/probes/unit_clock_probe.py and > $OUT/unit_clock_and.txt 2>&1; grep -A10 'unit clock: class' $OUT/unit_clock_and.txt | tail -50
DS2I_UNIT_CLOCK=1 timeout 400 python profiles/probes/unit_clock_probe.py ranked_and > $OUT/unit_clock_ranked_and.txt 2>&1; grep 'unit clock: class' $OUT/unit_clock_ranked_and.txt | tail -8
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3))
    except Exception as e: print(f, "FAILED", e)
PY
