set -u
OUT=gpurun_out/${1:-r6d}
mkdir -p $OUT
export TMPDIR=/tmp
DS2I_UNIT_CLOCK=1 timeout 400 python profiles/probes/unit_clock_probe.py wand > $OUT/unit_clock_wand.txt 2>&1; grep -A12 'unit clock: class' $OUT/unit_clock_wand.txt | grep -v '    unit' | tail -60
for v in 64 320; do
  DS2I_UT_BLOCKS=$v timeout 400 python bench.py --op wand --no-oracle --steps 30 --warmup 3 > $OUT/bench_wand_utb$v.json 2> $OUT/bench_wand_utb$v.err
done
timeout 400 python bench.py --op ranked_and --no-oracle --steps 40 --warmup 5 > $OUT/bench_ranked_and.json 2> $OUT/bench_ranked_and.err
timeout 400 python bench.py --op ranked_and --no-oracle --steps 40 --warmup 5 > $OUT/bench_ranked_and2.json 2> $OUT/bench_ranked_and2.err
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3))
    except Exception as e: print(f, "FAILED", e)
PY
