set -u
OUT=gpurun_out/${1:-r6f}
mkdir -p $OUT
export TMPDIR=/tmp
B="timeout 400 python bench.py --no-oracle --steps 30 --warmup 3"
$B --op wand > $OUT/bench_wand.json 2> $OUT/bench_wand.err
DS2I_UT_WARM=16 $B --op wand > $OUT/bench_wand_warm16.json 2> $OUT/bench_wand_warm16.err
DS2I_UT_WARM=32 DS2I_UT_BLOCKS=320 $B --op wand > $OUT/bench_wand_warm32_utb320.json 2> $OUT/bench_wand_warm32_utb320.err
DS2I_UT_BLOCKS=320 $B --op wand > $OUT/bench_wand_utb320.json 2> $OUT/bench_wand_utb320.err
$B --op and > $OUT/bench_and.json 2> $OUT/bench_and.err
DS2I_AND_UNIT_BLOCKS=48 $B --op and > $OUT/bench_and_48.json 2> $OUT/bench_and_48.err
DS2I_LIB_VARIANT=usphase timeout 500 python profiles/probes/us_phase_probe.py 2,3,4,6,8 > $OUT/us_phase.txt 2>&1; cat $OUT/us_phase.txt | tail -90
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3))
    except Exception as e: print(f, "FAILED", e)
PY
