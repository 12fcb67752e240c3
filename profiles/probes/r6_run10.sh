set -u
OUT=gpurun_out/${1:-r6k}
mkdir -p $OUT
export TMPDIR=/tmp
B="timeout 400 python bench.py --no-oracle --steps 40 --warmup 5"
for i in 1 2 3; do
  $B --op ranked_and > $OUT/bench_ra_occ6_$i.json 2> $OUT/bench_ra_occ6_$i.err
  DS2I_LIB_VARIANT=occ5 $B --op ranked_and > $OUT/bench_ra_occ5_$i.json 2> $OUT/bench_ra_occ5_$i.err
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", " ".join("%s=%.2f"%(k["kernel"][-3:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
