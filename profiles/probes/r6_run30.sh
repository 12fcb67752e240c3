set -u
OUT=gpurun_out/${1:-r6af}
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b wand_multi --op wand --steps 24 --warmup 3
b ranked_multi --steps 40 --warmup 4
export DS2I_ONE_STREAM=1
b wand_one --op wand --steps 24 --warmup 3
b ranked_one --steps 40 --warmup 4
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "resident", round(d["kernel_resident_qps"]), " ".join("%s=%.2f"%(k["kernel"][-9:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
