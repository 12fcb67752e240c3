set -u
OUT=gpurun_out/${1:-r6ad}
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
for aub in 96 48 24; do
DS2I_AUB=$aub b gov2_and_freq_aub$aub --op and_freq --steps 24 --warmup 3
done
for ut in 240 480; do
DS2I_UT_BLOCKS=$ut b gov2_wand_ut$ut --op wand --steps 24 --warmup 3
done
DS2I_AUB=192 b gov2_and_aub192 --op and --steps 30 --warmup 3
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "resident", round(d["kernel_resident_qps"]), " ".join("%s=%.2f"%(k["kernel"][-9:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
