set -u
OUT=gpurun_out/${1:-r6ah}
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
for ns in 2 3; do
export DS2I_NSETS=$ns
b gov2_b256_ns$ns --batch 256 --depth 9 --steps 200 --warmup 80
b gov2_b512_ns$ns --batch 512 --depth 9 --steps 160 --warmup 80
b gov2_b1024_ns$ns --batch 1024 --depth 6 --steps 120 --warmup 40
b gov2_wand_b512_ns$ns --op wand --batch 512 --depth 9 --steps 100 --warmup 40
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "host submit ms", round(d["host_submit_ms_per_step"],3), "spread", d["step_ms_spread"]["max"])
    except Exception as e: print(f, "FAILED", e)
PY
