set -u
OUT=gpurun_out/${1:-r6aa}
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
for alt in 1 0; do
if [ $alt = 0 ]; then export DS2I_NO_ALT=1; fi
b gov2_b256_alt$alt --batch 256 --depth 8 --steps 200 --warmup 80
b gov2_b512_alt$alt --batch 512 --depth 8 --steps 160 --warmup 80
b gov2_b1024_alt$alt --batch 1024 --depth 6 --steps 120 --warmup 40
b gov2_wand_b512_alt$alt --op wand --batch 512 --depth 8 --steps 100 --warmup 40
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "host submit ms", round(d["host_submit_ms_per_step"],3), "resident", round(d["kernel_resident_qps"]))
    except Exception as e: print(f, "FAILED", e)
PY
