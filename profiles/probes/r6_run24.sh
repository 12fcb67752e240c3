set -u
OUT=gpurun_out/${1:-r6z}
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
for cap in 1 1.5 2 3; do
export DS2I_POOL_CAP=$cap
b gov2_b256_cap$cap --batch 256 --depth 8 --steps 200 --warmup 80
b gov2_b512_cap$cap --batch 512 --depth 8 --steps 160 --warmup 80
b gov2_b1024_cap$cap --batch 1024 --depth 6 --steps 120 --warmup 40
b gov2_b2048_cap$cap --batch 2048 --depth 4 --steps 80 --warmup 20
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "host submit ms", round(d["host_submit_ms_per_step"],3), "resident", round(d["kernel_resident_qps"]))
    except Exception as e: print(f, "FAILED", e)
PY
