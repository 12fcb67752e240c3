set -u
OUT=gpurun_out/${1:-r6ac}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.txt
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b gov2_b512 --batch 512 --depth 8 --steps 160 --warmup 80
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "host submit ms", round(d["host_submit_ms_per_step"],3), "resident", round(d["kernel_resident_qps"]))
    except Exception as e: print(f, "FAILED", e)
PY
