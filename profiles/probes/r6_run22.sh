set -u
OUT=gpurun_out/${1:-r6w}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "pipeline or ticket or index_set or union_through" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
b gov2_b512 --batch 512 --depth 8 --steps 160 --warmup 80
b gov2_b1024 --batch 1024 --depth 6 --steps 120 --warmup 40
b gov2_b256 --batch 256 --depth 8 --steps 200 --warmup 80
b gov2_wand_b512 --op wand --batch 512 --depth 8 --steps 100 --warmup 40
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "resident", round(d["kernel_resident_qps"]), " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
