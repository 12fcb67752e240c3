set -u
OUT=gpurun_out/${1:-r6ab}
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-oracle > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
for ut in 320 160 96; do
export DS2I_UT_BLOCKS=$ut
b gov2_wand_b512_ut$ut --op wand --batch 512 --depth 8 --steps 100 --warmup 40
b gov2_wand_b1024_ut$ut --op wand --batch 1024 --depth 6 --steps 80 --warmup 30
done
unset DS2I_UT_BLOCKS
b gov2_and_b512 --op and --batch 512 --depth 8 --steps 100 --warmup 40
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", "host submit ms", round(d["host_submit_ms_per_step"],3), "resident", round(d["kernel_resident_qps"]))
    except Exception as e: print(f, "FAILED", e)
PY
timeout 1500 python -m pytest tests/test_gpu.py -x -q -k "pipeline or ticket or index_set or two_ranks or strong or small" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
