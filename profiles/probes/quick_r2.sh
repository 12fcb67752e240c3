export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "ranked or oracle or prun or full_size or gov2 or pipeline" 2>&1 | tail -3
python bench.py --workload gov2 --steps 10 --warmup 2 --no-oracle 2>&1 | grep -E "^class|^\{" | cut -c1-420
mkdir -p gpurun_out/q; timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/q/raw -o pmc -- python bench.py --workload gov2 --steps 3 --warmup 1 --no-oracle > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/q/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, n), v in sorted(agg.items()):
    if "conjunctive" in k and ", false>" in k: print(k[40:75], n, len(v), "mean KB=%.0f -> x2 = %.2f GB" % (sum(v)/len(v), 2*sum(v)/len(v)/1e6))
PY
rm -rf gpurun_out/q
