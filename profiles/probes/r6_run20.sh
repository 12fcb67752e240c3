set -u
OUT=gpurun_out/${1:-r6u}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tests/union_stream_probe.py 1 2 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -2 $OUT/union_probe.txt
for op in wand maxscore ranked_or; do
timeout 600 python bench.py --op $op --steps 30 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_gov2_$op.json 2> $OUT/bench_gov2_$op.err
done
timeout 600 python bench.py --workload gov2c --op wand --steps 30 --warmup 3 --no-cpu-baseline --no-oracle > $OUT/bench_gov2c_wand.json 2> $OUT/bench_gov2c_wand.err
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", " ".join("%s=%.2f"%(k["kernel"][-7:],k["ms_alone"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py --workload gov2 --op wand --steps 4 --warmup 1 --no-oracle --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc_fetch/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, n), v in sorted(agg.items()):
    if "stream" in k and ", false" in k: print(k[30:70], n, len(v), "GB/launch %.2f" % (sum(v)/len(v)*1024*2/1e9))
PY
rm -rf $OUT/pmc_fetch
