# round-4 closing pass on the GPU box: the whole GPU suite, the committed profile (kernel stats + PMC passes) for ranked_and,
# a FETCH_SIZE pass for wand, then every bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_close
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_close/tests.log 2>&1
tail -3 gpurun_out/r04_close/tests.log
bash profiles/probes/run_r4_prof.sh r04 gov2 ranked_and > gpurun_out/r04_close/prof.log 2>&1
tail -25 gpurun_out/r04_close/prof.log
OUT=gpurun_out/prof_r04_wand; mkdir -p $OUT
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- \
    python bench.py --workload gov2 --op wand --steps 4 --warmup 1 --no-oracle > /dev/null 2> $OUT/pmc_fetch.err
python - "$OUT" fetch <<'PY'
import csv, glob, collections, sys
out, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, name), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("%s/counters_%s.txt" % (out, name), "w") as fo:
    for (k, n), v in sorted(agg.items()):
        if "rocclr" in k: continue
        fo.write("%s\t%s\tdispatches=%d\tmean=%.1f\n" % (k[:78], n, len(v), sum(v) / len(v)))
PY
rm -rf $OUT/pmc_fetch
grep k_union $OUT/counters_fetch.txt | cut -c1-160
bash profiles/probes/run_r4_all.sh 2>&1 | tee gpurun_out/r04_close/all.log
