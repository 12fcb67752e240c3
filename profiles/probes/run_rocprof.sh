#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/probes/run_rocprof.sh <tag> <workload> [steps]
# Produces gpurun_out/prof_<tag>/{kernel_stats.csv, fetch_size.txt, write_size.txt, bench.json}
set -u
TAG=${1:-r01}; WL=${2:-gov2}; STEPS=${3:-10}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# pass 1: kernel trace + stats (no counters)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- \
    python bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$KS" ]; then cp "$KS" $OUT/kernel_stats.csv; fi
# pass 2/3: HBM traffic counters, each in its own run (MI355X_MICROARCH.md §HBM)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- \
      python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err
  python - "$OUT" "$C" <<'PY'
import csv, glob, collections, sys
out, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, c), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("%s/%s.txt" % (out, c.lower()), "w") as fo:
    for (k, n), v in sorted(agg.items()):
        fo.write("%s\t%s\tdispatches=%d\tmean=%.1f\n" % (k[:90], n, len(v), sum(v) / len(v)))
PY
done
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
