#!/bin/bash
# Usage: bash profiles/probes/run_pmc.sh <tag> <workload> "<counters>"   (one rocprofv3 --pmc pass, kernel-trace only)
set -u
TAG=$1; WL=$2; CTRS=$3
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw -o pmc -- \
    python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/err.txt
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob("%s/raw/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("%s/counters.txt" % out, "w") as fo:
    for (k, n), v in sorted(agg.items()):
        if "rocclr" in k: continue
        fo.write("%s\t%s\tdispatches=%d\tmean=%.1f\n" % (k[:70], n, len(v), sum(v) / len(v)))
print(open("%s/counters.txt" % out).read())
PY
rm -rf $OUT/raw
