#!/bin/bash
# Usage: bash profiles/probes/run_pmc_cmd.sh <tag> "<counters>" <python script + args...>   (one rocprofv3 --pmc pass over any script)
set -u
TAG=$1; CTRS=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/raw -o pmc -- python "$@" > $OUT/out.txt 2> $OUT/err.txt
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
        fo.write("%s\t%s\tdispatches=%d\tlast=%.1f\n" % (k[:70], n, len(v), v[-1]))
print(open("%s/counters.txt" % out).read())
PY
cat $OUT/out.txt
rm -rf $OUT/raw
