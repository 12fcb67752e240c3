#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/sq_pass.sh <tag> <workload> <op> [name-filter]   -- one SQ counter pass of bench.py
set -u
TAG=$1; WL=$2; OP=$3; FLT=${4:-k_}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/raw -o pmc -- \
    python bench.py --workload $WL --op $OP --steps 4 --warmup 1 --no-oracle > /dev/null 2> $OUT/err.txt
python - "$OUT" "$FLT" <<'PY'
import csv, glob, collections, sys
out, flt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("%s/raw/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("%s/counters.txt" % out, "w") as fo:
    for (k, n), v in sorted(agg.items()):
        if "rocclr" in k or flt not in k: continue
        line = "%s\t%s\tdispatches=%d\tmean=%.4g" % (k.replace("(anonymous namespace)::", "")[:60], n, len(v), sum(v) / len(v))
        fo.write(line + "\n"); print(line)
PY
rm -rf $OUT/raw
