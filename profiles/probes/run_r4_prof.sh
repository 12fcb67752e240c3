#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/run_r4_prof.sh <tag> [workload] [op]   (round 4)
# pass 1: rocprofv3 --kernel-trace --stats of the default bench command; passes 2-4: PMC counters, each in its own run
set -u
TAG=${1:-r04}; WL=${2:-gov2}; OP=${3:-ranked_and}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- \
    python bench.py --workload $WL --op $OP --steps 60 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$KS" ]; then cp "$KS" $OUT/kernel_stats.csv; fi
rm -rf $OUT/kt
for PASS in "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  NAME=${PASS%%:*}; CTRS=${PASS#*:}
  timeout 420 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- \
      python bench.py --workload $WL --op $OP --steps 4 --warmup 1 --no-oracle > /dev/null 2> $OUT/pmc_$NAME.err
  python - "$OUT" "$NAME" <<'PY'
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
  rm -rf $OUT/pmc_$NAME
done
ls -la $OUT; cat $OUT/kernel_stats.csv | head -12; grep -E "k_ranked_stream|k_conjunctive<true, true|k_union|k_disj" $OUT/counters_*.txt | grep -v ", true>(" | cut -c1-200
