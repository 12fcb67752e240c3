#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/pmc_mem.sh <tag> [workload] [op]
# L1 / L2 request counters of the class kernels, each group in its own rocprofv3 --pmc run (no trace domains)
set -u
TAG=${1:-mem}; WL=${2:-gov2}; OP=${3:-ranked_and}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# (a TCC_* pass -- TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_BUSY_avr -- left a dispatch incomplete and ran
# into the call's time limit on this pool: only the TCP group is collected)
for PASS in "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  NAME=${PASS%%:*}; CTRS=${PASS#*:}
  timeout 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- \
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
        if "rocclr" in k or ", true>(" in k: continue
        fo.write("%s\t%s\tdispatches=%d\tmean=%.1f\n" % (k[:78], n, len(v), sum(v) / len(v)))
PY
  tail -2 $OUT/pmc_$NAME.err | cut -c1-200
  rm -rf $OUT/pmc_$NAME
done
grep -E "k_conjunctive|k_union|k_disj" $OUT/counters_*.txt | awk -F'\t' '{split($1,a,"<"); printf "%-24s %-36s %s\n", substr(a[2],1,22), $2, $4}'
