#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r4_pmc.sh <tag> [op] -- issue / wait counters of the class kernels, one rocprofv3 --pmc pass per group
set -u
TAG=${1:-r04}; OP=${2:-ranked_and}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for PASS in "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "issue:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "fetch:FETCH_SIZE"; do
  NAME=${PASS%%:*}; CTRS=${PASS#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- \
      python bench.py --workload gov2 --op $OP --steps 4 --warmup 1 --no-oracle > /dev/null 2> $OUT/pmc_$NAME.err
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
  rm -rf $OUT/pmc_$NAME
done
grep -E "k_ranked_stream|k_conjunctive<true, true, [28]|k_union|k_disj" $OUT/counters_*.txt | cut -c1-200
