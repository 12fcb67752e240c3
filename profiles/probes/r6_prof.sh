#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r6_prof.sh <tag>   -- the round's evidence on the build that ships:
#   kernel trace + stats of the default bench command (60 steps), then rocprofv3 --pmc passes of their own (SQ wait / busy, TCC fabric
#   requests, L2 hit rate, FETCH_SIZE) for ranked_and and wand, summaries under gpurun_out/prof_<tag>/ (copied to profiles/r06_gov2*/)
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/ranked_and $OUT/wand
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 60 --warmup 5 > $OUT/ranked_and/bench.json 2> $OUT/ranked_and/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); if [ -n "$KS" ]; then cp "$KS" $OUT/ranked_and/kernel_stats.csv; fi; rm -rf $OUT/kt
head -14 $OUT/ranked_and/kernel_stats.csv | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --op wand --steps 40 --warmup 5 --no-oracle > $OUT/wand/bench.json 2> $OUT/wand/bench.err
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); if [ -n "$KS" ]; then cp "$KS" $OUT/wand/kernel_stats.csv; fi; rm -rf $OUT/kt
head -10 $OUT/wand/kernel_stats.csv | cut -c1-160
for OP in ranked_and wand; do
for PASS in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
            "sq2:SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
            "tcc1:TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_DRAM_32B" \
            "tcc2:TCC_HIT TCC_MISS TCC_REQ TCC_READ" \
            "fetch:FETCH_SIZE"; do
  NAME=${PASS%%:*}; CTRS=${PASS#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- \
      python bench.py --workload gov2 --op $OP --steps 4 --warmup 1 --no-oracle > /dev/null 2> $OUT/$OP/pmc_$NAME.err
  python - "$OUT" "$NAME" "$OP" <<'PY'
import csv, glob, collections, sys
out, name, op = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, name), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("%s/%s/counters_%s.txt" % (out, op, name), "w") as fo:
    for (k, n), v in sorted(agg.items()):
        if "rocclr" in k: continue
        fo.write("%s\t%s\tdispatches=%d\tmean=%.1f\n" % (k[:78], n, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/pmc_$NAME
done
done
grep -E "k_ranked_stream<2, false|k_union_stream<2, false" $OUT/*/counters_sq1.txt | cut -c1-200
