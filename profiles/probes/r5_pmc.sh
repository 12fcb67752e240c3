#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/r5_pmc.sh <tag> [op] [extra bench args]   (round 5)
# Utilisation MEASURED, not derived: issue-busy cycles of the vector / scalar / LDS / VMEM pipes, what waves
# wait for, and how much of the L2's fabric traffic is DRAM. One rocprofv3 --pmc pass per group (own runs, --kernel-trace only).
set -u
TAG=${1:-r05}; OP=${2:-ranked_and}; shift; if [ $# -gt 0 ]; then shift; fi
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for PASS in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
            "sq2:SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
            "sq3:SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
            "sq4:SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
            "tcc1:TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ_DRAM_32B" \
            "tcc2:TCC_HIT TCC_MISS TCC_REQ TCC_READ" \
            "ta:TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_FLAT_READ_WAVEFRONTS TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES" \
            "fetch:FETCH_SIZE"; do
  NAME=${PASS%%:*}; CTRS=${PASS#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- \
      python bench.py --workload gov2 --op $OP --steps 4 --warmup 1 --no-oracle "$@" > /dev/null 2> $OUT/pmc_$NAME.err
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
  tail -2 $OUT/pmc_$NAME.err | cut -c1-300
done
grep -E "k_ranked_stream<2|k_union_topk<.*4," $OUT/counters_*.txt | cut -c1-220
