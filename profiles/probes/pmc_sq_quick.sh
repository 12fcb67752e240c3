#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probes/pmc_sq_quick.sh [op]  -- SALU / VALU wave instructions per batch of the class kernels + a 30-step line
set -u
OP=${1:-ranked_and}
OUT=gpurun_out/pmc_sq_quick
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -o pmc -- \
    python bench.py --workload gov2 --op $OP --steps 4 --warmup 1 --no-oracle > /dev/null 2> $OUT/pmc.err
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc/**/*counter_collection.csv" % out, recursive=True):
    for r in csv.DictReader(open(f)):
        if ", true>(" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]: continue
        if not any(k in r["Kernel_Name"] for k in ("k_conjunctive", "k_union", "k_disjunctive", "k_ranked_stream")): continue
        agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
tot = collections.defaultdict(float)
for (k, n), v in sorted(agg.items()):
    m = sum(v) / len(v); tot[n] += m
    print("%-72s %-14s %.4g" % (k, n, m))
print("per batch:", {n: "%.4g" % v for n, v in tot.items()}, "SALU/VALU %.3f" % (tot["SQ_INSTS_SALU"] / max(1.0, tot["SQ_INSTS_VALU"])))
PY
rm -rf $OUT/pmc
python bench.py --workload gov2 --op $OP --steps 30 --warmup 4 --no-oracle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$OP', round(d['value']), round(d['ms_per_step'],2), d['step_ms_spread'])"
