# FETCH_SIZE by access shape: one rocprofv3 --pmc pass per counter group (no trace domains besides --kernel-trace)
export TMPDIR=/tmp
OUT=gpurun_out/r04_calib
rm -rf $OUT; mkdir -p $OUT
for PASS in "fetch:FETCH_SIZE" "req:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  NAME=${PASS%%:*}; CTRS=${PASS#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- profiles/probes/calib_gather 8 28 > $OUT/run_$NAME.txt 2>&1
  python - "$OUT" "$NAME" <<'PY'
import csv, glob, collections, sys
out, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, name), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("%s/counters_%s.txt" % (out, name), "w") as fo:
    for (k, n), v in sorted(agg.items()):
        fo.write("%s\t%s\tdispatches=%d\tmean=%.1f\n" % (k[:60], n, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/pmc_$NAME
done
cat $OUT/run_fetch.txt | grep calib; cat $OUT/counters_*.txt | grep -v "__amd"
