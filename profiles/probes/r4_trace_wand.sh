export TMPDIR=/tmp
OUT=gpurun_out/r04_trace; rm -rf $OUT; mkdir -p $OUT
OP=${1:-wand}
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/kt -o kt -- python bench.py --workload gov2 --op $OP --steps 12 --warmup 3 --no-oracle > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
for f in glob.glob(out + "/kt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:30], "-"))
rows.sort()
# the last ~60 events before the final ones of the timed region: take a window in the middle of the pipelined region
ks = [r for r in rows if "k_block_max" not in r[2] and "interleave" not in r[2]]
mid = len(ks) // 3
t0 = ks[mid][0]
for s, e, n, q in ks[mid:mid + 70]:
    print("%9.3f ms  +%8.3f ms  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
PY
