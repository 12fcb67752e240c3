set -u
OUT=gpurun_out/${1:-r6ai}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --workload c2 --steps 5 --warmup 1 > $OUT/b1.json 2> $OUT/b1.err; echo rc=$?
timeout 300 python bench.py --workload c2 --op wand --steps 5 --warmup 1 --no-oracle > $OUT/b2.json 2> $OUT/b2.err; echo rc=$?
timeout 300 python bench.py --workload c2 --op or --steps 5 --warmup 1 --no-cpu-baseline > $OUT/b3.json 2> $OUT/b3.err; echo rc=$?
python - $OUT <<'PY'
import json,sys
for n in ("b1","b2","b3"):
    d=json.loads(open(sys.argv[1]+"/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["value"]), [r["kernel"] for r in d["roofline"]["per_class"]])
PY
