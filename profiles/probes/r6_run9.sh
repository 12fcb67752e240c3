set -u
OUT=gpurun_out/${1:-r6j}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/union_stream_probe.py 1 2 > $OUT/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -1 $OUT/union_probe.txt
B="timeout 400 python bench.py --no-oracle --steps 40 --warmup 5"
for i in 1 2; do
  $B --op wand > $OUT/bench_wand_l1_$i.json 2> $OUT/bench_wand_l1_$i.err
  DS2I_LIB_VARIANT=fwsc1 $B --op wand > $OUT/bench_wand_sc1_$i.json 2> $OUT/bench_wand_sc1_$i.err
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], round(k["ms_per_launch"],3))
    except Exception as e: print(f, "FAILED", e)
PY
