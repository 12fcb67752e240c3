set -u
OUT=gpurun_out/${1:-r6l}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/and_stream_probe.py 1 2 3 > $OUT/and_probe.txt 2>&1; echo "and probe rc=$?"; tail -2 $OUT/and_probe.txt
DS2I_UNIT_CAP=8 timeout 600 python tests/and_stream_probe.py 1 > $OUT/and_probe_cap.txt 2>&1; echo "and probe cap rc=$?"; tail -1 $OUT/and_probe_cap.txt
DS2I_NO_RMH=1 timeout 600 python tests/and_stream_probe.py 2 > $OUT/and_probe_normh.txt 2>&1; echo "and probe normh rc=$?"; tail -1 $OUT/and_probe_normh.txt
B="timeout 400 python bench.py --steps 30 --warmup 3"
$B --op and_freq > $OUT/bench_and_freq.json 2> $OUT/bench_and_freq.err
DS2I_NO_RANKED_STREAM=1 $B --no-oracle --op and_freq > $OUT/bench_and_freq_old.json 2> $OUT/bench_and_freq_old.err
for bs in 512 1024 2048; do
  timeout 400 python bench.py --no-oracle --batch $bs --depth 8 --steps 120 --warmup 20 > $OUT/bench_b$bs.json 2> $OUT/bench_b$bs.err
done
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step", " ".join("%s=%.2f"%(k["kernel"][-6:],k["ms_per_launch"]) for k in d["roofline"].get("per_kernel",[])))
    except Exception as e: print(f, "FAILED", e)
PY
