set -u
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
timeout 600 python tests/union_stream_probe.py 1 2 3 > gpurun_out/r6a/union_probe.txt 2>&1; echo "union probe rc=$?"; tail -5 gpurun_out/r6a/union_probe.txt
timeout 300 python tests/and_stream_probe.py 1 2 > gpurun_out/r6a/and_probe.txt 2>&1; echo "and probe rc=$?"; tail -3 gpurun_out/r6a/and_probe.txt
for op in wand maxscore and; do
  timeout 400 python bench.py --op $op --steps 30 --warmup 3 > gpurun_out/r6a/bench_$op.json 2> gpurun_out/r6a/bench_$op.err; echo "bench $op rc=$?"; tail -c 600 gpurun_out/r6a/bench_$op.json | head -c 300; echo
done
DS2I_NO_UNION_RSTREAM=1 timeout 400 python bench.py --op wand --no-oracle --steps 30 --warmup 3 > gpurun_out/r6a/bench_wand_old.json 2> gpurun_out/r6a/bench_wand_old.err
python - <<'PY'
import json
for n in ["wand","maxscore","and","wand_old"]:
    try:
        d=json.loads(open("gpurun_out/r6a/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), "q/s", round(d["ms_per_step"],3), "ms/step")
        for k in d["roofline"].get("per_kernel",[]): print("   ", k["kernel"], k["class"], k["queries"], round(k["ms_per_launch"],3))
    except Exception as e: print(n, "FAILED", e)
PY
