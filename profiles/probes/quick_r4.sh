# round-4 quick loop on the GPU box: parity tests that exercise the ranked conjunctive kernels, then the default bench
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04_q}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "${2:-query_ops or fuzz or full_size_c2_prop or prun or uninstrumented or topk or alternative}" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
python bench.py --workload gov2 --steps ${3:-20} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
grep -E "^class|postings|docs_blocks" $OUT/bench.err | cut -c1-300 | head -20
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("qps", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "resident", round(d.get("kernel_resident_qps",0)))
for c in d["roofline"]["per_class"]: print({k:(round(v,2) if isinstance(v,float) else v) for k,v in c.items()})
PY
