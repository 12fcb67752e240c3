export TMPDIR=/tmp
OUT=gpurun_out/r04_t3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu.py -m gpu -x -q -k "fuzz or (query_ops and ranked_and) or full_size_c2_prop or prun or uninstrumented" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
STEPS=40 bash profiles/probes/r4_ablib.sh main base
