export TMPDIR=/tmp
OUT=gpurun_out/${1:-r03_andor}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu.py -m gpu -x -q -k "${2:-query_ops and (and or or-) or pruning_tables or brute or full_size_c2 or long_queries or uninstrumented or alternative}" > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
bash profiles/probes/ab_env_op.sh and "X=1" "DS2I_NO_BITMAP_USE=1"
bash profiles/probes/ab_env_op.sh or "X=1" "DS2I_NO_BITMAP_USE=1"
bash profiles/probes/ab_env_op.sh and_freq "X=1"
bash profiles/probes/ab_env_op.sh or_freq "X=1"
