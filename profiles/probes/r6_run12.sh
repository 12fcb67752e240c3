set -u
OUT=gpurun_out/${1:-r6n}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/and_stream_probe.py 1 2 3 > $OUT/and_probe.txt 2>&1; echo "and probe rc=$?"; tail -2 $OUT/and_probe.txt
DS2I_UNIT_CAP=8 timeout 600 python tests/and_stream_probe.py 1 > $OUT/and_probe_cap.txt 2>&1; echo "and probe cap rc=$?"; tail -1 $OUT/and_probe_cap.txt
timeout 900 python -m pytest tests/test_gpu.py -x -q -k "test_query_ops_match_oracle or uninstrumented or correlated or brute_force or full_size_c2_properties" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
