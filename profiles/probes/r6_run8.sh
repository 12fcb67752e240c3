set -u
OUT=gpurun_out/${1:-r6i}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu.py -x -q -k "alternative_paths or union_through or and_through or ranked_stream_5_to_8 or table_budget or transcoded or error_behaviour or without_side_tables or both_decoders or block_mixed" > $OUT/pytest_knobs.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_knobs.txt
DS2I_LIB_VARIANT=rsphase timeout 400 python profiles/probes/rs_phase_probe.py > $OUT/rs_phase.txt 2>&1; grep -A14 '^2 terms' $OUT/rs_phase.txt
DS2I_LIB_VARIANT=usphase timeout 400 python profiles/probes/us_phase_probe.py 2 > $OUT/us_phase.txt 2>&1; grep -A14 '^2 terms' $OUT/us_phase.txt
