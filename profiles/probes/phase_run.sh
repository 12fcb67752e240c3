# diagnostic: the -DDS2I_PHASE_TIMING build of the library (profiles/tmp_libs/lib_phase.so) under bench.py
export TMPDIR=/tmp
cp ds2i_amd/libds2i_hip.so /tmp/orig.so
cp profiles/tmp_libs/lib_phase.so ds2i_amd/libds2i_hip.so
python bench.py --workload gov2 --steps 3 --warmup 1 --no-oracle "$@" 2>&1 | grep -E "^class|phase cycles" | cut -c1-400
cp /tmp/orig.so ds2i_amd/libds2i_hip.so
