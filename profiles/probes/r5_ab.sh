#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_ab.sh <variant> [bench args]  -- product library against profiles/tmp_libs/lib_<variant>.so: parity subset on the variant, then alternating default-bench runs
export TMPDIR=/tmp
V=$1; shift
mkdir -p gpurun_out/r5_ab_$V
DS2I_LIB_VARIANT=$V timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "test_query_ops_match_oracle or fuzz_bit_identical or pruning_prunes or test_full_size_c2_properties or correlated" > gpurun_out/r5_ab_$V/pytest.txt 2>&1
tail -3 gpurun_out/r5_ab_$V/pytest.txt
run() { python bench.py --no-oracle --no-cpu-baseline --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)), [(k['kernel'][:22], round(k['ms_per_launch'],2)) for k in d['roofline']['per_kernel']])"; }
{
for i in 1 2 3; do
echo "== product"; run "$@"
echo "== $V"; DS2I_LIB_VARIANT=$V run "$@"
done
} | tee gpurun_out/r5_ab_$V/out.txt
