#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_defer.sh -- stage C deferred (survivor queue; apply profiles/probes/r5_defer_stage_c.patch and build the variants defer16 .. defer64 with DS2I_BUILD_VARIANT / DS2I_EXTRA_CFLAGS=-DDS2I_RS_DEFER=N first) against stage C per block: parity subset, then the default bench per flush size
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_defer
DS2I_LIB_VARIANT=defer32 timeout 900 python -m pytest tests/test_gpu.py -x -q -m gpu -k "test_query_ops_match_oracle or fuzz_bit_identical or pruning_prunes or test_full_size_c2_properties or correlated" > gpurun_out/r5_defer/pytest.txt 2>&1
tail -5 gpurun_out/r5_defer/pytest.txt
run() { python bench.py --no-oracle --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)), [(k['kernel'][:22], round(k['ms_per_launch'],2)) for k in d['roofline']['per_kernel']])"; }
{
echo "== shipped"; run
for n in 16 32 48 64; do echo "== defer$n"; DS2I_LIB_VARIANT=defer$n run; done
echo "== shipped again"; run
} | tee gpurun_out/r5_defer/out.txt
