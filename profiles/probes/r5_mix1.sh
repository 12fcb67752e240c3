#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_mix1
run() { echo "== $* $EXTRA"; env "$@" python bench.py --workload gov2 --no-oracle $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],3), 'resident', round(d.get('kernel_resident_qps',0)), d.get('step_ms_spread'))"; }
EXTRA="--batch 512 --depth 8 --steps 160 --warmup 80" run A=1
EXTRA="--batch 512 --depth 3 --steps 160 --warmup 80" run A=1
EXTRA="--batch 512 --depth 8 --steps 160 --warmup 80" run DS2I_UNIT_FACTOR=2
EXTRA="--steps 40 --warmup 5" run A=1
bash profiles/probes/run_pmc_cmd.sh instr1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY" profiles/probes/instr_probe.py 1 2>&1 | grep -E "k_conjunctive|^and|^ranked" | cut -c1-200
