#!/bin/bash
# Usage (GPU box): bash profiles/probes/r5_prio.sh -- stream priorities of the class streams x (first / second upload of the process)
export TMPDIR=/tmp
mkdir -p gpurun_out/r5_prio
for m in 0 1 2 3; do
  echo "== DS2I_FLAT_PRIORITY=$m (0 = unset: class priorities; 1 = equal; 2 = class 0 high; 3 = classes 0-1 high)"
  if [ $m = 0 ]; then timeout 600 python profiles/probes/reopen_probe.py short 2>&1 | grep -E "q/s|upload" 
  else DS2I_FLAT_PRIORITY=$m timeout 600 python profiles/probes/reopen_probe.py short 2>&1 | grep -E "q/s|upload"; fi
done | tee gpurun_out/r5_prio/out.txt
