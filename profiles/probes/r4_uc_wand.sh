export TMPDIR=/tmp
mkdir -p gpurun_out/r04_uc
DS2I_DEBUG_PLAN=1 python profiles/probes/unit_clock_probe.py wand > gpurun_out/r04_uc/uc_wand.txt 2>&1
grep -E "unit clock|unit [0-9]+:|kernel_ms|ds2i plan" gpurun_out/r04_uc/uc_wand.txt | head -48
