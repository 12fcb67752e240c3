export TMPDIR=/tmp
mkdir -p gpurun_out/r04_phase
DS2I_LIB_VARIANT=phase timeout 900 python profiles/probes/phase_probe2.py > gpurun_out/r04_phase/phase.txt 2>&1
cat gpurun_out/r04_phase/phase.txt
