bash profiles/probes/r6_final.sh opt 2>&1 | tee gpurun_out/r6_final_opt_log.txt
DS2I_LIB_VARIANT=usphase timeout 400 python profiles/probes/us_phase_probe.py 2,3 > gpurun_out/r6_final/us_phase2.txt 2>&1; cat gpurun_out/r6_final/us_phase2.txt | head -34
