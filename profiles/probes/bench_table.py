"""profiles/r<NN>_bench_*.json -> the rows of DESIGN.md section 6's table. usage: bench_table.py 04"""
import json, glob, sys, os
rnd = sys.argv[1] if len(sys.argv) > 1 else "04"
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "r%s_bench_*.json" % rnd))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
        continue
    r, c = d["roofline"], d.get("cpu_baseline") or {}
    print("%-40s %9.0f q/s  %6.2f ms/step  resident %9.0f  e2e/res %.2f  frac %.3f step_frac %s  kernel %s %.2f ms  cpu %s q/s (x%s)" % (
        os.path.basename(f)[len("r%s_bench_" % rnd):-5], d["value"], d["ms_per_step"], d.get("kernel_resident_qps", 0), d.get("end_to_end_over_resident", 0),
        r["frac"], r.get("step_frac") and round(r["step_frac"], 3), r.get("kernel"), r.get("kernel_ms", 0),
        c.get("value") and round(c["value"], 1), c.get("value") and round(d["value"] / c["value"])))
