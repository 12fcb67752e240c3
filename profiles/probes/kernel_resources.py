"""Per-kernel register / scratch / LDS / occupancy table of one kernels.hip translation unit (no GPU needed):
   python profiles/probes/kernel_resources.py <TMAX> [name-filter] [source]   (hipcc -Rpass-analysis=kernel-resource-usage)"""
import re, subprocess, sys
tmax = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
src = sys.argv[3] if len(sys.argv) > 3 else "/root/repo/ds2i_amd/csrc/kernels.hip"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DDS2I_TU_TMAX=" + tmax,
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[4:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\S+: )?\s*(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if flt in name:
        short = name.replace("(anonymous namespace)::", "").replace("(ds2i_dev::BatchArgs)", "").replace("void ", "")
        print(f"{short:48s} VGPR {r.get('VGPRs','?'):>4} SGPR {r.get('SGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} LDS {r.get('LDS Size [bytes/block]','?'):>6} occ {r.get('Occupancy [waves/SIMD]','?')}")
