#!/usr/bin/env python
"""Fills the @NAME@ placeholders of DESIGN.md / README.md from the bench lines of profiles/probes/r5_close2.sh
(gpurun_out/r5_close2/bench_*.json, copied to profiles/r05_bench_*.json first). Usage: python profiles/probes/r5_fill_design.py"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "r5_close2")


def load(name):
    p = os.path.join(SRC, "bench_%s.json" % name)
    if not os.path.exists(p):
        p = os.path.join(ROOT, "profiles", "r05_bench_%s.json" % name)
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:  # a line that did not run stays a placeholder
        print("missing", name, e, file=sys.stderr)
        return None


def k(v):
    return ("%d k" % round(v / 1e3)) if v < 999500 else ("%.0f %03d k" % (v // 1e6, round((v % 1e6) / 1e3)))


def main():
    if os.path.isdir(SRC):
        for f in os.listdir(SRC):
            if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(os.path.join(SRC, f)) > 0:
                shutil.copy(os.path.join(SRC, f), os.path.join(ROOT, "profiles", "r05_" + f))
    rep = {}
    d = load("gov2_opt")
    if d:
        cpu = (d.get("cpu_baseline") or {}).get("value")
        rep.update(OPT_T=k(d["value"]), OPT_T_MS="%.2f" % d["ms_per_step"], OPT_HBM="%.1f" % (d["config"]["device_bytes"] / 1e9))
        if cpu:
            rep["OPT_T_X"] = "{:,}".format(int(round(d["value"] / cpu, -1))).replace(",", " ")
    d = load("gov2_opt_wand")
    if d:
        rep["OPT_W"] = k(d["value"])
    o = load("cw09_optpfor")
    if o:
        cpu = (o.get("cpu_baseline") or {}).get("value")
        rep.update(CW_O=k(o["value"]), CW_O_MS="%.2f" % o["ms_per_step"], CW_O_CPU=("%d q/s" % round(cpu)) if cpu else "")
    for tag, name in (("CW_MF", "cw09_mixed_fixed"), ("CW_MN", "cw09_mixed_fixed_native"), ("CW_MO", "cw09_mixed_optimised")):
        d = load(name)
        if d:
            rep[tag] = k(d["value"])
            rep[tag + "_MS"] = "%.2f" % d["ms_per_step"]
            if tag == "CW_MF":
                rep["CW_MF_CPU"] = ""
                if o:
                    rep["CW_MF_PCT"] = "%d %%" % round(100.0 * d["value"] / o["value"])
    for tag, name in (("B512", "gov2_b512"), ("B1024", "gov2_b1024"), ("B2048", "gov2_b2048")):
        d = load(name)
        if d:
            rep[tag] = k(d["value"])
    for doc in ("DESIGN.md", "README.md"):
        p = os.path.join(ROOT, doc)
        s = open(p).read()
        s2 = re.sub(r"@([A-Z0-9_]+)@", lambda m: rep.get(m.group(1), m.group(0)), s)
        if s2 != s:
            open(p, "w").write(s2)
        left = sorted(set(re.findall(r"@([A-Z0-9_]+)@", s2)))
        print(doc, "filled", len(set(re.findall(r"@([A-Z0-9_]+)@", s))) - len(left), "left:", left)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
