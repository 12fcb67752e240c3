"""profiles/<dir>/counters_fetch.txt (rocprofv3 --pmc FETCH_SIZE pass of the bench command) -> profiles/r04_traffic_<wl>_<op>.json:
HBM-side bytes per launch of every query kernel and per batch (step). FETCH_SIZE is in KB and, on gfx950, counts every
128-byte request at 64 bytes -- measured for streams AND for 1-byte gathers (profiles/r04_calib): x 2.
usage: make_traffic_json.py <profile dir> <workload> <op> [launches of each kernel per batch = 1]"""
import json, re, sys
d, wl, op = sys.argv[1], sys.argv[2], sys.argv[3]
per = {}
for line in open(d + "/counters_fetch.txt"):
    k, name, disp, mean = line.rstrip("\n").split("\t")
    if ", true>(" in k or re.search(r"k_ranked_stream<\d, true", k):  # the instrumented instantiations (one launch per bench run): not what a step runs
        continue
    if name != "FETCH_SIZE" or not re.search(r"k_ranked_stream|k_conjunctive|k_union|k_disjunctive|k_daat|k_merge", k):
        continue
    m = re.search(r"(k_\w+)<([^>(]*)", k)  # (rocprofv3 names are cut at 78 characters: the closing '>' may be missing)
    if not m:
        continue
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) == "k_ranked_stream":
        short = "k_ranked_stream<%s>" % args[0]
    elif m.group(1) == "k_conjunctive":
        short = "k_conjunctive<%s,%s,TMAX=%s>" % (args[0], args[1], args[2])
    elif m.group(1) == "k_union_topk":
        short = "k_union_topk<TMAX=%s> (%s)" % (args[0], op)
    else:
        short = m.group(1) + "<" + ",".join(args) + ">"
    per[short] = per.get(short, 0) + float(mean.split("=")[1]) * 1024.0 * 2.0
out = {"source": d + "/counters_fetch.txt", "factor": 2.0, "factor_calibration": "profiles/r04_calib (coalesced streams and scattered 1-byte gathers: one 128-B request tallied as 64 B)",
       "hbm_bytes_per_launch": {k: int(v) for k, v in sorted(per.items())}, "hbm_bytes_per_step": int(sum(per.values()))}
json.dump(out, open("profiles/r04_traffic_%s_%s.json" % (wl, op), "w"), indent=1)
print(json.dumps(out, indent=1))
