"""profiles/<dir>/counters_fetch.txt (rocprofv3 --pmc FETCH_SIZE pass of the bench command) -> profiles/r04_traffic_<wl>_<op>.json:
HBM-side bytes per launch of every query kernel and per batch (step). FETCH_SIZE is in KB and, on gfx950, counts every
128-byte request at 64 bytes -- measured for streams AND for 1-byte gathers (profiles/r04_calib): x 2.
usage: make_traffic_json.py <profile dir> <workload> <op> [round prefix = r05]
When the directory also holds counters_tcc1.txt (TCC_EA0_RDREQ / _DRAM / _32B) the request counts are carried along: every
fabric read of these kernels is a 128-byte request addressed to DRAM (TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ, _32B == 0); whether the
Infinity Cache served it is invisible to the L2's counters."""
import json, os, re, sys
d, wl, op = sys.argv[1], sys.argv[2], sys.argv[3]
prefix = sys.argv[4] if len(sys.argv) > 4 else "r05"
per = {}
for line in open(d + "/counters_fetch.txt"):
    k, name, disp, mean = line.rstrip("\n").split("\t")
    if ", true>(" in k or re.search(r"k_(ranked|union)_stream<\d+, true", k):  # the instrumented instantiations (one launch per bench run): not what a step runs
        continue
    if name != "FETCH_SIZE" or not re.search(r"k_ranked_stream|k_conjunctive|k_union|k_disjunctive|k_daat|k_merge|k_freq_stream|k_and_stream", k):
        continue
    m = re.search(r"(k_\w+)<([^>(]*)", k)  # (rocprofv3 names are cut at 78 characters: the closing '>' may be missing)
    if not m:
        continue
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) == "k_ranked_stream":
        short = "k_ranked_stream<%s%s>" % (args[0], ",AND" if len(args) > 2 and args[2] == "true" else "")
    elif m.group(1) == "k_union_stream":
        short = "k_union_stream<%s>" % args[0]
    elif m.group(1) == "k_ranked_stream_mixed":
        short = "k_ranked_stream_mixed<%s>" % args[0]
    elif m.group(1) == "k_conjunctive":
        short = "k_conjunctive<%s,%s,TMAX=%s>" % (args[0], args[1], args[2])
    elif m.group(1) == "k_union_topk":
        short = "k_union_topk<TMAX=%s> (%s)" % (args[0], op)
    else:
        short = m.group(1) + "<" + ",".join(args) + ">"
    per[short] = per.get(short, 0) + float(mean.split("=")[1]) * 1024.0 * 2.0
out = {"source": d + "/counters_fetch.txt", "factor": 2.0, "factor_calibration": "profiles/r04_calib (coalesced streams and scattered 1-byte gathers: one 128-B request tallied as 64 B)",
       "hbm_bytes_per_launch": {k: int(v) for k, v in sorted(per.items())}, "hbm_bytes_per_step": int(sum(per.values()))}
req = {}
tcc = d + "/counters_tcc1.txt"
if os.path.exists(tcc):
    for line in open(tcc):
        k, name, disp, mean = line.rstrip("\n").split("\t")
        if re.search(r"k_ranked_stream|k_conjunctive|k_union|k_disjunctive|k_daat|k_freq_stream|k_and_stream", k) and ", true>(" not in k and not re.search(r"k_(ranked|union)_stream<\d+, true", k):
            req[name] = req.get(name, 0.0) + float(mean.split("=")[1])
    out["l2_fabric_read_requests_per_step"] = {k: int(v) for k, v in sorted(req.items())}
    out["l2_fabric_note"] = ("TCC_EA0_RDREQ = read requests the L2s sent to the fabric, all of them 128-byte (TCC_EA0_RDREQ_32B = 0; _DRAM_32B = 4 x) and all "
                             "addressed to local DRAM (TCC_EA0_RDREQ_DRAM equal): x 128 B = the FETCH_SIZE x 2 figure. The Infinity Cache sits behind that "
                             "interface: its hits are not separable with the TCC counters this rocprofv3 exposes, so hbm_bytes_per_step is an UPPER bound of "
                             "what reached HBM")
json.dump(out, open("profiles/%s_traffic_%s_%s.json" % (prefix, wl, op), "w"), indent=1)
print(json.dumps(out, indent=1))
