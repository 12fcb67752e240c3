#!/usr/bin/env python
"""bench.py -- queries/sec of ranked_and over a synthetic block_optpfor index on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU. A "step" = one pass of the hot path (ranked_and, k=10)
over one 4096-query batch per GPU, with the index, the prepared query terms and the output buffers
already resident in HBM. Rank 0 prints ONE JSON line.

  value        whole-job queries/s = (queries all ranks processed) / max-over-ranks wall time
  roofline     dominant kernel (k_conjunctive<ranked>, <=4-term class): algorithmic bytes (the reference
               traversal's A_skip, SURVEY.md §8(d), counted by the instrumented oracle) / that kernel's
               mean hipEvent duration, against the 8 TB/s HBM peak
  cpu_baseline the oracle (CPU restatement of the reference path, oracle/) driven like op_perftest
               (queries.cpp:13-62) on ONE host core over a bounded sample of the same batch -- the only
               place this file touches oracle/.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: synthetic 1M-doc Zipf collection, batch = 4096 queries
    "c2": dict(num_docs=1_000_000, num_terms=65536, zipf_exp=0.75, top_df_frac=0.5, min_len=128, clustered_every=4,
               seed=0xD5210002, label="synthetic 1M-doc Zipf (configs[1])"),
    # BASELINE.json metric ("GOV2-scale"): 25M docs, ~1.0 B postings (SURVEY.md §8(d) C3/C4 shape)
    "gov2": dict(num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4,
                 seed=0xD5210004, label="synthetic GOV2-scale 25M-doc Zipf (metric config)"),
    # BASELINE.json configs[4] ("ClueWeb09-B-scale"): 50M docs, ~3.5 B postings; meant for --codec block_mixed and
    # --gpus 8 (every rank holds a replica and answers its own batch, so one rank alone runs the per-GPU work)
    "cw09": dict(num_docs=50_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4,
                 seed=0xD5210005, label="synthetic ClueWeb09-B-scale 50M-doc Zipf (configs[4])"),
}
NCLS = 4  # kernel classes by distinct query terms: <=2, <=4, <=8, <=16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mixed-policy", default="optimised", choices=["optimised", "fixed"],
                    help="block_mixed only: ds2i_hybrid optimiser (default) or the fixed per-block size policy")
    ap.add_argument("--workload", default=os.environ.get("DS2I_BENCH_WORKLOAD", "auto"), choices=["auto", "c2", "gov2", "cw09"])
    ap.add_argument("--op", default="ranked_and")
    ap.add_argument("--codec", default="block_optpfor")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-json", default=None, help="rocprofv3 --pmc derived HBM bytes per launch (profiles/*.json)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ds2i_amd as d

    from ds2i_amd import sharding as sh
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the query path has no CPU fallback)")
    # "nccl" is RCCL on ROCm; dist is None when N == 1. DS2I_BENCH_BACKEND / DS2I_BENCH_ONE_DEVICE exist only to
    # exercise the multi-rank code path on a single-GPU box (tests): gloo rendezvous, every rank on cuda:0.
    rank, local_rank, world, dist = sh.init_distributed(os.environ.get("DS2I_BENCH_BACKEND", "nccl"))
    if os.environ.get("DS2I_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    d.lib()  # fail loudly if the HIP extension is missing

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- workload
    wl = args.workload
    threads = max(1, (os.cpu_count() or 8))
    if wl == "auto":
        # probe the host's index-build rate; GOV2-scale (~1 B postings) must build in ~2 minutes
        if rank == 0:
            pp = d.SynthParams(seed=1, num_docs=2_000_000, num_terms=2048, zipf_exp=0.6, top_df_frac=0.25, min_len=4096,
                               clustered_every=4)
            t0 = time.time()
            _, _, n = d.synth_build(pp, args.codec, threads)
            rate = n / (time.time() - t0)
            wl = "gov2" if 1.0e9 / rate < 150 else "c2"
            log("index build rate %.1f Mpostings/s on %d threads -> workload %s" % (rate / 1e6, threads, wl))
        wl = "gov2" if sh.broadcast_int(dist, 1 if wl == "gov2" else 0) else "c2"
    W = WORKLOADS[wl]
    p = d.SynthParams(seed=W["seed"], num_docs=W["num_docs"], num_terms=W["num_terms"], zipf_exp=W["zipf_exp"],
                      top_df_frac=W["top_df_frac"], min_len=W["min_len"], clustered_every=W["clustered_every"])
    import tempfile
    need = 4 << 30  # rank 0 hands the index image (~3 GB at GOV2 scale) to the other ranks through a file
    shm = tempfile.gettempdir()
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(cand)
            if st.f_bavail * st.f_frsize > need:
                shm = cand
                break
        except OSError:
            pass
    tag = "ds2i_bench_%s_%s_%d" % (wl, args.codec, os.getppid() if world > 1 else os.getpid())
    f_idx, f_wand = os.path.join(shm, tag + ".idx"), os.path.join(shm, tag + ".wand")
    postings = 0
    img = wand = None
    if rank == 0:
        t0 = time.time()
        if args.codec == "block_mixed" and args.mixed_policy == "optimised":
            # block_mixed images come out of the space/time optimiser (the reference builds them with
            # optimal_hybrid_index.cpp): MI355X decode-time model, uniform access, budget halfway between the
            # smallest and the fastest index
            img, wand, postings, tc = d.synth_build_hybrid(p, budget_frac=0.5, threads=threads)
            log("block_mixed optimiser: full blocks by type (pfor, varint, interpolative) docs %s freqs %s" % (tc["docs"], tc["freqs"]))
        else:
            img, wand, postings = d.synth_build(p, args.codec, threads)
        log("built %s index: %d postings, %.1f MB, %.1fs" % (wl, postings, len(img) / 1e6, time.time() - t0))
    img = sh.share_bytes(dist, rank, img, f_idx)
    wand = sh.share_bytes(dist, rank, wand, f_wand)
    # each rank: full index replica in its GPU's HBM, its own 4096-query batch (weak scaling, no collective)
    queries = d.synth_queries(0x51E21 + rank, p.num_terms, args.batch)
    idx = d.Index(args.codec, img, wand, device=local_rank)
    batch = d.Batch(idx, args.op, queries, k=10)

    # ---------------------------------------------------------------- timed region
    # The timed steps run the uninstrumented kernels (statistics are a compile-time option, like the reference's
    # block_profiler); one extra untimed, instrumented step afterwards collects the block / byte counters.
    batch.set_instrumented(False)
    for _ in range(args.warmup):
        batch.run()
    barrier()
    t0 = time.perf_counter()
    kern_ms = [0.0] * NCLS
    for _ in range(args.steps):
        st = batch.run()
        for c in range(NCLS):
            kern_ms[c] += batch.class_stats(c)[0].kernel_ms
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sh.max_over_ranks(dist, elapsed)
    count, topk, tlen, _ = batch.fetch()
    batch.set_instrumented(True)
    batch.run()
    count_i, topk_i, tlen_i, _ = batch.fetch()
    assert np.array_equal(count, count_i) and np.array_equal(tlen, tlen_i), "instrumented / uninstrumented kernels disagree"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_q = args.batch * world * args.steps
    qps = total_q / elapsed
    cls_stats = [batch.class_stats(c) for c in range(NCLS)]
    dom = max(range(NCLS), key=lambda c: cls_stats[c][0].algorithmic_bytes)
    for c in range(NCLS):
        log("class %d: %d queries, kernel %.3f ms/step, %s" % (c, cls_stats[c][1], kern_ms[c] / args.steps, cls_stats[c][0].as_dict()))
        pc = batch.phase_cycles(c)
        if pc["total"]:
            log("   phase cycles (diagnostic build): " + ", ".join("%s %.1f%%" % (k, 100.0 * v / pc["total"]) for k, v in pc.items()))
    dom_ms = kern_ms[dom] / args.steps
    out = {
        "metric": "queries/sec (%s, %s)" % (args.op, args.codec), "value": qps, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "mean_us_per_query": 1e6 * elapsed / (args.batch * args.steps),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32+f32", "data": "synthetic",
        "config": {"workload": "%s, %s, %s, batch=%d" % (W["label"], args.codec, args.op, args.batch), "num_docs": W["num_docs"], "postings": int(postings), "index_bytes": len(img),
                   "batch_per_gpu": args.batch, "k": 10, "parallelism": "query-batch sharding x%d, index replicated" % world},
    }

    # ---------------------------------------------------------------- cpu baseline + algorithmic bytes (oracle; rank 0 only)
    a_skip_dom = None
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import oracle as o
        try:
            import tempfile
            opath = o.build(native=True, out=os.path.join(tempfile.gettempdir(), tag + "_oracle.so"))  # -O3 -march=native here
        except Exception as e:  # no compiler on the box: use the prebuilt generic library
            log("native oracle build failed (%s); using prebuilt liboracle.so" % e)
            opath = None
        oidx = o.Index(args.codec, img, wand, libpath=opath)
        nterms = [len(set(q)) for q in queries]
        cls_of = lambda n: 0 if n <= 2 else 1 if n <= 4 else 2 if n <= 8 else 3
        cls_q = [[q for q, n in zip(queries, nterms) if cls_of(n) == c] for c in range(NCLS)]
        t0 = time.time()
        prof = [oidx.query_batch(args.op, cq, k=10, profile=True)[4] if cq else None for cq in cls_q]
        log("oracle profile pass (reference traversal A_skip): %.1fs" % (time.time() - t0))
        a_skip_dom = prof[dom]["algorithmic_bytes"] if prof[dom] else 0
        out["a_skip_bytes_per_step"] = sum(pr["algorithmic_bytes"] for pr in prof if pr)
        # parity spot check in the same run (count + top-k within 1e-5) on the sample below
        probe = queries[:64]
        t0 = time.time()
        oidx.query_batch(args.op, probe, k=10)
        per_q = (time.time() - t0) / len(probe)
        nsample = int(max(64, min(len(queries), 15.0 / (3 * per_q))))
        sample = queries[:nsample]
        oc, otopk, otlen, _, _ = oidx.query_batch(args.op, sample, k=10)
        assert np.array_equal(count[:nsample], oc), "GPU/oracle count mismatch"
        fin = np.isfinite(otopk)
        np.testing.assert_allclose(topk[:nsample][fin], otopk[fin], rtol=1e-5)
        pt = oidx.perftest(args.op, sample, k=10, runs=2)
        cpu = {"value": 1e6 / pt["avg"], "unit": "queries/s", "cores": 1, "kind": "port",
               "sample": "first %d of the %d-query batch, op_perftest: 1 untimed + 2 timed passes, %.1fs timed; "
                         "mean %.1f us q50 %.1f q90 %.1f q95 %.1f" % (nsample, len(queries), pt["seconds"], pt["avg"],
                                                                   pt["q50"], pt["q90"], pt["q95"]),
               "host_cpus": os.cpu_count()}
        out["speedup_vs_cpu_1core"] = qps / cpu["value"]
        # N-thread replay in the style of profile_queries.cpp:21-39: one functor copy per thread, index shared read-only,
        # every thread runs op_perftest (1 untimed + 1 timed pass) over its own slice of the batch
        import threading
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:  # a container may grant fewer CPUs than it shows (cgroup v2 cpu.max = "<quota> <period>" or "max <period>")
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
        except Exception:  # noqa: BLE001
            pass
        nthreads = max(1, min(ncores, quota or ncores, 256))
        # bounded: a quarter of the single-thread sample per thread keeps this leg to tens of seconds even when the
        # threads share memory bandwidth
        per_thread = max(16, min(nsample // 4, len(queries)))
        slices = [[queries[(t * per_thread + i) % len(queries)] for i in range(per_thread)] for t in range(nthreads)]
        errs = []

        def replay(sl):
            try:
                oidx.perftest(args.op, sl, k=10, runs=1)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        threads = [threading.Thread(target=replay, args=(sl,)) for sl in slices]
        t0 = time.time()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        wall = time.time() - t0
        if not errs and wall > 0:
            mt_qps = 2.0 * nthreads * per_thread / wall
            out["cpu_baseline_threads"] = {"value": mt_qps, "unit": "queries/s", "cores": nthreads, "kind": "port",
                                           "sample": "%d threads x %d queries x 2 passes in %.1fs wall (all passes counted); "
                                                     "%d logical CPUs visible, cgroup quota %s"
                                                     % (nthreads, per_thread, wall, ncores, quota if quota else "none")}
            out["speedup_vs_cpu_all_cores"] = qps / mt_qps
        if opath and os.path.exists(opath):
            os.remove(opath)
    if a_skip_dom is None:  # N>1 or baseline skipped: price the device's own traversal with the same pricing
        a_skip_dom = cls_stats[dom][0].algorithmic_bytes
        src = "device-counted (block-synchronous traversal, same pricing)"
    else:
        src = "oracle-counted reference traversal (A_skip)"
    achieved = a_skip_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None  # PMC counters cannot be collected inside this process: taken from the committed rocprofv3 passes
    valu_issue = None
    tj = args.traffic_json or os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % wl)
    if os.path.exists(tj) and args.op == "ranked_and" and args.codec == "block_optpfor" and dom == 0:
        tjson = json.load(open(tj))
        traffic = tjson.get("hbm_bytes_per_launch")
        # the binding resource of this path is vector-instruction issue, not bandwidth: carry the
        # measured figure of the committed counter pass next to the HBM roofline
        valu_issue = (tjson.get("valu_issue") or {}).get("frac")
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu_issue_frac": valu_issue,
                       "kernel": "%s<%s,TMAX=%d>" % ("k_conjunctive" if args.op in ("and", "and_freq", "ranked_and") else "k_daat",
                                                     args.op, (2, 4, 8, 16)[dom]),
                       "kernel_ms": dom_ms, "algorithmic_bytes": int(a_skip_dom), "bytes_source": src,
                       "device_counted_bytes": int(cls_stats[dom][0].algorithmic_bytes),
                       "queries_in_kernel": cls_stats[dom][1]}
    out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
