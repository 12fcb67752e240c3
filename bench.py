#!/usr/bin/env python
"""bench.py -- queries/sec of ranked_and over a synthetic block_optpfor index on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU. A "step" = one pass of the hot path (ranked_and, k=10)
over one FRESH 4096-query batch per GPU, end to end through the public pipelined ABI (SURVEY.md §8(d)
"Timing": host-side query normalisation and planning + H2D of the terms + kernels + D2H of the results;
only the index is resident in HBM). K distinct batches are timed, `--depth` of them in flight.
Rank 0 prints ONE JSON line.

  value        whole-job queries/s = (queries all ranks processed) / max-over-ranks wall time
  kernel_resident_qps  the same kernels over ONE prepared batch re-run K times (round 1's figure), beside it
  roofline     the class kernel with the most GPU time: algorithmic bytes (the reference traversal's A_skip,
               SURVEY.md §8(d), counted by the instrumented oracle) / that kernel's mean hipEvent duration
               over the launches of the timed region, against the 8 TB/s HBM peak
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
    # robustness workload (no BASELINE config): the GOV2-scale collection with CORRELATED terms -- 256 topics, a term 64 times
    # as likely in the documents of its home topic, 25 % of the multi-term queries drawn from one topic. The independent
    # lists of "gov2" are the easy case for range-table pruning; this says how much of the rate hangs on that
    "gov2c": dict(num_docs=25_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4,
                  topics=256, topic_boost=64, same_topic_pct=25,
                  seed=0xD5210004, label="synthetic GOV2-scale 25M-doc Zipf, correlated terms (256 topics x64, 25% same-topic queries)"),
    "cw09": dict(num_docs=50_000_000, num_terms=32768, zipf_exp=0.6, top_df_frac=0.25, min_len=4096, clustered_every=4,
                 seed=0xD5210005, label="synthetic ClueWeb09-B-scale 50M-doc Zipf (configs[4])"),
}
NCLS = 5  # kernel classes by distinct query terms: <=2, <=4, <=8, <=16, more
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
GATHER_PEAK_GBS = 6320.0  # measured: 49.4 G distinct 128-byte lines/s (profiles/probes/r4_gather_rate.sh), the same as a 16-byte-per-lane copy


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mixed-policy", default="optimised", choices=["optimised", "fixed"],
                    help="block_mixed only: ds2i_hybrid optimiser (default) or the fixed per-block size policy")
    ap.add_argument("--workload", default=os.environ.get("DS2I_BENCH_WORKLOAD", "auto"), choices=["auto", "c2", "gov2", "gov2c", "cw09"])
    ap.add_argument("--op", default="ranked_and")
    ap.add_argument("--codec", default="block_optpfor")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--k", type=int, default=10, help="results kept per query (the metric is quoted at 10; > 64 runs the big-heap kernels, queries.hpp:152-197)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle", action="store_true", help="skip every oracle leg (A_skip profile, parity sample, cpu baseline)")
    ap.add_argument("--depth", type=int, default=3, help="batches in flight in the pipelined timed region")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = every rank its own 4096-query batches; strong = one stream of batches cut into N slices")
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
                      top_df_frac=W["top_df_frac"], min_len=W["min_len"], clustered_every=W["clustered_every"],
                      topics=W.get("topics", 0), topic_boost=W.get("topic_boost", 0))
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
    # each rank: full index replica in its GPU's HBM; no data-path collective.
    #   weak (default): every rank answers its own stream of 4096-query batches
    #   strong: ONE stream of 4096-query batches for the whole job; rank r answers slice r of every batch
    #           (sharding.query_slice) and rank 0 gathers the results (sharding.gather_concat)
    nbatches = args.steps + args.warmup
    strong = args.scaling == "strong"
    qseed = lambda i: 0x51E21 + (0 if strong else 1000003 * rank) + 7919 * i
    if W.get("topics"):
        all_queries = [d.synth_queries_topical(p, qseed(i), args.batch, W["same_topic_pct"]) for i in range(nbatches)]
    else:
        all_queries = [d.synth_queries(qseed(i), p.num_terms, args.batch) for i in range(nbatches)]
    if strong:
        lo_, hi_ = sh.query_slice(args.batch, rank, world)
        my_queries = [q[lo_:hi_] for q in all_queries]
    else:
        my_queries = all_queries
    flat = [d.flatten_queries(q) for q in my_queries]  # the query log is read before the clock starts (queries.cpp:74-88)
    queries = all_queries[args.warmup]                  # first timed batch: oracle profile / cpu baseline / parity sample
    t0 = time.time()
    idx = d.Index(args.codec, img, wand, device=local_rank)
    log("index upload + upload-time tables (list offsets, skip table, block-max weights): %.2fs, %.1f MB in HBM"
        % (time.time() - t0, idx.device_bytes() / 1e6))
    pipe = d.Pipeline(idx, depth=args.depth)

    # ---------------------------------------------------------------- timed region (SURVEY.md §8(d) "Timing")
    # Every step is a FRESH batch through the public pipelined ABI: host-side query normalisation + BM25 query weights
    # + work-unit planning, H2D of the plan, kernels, merge, D2H of the results. Nothing is pre-staged. `depth` batches
    # are in flight: the host plans batch i+1 while batch i runs. The kernels are the uninstrumented instantiations
    # (statistics are a compile-time option, like the reference's block_profiler).
    def run_stream(first, n, collect):
        tickets, results, cls_ms = [], [], [[0.0, 0] for _ in range(NCLS)]
        def reap():
            r = pipe.wait(tickets.pop(0))
            done_at.append(time.perf_counter())
            if collect:
                results.append(r)
                for c in range(NCLS):
                    st, nqc = pipe.class_stats(c)
                    if nqc:
                        cls_ms[c][0] += st.kernel_ms
                        cls_ms[c][1] += 1
                        for g in pipe.class_groups(c):  # a class may run several kernels (one per exact list count): time each
                            acc = grp_ms.setdefault((c, g["lists"], g["pipelined_stream"]), [0.0, 0])
                            acc[0] += g["kernel_ms"]
                            acc[1] += 1
        for i in range(first, first + n):
            if len(tickets) == args.depth:
                reap()
            ts = time.perf_counter()
            tickets.append(pipe.submit(args.op, flat[i], k=args.k))
            host_s[0] += time.perf_counter() - ts
        while tickets:
            reap()
        return results, cls_ms

    done_at = []
    grp_ms = {}
    host_s = [0.0]  # seconds the caller's thread spent inside ds2i_hip_pipeline_submit (normalisation + planning + H2D + launches)
    _, warm_ms = run_stream(0, args.warmup, True)
    warm_grp_ms = grp_ms
    barrier()
    t0 = time.perf_counter()
    done_at = []
    grp_ms = {}
    host_s[0] = 0.0
    results, cls_ms = run_stream(args.warmup, args.steps, True)
    host_submit_ms = 1e3 * host_s[0] / max(1, args.steps)
    timed_grp_ms = grp_ms
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sh.max_over_ranks(dist, elapsed)
    count, topk, tlen = results[0]
    # spread of the step time: intervals between consecutive batch completions in the pipelined region (the first `depth`
    # completions include the ramp-up of the pipeline and are left out when there are enough steps)
    gaps = np.diff(np.array([t0] + done_at))
    steady = gaps[args.depth:] if len(gaps) > 2 * args.depth else gaps
    step_spread = {"min": 1e3 * float(steady.min()), "median": 1e3 * float(np.median(steady)), "max": 1e3 * float(steady.max()),
                   "n": int(len(steady))}
    # 95 % interval of the mean step time from the same intervals (normal approximation; the batches are distinct, so the spread is
    # the workload's as much as the machine's), carried over to `value` as a relative +-
    rel_ci = float(1.96 * steady.std(ddof=1) / np.sqrt(len(steady)) / steady.mean()) if len(steady) > 2 else None

    # ---- untimed extras: kernel-resident rate (one prepared batch re-run, round-1's figure) and the instrumented pass
    batch = d.Batch(idx, args.op, my_queries[args.warmup], k=args.k)
    batch.set_instrumented(False)
    batch.run()
    first_res_ms = [batch.class_stats(c)[0].kernel_ms for c in range(NCLS)]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res_ms = [0.0] * NCLS
    res_grp_ms = {}
    for _ in range(args.steps):
        batch.run()
        for c in range(NCLS):
            res_ms[c] += batch.class_stats(c)[0].kernel_ms
            for g in batch.class_groups(c):
                acc = res_grp_ms.setdefault((c, g["lists"], g["pipelined_stream"]), [0.0, 0])
                acc[0] += g["kernel_ms"]
                acc[1] += 1
    torch.cuda.synchronize()
    resident_s = (time.perf_counter() - t1) / args.steps
    count_r, topk_r, tlen_r, _ = batch.fetch()
    conj = args.op in ("and", "and_freq", "ranked_and")
    if args.op in ("and", "and_freq", "or", "or_freq"):  # no top-k: counts only
        topk = topk_r = np.zeros((len(count), 1), dtype=np.float32)
    fin = np.isfinite(topk_r)
    # every operator is bit-stable from run to run (the disjunctive kernel sums term scores in fixed point, so the order
    # its parts meet a document's terms in does not show)
    same = np.array_equal(topk, topk_r)
    assert np.array_equal(count, count_r) and np.array_equal(tlen, tlen_r) and same, "pipelined / prepared-batch results disagree"
    batch.set_instrumented(True)
    batch.run()
    count_i, topk_i, tlen_i, _ = batch.fetch()
    assert np.array_equal(count, count_i) and np.array_equal(tlen, tlen_i), "instrumented / uninstrumented kernels disagree"
    if strong:  # the gathered answer of the first timed batch must be the whole batch, in order
        g_count = sh.gather_concat(dist, rank, world, count)
        g_topk = sh.gather_concat(dist, rank, world, topk)
        assert g_count.shape[0] == args.batch and g_topk.shape[0] == args.batch
        count, topk, tlen = g_count, g_topk, sh.gather_concat(dist, rank, world, tlen)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    import zlib
    # bit-level fingerprint of the first timed batch's answer (rank 0's batch; the gathered whole batch in strong mode):
    # equal across --gpus for the same scaling mode, since every rank answers the same queries it would answer alone
    checksum = zlib.crc32(np.ascontiguousarray(topk).tobytes(), zlib.crc32(np.ascontiguousarray(count).tobytes()))
    per_rank_q = (hi_ - lo_) if strong else args.batch
    total_q = (args.batch if strong else args.batch * world) * args.steps
    qps = total_q / elapsed
    cls_stats = [batch.class_stats(c) for c in range(NCLS)]
    mean_ms = [cls_ms[c][0] / cls_ms[c][1] if cls_ms[c][1] else 0.0 for c in range(NCLS)]
    dom = max(range(NCLS), key=lambda c: cls_ms[c][0])  # provisional: re-chosen below by algorithmic bytes when the oracle ran
    for c in range(NCLS):
        log("class %d: %d queries, kernel %.3f ms/launch in the pipelined region (%.3f ms alone), %s"
            % (c, cls_stats[c][1], mean_ms[c], res_ms[c] / args.steps, cls_stats[c][0].as_dict()))
        pc = batch.phase_cycles(c)
        if pc["total"]:
            log("   phase cycles (diagnostic build): " + ", ".join("%s %.1f%%" % (k, 100.0 * v / pc["total"]) for k, v in pc.items()))
    dom_ms = mean_ms[dom]
    out = {
        "metric": "queries/sec (%s, %s)" % (args.op, args.codec), "value": qps, "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "value_ci95": ({"rel": rel_ci, "low": qps / (1.0 + rel_ci), "high": qps / max(1e-9, 1.0 - rel_ci),
                        "note": "95 %% interval of the mean step time over %d completion intervals of the timed region" % len(steady)} if rel_ci is not None else None),
        "mean_us_per_query": 1e6 / qps, "step_ms_spread": step_spread,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32+f32", "data": "synthetic",
        "timing": "end-to-end over %d distinct batches through ds2i_hip_pipeline_submit/wait (host planning + H2D + kernels + D2H), "
                  "%d in flight" % (args.steps, args.depth),
        "host_submit_ms_per_step": host_submit_ms,  # the caller's thread inside submit: when this approaches ms_per_step the step is host-bound
        "first_batch_checksum": checksum,
        "kernel_resident_qps": per_rank_q * world / resident_s,  # one prepared batch re-run (rank 0's rate x ranks)
        "end_to_end_over_resident": qps / (per_rank_q * world / resident_s),
        "config": {"workload": "%s, %s, %s, batch=%d" % (W["label"], args.codec, args.op, args.batch), "num_docs": W["num_docs"], "postings": int(postings), "index_bytes": len(img),
                   "batch_per_gpu": per_rank_q, "k": args.k, "parallelism": "query-batch sharding x%d (%s), index replicated" % (world, args.scaling)},
    }
    info = idx.info()
    out["config"]["device_bytes"] = int(idx.device_bytes())
    if info["transcoded_from"] >= 0:  # block_mixed / opt / ef / single / uniform: decoded once at upload, queried as block_optpfor + side tables
        out["config"]["upload"] = ("%s image transcoded to block_optpfor at upload (DS2I_MIXED_NATIVE=1 / DS2I_PEF_NATIVE=1 "
                                   "query the image as it is)" % args.codec)

    # ---------------------------------------------------------------- oracle legs (rank 0 only; the only place this file touches oracle/)
    cls_of = lambda n: 0 if n <= 2 else 1 if n <= 4 else 2 if n <= 8 else 3 if n <= 16 else 4
    if args.k > 64 and not any(batch.class_groups(c) for c in range(NCLS - 1)):
        # k > 64 with a query of more than 16 lists (or a natively queried block_mixed image): the whole batch runs the big-heap kernel of
        # the last class (ds2i_hip.h: DS2I_HIP_MAX_K). Otherwise k <= 1024 stays on the stream kernels, 4 or 16 scores per lane.
        cls_of = lambda n: NCLS - 1
    nterms = [len(set(q)) for q in queries]
    a_skip_q = [None] * NCLS  # reference-traversal bytes per query of each class (SURVEY.md §8(d) A_skip)
    a_skip_g = {}             # ... and per launch group (class, lists, pipelined): (bytes of its queries, queries)
    cpu = None
    if not args.no_oracle:
        import oracle as o
        try:
            opath = o.build(native=True, out=os.path.join(tempfile.gettempdir(), tag + "_oracle.so"))  # -O3 -march=native here
        except Exception as e:  # no compiler on the box: use the prebuilt generic library
            log("native oracle build failed (%s); using prebuilt liboracle.so" % e)
            opath = None
        oidx = o.Index(args.codec, img, wand, libpath=opath)
        # (1) algorithmic bytes: the instrumented oracle (Profile=true equivalent) over the first timed batch, per class
        cls_q = [[q for q, n in zip(queries, nterms) if cls_of(n) == c] for c in range(NCLS)]
        t0 = time.time()
        prof = [oidx.query_batch(args.op, cq, k=args.k, profile=True)[4] if cq else None for cq in cls_q]
        log("oracle profile pass (reference traversal A_skip): %.1fs" % (time.time() - t0))
        a_skip_q = [prof[c]["algorithmic_bytes"] / len(cls_q[c]) if prof[c] else None for c in range(NCLS)]
        out["a_skip_bytes_per_step"] = sum(pr["algorithmic_bytes"] for pr in prof if pr)
        # a class may run several kernels (ranked_and: k_ranked_stream<n> per exact list count n + the class kernel for
        # the rest): the reference traversal's bytes of exactly the queries each of them answers
        for c in range(NCLS):
            gs = batch.class_groups(c)
            exact = set(g["lists"] for g in gs if g["pipelined_stream"])
            for g in gs:
                if g["pipelined_stream"]:
                    gq = [q for q in cls_q[c] if len(set(q)) == g["lists"]]
                else:
                    gq = [q for q in cls_q[c] if len(set(q)) not in exact]
                if len(gs) == 1:
                    a_skip_g[(c, g["lists"], g["pipelined_stream"])] = (prof[c]["algorithmic_bytes"] if prof[c] else None, len(cls_q[c]))
                else:
                    a_skip_g[(c, g["lists"], g["pipelined_stream"])] = (oidx.query_batch(args.op, gq, k=args.k, profile=True)[4]["algorithmic_bytes"] if gq else 0, len(gq))
        # (2) parity spot check in the same run (count + top-k within 1e-5) on the sample the CPU baseline is timed on
        probe = queries[:64]
        t0 = time.time()
        oidx.query_batch(args.op, probe, k=args.k)
        per_q = (time.time() - t0) / len(probe)
        nsample = int(max(64, min(len(queries), 15.0 / (3 * per_q))))
        if not strong:
            nsample = min(nsample, len(count))
        sample = queries[:nsample]
        oc, otopk, otlen, _, _ = oidx.query_batch(args.op, sample, k=args.k)
        assert np.array_equal(count[:nsample], oc), "GPU/oracle count mismatch"
        if args.op in ("ranked_and", "wand", "maxscore", "ranked_or"):
            fin = np.isfinite(otopk)
            np.testing.assert_allclose(topk[:nsample][fin], otopk[fin], rtol=1e-5)
        if not args.no_cpu_baseline and world == 1:
            # (3) cpu_baseline: the oracle driven like op_perftest (queries.cpp:13-62) on ONE core
            pt = oidx.perftest(args.op, sample, k=args.k, runs=2)
            cpu_model = "unknown"
            try:
                cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:  # noqa: BLE001
                pass
            cpu = {"value": 1e6 / pt["avg"], "unit": "queries/s", "cores": 1, "kind": "port",
                   "sample": "first %d of the %d-query batch, op_perftest: 1 untimed + 2 timed passes, %.1fs timed; "
                             "mean %.1f us q50 %.1f q90 %.1f q95 %.1f" % (nsample, len(queries), pt["seconds"], pt["avg"],
                                                                       pt["q50"], pt["q90"], pt["q95"]),
                   "host_cpus": os.cpu_count(), "cpu_model": cpu_model}
            out["speedup_vs_cpu_1core"] = qps / cpu["value"]
            # N-thread replay in the style of profile_queries.cpp:21-39: one functor copy per thread, index shared
            # read-only, every thread runs op_perftest (1 untimed + 1 timed pass) over its own slice of the batch.
            # N = the CPUs this process may actually use (affinity mask, cgroup quota), i.e. physical cores when the
            # container is granted them.
            import threading
            ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            quota = None
            try:  # cgroup v2 cpu.max = "<quota> <period>" or "max <period>"
                qv, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if qv != "max":
                    quota = max(1, int(int(qv) / int(per)))
            except Exception:  # noqa: BLE001
                pass
            nthreads = max(1, min(ncores, quota or ncores, 256))
            per_thread = max(16, min(nsample // 4, len(queries)))
            slices = [[queries[(t * per_thread + i) % len(queries)] for i in range(per_thread)] for t in range(nthreads)]
            errs = []

            def replay(sl):
                try:
                    oidx.perftest(args.op, sl, k=args.k, runs=1)
                except Exception as e:  # noqa: BLE001
                    errs.append(e)

            ths = [threading.Thread(target=replay, args=(sl,)) for sl in slices]
            t0 = time.time()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            wall = time.time() - t0
            if not errs and wall > 0:
                mt_qps = 2.0 * nthreads * per_thread / wall
                out["cpu_baseline_threads"] = {"value": mt_qps, "unit": "queries/s", "cores": nthreads, "kind": "port",
                                               "cpu_model": cpu_model,
                                               "sample": "%d threads x %d queries x 2 passes in %.1fs wall (all passes counted); "
                                                         "%d logical CPUs visible, cgroup quota %s"
                                                         % (nthreads, per_thread, wall, ncores, quota if quota else "none")}
                out["speedup_vs_cpu_all_cores"] = qps / mt_qps
        if opath and os.path.exists(opath):
            os.remove(opath)

    # ---------------------------------------------------------------- roofline of the dominant kernel
    # achieved = (A_skip per query of the class, from the oracle) x (queries of that class in one launch) / (mean
    # hipEvent duration of the class kernel over the launches of the timed region). The kernel's own traversal touches
    # fewer bytes than that (block-max pruning skips what cannot enter the heap): device_counted_bytes.
    kname = {"and": "k_conjunctive<false,false", "and_freq": "k_conjunctive<false,true", "ranked_and": "k_conjunctive<true,true"}
    def kernel_name(c):
        if c == 4:
            return "k_daat_long<%s>" % args.op
        if args.op in kname:
            return "%s,TMAX=%d>" % (kname[args.op], (2, 4, 8, 16)[c])
        if args.op in ("or", "or_freq"):
            stream = True
            return ("k_union<%s> (<=%d lists)" if stream else "k_disjunctive<TMAX=%d> (%s)") % (
                ("true" if args.op == "or_freq" else "false", (2, 4, 8, 16)[c]) if stream else ((2, 4, 8, 16)[c], args.op))
        stream = not any(os.environ.get(e) for e in ("DS2I_NO_BMW", "DS2I_NO_RMW"))
        if stream and not os.environ.get("DS2I_NO_UNION_RSTREAM") and not os.environ.get("DS2I_NO_XSLOTS"):  # (every index kind is queried as block_optpfor + side tables by default)
            return "k_union_stream, class of <=%d lists (%s)" % ((2, 4, 8, 16)[c], args.op)
        return ("k_union_topk<TMAX=%d> (%s)" if stream else "k_disjunctive<TMAX=%d> (%s)") % ((2, 4, 8, 16)[c], args.op)
    per_class = []
    for c in range(NCLS):
        nqc = cls_stats[c][1]
        if not nqc or not mean_ms[c]:
            continue
        bytes_c = a_skip_q[c] * nqc if a_skip_q[c] is not None else None
        per_class.append({"kernel": kernel_name(c), "queries": nqc, "ms_per_launch": mean_ms[c], "ms_alone": res_ms[c] / args.steps,
                          "algorithmic_bytes": int(bytes_c) if bytes_c is not None else None,
                          "achieved_gbs": (bytes_c / (mean_ms[c] * 1e-3) / 1e9) if bytes_c is not None else None,
                          "device_counted_bytes": int(cls_stats[c][0].algorithmic_bytes),
                          "postings_scored": int(cls_stats[c][0].postings_scored),
                          "docs_blocks_decoded": int(cls_stats[c][0].docs_blocks_decoded),
                          "freqs_blocks_decoded": int(cls_stats[c][0].freqs_blocks_decoded)})
    # ... and per KERNEL: a class stream runs its launch groups back to back, each timed with its own pair of hipEvents on
    # that stream (ds2i_hip_pipeline_class_groups); rocprofv3 --kernel-trace --stats reports the same kernels by name
    def group_kernel_name(c, lists, pipelined):
        # `lists` of a pipelined group = the list CAPACITY of its instantiation (2 | 4 | 6 | 8 (| 16): it holds the queries of cap - 1 and cap lists)
        if not pipelined:
            return kernel_name(c)
        if args.op in ("wand", "maxscore", "ranked_or"):
            return "k_union_stream<%d>" % lists
        return "k_ranked_stream<%d%s>" % (lists, ",AND" if args.op == "and" else "")
    per_kernel = []
    for key, (tot, n) in sorted(timed_grp_ms.items()):
        c, lists, pipelined = key
        if not n:
            continue
        ms = tot / n
        ab, nqg = a_skip_g.get(key, (None, None))
        ra = res_grp_ms.get(key, [0.0, 0])
        wa = warm_grp_ms.get(key, [0.0, 0])
        n_all = wa[1] + n + ra[1]
        per_kernel.append({"kernel": group_kernel_name(c, lists, pipelined), "class": c, "queries": nqg, "ms_per_launch": ms,
                           "ms_alone": ra[0] / ra[1] if ra[1] else None,
                           "ms_all_launches": (wa[0] + tot + ra[0]) / n_all if n_all else None, "launches_all": n_all, "launches_timed": n,
                           "algorithmic_bytes": int(ab) if ab is not None else None,
                           "achieved_gbs": (ab / (ms * 1e-3) / 1e9) if ab is not None and ms > 0 else None})
    # (a class row names the kernels its stream actually ran -- the launch groups above -- not the class-kernel family)
    cls_of_row = [c for c in range(NCLS) if cls_stats[c][1] and mean_ms[c]]
    for row, c in zip(per_class, cls_of_row):
        names = [k["kernel"] for k in per_kernel if k["class"] == c]
        if names:
            row["kernel"] = " + ".join(names)
    # dominant kernel = the kernel that moves the most algorithmic bytes per launch. (Launch durations are not a good
    # criterion here: the class kernels of a batch run concurrently and the small many-list classes are stretched to the
    # length of the step by the big ones.)
    # dominant kernel = the one that takes the most device time per batch among the kernels that carry a real share (>= 10 %) of
    # the batch's algorithmic bytes. (Bytes alone would pick the one-term kernel on some workloads: it owns a third of the
    # reference traversal's bytes and prunes nearly all of them in half a millisecond; time alone would pick a many-list
    # class whose few units are stretched by sharing the GPU.)
    dom_k = None
    if any(k["algorithmic_bytes"] for k in per_kernel):
        tot_b = sum(k["algorithmic_bytes"] or 0 for k in per_kernel)
        cand = [k for k in per_kernel if (k["algorithmic_bytes"] or 0) >= 0.1 * tot_b] or per_kernel
        dom_k = max(cand, key=lambda k: k["ms_per_launch"])
        dom = dom_k["class"]
        dom_ms = dom_k["ms_per_launch"]
        a_skip_dom = dom_k["algorithmic_bytes"]
        src = "oracle-counted reference traversal (A_skip) of the queries this kernel answers, per launch"
    elif any(x is not None for x in a_skip_q):
        dom = max(range(NCLS), key=lambda c: (a_skip_q[c] or 0) * cls_stats[c][1] if mean_ms[c] else -1)
        dom_ms = mean_ms[dom]
        a_skip_dom = a_skip_q[dom] * cls_stats[dom][1]
        src = "oracle-counted reference traversal (A_skip) per query x queries in the launch"
    else:  # oracle skipped: price the device's own (pruned) traversal with the same pricing
        a_skip_dom = cls_stats[dom][0].algorithmic_bytes
        src = "device-counted (block-synchronous pruned traversal, same pricing)"
    achieved = a_skip_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM traffic from PMC counters cannot be collected inside this process (rocprofv3 wraps the command): when a
    # counter pass of the same command has been committed under profiles/, its per-launch figure is carried along and
    # labelled as such; otherwise null.
    traffic = None
    traffic_src = None
    traffic_step = None
    tjson = {}
    tj = args.traffic_json
    for rnd in ("r06", "r05"):  # the newest committed counter pass of this workload / operator
        if not tj and os.path.exists(os.path.join(ROOT, "profiles", "%s_traffic_%s_%s.json" % (rnd, wl, args.op))):
            tj = os.path.join(ROOT, "profiles", "%s_traffic_%s_%s.json" % (rnd, wl, args.op))
    tj = tj or ""
    if os.path.exists(tj) and args.codec == "block_optpfor":
        tjson = json.load(open(tj))
        name = dom_k["kernel"] if dom_k else kernel_name(dom)
        traffic = tjson.get("hbm_bytes_per_launch", {}).get(name)
        traffic_step = tjson.get("hbm_bytes_per_step")
        traffic_src = "NOT measured in this run: rocprofv3 --pmc FETCH_SIZE pass of the same command, committed as " + os.path.relpath(tj, ROOT)
    # mean over EVERY launch of that kernel in this process (warm-up + timed + the prepared-batch re-runs): the figure a
    # `rocprofv3 --kernel-trace --stats` of this command reports as the kernel's average duration
    if dom_k:
        n_all, ms_all, ms_alone = dom_k["launches_all"], dom_k["ms_all_launches"], dom_k["ms_alone"]
        n_timed, nq_dom, kname_dom = dom_k["launches_timed"], dom_k["queries"], dom_k["kernel"]
    else:
        n_all = warm_ms[dom][1] + cls_ms[dom][1] + 1 + args.steps
        ms_all = (warm_ms[dom][0] + cls_ms[dom][0] + first_res_ms[dom] + res_ms[dom]) / n_all
        ms_alone, n_timed, nq_dom, kname_dom = res_ms[dom] / args.steps, cls_ms[dom][1], cls_stats[dom][1], kernel_name(dom)
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_per_step": traffic_step, "traffic_source": traffic_src,
                       #  what of that traffic reached HBM: the L2's fabric-side counters cannot tell an Infinity-Cache hit from
                       # a DRAM access (TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ for these kernels), so the same figure is an UPPER bound
                       "hbm_bytes_per_step": traffic_step, "hbm_bytes_source": (tjson.get("l2_fabric_note") if traffic_step else None),
                       "kernel": kname_dom, "kernel_ms": dom_ms, "kernel_ms_alone": ms_alone,
                       "kernel_ms_all_launches": ms_all, "launches_all": n_all,
                       "launches_timed": n_timed, "algorithmic_bytes": int(a_skip_dom), "bytes_source": src,
                       "device_counted_bytes_of_class": int(cls_stats[dom][0].algorithmic_bytes),
                       "queries_in_kernel": nq_dom, "per_kernel": per_kernel, "per_class": per_class}
    # beside the dominant kernel BY TIME (above: the longest launch among the kernels that answer >= 10 % of the step's bytes -- since
    # the launch groups became capacities that is the 3-4-list group) the kernel that answers the MOST bytes, priced the same way
    kb = [k for k in per_kernel if k.get("algorithmic_bytes") and k.get("ms_per_launch")]
    if kb:
        big = max(kb, key=lambda k: k["algorithmic_bytes"])
        gbs = big["algorithmic_bytes"] / (big["ms_per_launch"] * 1e-3) / 1e9
        out["roofline"]["largest_kernel"] = {"kernel": big["kernel"], "queries": big["queries"], "kernel_ms": big["ms_per_launch"],
                                             "algorithmic_bytes": int(big["algorithmic_bytes"]), "achieved": gbs, "frac": gbs / HBM_PEAK_GBS}
    # the class kernels of a batch (and of the neighbouring batches) overlap, so one kernel's launch duration stretches
    # when another class is given more of the GPU; the whole step is the figure that cannot: every class's algorithmic
    # bytes over the wall time of a step
    # what the kernels' OWN (pruned) traversal decodes per step, priced like A_skip (device-counted, the instrumented pass), and --
    # when a counter pass is committed -- how many bytes of 128-byte lines the fabric moved per byte of it
    own = int(sum(cls_stats[c][0].algorithmic_bytes for c in range(NCLS)))
    out["roofline"]["own_traversal_bytes_per_step"] = own
    if info["transcoded_from"] >= 0:
        # A_skip (algorithmic_bytes, achieved, frac, step_frac) prices the reference traversal of the CALLER's image; the kernels that ran
        # decode its block_optpfor re-encoding -- own_traversal_bytes_per_step is what THEY decoded. Neither frac is a utilisation
        # figure of the native (partitioned Elias-Fano / mixed-codec) kernels: DS2I_PEF_NATIVE=1 / DS2I_MIXED_NATIVE=1 run those.
        out["roofline"]["bytes_priced_on"] = "the caller's %s image (reference traversal); the kernels ran on its block_optpfor transcoding" % args.codec
    out["roofline"]["lines_per_useful_byte"] = (traffic_step / own) if (traffic_step and own) else None
    if out.get("a_skip_bytes_per_step"):
        out["roofline"]["step_algorithmic_bytes"] = int(out["a_skip_bytes_per_step"])
        out["roofline"]["step_achieved"] = out["a_skip_bytes_per_step"] / (out["ms_per_step"] * 1e-3) / 1e9
        out["roofline"]["step_frac"] = out["roofline"]["step_achieved"] / HBM_PEAK_GBS
    if traffic_step and out.get("ms_per_step"):
        # the memory side of the same step: the committed profile's bytes per batch over THIS run's step time (per GPU)
        rate = traffic_step / (out["ms_per_step"] * 1e-3) / 1e9
        out["roofline"]["traffic_rate"] = {"gbs": rate, "of_hbm_peak": rate / HBM_PEAK_GBS, "of_measured_gather_peak": rate / GATHER_PEAK_GBS,
                                           "gather_peak_gbs": GATHER_PEAK_GBS,
                                           "note": "FETCH_SIZE bytes per batch from the committed profile / this run's ms_per_step; gather peak = 2^30 scattered "
                                                   "one-byte gathers over 8 GiB on this device (profiles/probes/r4_gather_rate.sh)"}
    out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
