"""N>1 path on CPU: world_size-2 gloo processes shard one query batch into contiguous slices, every rank loads the
shared index image, answers its slice, and the gathered results equal the unsharded answer. The per-rank query
engine here is the CPU oracle (there is no GPU in this container); bench.py runs the same sharding code with the
HIP path and the RCCL backend."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ds2i_amd as d
    import oracle as o
    from ds2i_amd import sharding as sh
    from helpers import Collection, queries_for, small_params
    rank_, local_rank, world_, dist = sh.init_distributed("gloo")
    assert (rank_, world_) == (rank, world) and dist is not None
    coll = Collection(small_params(num_docs=8000, num_terms=120))
    queries = queries_for(coll, 101)
    img = wand = None
    if rank == 0:
        img, wand = coll.index_image("block_optpfor"), coll.wand_image()
    img = sh.share_bytes(dist, rank, img, os.path.join(tmpdir, "idx"))
    wand = sh.share_bytes(dist, rank, wand, os.path.join(tmpdir, "wand"))
    assert sh.broadcast_int(dist, 42 if rank == 0 else 7) == 42
    b, e = sh.query_slice(len(queries), rank, world)
    idx = o.Index("block_optpfor", img, wand)
    count, topk, tlen, _, _ = idx.query_batch("ranked_and", queries[b:e])
    all_count = sh.gather_concat(dist, rank, world, count)
    all_topk = sh.gather_concat(dist, rank, world, topk)
    slowest = sh.max_over_ranks(dist, 1.0 + rank)
    if rank == 0:
        full_count, full_topk, _, _, _ = idx.query_batch("ranked_and", queries)
        ret["ok"] = bool(np.array_equal(all_count, full_count) and np.array_equal(all_topk, full_topk)
                         and slowest == float(world) and len(all_count) == len(queries))
    dist.barrier()
    dist.destroy_process_group()


def test_query_slices_cover_batch():
    from ds2i_amd.sharding import query_slice
    for nq in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            cuts = [query_slice(nq, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == nq
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharding(built_lib, tmp_path):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 400)
    mp.spawn(_worker, args=(2, port, str(tmp_path), ret), nprocs=2, join=True)
    assert ret.get("ok") is True
