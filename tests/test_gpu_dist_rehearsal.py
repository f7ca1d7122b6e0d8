"""Rehearsal of the multi-rank search on ONE GPU: 2 processes (gloo rendezvous on 127.0.0.1) each build
their shard of the same index on cuda:0 with the GPU k-means, search the whole batch, exchange the
candidate lists (zvec_amd.dist.ShardedIVF, the code bench.py runs under RCCL) and merge on the GPU.
Every rank must end with the unsharded answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zvec_amd
    from zvec_amd.dist import ShardedIVF
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if rank == 0:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))   # rank 0: explicit stream; rank 1: legacy default stream
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    n, dim, nlist, nq, k, nprobe = 40000, 64, 64, 200, 10, 8
    proj = torch.randn((8, dim), generator=g, device=dev)
    base = torch.randn((n, 8), generator=g, device=dev) @ proj
    q = (torch.randn((nq, 8), generator=g, device=dev) @ proj).contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    # reference answer: unsharded index in this process
    full = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean", device=0)
    assert full.build_dev(base.data_ptr(), n, nlist, kmeans_iters=5, sample_per_list=64, seed=9, stream=stream) == 0
    fctx = full.create_context()
    fctx.set_stream(stream)
    one = ShardedIVF(full, fctx, 0, 1)
    fk, fs, fc = [t.clone() for t in one.search(q, k, nprobe, n, stream)]
    # this rank's shard + exchange
    sh = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean", device=0)
    assert sh.set_shard(rank, world) == 0
    assert sh.build_dev(base.data_ptr(), n, nlist, kmeans_iters=5, sample_per_list=64, seed=9, stream=stream) == 0
    ctx = sh.create_context()
    ctx.set_stream(stream)
    sharded = ShardedIVF(sh, ctx, rank, world)
    mk, ms, mc = sharded.search(q, k, nprobe, n, stream)
    torch.cuda.synchronize()
    ok = bool(torch.equal(mc, fc)) and bool(torch.equal(ms, fs))
    same_keys = (mk == fk)
    tied = torch.zeros_like(same_keys)
    tied[:, 1:] |= fs[:, 1:] == fs[:, :-1]
    tied[:, :-1] |= fs[:, 1:] == fs[:, :-1]
    ok = ok and bool((same_keys | tied).all())
    if not ok:
        bad = (ms != fs).nonzero()
        print("rank", rank, "counts equal", bool(torch.equal(mc, fc)), "score mismatches", bad.shape[0],
              "first", bad[:3].tolist(), ms[bad[0, 0]].tolist() if bad.shape[0] else None,
              fs[bad[0, 0]].tolist() if bad.shape[0] else None, "key mismatches", int((~(same_keys | tied)).sum()), flush=True)
    # the coarse pass DEALT over the ranks (default off): rank r scores its slice of the batch, one all-gather of the probe lists,
    # every rank plans from them — must give the very same lists
    dealt = ShardedIVF(sh, ctx, rank, world, deal_coarse=True)
    dk, ds, dc = dealt.search(q, k, nprobe, n, stream)
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(dc, mc)) and bool(torch.equal(ds, ms)) and bool(torch.equal(dk, mk))
    # every rank's shard with shadow lists (zvec_hip_ivf_set_shadow): pre-selected on fp16 twins of the local lists, re-scored in fp32,
    # certified (uncertified queries re-run on the fp32 lists) BEFORE the exchange — the merged lists must be the very same, with the
    # replicated and with the dealt coarse pass
    sh.set_shadow(True)
    for variant in (sharded, dealt):
        hk, hs, hc = variant.search(q, k, nprobe, n, stream)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(hc, mc)) and bool(torch.equal(hs, ms)) and bool(((hk == mk) | tied).all())
    sh.set_shadow(False)
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {r: True for r in range(world)}
