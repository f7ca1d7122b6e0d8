"""N>1 plumbing on CPU: 2 processes over gloo exchange their per-shard candidate lists with the same
pack / all_gather_into_tensor / unpack code the GPU path uses (zvec_amd/dist.py); the merged result is
checked with the oracle's concat-sort-truncate merge (combined_vector_column_indexer.cc:172-232) against
the single-index answer.  The shards are produced by the oracle scanning the lists the byte-balanced list -> shard map gives each
rank (tests/util.py::lpt_owner, the restatement of zvec_hip_ivf_shard_map that zvec_hip_ivf_keep_shard applies on the GPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
    from oracle import oracle as O
    from tests.util import kmeans_lists, lpt_owner
    from zvec_amd.dist import all_gather_candidates, packed_bytes, pack_candidates
    o = O.get()
    rng = np.random.default_rng(41)                      # same data on every rank
    n, dim, nlist, nq, k, nprobe = 3000, 16, 24, 20, 10, 6
    base = rng.integers(0, 60, (n, dim)).astype(np.float32)
    q = rng.integers(0, 60, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    vecs, keys = base[order], order.astype(np.uint64)
    # full answer (every rank computes it for the check)
    fk, fs, _, fc, _, probes = o.ivf_search(cent, offs, vecs, q, k, nprobe, n, keys=keys, want_probes=True)
    # this rank's shard: keep only the lists it owns (others become empty), same centroids
    sizes = np.diff(offs.astype(np.int64))
    owner = lpt_owner(sizes, world)
    keep = np.concatenate([np.arange(offs[l], offs[l + 1]) for l in range(nlist) if owner[l] == rank]).astype(np.int64)
    soffs = np.concatenate([[0], np.cumsum([sizes[l] if owner[l] == rank else 0 for l in range(nlist)])]).astype(np.uint64)
    sk, ss, _, sc, _ = o.ivf_search(cent, soffs, vecs[keep], q, k, nprobe, n, keys=keys[keep])
    tk = torch.from_numpy(sk.astype(np.int64))
    ts = torch.from_numpy(ss)
    tc = torch.from_numpy(sc.astype(np.int32))
    assert pack_candidates(tk, ts, tc).numel() == packed_bytes(nq, k)
    gk, gs, gc = all_gather_candidates(tk, ts, tc)
    assert gk.shape == (world, nq, k) and gc.shape == (world, nq)
    mk, ms, mc = o.merge_topk(gk.numpy().astype(np.uint64), gs.numpy(), gc.numpy().astype(np.uint32), k)
    ok = bool(np.array_equal(mc, fc))
    for i in range(nq):
        ok = ok and np.array_equal(ms[i, :fc[i]], fs[i, :fc[i]])
        # ids equal wherever scores are not tied
        uniq = np.concatenate([[True], np.diff(fs[i, :fc[i]]) != 0]) & np.concatenate([np.diff(fs[i, :fc[i]]) != 0, [True]])
        ok = ok and np.array_equal(mk[i, :fc[i]][uniq], fk[i, :fc[i]][uniq])
    # the DEALT coarse pass (ShardedIVF(deal_coarse=True)): every rank computes the probe lists of ITS slice of the batch only,
    # one all-gather (zvec_amd.dist.all_gather_probe_lists) must hand every rank the probe table of the whole batch — the same
    # table the local coarse pass writes (here: the oracle's), hence the same plan and the same results
    from zvec_amd.dist import all_gather_probe_lists
    per = (nq + world - 1) // world
    lo, hi = min(nq, rank * per), min(nq, (rank + 1) * per)
    mine = torch.zeros(per * (nprobe + 1), dtype=torch.int32)
    if hi > lo:
        _, _, _, _, _, pr = o.ivf_search(cent, offs, vecs, q[lo:hi], k, nprobe, n, keys=keys, want_probes=True)
        mine[:(hi - lo) * nprobe] = torch.from_numpy(pr.astype(np.int64).reshape(-1).astype(np.int32))
        mine[per * nprobe:per * nprobe + (hi - lo)] = nprobe
    idx, cnt = all_gather_probe_lists(mine, per, nprobe)
    ok = ok and np.array_equal(idx.numpy()[:nq].astype(np.uint32), probes.astype(np.uint32)) and bool((cnt[:nq] == nprobe).all())
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_candidate_exchange_and_merge():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
