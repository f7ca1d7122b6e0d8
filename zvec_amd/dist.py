"""One-process-per-GPU sharding of the IVF scan + the candidate-list exchange.

The IVF index shards by inverted list (whole lists, dealt to ranks by the byte-balanced map of
zvec_hip_ivf_shard_map; centroids are replicated), a flat index by contiguous row ranges; every
rank sees the whole query batch, scans only the probed lists it owns and produces a partial top-k per
query.  The only exchange on the path is ONE all-gather of the small candidate lists
([count][topk] keys u64 + scores f32 + counts u32 per rank; 1024x10 => 124 KiB per rank) over RCCL
(`torch.distributed` backend "nccl" on ROCm), followed by a local k-way merge with the same order rule
as CombinedVectorColumnIndexer::Search (combined_vector_column_indexer.cc:172-232): concatenate in
part order, sort by score, truncate.  No all-reduce anywhere; the collective is latency-bound.

`torch` is used for device buffers and the collective only.
"""
import torch
import torch.distributed as dist


def packed_bytes(count, topk):
    return count * topk * 8 + count * topk * 4 + count * 4


def pack_candidates(keys, scores, counts):
    """keys int64 [count][topk], scores f32 [count][topk], counts int32 [count] -> one uint8 buffer."""
    return torch.cat([keys.reshape(-1).view(torch.uint8), scores.reshape(-1).view(torch.uint8),
                      counts.reshape(-1).view(torch.uint8)])


def unpack_candidates(gathered, world, count, topk):
    """gathered uint8 [world][packed_bytes] -> (keys [world][count][topk] i64, scores f32, counts i32)."""
    kb, sb = count * topk * 8, count * topk * 4
    g = gathered.view(world, -1)
    keys = g[:, :kb].contiguous().view(torch.int64).view(world, count, topk)
    scores = g[:, kb:kb + sb].contiguous().view(torch.float32).view(world, count, topk)
    counts = g[:, kb + sb:].contiguous().view(torch.int32).view(world, count)
    return keys, scores, counts


def all_gather_candidates(keys, scores, counts, group=None):
    """the one collective of the path.  Works on any backend (RCCL on GPUs, gloo in the CPU tests)."""
    world = dist.get_world_size(group)
    count, topk = keys.shape
    mine = pack_candidates(keys, scores, counts)
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the N>1 path on a box without RCCL peers: stage through the host
        host = mine.cpu()
        out = torch.empty(world * host.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, host, group=group)
        out = out.to(mine.device)
    else:
        out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)
    return unpack_candidates(out, world, count, topk)


def all_gather_probe_lists(mine, per, np_, group=None, out=None):
    """The exchange of a DEALT coarse pass: `mine` = this rank's slice of the probe lists, int32 [per x np_] centroid ids (coarse-score
    order) followed by [per] valid counts, for queries [rank x per, (rank + 1) x per) of the batch.  ONE all-gather; returns
    (idx [world x per][np_], cnt [world x per]) in query order — rank-major slices are query order.  Any backend (RCCL on the GPUs,
    gloo staged through the host in the CPU tests / one-GPU rehearsals)."""
    world = dist.get_world_size(group)
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(world * mine.numel(), dtype=mine.dtype)
        dist.all_gather_into_tensor(host, mine.cpu(), group=group)
        gathered = host.to(mine.device)
    else:
        gathered = out["gathered"] if out is not None else torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(gathered, mine, group=group)
    g = gathered.view(world, per * (np_ + 1))
    idx = out["idx"] if out is not None else torch.empty((world * per, np_), dtype=mine.dtype, device=mine.device)
    cnt = out["cnt"] if out is not None else torch.empty((world * per,), dtype=mine.dtype, device=mine.device)
    idx.view(world, per * np_).copy_(g[:, :per * np_])
    cnt.view(world, per).copy_(g[:, per * np_:])
    return idx, cnt


class _Sharded:
    """rank-local shard + the exchange/merge step.

    The scan writes keys / scores / counts straight into one packed per-rank buffer (zvec_hip_packed_bytes
    layout), that buffer is all-gathered as is, and the merge kernel reads the gathered buffer with a part
    stride: search -> RCCL -> merge with no repacking kernels in between."""

    def __init__(self, searcher, ctx, rank, world, group=None):
        self.searcher = searcher
        self.ctx = ctx
        self.rank, self.world, self.group = rank, world, group
        self._buf = {}
        # An index with shadow rows (zvec_hip_ivf_set_shadow / zvec_hip_flat_set_shadow): the device-pointer search only enqueues, its
        # result becomes final in the certify step (waits, re-runs uncertified queries on the fp32 rows).  Done inside search() — always
        # before an exchange; a single-rank caller that pipelines lanes may set defer_certify and call searcher.shadow_certify itself
        # before it reads a lane's result (bench.py does).
        self.defer_certify = False

    def _buffers(self, count, topk, device):
        key = (count, topk)
        if key not in self._buf:
            from . import _lib
            pb = int(_lib.lib().zvec_hip_packed_bytes(count, topk))
            mine = torch.zeros(pb, dtype=torch.uint8, device=device)
            kb, sb = count * topk * 8, count * topk * 4
            self._buf[key] = dict(
                pb=pb, mine=mine,
                keys=mine[:kb].view(torch.int64).view(count, topk),
                scores=mine[kb:kb + sb].view(torch.float32).view(count, topk),
                counts=mine[kb + sb:kb + sb + count * 4].view(torch.int32),
                gathered=torch.empty(self.world * pb, dtype=torch.uint8, device=device),
                okeys=torch.empty((count, topk), dtype=torch.int64, device=device),
                oscores=torch.empty((count, topk), dtype=torch.float32, device=device),
                ocounts=torch.empty((count,), dtype=torch.int32, device=device))
        return self._buf[key]

    def _local_search(self, d_queries, count, topk, b, stream_ptr, **kw):
        raise NotImplementedError

    def search(self, d_queries, topk, *args, stream_ptr=None, **kw):
        """d_queries: torch tensor [count][dim] (index element type) on this rank's GPU.  Returns
        (keys, scores, counts) tensors of the GLOBAL top-k (identical on every rank)."""
        from . import _lib
        import ctypes as C
        count = d_queries.shape[0]
        b = self._buffers(count, topk, d_queries.device)
        # torch's legacy default stream has the handle 0, which the C ABI reads as "use the context's own
        # stream": callers should run torch on an explicit stream (bench.py does); if they do not, order the
        # two streams by hand so that torch never reads candidate buffers the scan is still writing
        legacy = not stream_ptr
        if legacy:
            torch.cuda.current_stream().synchronize()
        self._local_search(d_queries, count, topk, b, stream_ptr, *args, **kw)
        if legacy:
            self.ctx.synchronize()
        if self.world == 1:
            return b["keys"], b["scores"], b["counts"]
        if dist.get_backend(self.group) == "gloo":
            # rehearsal of the N>1 path on a box without RCCL peers: stage through the host
            host = torch.empty(self.world * b["pb"], dtype=torch.uint8)
            dist.all_gather_into_tensor(host, b["mine"].cpu(), group=self.group)
            b["gathered"].copy_(host)
        else:
            dist.all_gather_into_tensor(b["gathered"], b["mine"], group=self.group)
        if legacy:
            torch.cuda.current_stream().synchronize()
        rc = _lib.lib().zvec_hip_merge_topk_packed_dev(self.ctx._h, C.c_void_p(b["gathered"].data_ptr()), b["pb"], self.world,
                                                       count, topk, C.c_void_p(b["okeys"].data_ptr()),
                                                       C.c_void_p(b["oscores"].data_ptr()), C.c_void_p(b["ocounts"].data_ptr()),
                                                       C.c_void_p(stream_ptr))
        _lib.check(rc, "zvec_hip_merge_topk_packed_dev")
        if legacy:
            self.ctx.synchronize()
        return b["okeys"], b["oscores"], b["ocounts"]


class ShardedIVF(_Sharded):
    """IVF: the shard holds the inverted lists the byte-balanced list -> shard map gives this rank
    (zvec_hip_ivf_keep_shard / zvec_hip_ivf_shard_map), centroids replicated; probe sets are global.

    deal_coarse (default off): the coarse pass — the one part of a rank's step that does not shrink with the number of ranks
    (2 x Q x nlist x d flop against the replicated centroids; 67 us of nearly whole-chip MFMA work at 1024 x 4096 x 768, which
    cannot share a CU with the resident list scan: DESIGN §7) — is DEALT over the ranks: rank r scores queries
    [r x ceil(Q / G), (r + 1) x ceil(Q / G)), ONE all-gather brings every rank the probe lists of the whole batch
    (Q x (nprobe + 1) x 4 bytes: 132 KB at 1024 x 32), every rank plans from them (zvec_hip_ivf_search_probes_dev).  Identical
    results (the probe lists are the same arrays the local pass would have written); a second small collective per step is the
    price, which only a run on N GPUs can weigh against 7/8 of the coarse pass — cost model in DESIGN §7."""

    def __init__(self, searcher, ctx, rank, world, group=None, deal_coarse=False):
        super().__init__(searcher, ctx, rank, world, group)
        self.deal_coarse = bool(deal_coarse)
        self.shadow_reruns = 0
        self._probe = {}

    def _probe_buffers(self, count, nprobe, device):
        key = (count, nprobe)
        if key not in self._probe:
            nlist = int(self.searcher.info()[1])
            np_ = max(1, min(nprobe, nlist))
            per = (count + self.world - 1) // self.world
            self._probe[key] = dict(
                np=np_, per=per,
                mine=torch.zeros(per * (np_ + 1), dtype=torch.int32, device=device),                # [per][np] ids, then [per] counts
                gathered=torch.zeros(self.world * per * (np_ + 1), dtype=torch.int32, device=device),
                idx=torch.zeros((self.world * per, np_), dtype=torch.int32, device=device),
                cnt=torch.zeros((self.world * per,), dtype=torch.int32, device=device))
        return self._probe[key]

    def _local_search(self, d_queries, count, topk, b, stream_ptr, nprobe, max_scan):
        from . import _lib
        if not (self.deal_coarse and self.world > 1):
            rc = self.searcher.search_dev(d_queries.data_ptr(), count, topk, nprobe, max_scan, b["keys"].data_ptr(),
                                          b["scores"].data_ptr(), b["counts"].data_ptr(), self.ctx, stream=stream_ptr)
            _lib.check(rc, "zvec_hip_ivf_search_dev")
            if self.world > 1 or not self.defer_certify:
                # an index with shadow lists (zvec_hip_ivf_set_shadow): the local lists must be certified (uncertified queries re-run
                # on the fp32 lists) BEFORE they are exchanged; returns at once when the search did not use the shadow lists
                self.shadow_reruns += self.searcher.shadow_certify(d_queries.data_ptr(), count, topk, nprobe, max_scan, b["keys"].data_ptr(),
                                                                   b["scores"].data_ptr(), b["counts"].data_ptr(), self.ctx, stream=stream_ptr)
            return
        p = self._probe_buffers(count, nprobe, d_queries.device)
        per, np_ = p["per"], p["np"]
        lo, hi = min(count, self.rank * per), min(count, (self.rank + 1) * per)
        if hi > lo:
            row_bytes = d_queries.shape[1] * d_queries.element_size()
            rc = self.searcher.coarse_dev(d_queries.data_ptr() + lo * row_bytes, hi - lo, nprobe, p["mine"].data_ptr(),
                                          p["mine"].data_ptr() + per * np_ * 4, self.ctx, stream=stream_ptr)
            _lib.check(rc, "zvec_hip_ivf_coarse_dev")
        legacy = not stream_ptr
        if legacy:
            self.ctx.synchronize()
        all_gather_probe_lists(p["mine"], per, np_, self.group, out=p)
        if legacy:
            torch.cuda.current_stream().synchronize()
        rc = self.searcher.search_probes_dev(d_queries.data_ptr(), count, topk, nprobe, max_scan, p["idx"].data_ptr(), p["cnt"].data_ptr(),
                                             b["keys"].data_ptr(), b["scores"].data_ptr(), b["counts"].data_ptr(), self.ctx,
                                             stream=stream_ptr)
        _lib.check(rc, "zvec_hip_ivf_search_probes_dev")
        self.shadow_reruns += self.searcher.shadow_certify(d_queries.data_ptr(), count, topk, nprobe, max_scan, b["keys"].data_ptr(),
                                                           b["scores"].data_ptr(), b["counts"].data_ptr(), self.ctx, stream=stream_ptr)

    def search(self, d_queries, topk, nprobe, max_scan, stream_ptr):
        return super().search(d_queries, topk, nprobe, max_scan, stream_ptr=stream_ptr)


def flat_row_range(n, rank, world):
    """rows [rank*n/world, (rank+1)*n/world) — SURVEY §8(e): contiguous row ranges, key = local + offset
    (combined_vector_column_indexer.cc:140-145)."""
    return (rank * n) // world, ((rank + 1) * n) // world


class ShardedFlat(_Sharded):
    """Flat: rank g holds the contiguous row range flat_row_range(n, g, world) with GLOBAL keys (the caller appends its
    range with keys = global row numbers, i.e. local position + range start), every rank scans the whole batch over its
    rows, same exchange + merge.  Part order = rank order = ascending row ranges, so ties resolve as in one scan."""

    def _local_search(self, d_queries, count, topk, b, stream_ptr, d_exclude=None, threshold=None):
        from . import _lib
        kw = {} if threshold is None else {"threshold": threshold}
        rc = self.searcher.search_dev(d_queries.data_ptr(), count, topk, b["keys"].data_ptr(), b["scores"].data_ptr(),
                                      b["counts"].data_ptr(), self.ctx, d_exclude=d_exclude, stream=stream_ptr, **kw)
        _lib.check(rc, "zvec_hip_flat_search_dev")
        if self.world > 1 or not self.defer_certify:      # a store with shadow rows: certified (uncertified queries re-run in fp32) BEFORE the exchange
            self.searcher.shadow_certify(d_queries.data_ptr(), count, topk, b["keys"].data_ptr(), b["scores"].data_ptr(),
                                         b["counts"].data_ptr(), self.ctx, d_exclude=d_exclude, stream=stream_ptr)

    def search(self, d_queries, topk, stream_ptr, d_exclude=None, threshold=None):
        return super().search(d_queries, topk, stream_ptr=stream_ptr, d_exclude=d_exclude, threshold=threshold)
