#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X zvec scan core (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a child, before
anything here touches the GPU), relays their output and exits with their status.  Under a launcher
(`torch.distributed.run ... bench.py --gpus N`) each process is one rank; WORLD_SIZE != --gpus is an error.

A "step" = one pass of the hot path over one batch of synthetic queries already resident in HBM:
coarse assign -> plan -> list-major IVF scan (MFMA distance + fused top-k) -> merge [-> all-gather of
the candidate lists over RCCL + shard merge when N>1].  Workload at N=1 = BASELINE.json configs[2]:
IVF-Flat nlist=4096 nprobe=32, 10M x 768 fp32, batch=1024, k=10 (the configuration the metric
"QPS @ recall@10 >= 0.99, 10M x 768 fp32, batch=1024" is quoted on).  With N GPUs the SAME 10M index
is sharded by inverted list, byte-balanced (strong scaling).

The corpus is never resident as a whole: it is generated chunk by chunk (counter-based generator, identical on
every rank) three times — (A) ground-truth flat shard + k-means sample, (B) nearest-centroid labels (chunks dealt
round-robin to the ranks, labels summed over RCCL), (C) list fill (every rank keeps the rows of the lists it owns).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the list scan, timed with HIP events
on its launch stream inside the library) and `cpu_baseline` (kind "reference": the reference's OWN IVFSearcher /
FlatSearcher — oracle/_ref/libzvec_ref_core.so, its core library compiled in place — on the host cores, same index /
queries; N=1 only; kind "port" = the oracle's restated loops when that library did not travel).  oracle/ is used ONLY
for that baseline and the parity cross-check — never on the timed GPU path.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260320
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # dense fp32 MFMA peak (no xf32 on gfx950)
MFMA_F16_PEAK_TF = 2500.0    # dense fp16 MFMA peak (MI355X_MICROARCH.md; not the 2:1-sparsity figure)
CHUNK = 1 << 20

WORKLOADS = {
    # name: (kind, n, dim, nlist, nprobe, batch, dtype)
    "ivf10m": ("ivf", 10_000_000, 768, 4096, 32, 1024, "fp32"),       # BASELINE configs[2]  (headline)
    "ivf1m": ("ivf", 1_000_000, 768, 1024, 16, 1024, "fp32"),         # small rehearsal
    "flat_sift1m": ("flat", 1_000_000, 128, 0, 0, 1, "fp32"),         # BASELINE configs[0] shape: 1M x 128-d, one query at a time
    "flat1m": ("flat", 1_000_000, 768, 0, 0, 256, "fp32"),            # BASELINE configs[1]
    "ivf10m_fp16": ("ivf", 10_000_000, 768, 4096, 32, 1024, "fp16"),  # BASELINE configs[3]'s storage type at 1-GPU size
    "ivf100m_fp16": ("ivf", 100_000_000, 768, 16384, 64, 1024, "fp16"),  # BASELINE configs[3]: 8 ranks, or --shard-of 8 for one rank's share
    "filter10m": ("flat", 10_000_000, 768, 0, 0, 512, "fp32"),        # BASELINE configs[4]: bitmap-gated scan, keep 10 %
    "flat1m_fp16": ("flat", 1_000_000, 768, 0, 0, 256, "fp16"),       # (not a BASELINE config: fp16 rows through the wide flat tile)
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ivf10m", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--nprobe", type=int, default=0)
    ap.add_argument("--nlist", type=int, default=0)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--kmeans-iters", type=int, default=0, help="0 = what --train says")
    ap.add_argument("--train", default="auto", choices=["auto", "reference", "quick"],
                    help="IVF training: reference = the reference trainer's setting, 20 Lloyd rounds over EVERY row (tests/golden/"
                         "kmeans_quality.json: the GPU build then lands inside the spread of the reference's own trainer); quick = 10 rounds "
                         "on a strided 256-rows-per-list sample (SSE 1.2 %% higher, recall@10 at equal nprobe ~3 points lower on that "
                         "fixture); auto = reference when the raw rows fit beside the index (<= 48 GB), else quick")
    ap.add_argument("--gt-queries", type=int, default=0, help="queries recall is measured on (0 = the timed batch)")
    ap.add_argument("--intrinsic-dim", type=int, default=12)
    ap.add_argument("--target-recall", type=float, default=0.99)
    ap.add_argument("--nprobe-step", type=int, default=2, help="widening step of the recall sweep")
    ap.add_argument("--gate", action="store_true",
                    help="IVF workloads with --streams > 1: the lanes share a gate (zvec_hip_gate_t): their list scans take turns "
                         "strictly one after the other.  Off by default since round 3: the persistent scan keeps the other lane's "
                         "scan off the device anyway (registers), and without the gate's event hand-over (~20 us) the next scan "
                         "starts under the tail of the previous one: +2.5 %% QPS, the scan's own HIP-event time within 1 %%")
    ap.add_argument("--no-gate", action="store_true", help="(the default now; kept so that recorded command lines still run)")
    ap.add_argument("--streams", type=int, default=2,
                    help="consecutive (independent) batches alternate over this many contexts / HIP streams: the small kernels of "
                         "batch i+-1 run around batch i's scan (IVF workloads; flat workloads on one GPU without a predicate)")
    ap.add_argument("--deal-coarse", action="store_true",
                    help="N > 1 IVF: the coarse pass dealt over the ranks (rank r scores 1/N of the batch, one all-gather of the probe "
                         "lists, every rank plans from them; zvec_amd.dist.ShardedIVF(deal_coarse=True)).  Off by default: it trades "
                         "7/8 of the 67 us coarse pass for a second small collective per step, which only an N-GPU run can price")
    ap.add_argument("--keep", type=float, default=0.1, help="filter workloads: fraction of rows the bitmap keeps")
    ap.add_argument("--flat-threshold", type=float, default=None,
                    help="diagnostic: RNN radius for the flat workloads (a huge negative value admits nothing => distance-only time)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="on ONE GPU, hold and time only shard 0 of an N-way byte-balanced list sharding (no exchange; "
                         "recall is not computed) — the per-rank compute of an N-GPU run; with --workload ivf100m_fp16 "
                         "--shard-of 8 this is one rank's real share of BASELINE configs[3]")
    ap.add_argument("--no-shadow-leg", action="store_true",
                    help="skip the second measurement of fp32 IVF workloads through the half-width pre-selection (zvec_hip_ivf_set_shadow: "
                         "fp16 shadow lists + fp32 re-scoring + certificate + fp32 re-run of uncertified queries; reported beside `value`, never as it)")
    ap.add_argument("--shadow-preselect", type=int, default=0, help="rows pre-selected per query on the shadow lists (0 = max(32, 3k))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-port", action="store_true", help="time the oracle's restated loops (kind \"port\") even when the reference's own classes are available")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host-pointer measurement")
    ap.add_argument("--cpu-queries", type=int, default=0, help="0 = one whole batch")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, form the process group (backend: $ZVEC_BENCH_BACKEND, default nccl = RCCL), "
                         "all-gather the rank ids and print the JSON line; no GPU work (the CPU test of the launcher path)")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_launch(args):
    """--gpus N > 1 and no launcher: become the launcher.  Nothing in this process has touched torch / HIP yet; the
    ranks are fresh children of a torch.distributed.run child (never an exec of a process that initialised the GPU)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            print("bench.py: WORLD_SIZE=%s but --gpus %d; launch with `python -m torch.distributed.run --nnodes=1 "
                  "--nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...` or run "
                  "`python bench.py --gpus %d` and let it start the ranks" % (env_world, args.gpus, args.gpus, args.gpus, args.gpus),
                  file=sys.stderr, flush=True)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def corpus_proj(torch, dim, device, intrinsic_dim):
    pg = torch.Generator(device=device)
    pg.manual_seed(SEED + 7)
    return torch.randn((intrinsic_dim, dim), generator=pg, device=device, dtype=torch.float32)


def corpus_chunks(torch, n, dim, device, seed, proj, out_dtype, noise=0.02, chunk=CHUNK):
    """Seeded synthetic corpus with realistic neighbourhood structure, one chunk at a time: a Gaussian of low intrinsic
    dimension embedded in R^dim by a fixed random projection, plus small isotropic noise (x = z A + noise,
    z ~ N(0, I_r)).  i.i.d. Gaussians in 768-d have no cluster structure at all (no IVF can reach recall 0.99 at
    nprobe 32/4096) and a mixture of a few thousand well separated blobs makes recall trivially 1 at
    nprobe 1; a low-rank Gaussian gives k-means cells of moderate imbalance and true neighbours that spill into
    adjacent cells, like embedding data.  Counter-based generator (torch Philox), identical on every rank and on every
    pass; yields (first_row, rows[m][dim]) with fp16 workloads rounded to nearest even (HalfFloatConverter)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        z = torch.randn((m, proj.shape[0]), generator=g, device=device, dtype=torch.float32)
        x = torch.mm(z, proj)
        x += torch.randn((m, dim), generator=g, device=device, dtype=torch.float32) * noise
        yield o, (x if out_dtype == torch.float32 else x.to(out_dtype))


def gen_corpus(torch, n, dim, device, seed, proj, out_dtype):
    out = torch.empty((n, dim), device=device, dtype=out_dtype)
    for o, x in corpus_chunks(torch, n, dim, device, seed, proj, out_dtype):
        out[o:o + x.shape[0]] = x
    return out


def launch_check(args):
    import torch
    import torch.distributed as dist
    backend = os.environ.get("ZVEC_BENCH_BACKEND", "nccl")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group(backend)
    mine = torch.tensor([rank, local_rank], dtype=torch.int64, device=dev)
    got = torch.zeros((dist.get_world_size(), 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(got.view(-1), mine)
    ranks = sorted(int(r) for r in got[:, 0].cpu().tolist())
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks": len(set(ranks)), "ranks": ranks,
                          "backend": dist.get_backend()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if (len(set(ranks)) == world == args.gpus) else 3)


def main():
    args = parse_args()
    maybe_launch(args)
    if args.launch_check:
        launch_check(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the scan core has no CPU fallback")
    # rehearsal of the N > 1 path on a ONE-GPU box: ZVEC_BENCH_BACKEND=gloo puts every rank on cuda:0 and runs the
    # collectives through the host (the RCCL transport itself is then the only thing not exercised)
    backend = os.environ.get("ZVEC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one explicit HIP stream for everything (torch ops, the scan library, RCCL): torch's legacy default
    # stream has handle 0, which the C ABI would read as "the context's own stream"
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    rccl = None

    def coll(fn, t, *a, **kw):
        """a collective on a device tensor: directly over RCCL, or staged through the host for the gloo rehearsal"""
        if backend == "nccl":
            return fn(t, *a, **kw)
        h = t.cpu()
        fn(h, *a, **kw)
        t.copy_(h)

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # proof that RCCL sees every rank: one all-gather of (rank, device) over the process group
        mine = torch.tensor([rank, local_rank], dtype=torch.int64, device=dev)
        got = torch.zeros((world, 2), dtype=torch.int64, device=dev)
        if backend == "nccl":
            dist.all_gather_into_tensor(got.view(-1), mine)
        else:
            hg = got.cpu()
            dist.all_gather_into_tensor(hg.view(-1), mine.cpu())
            got.copy_(hg)
        torch.cuda.synchronize()
        rows = got.cpu().tolist()
        rccl = {"ranks": len(set(int(r[0]) for r in rows)), "devices": [int(r[1]) for r in rows], "backend": dist.get_backend()}
        if rccl["ranks"] != args.gpus:
            raise SystemExit("RCCL process group has %d ranks, --gpus %d" % (rccl["ranks"], args.gpus))

    import zvec_amd
    from zvec_amd.dist import ShardedIVF, ShardedFlat, flat_row_range

    kind, n, dim, nlist, nprobe, batch, dtype = WORKLOADS[args.workload]
    tdtype = torch.float16 if dtype == "fp16" else torch.float32
    n = args.n or n
    batch = args.batch or batch
    nprobe = args.nprobe or nprobe
    topk = args.topk
    if args.nlist and kind == "ivf":
        nlist = args.nlist
    elif args.n and kind == "ivf":
        nlist = max(16, min(nlist, int(round(math.sqrt(n) * 1.3))))
    t0 = time.time()
    log("workload %s: n=%d dim=%d nlist=%d nprobe=%d batch=%d k=%d world=%d" % (args.workload, n, dim, nlist, nprobe, batch, topk, world))
    stream_ptr = torch.cuda.current_stream().cuda_stream
    proj = corpus_proj(torch, dim, dev, args.intrinsic_dim)
    shard_mode = args.shard_of > 1 and world == 1 and kind == "ivf"
    if shard_mode:
        args.no_cpu_baseline = True
    # small batches (the product's count = 1 calls): the timed steps ROTATE over a pool of different query batches — the same
    # single query every step would find its ~290 MB of probed rows half-resident in the 256 MB Infinity Cache — and recall is
    # measured on the whole pool.  (A 1024-query batch reads every list of the index each step: nothing to rotate for.)
    qpool_n = 64 if (kind == "ivf" and batch <= 16) else 1
    ngt = 0 if shard_mode else max(batch * qpool_n, args.gt_queries or batch)
    nqueries = max(batch * qpool_n, ngt)
    queries = gen_corpus(torch, nqueries, dim, dev, SEED + 1, proj, tdtype)   # held-out draws (fp16: HalfFloatReformer = RNE cast)
    # BASELINE configs[1] is a flat INNER-PRODUCT scan; every other workload is L2
    flat_metric = "InnerProduct" if args.workload == "flat1m" else "SquaredEuclidean"
    extra_cfg = {}

    if kind == "flat":
        # ---------------- flat workloads: rank g holds the contiguous row range g ----------------
        lo, hi = flat_row_range(n, rank, world)
        flat = zvec_amd.HipFlatSearcher(dim, flat_metric, device=local_rank, dtype=dtype)
        base = None
        keep_base = world == 1 and n <= 2_000_000 and not args.no_cpu_baseline
        parts = []
        for o, x in corpus_chunks(torch, n, dim, dev, SEED, proj, tdtype):
            a, b = max(o, lo), min(o + x.shape[0], hi)
            if a < b:
                rows = x[a - o:b - o].contiguous()
                keys = None if world == 1 else torch.arange(a, b, dtype=torch.int64, device=dev)
                zvec_amd._lib.check(flat.add_batch_dev(rows.data_ptr(), b - a, d_keys_ptr=keys.data_ptr() if keys is not None else None,
                                                       stream=stream_ptr), "flat append")
                torch.cuda.synchronize()
                if keep_base:
                    parts.append(rows)
        if keep_base:
            base = torch.cat(parts)
            del parts
        log("flat rows [%d, %d) packed in %.1fs" % (lo, hi, time.time() - t0))
        fctx = flat.create_context()
        fctx.set_stream(stream_ptr)
        result = run_flat(torch, dist, zvec_amd, ShardedFlat(flat, fctx, rank, world), flat, fctx, queries[:batch].contiguous(), n, hi - lo,
                          dim, topk, args, dev, stream_ptr, world, rank, base, flat_metric, coll)
        recall = 1.0
        nprobe_base = recall_base = None
    else:
        nshards = args.shard_of if shard_mode else world
        shard = 0 if shard_mode else rank
        ivf = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean", device=local_rank, dtype=dtype)
        zvec_amd._lib.check(ivf.set_shard(shard, nshards), "set_shard")
        # ---------------- pass A: ground-truth flat shard + the k-means sample ----------------
        train = args.train
        if train == "auto":
            train = "reference" if n * dim * (2 if dtype == "fp16" else 4) <= 48e9 else "quick"
        kmeans_iters = args.kmeans_iters or (20 if train == "reference" else 10)
        S = n if train == "reference" else min(n, 256 * nlist)
        extra_cfg["index_training"] = (
            "GPU k-means, %d Lloyd rounds over %s" % (kmeans_iters, "every row (the reference trainer's setting)" if S == n else
                                                     "a strided sample of %d rows (256 per list: the quick setting, below the reference "
                                                     "trainer's clustering quality; QPS does not depend on it — a 1024-query batch streams "
                                                     "every list either way — the nprobe needed for the recall target does)" % S))
        sample_ids = (np.arange(S, dtype=np.uint64) * np.uint64(n)) // np.uint64(S)    # the strided sample of zvec_hip_ivf_build
        sample = torch.empty((S, dim), device=dev, dtype=tdtype)
        flat = None
        lo, hi = flat_row_range(n, rank, world)
        if ngt:
            flat = zvec_amd.HipFlatSearcher(dim, flat_metric, device=local_rank, dtype=dtype)
            zvec_amd._lib.check(flat.reserve(hi - lo), "flat reserve")      # no growth copies: at 100M x 768 fp16 the rows are 154 GB
        t1 = time.time()
        for o, x in corpus_chunks(torch, n, dim, dev, SEED, proj, tdtype):
            m = x.shape[0]
            if S == n:
                sample[o:o + m] = x
            else:
                i0, i1 = np.searchsorted(sample_ids, [o, o + m])
                if i1 > i0:
                    sample[i0:i1] = x[torch.from_numpy((sample_ids[i0:i1] - np.uint64(o)).astype(np.int64)).to(dev)]
            a, b = max(o, lo), min(o + m, hi)
            if flat is not None and a < b:
                rows = x[a - o:b - o].contiguous()
                keys = None if world == 1 else torch.arange(a, b, dtype=torch.int64, device=dev)
                zvec_amd._lib.check(flat.add_batch_dev(rows.data_ptr(), b - a, d_keys_ptr=keys.data_ptr() if keys is not None else None,
                                                       stream=stream_ptr), "flat append")
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        log("pass A (sample%s) in %.1fs" % (", ground-truth flat rows [%d, %d)" % (lo, hi) if flat is not None else "", time.time() - t1))
        # exact ground truth for recall: the flat path over the same corpus (sharded by row range when N > 1)
        gt = None
        if flat is not None:
            fctx = flat.create_context()
            fctx.set_stream(stream_ptr)
            gk, _, _ = ShardedFlat(flat, fctx, rank, world).search(queries[:ngt].contiguous(), topk, stream_ptr)
            torch.cuda.synchronize()
            gt = gk.cpu().numpy()
            del fctx, flat
            torch.cuda.empty_cache()
        # ---------------- train: k-means on the sample (same seed, same rows on every rank) ----------------
        t1 = time.time()
        zvec_amd._lib.check(ivf.train_dev(sample.data_ptr(), S, nlist, kmeans_iters=kmeans_iters, seed=SEED, stream=stream_ptr), "ivf train")
        del sample
        if world > 1:      # one set of centroids for everybody: rank 0's, bit for bit
            cent = torch.from_numpy(ivf.get_centroids().view(np.uint8)).to(dev)
            coll(dist.broadcast, cent, 0)
            if rank != 0:
                zvec_amd._lib.check(ivf.set_centroids(cent.cpu().numpy().view(np.float16 if dtype == "fp16" else np.float32)), "set_centroids")
        log("k-means (%d iters, %d lists, %d sample rows) in %.1fs" % (kmeans_iters, nlist, S, time.time() - t1))
        # ---------------- pass B: nearest-centroid labels (chunks dealt round-robin to the ranks) ----------------
        t1 = time.time()
        labels = torch.zeros((n,), dtype=torch.int32, device=dev)
        for ci, (o, x) in enumerate(corpus_chunks(torch, n, dim, dev, SEED, proj, tdtype)):
            if ci % world == rank:
                zvec_amd._lib.check(ivf.label_dev(x.data_ptr(), x.shape[0], labels[o:o + x.shape[0]].data_ptr(), stream=stream_ptr), "label")
        if world > 1:
            coll(dist.all_reduce, labels)  # every row was labelled by exactly one rank, the others hold 0
        torch.cuda.synchronize()
        labels_h = labels.cpu().numpy().astype(np.uint32)
        del labels
        sizes = np.bincount(labels_h, minlength=nlist).astype(np.uint32)
        log("pass B (labels) in %.1fs; list sizes min %d / mean %.0f / max %d" % (time.time() - t1, sizes.min(), sizes.mean(), sizes.max()))
        # ---------------- pass C: fill the lists this rank owns ----------------
        t1 = time.time()
        zvec_amd._lib.check(ivf.begin_lists(sizes), "begin_lists")
        for o, x in corpus_chunks(torch, n, dim, dev, SEED, proj, tdtype):
            zvec_amd._lib.check(ivf.add_dev(x.data_ptr(), x.shape[0], labels_h[o:o + x.shape[0]], o, stream=stream_ptr), "add")
        zvec_amd._lib.check(ivf.end_lists(), "end_lists")
        del labels_h
        torch.cuda.empty_cache()
        _, shard_rows = zvec_amd.shard_map(sizes, nshards)
        local_rows = ivf.info()[0]
        assert local_rows == int(shard_rows[shard])
        extra_cfg["shard_rows_max_over_mean"] = float(shard_rows.max() / shard_rows.mean())
        log("pass C (lists of shard %d/%d: %d rows = %.2f GB; max/mean over shards %.4f) in %.1fs" % (
            shard, nshards, local_rows, local_rows * dim * (2 if dtype == "fp16" else 4) / 1e9, extra_cfg["shard_rows_max_over_mean"],
            time.time() - t1))
        ctx = ivf.create_context()
        ctx.set_stream(stream_ptr)
        sh = ShardedIVF(ivf, ctx, rank, world, deal_coarse=args.deal_coarse)
        max_scan = n        # brute_force_threshold = N-1 => exactly nprobe lists are probed
        q = queries[:batch].contiguous()
        qpool = [queries[j * batch:(j + 1) * batch].contiguous() for j in range(qpool_n)]

        # recall@10 of the configuration being timed, measured on the timed queries; the metric demands >= 0.99: if
        # nprobe (BASELINE: 32) does not reach it on this corpus, widen nprobe until it does and time THAT (the
        # nprobe=32 recall and QPS are reported too)
        def recall_at(np_):
            k_, s_, c_ = sh.search(queries[:ngt].contiguous(), topk, np_, max_scan, stream_ptr)
            torch.cuda.synchronize()
            got = k_.cpu().numpy()
            return float(np.mean([len(set(got[i].tolist()) & set(gt[i].tolist())) / float(topk) for i in range(ngt)]))
        nprobe_base = nprobe
        recall_base = recall = None
        if gt is not None:
            recall_base = recall = recall_at(nprobe)
            log("recall@%d = %.4f (nprobe=%d, %d queries)" % (topk, recall, nprobe, ngt))
            while recall < args.target_recall and nprobe < nlist:
                nprobe = min(nlist, nprobe + max(args.nprobe_step, 1))
                recall = recall_at(nprobe)
                log("recall@%d = %.4f (nprobe=%d)" % (topk, recall, nprobe))
            scanned, probes = ivf.last_stats(ctx, ngt)
            log("rows scanned per query: mean %.0f (%.3f%% of the corpus), lists probed %.1f" % (
                scanned.mean(), 100.0 * scanned.mean() / n, probes.mean()))

        cpu = None
        if not args.no_cpu_baseline and world == 1:
            gk, gs, gc = sh.search(q, topk, nprobe, max_scan, stream_ptr)
            torch.cuda.synchronize()
            cpu = cpu_baseline_ivf(torch, ivf, q, topk, nprobe, max_scan, args, dtype,
                                   gpu=(gk.cpu().numpy(), gs.cpu().numpy(), gc.cpu().numpy()))

        # ---------------- timed region ----------------
        # --streams 2 (default): consecutive (independent) batches alternate between two contexts on two HIP streams
        # (serving-style pipelining): the coarse pass / plan of batch i+1 and the merge / refine / exchange of batch i-1
        # run around batch i's list scan.  The list scan is a persistent kernel that holds every CU slot, so two scans
        # overlap only where one drains and the next fills (~80 us of 4.9 ms; --gate forbids even that, at the price of
        # an event hand-over per step).  Every step is still one complete pass over one batch and all K steps complete
        # inside the timed region; the scan's HIP-event duration includes that short shared head / tail
        lanes = [(sh, stream_ptr, None)]
        if args.streams > 1:
            gate = zvec_amd.Gate(local_rank) if args.gate else None
            ctx.set_gate(gate)
            for _ in range(args.streams - 1):
                s2 = torch.cuda.Stream(device=dev)
                ctx2 = ivf.create_context()
                ctx2.set_stream(s2.cuda_stream)
                ctx2.set_gate(gate)
                lanes.append((ShardedIVF(ivf, ctx2, rank, world, deal_coarse=args.deal_coarse), s2.cuda_stream, s2))

        def run_step(i, np_):
            sh_i, sp, ts = lanes[i % len(lanes)]
            qi = qpool[i % qpool_n]
            if ts is None:
                sh_i.search(qi, topk, np_, max_scan, sp)
            else:
                with torch.cuda.stream(ts):
                    sh_i.search(qi, topk, np_, max_scan, sp)

        host_issue = [0.0]

        def timed(np_, steps, warmup):
            for i in range(warmup * len(lanes)):
                run_step(i, np_)
                if i == 0:
                    # the lanes must not START together: two list scans launched at the same instant share the CUs for their whole
                    # length (each then "lasts" ~10 ms in a kernel trace); one step apart they alternate, as they do in steady state
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            for sh_l, _, _ in lanes:                     # EVERY lane's scans are timed (HIP events on the lane's own stream)
                sh_l.ctx.profile(True)
                sh_l.ctx.profile_read(reset=True)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t_start = time.perf_counter()
            for i in range(steps):
                run_step(i, np_)
            host_issue[0] = (time.perf_counter() - t_start) / steps * 1e3      # ms of host time to ISSUE one step
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t_start
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                coll(dist.all_reduce, t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            pr = {"scan_ms": 0.0, "launches": 0, "bytes": 0.0, "flops": 0.0, "per_lane_ms": []}
            for sh_l, _, _ in lanes:
                p1 = sh_l.ctx.profile_read(reset=True)
                sh_l.ctx.profile(False)
                for key in ("scan_ms", "launches", "bytes", "flops"):
                    pr[key] += p1[key]
                pr["per_lane_ms"].append(p1["scan_ms"] / max(p1["launches"], 1))
            return el, pr
        elapsed, prof = timed(nprobe, args.steps, args.warmup)
        extra_cfg["host_issue_ms_per_step"] = host_issue[0]
        qps_base = None
        if nprobe != nprobe_base:      # the BASELINE nprobe as well (outside the reported region)
            el32, _ = timed(nprobe_base, max(3, min(args.steps, 10)), 1)
            qps_base = batch * max(3, min(args.steps, 10)) / el32
        # the boundary's HOST-pointer entry (what IndexRunner::search_impl hands over: queries in host memory, results
        # back into host buffers) on the same batch: H2D of the queries + D2H of the lists per call.  Reported next to
        # `value`, never as it.
        host_qps = None
        if world == 1 and not args.no_host_path:
            hctx = ivf.create_context()
            qhs = [np.ascontiguousarray(x.cpu().numpy()) for x in qpool]      # (small batches: a different batch every call)
            hcall = [0]
            hk = np.zeros((batch, topk), np.uint64)
            hs = np.zeros((batch, topk), np.float32)
            hc = np.zeros(batch, np.uint32)
            L = zvec_amd._lib.lib()

            def host_call():      # zvec_hip_ivf_search: the C ABI entry itself (no Python result objects)
                qh = qhs[hcall[0] % len(qhs)]
                hcall[0] += 1
                zvec_amd._lib.check(L.zvec_hip_ivf_search(ivf._h, hctx._h, qh.ctypes.data, batch, topk, 3.4028234663852886e38,
                                                          nprobe, n - 1, None, hk.ctypes.data, hs.ctypes.data, hc.ctypes.data),
                                    "zvec_hip_ivf_search")
            nhost = 5 if batch > 16 else 200
            for _ in range(1 if batch > 16 else 20):
                host_call()
            th = time.perf_counter()
            for _ in range(nhost):
                host_call()
            host_qps = nhost * batch / (time.perf_counter() - th)
            log("host-pointer entry zvec_hip_ivf_search (PCIe-inclusive): %.0f QPS" % host_qps)
        # ---------------- second measurement: the same steps through the half-width pre-selection ----------------
        # (zvec_hip_ivf_set_shadow: the list scan streams an fp16 twin of the lists for k' rows per query, those are re-scored on the
        # fp32 rows, the k best are CERTIFIED against the measured rounding, uncertified queries are re-run on the fp32 lists —
        # inside the timed region.  Same answer, about half the bytes.  Reported as `certified_half_scan`, never as `value`.)
        shadow_leg = None
        if dtype == "fp32" and world == 1 and batch > 8 and not args.no_shadow_leg:
            try:
                # the fp32 route's answers on every batch of the pool first (the parity reference of this leg)
                ref32 = []
                for qi in qpool:
                    ref32.append([x.clone() for x in sh.search(qi, topk, nprobe, max_scan, stream_ptr)])
                torch.cuda.synchronize()
                t1 = time.time()
                ivf.set_shadow(True, args.shadow_preselect)
                torch.cuda.synchronize()
                sinfo = ivf.shadow_info()
                log("shadow lists: %.2f GB in %.2fs; max |b - b16| %.4g, max |b16| %.4g" % (sinfo["bytes"] / 1e9, time.time() - t1,
                                                                                          sinfo["max_row_error"], sinfo["max_row_norm"]))
                rer = [0]
                for sh_l, _, _ in lanes:                 # (the lanes certify a step when they come back to it, not inside search())
                    sh_l.defer_certify = True

                def certify(pend, np_):
                    sh_i, qi, k_, s_, c_, sp = pend
                    return ivf.shadow_certify(qi.data_ptr(), qi.shape[0], topk, np_, max_scan, k_.data_ptr(), s_.data_ptr(), c_.data_ptr(),
                                              sh_i.ctx, stream=sp)
                same_ids = same_bits = total = 0
                for qi, (k0, s0, c0) in zip(qpool, ref32):
                    k1, s1, c1 = sh.search(qi, topk, nprobe, max_scan, stream_ptr)
                    rer[0] += certify((sh, qi, k1, s1, c1, stream_ptr), nprobe)
                    torch.cuda.synchronize()
                    same_ids += int((k0 == k1).all(1).sum().item())
                    same_bits += int((s0.view(torch.int32) == s1.view(torch.int32)).all(1).sum().item())
                    total += qi.shape[0]
                del ref32
                parity_rerun = rer[0]
                log("shadow route vs fp32 route: %d / %d queries with identical key lists, %d with identical score bits, %d re-run in fp32" % (
                    same_ids, total, same_bits, parity_rerun))
                pending = [None] * len(lanes)

                def run_step_shadow(i, np_):
                    li = i % len(lanes)
                    sh_i, sp, ts = lanes[li]
                    if pending[li] is not None:            # the lane's previous step: wait for it, re-run what it could not certify
                        rer[0] += certify(pending[li], np_)
                    qi = qpool[i % qpool_n]
                    if ts is None:
                        k_, s_, c_ = sh_i.search(qi, topk, np_, max_scan, sp)
                    else:
                        with torch.cuda.stream(ts):
                            k_, s_, c_ = sh_i.search(qi, topk, np_, max_scan, sp)
                    pending[li] = (sh_i, qi, k_, s_, c_, sp)

                def drain(np_):
                    for li in range(len(lanes)):
                        if pending[li] is not None:
                            rer[0] += certify(pending[li], np_)
                            pending[li] = None
                for i in range(args.warmup * len(lanes)):
                    run_step_shadow(i, nprobe)
                    if i == 0:
                        torch.cuda.synchronize()
                drain(nprobe)
                torch.cuda.synchronize()
                for sh_l, _, _ in lanes:
                    sh_l.ctx.profile(True)
                    sh_l.ctx.profile_read(reset=True)
                rer[0] = 0
                torch.cuda.synchronize()
                t_start = time.perf_counter()
                for i in range(args.steps):
                    run_step_shadow(i, nprobe)
                drain(nprobe)                               # every step certified (and re-run where needed) inside the timed region
                torch.cuda.synchronize()
                el_s = time.perf_counter() - t_start
                spr = {"scan_ms": 0.0, "launches": 0, "bytes": 0.0}
                for sh_l, _, _ in lanes:
                    p1 = sh_l.ctx.profile_read(reset=True)
                    sh_l.ctx.profile(False)
                    for key in ("scan_ms", "launches", "bytes"):
                        spr[key] += p1[key]
                s_ms = el_s / args.steps * 1e3
                s_kernel_ms = spr["scan_ms"] / max(spr["launches"], 1)
                s_bytes = spr["bytes"] / max(spr["launches"], 1)
                s_basis = min(s_kernel_ms, s_ms) if s_kernel_ms > 0 else s_ms
                shadow_leg = {
                    "qps": batch * args.steps / el_s, "ms_per_step": s_ms, "unit": "queries/s",
                    "what": "the same steps with zvec_hip_ivf_set_shadow on: list scan over an fp16 twin of the lists for k' rows per query, fp32 "
                            "re-scoring, per-query certificate from the measured rounding, fp32 re-run of uncertified queries inside the timed region",
                    "preselect": args.shadow_preselect or "chosen by the index (from max(32, 3k): narrowed after clean certify steps, widened after re-runs)",
                    "preselect_rows_at_end": ivf.shadow_width(topk), "shadow_bytes": sinfo["bytes"],
                    "max_row_rounding": sinfo["max_row_error"], "max_row_norm": sinfo["max_row_norm"],
                    "rerun_queries_per_step": rer[0] / float(args.steps),
                    "parity_vs_fp32_route": {"queries": total, "identical_key_lists": same_ids, "identical_score_bits": same_bits,
                                             "rerun_in_fp32": parity_rerun},
                    "kernel": "zvk::scan_kernel<1> over the fp16 shadow lists", "kernel_ms": s_kernel_ms, "algorithmic_bytes": s_bytes,
                    "achieved_gbs": s_bytes / (s_basis * 1e-3) / 1e9 if s_basis > 0 else None,
                    "frac_of_hbm_peak": s_bytes / (s_basis * 1e-3) / 1e9 / HBM_PEAK_GBS if s_basis > 0 else None,
                    "speedup_over_value": None,
                }
                try:      # the fp16 scan's HBM-side bytes per launch, from separate PMC passes (tools/pmc_shadow_traffic.sh), as roofline.traffic
                    if committed_traffic(args, world, shard_mode)[0] is not None and not args.shadow_preselect:
                        shadow_leg["traffic"] = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload + "_shadow")
                        shadow_leg["traffic_source"] = "profiles/r4_ivf10m_shadow_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)"
                except (OSError, ValueError):
                    pass
                if host_qps is not None and batch > 16:
                    # the boundary's host-pointer entry on the same index: zvec_hip_ivf_search certifies (and re-runs) inside the call
                    host_call()
                    th = time.perf_counter()
                    for _ in range(5):
                        host_call()
                    shadow_leg["host_pointer_qps"] = 5 * batch / (time.perf_counter() - th)
                ivf.set_shadow(False)
                for sh_l, _, _ in lanes:
                    sh_l.defer_certify = False
                log("certified half-width scan: %.0f QPS (%.3f ms per step, list scan %.3f ms, %.2f queries re-run per step)" % (
                    shadow_leg["qps"], s_ms, s_kernel_ms, shadow_leg["rerun_queries_per_step"]))
            except Exception as e:      # (the second measurement must never cost the line its first)
                log("certified half-width leg failed: %r" % (e,))
                shadow_leg = {"error": repr(e)}
                try:
                    ivf.set_shadow(False)
                except Exception:
                    pass
        per_launch_ms = prof["scan_ms"] / max(prof["launches"], 1)
        bytes_per_launch = prof["bytes"] / max(prof["launches"], 1)
        flops_per_launch = prof["flops"] / max(prof["launches"], 1)
        traffic, traffic_source = committed_traffic(args, world, shard_mode)
        ms_per_step = elapsed / args.steps * 1e3
        # un-gated lanes: one scan fills the CUs while the previous one drains, so a launch's HIP-event time can exceed the step it
        # belongs to; a launch cannot take longer than a step in steady state, so the roofline then divides by the STEP time
        frac_ms, frac_basis = per_launch_ms, "kernel_ms (HIP events around the launch, mean over every lane's launches)"
        if 0 < ms_per_step < per_launch_ms:
            # (un-gated lanes: a lane's scan is dispatched while the other lane's still holds the CUs, so its own event time — and its
            # duration in a kernel trace — includes that wait; the step time is then the honest per-launch figure: it cannot understate)
            frac_ms, frac_basis = ms_per_step, "ms_per_step (overlapping lanes: kernel_ms > ms_per_step, see kernel_ms_per_lane)"
        achieved = bytes_per_launch / (frac_ms * 1e-3) / 1e9 if frac_ms > 0 else 0.0
        box = box_calibration(zvec_amd, local_rank)
        result = {
            "value": batch * args.steps / elapsed,
            "ms_per_step": ms_per_step,
            "roofline": {"bound": "hbm", "kernel": ("zvk::pkeys_topk_kernel (IVF small-batch route: a wave per four probed rows, a block keeps the k best of its 128%s)" if batch <= 8 else "zvk::scan_kernel<1> (IVF list scan%s)") % (", rank 0's shard" if world > 1 or shard_mode else ""),
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source, "frac_basis": frac_basis,
                         "box_clock_mhz": box.get("clock_mhz"), "box_stream_gbs": box.get("stream_gbs"), "box_note": box.get("note"),
                         "frac_of_box_stream": (achieved / box["stream_gbs"]) if box.get("stream_gbs") else None,
                         # (un-gated lanes: one scan fills the CUs while the previous one drains, so the step can be
                         # SHORTER than one scan's own HIP-event time; the difference is then reported as overlap)
                         "kernel_ms": per_launch_ms, "kernel_ms_per_lane": prof.get("per_lane_ms"),
                         "fixed_ms_per_step": max(ms_per_step - per_launch_ms, 0.0),
                         "scan_overlap_ms_per_step": max(per_launch_ms - ms_per_step, 0.0),
                         "algorithmic_bytes": bytes_per_launch,
                         "algorithmic_flops": flops_per_launch,
                         "mfma_tflops": flops_per_launch / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0},
            "cpu_baseline": cpu,
            "host_pointer_qps": host_qps,
            "certified_half_scan": shadow_leg,
        }
        if shadow_leg and "qps" in shadow_leg:
            shadow_leg["speedup_over_value"] = shadow_leg["qps"] / result["value"]
        extra_cfg["streams"] = len(lanes)
        if world > 1:
            extra_cfg["coarse_pass"] = "dealt over the ranks + all-gather of the probe lists" if args.deal_coarse else "replicated on every rank"
        extra_cfg.update({"recall_at_nprobe%d" % nprobe_base: recall_base, "qps_nprobe%d" % nprobe_base: qps_base if qps_base else
                          (result["value"] if nprobe == nprobe_base else None), "recall_queries": ngt})
        if shard_mode:
            extra_cfg["shard_of"] = args.shard_of

    if rank == 0:
        sharding = "single GPU"
        if world > 1:
            sharding = ("inverted lists dealt to %d ranks by bytes (largest first), one all-gather of the candidate lists" % world
                        if kind == "ivf" else "contiguous row ranges over %d ranks, one all-gather of the candidate lists" % world)
        elif shard_mode:
            sharding = "shard 0 of %d (byte-balanced list map), no exchange: one rank's share" % args.shard_of
        cfg = {"workload": "%s: %s n=%d dim=%d%s batch=%d k=%d" % (
            args.workload, "IVF-Flat L2" if kind == "ivf" else ("Flat IP" if args.workload == "flat1m" else "Flat L2"), n, dim,
            (" nlist=%d nprobe=%d" % (nlist, nprobe)) if kind == "ivf" else "", batch, topk),
            "recall_at_10": recall, "nprobe_timed": nprobe if kind == "ivf" else None, "sharding": sharding}
        cfg.update(extra_cfg)
        line = {
            "metric": "QPS @ recall@10>=0.99, 10Mx768 fp32, batch=1024" if args.workload == "ivf10m" else "QPS (%s)" % args.workload,
            "value": result["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": result["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if dtype == "fp32" else "f16 rows, f32 accumulate", "data": "synthetic",
            "config": cfg,
            "rccl_ranks": rccl["ranks"] if rccl else 1, "rccl": rccl,
            "roofline": result["roofline"], "cpu_baseline": result.get("cpu_baseline"),
            "host_pointer_qps": result.get("host_pointer_qps"),
        }
        if result.get("certified_half_scan"):
            line["certified_half_scan"] = result["certified_half_scan"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def committed_traffic(args, world, shard_mode):
    """HBM traffic of one launch from the PMC counters: rocprofv3 cannot run inside this process, so the value is the one
    measured by separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command (profiles/README.md) and
    corrected as MI355X_MICROARCH.md §HBM prescribes; null for any other configuration."""
    try:
        if world == 1 and not shard_mode and not args.n and args.batch in (0, None, 1) and not args.nprobe and not args.nlist:
            key = args.workload + ("_b1" if args.batch == 1 else "")      # (the single-query route has its own dominant kernel)
            v = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
            if v is not None:
                return v, "profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, not this run)"
    except (OSError, ValueError):
        pass
    return None, None


def roaring_portable(ids):
    """Synthetic-data helper: the bytes roaring_bitmap_portable_serialize would produce for a set of uint32 ids
    (RoaringFormatSpec, no run containers: cookie 12346, container count, (key, cardinality-1) pairs, offsets, then
    per 64K chunk either the sorted low 16 bits (<= 4096 ids) or a 8 KiB bitset)."""
    import struct
    ids = np.unique(np.asarray(ids, np.uint32))
    hi = (ids >> 16).astype(np.uint32)
    keys, starts = np.unique(hi, return_index=True)
    ends = np.append(starts[1:], ids.size)
    payloads = []
    for a_, b_ in zip(starts, ends):
        low = (ids[a_:b_] & 0xFFFF).astype(np.uint16)
        if low.size > 4096:
            bits = np.zeros(65536, np.uint8)
            bits[low] = 1
            payloads.append(np.packbits(bits, bitorder="little").tobytes())
        else:
            payloads.append(low.astype("<u2").tobytes())
    n = len(keys)
    out = bytearray(struct.pack("<II", 12346, n))
    for k_, a_, b_ in zip(keys, starts, ends):
        out += struct.pack("<HH", int(k_), int(b_ - a_ - 1))
    off = len(out) + 4 * n
    for p_ in payloads:
        out += struct.pack("<I", off)
        off += len(p_)
    for p_ in payloads:
        out += p_
    return bytes(out)


def parity_vs_cpu(gpu, ok, os_, oc, nq):
    """the oracle as the checker: the GPU's answers for the same queries at full size (ids as sets per query; scores of
    the common ids; the CPU sums in AVX-512 lane order, the GPU refines L2 scores directly: ~1e-6 relative apart)"""
    gk, gs, gc = gpu
    same = rel = 0.0
    for i in range(nq):
        a, b = set(gk[i, :gc[i]].astype(np.uint64).tolist()), set(ok[i, :oc[i]].tolist())
        same += len(a & b) / float(max(len(b), 1))
        cs = dict(zip(ok[i, :oc[i]].tolist(), os_[i, :oc[i]].tolist()))
        for key, sc in zip(gk[i, :gc[i]].astype(np.uint64).tolist(), gs[i, :gc[i]].tolist()):
            if key in cs:
                rel = max(rel, abs(sc - cs[key]) / max(abs(cs[key]), 1e-30))
    log("parity vs the CPU path: ids in common %.6f, max relative score difference %.3g" % (same / nq, rel))
    return {"queries": int(nq), "topk_ids_in_common": same / nq, "max_rel_score_diff": rel}


def cpu_baseline_flat(torch, base, q, topk, metric_name, args, gpu=None):
    """The reference's CPU flat scan on a bounded sample of the timed batch over the same base rows.  kind "reference": the
    reference's own FlatBuilder dumps the rows, its FlatSearcher searches them (one query per call and thread, as the product calls
    boundary B); kind "port" when that library did not travel: oracle loops + the reference's AVX-512 1x1 kernels."""
    from oracle import oracle as O
    from oracle import refcore as R
    o = O.get()
    host = base.cpu().numpy()
    threads = host_threads()
    cores = host_cores()
    nq = min(q.shape[0], args.cpu_queries or 2 * threads)
    qh = q[:nq].cpu().numpy()
    best = None
    if R.available() and not args.cpu_port:
        R.build("FlatBuilder", host, metric_name, "bench_flat")
        ref = R.Runner.searcher("FlatSearcher", "bench_flat", host.shape[1], host.dtype)
        for _ in range(2):
            ok, os_, oc, dt = ref.search_mt(qh, topk, threads)
            best = dt if best is None else min(best, dt)
        n1 = min(4, nq)
        one = n1 / ref.search_mt(qh[:n1], topk, 1)[3]
        phys = all_physical_cores_run(lambda t: ref.search_mt(q[:max(nq, min(q.shape[0], 4 * t))].cpu().numpy(), topk, t)[3],
                                      max(nq, min(q.shape[0], 4 * (cores.get("physical") or 0))), threads, cores)
        ref.close()
        R.mem_remove("bench_flat")
        kind, what = "reference", "the reference's own FlatBuilder + FlatSearcher::search_impl (libzvec_ref_core.so, -O2 -march=skylake-avx512)"
    else:
        used_ref = o.use_reference_kernels(True)
        metric = O.METRIC_IP if metric_name == "InnerProduct" else O.METRIC_L2
        for _ in range(2):
            t1 = time.perf_counter()
            ok, os_, _, oc = o.flat_search(host, qh, topk, metric, threads=threads)
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        o.use_reference_kernels(False)
        one = phys = None
        kind, what = "port", "scan loop = oracle restatement, 1x1 distance kernel = %s" % (
            "reference ailego AVX-512 (oracle/_ref)" if used_ref else "oracle C (-O3 -mavx2)")
    parity = parity_vs_cpu(gpu, ok, os_, oc, nq) if gpu is not None else None
    return {"value": nq / best, "unit": "queries/s", "cores": threads, "kind": kind, "value_1_thread": one, "parity": parity,
            "all_physical_cores": phys, "cpu_quota_cpus": cpu_quota_cpus(),
            "host_physical_cores": cores.get("physical"), "host_logical_cpus": cores.get("logical"), "host_usable_cpus": cores.get("usable"),
            "sample": "%d queries of the timed batch over the same %d rows, one query per call, %d threads across queries, best of 2; %s" % (
                nq, host.shape[0], threads, what)}


def run_flat(torch, dist, zvec_amd, sharded, flat, fctx, q, n, n_local, dim, topk, args, dev, stream_ptr, world, rank, base, metric_name, coll):
    batch = q.shape[0]
    excl = None
    doc_filter = None
    if args.workload.startswith("filter"):
        # BASELINE configs[4]: CRoaring bitmap predicate.  Bernoulli keep-mask p = --keep as the result
        # bitmap of an inverted-index condition (ids that MATCH, InvertedSearchResult) in roaring portable form; every
        # step materialises it on the GPU into the 1-bit-per-position exclude set (zvec_hip_flat_build_filter) and
        # runs the gated scan — both inside the timed region.  (N > 1: every rank draws the mask of the whole corpus and
        # hands its own row range's keys to the predicate; the bitmap is over KEYS = global row numbers.)
        g = torch.Generator(device=dev)
        g.manual_seed(SEED + 2)
        keep = (torch.rand((n,), generator=g, device=dev) < args.keep).cpu().numpy()
        blob = roaring_portable(np.nonzero(keep)[0])
        doc_filter = zvec_amd.DocFilter(invert=blob)
        excl = torch.zeros(((n_local + 63) // 64,), dtype=torch.int64, device=dev)
        log("filter: keep %.3f of %d rows; predicate = %d bytes of roaring" % (keep.mean(), n, len(blob)))

    thr = None if args.flat_threshold is None else args.flat_threshold

    # --streams 2 (default), one GPU, no predicate: consecutive (independent) batches alternate between two contexts on two HIP
    # streams, as the IVF workloads do — the small kernels around the scan (query preparation, the bound-seeding prefix scan and its
    # selection, the merge of the chunks' lists, the L2 refinement) of batch i+-1 run beside batch i's scan.  Every step is still one
    # complete pass over one batch and all K steps complete inside the timed region.
    lanes = [(sharded, fctx, stream_ptr, None)]
    if args.streams >= 2 and world == 1 and doc_filter is None:
        for _ in range(args.streams - 1):
            s2 = torch.cuda.Stream()
            c2 = flat.create_context()
            c2.set_stream(s2.cuda_stream)
            lanes.append((type(sharded)(flat, c2, rank, world), c2, s2.cuda_stream, s2))
    step_no = [0]

    def step():
        if doc_filter is not None:
            flat.build_filter(doc_filter, fctx, d_out=excl.data_ptr(), stream=stream_ptr)
        sh_i, _, sp, ts = lanes[step_no[0] % len(lanes)]
        step_no[0] += 1
        if ts is None:
            return sh_i.search(q, topk, sp, d_exclude=excl.data_ptr() if excl is not None else None, threshold=thr)
        with torch.cuda.stream(ts):
            return sh_i.search(q, topk, sp, d_exclude=excl.data_ptr() if excl is not None else None, threshold=thr)

    cpu = None
    if not args.no_cpu_baseline and world == 1 and doc_filter is None and base is not None:
        ok, os_, oc = step()
        torch.cuda.synchronize()
        cpu = cpu_baseline_flat(torch, base, q, topk, metric_name, args, gpu=(ok.cpu().numpy(), os_.cpu().numpy(), oc.cpu().numpy()))
    for i in range(args.warmup * len(lanes)):
        step()
        if i == 0:
            torch.cuda.synchronize()              # (the lanes must not START together: see the IVF lanes)
    torch.cuda.synchronize()
    step_no[0] = 0
    for _, c_l, _, _ in lanes:
        c_l.profile(True)
        c_l.profile_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        coll(dist.all_reduce, t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {"scan_ms": 0.0, "launches": 0, "bytes": 0.0, "flops": 0.0}
    per_lane_ms = []
    for _, c_l, _, _ in lanes:                  # EVERY lane's scans are timed (HIP events on the lane's own stream)
        p1 = c_l.profile_read(reset=True)
        for key in prof:
            prof[key] += p1[key]
        per_lane_ms.append(p1["scan_ms"] / max(p1["launches"], 1))
    ms = prof["scan_ms"] / max(prof["launches"], 1)
    fl = prof["flops"] / max(prof["launches"], 1)
    by = prof["bytes"] / max(prof["launches"], 1)
    traffic, traffic_source = committed_traffic(args, world, False)
    ms_per_step = elapsed / args.steps * 1e3
    # overlapping lanes: a lane's scan is dispatched while the other lane's still holds the CUs, so its own event time includes that
    # wait; a launch cannot take longer than a step in steady state, so the roofline then divides by the STEP time (cannot understate)
    frac_ms, frac_basis = ms, "kernel_ms (HIP events around the launch, mean over every lane's launches)"
    if 0 < ms_per_step < ms:
        frac_ms, frac_basis = ms_per_step, "ms_per_step (overlapping lanes: kernel_ms > ms_per_step, see kernel_ms_per_lane)"
    tf = fl / (frac_ms * 1e-3) / 1e12 if frac_ms > 0 else 0.0
    f16 = args.workload.endswith("fp16")
    small = batch <= 16         # a handful of queries: the scan streams the base once => HBM-bound
    peak = MFMA_F16_PEAK_TF if f16 else MFMA_F32_PEAK_TF
    box = box_calibration(zvec_amd, torch.cuda.current_device())
    # which wide kernel the dispatch takes (api_flat_scan.inc.h: fp16 rows, >= 256 queries, k <= 11, unfiltered, option scan256 on)
    import ctypes as C
    opt256 = C.c_int(0)
    zvec_amd._lib.lib().zvec_hip_get_option(b"scan256", C.byref(opt256))
    on256 = f16 and opt256.value != 0 and batch >= 256 and topk <= 11 and doc_filter is None and dim > 64
    kname = "zvk::scan256_f16_kernel (flat scan, 256 x 256 multi-phase tile)" if on256 else "zvk::scan8_kernel (flat scan)"
    roof = {"bound": "mfma", "kernel": kname, "achieved": tf, "peak": peak,
            "unit": "TFLOP/s", "frac": tf / peak, "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": ms,
            "frac_basis": frac_basis, "kernel_ms_per_lane": per_lane_ms, "lanes": len(lanes),
            "box_clock_mhz": box.get("clock_mhz"), "box_stream_gbs": box.get("stream_gbs"), "box_note": box.get("note"),
            "fixed_ms_per_step": max(ms_per_step - ms, 0.0), "algorithmic_bytes": by, "algorithmic_flops": fl,
            "hbm_gbs": by / (frac_ms * 1e-3) / 1e9 if frac_ms > 0 else 0.0}
    if small:
        gbs = by / (frac_ms * 1e-3) / 1e9 if frac_ms > 0 else 0.0
        roof.update({"bound": "hbm", "kernel": "zvk::scan_kernel<1, M16> (flat scan, <= 16 queries)", "achieved": gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "mfma_tflops": tf})
    # ---------------- second measurement: the same steps through the half-width pre-selection (zvec_hip_flat_set_shadow) ----------------
    shadow_leg = None
    if not f16 and world == 1 and thr is None and not args.no_shadow_leg:
        try:
            ex_ptr = excl.data_ptr() if excl is not None else None      # (filter workloads: the exclude set every step materialises)
            if doc_filter is not None:
                flat.build_filter(doc_filter, fctx, d_out=excl.data_ptr(), stream=stream_ptr)
            ref32 = [x.clone() for x in lanes[0][0].search(q, topk, lanes[0][2], d_exclude=ex_ptr, threshold=None)]
            torch.cuda.synchronize()
            t1 = time.time()
            flat.set_shadow(True, args.shadow_preselect)
            sinfo = flat.shadow_info()
            log("shadow rows: %.2f GB in %.2fs; max |b - b16| %.4g, max |b16| %.4g" % (sinfo["bytes"] / 1e9, time.time() - t1,
                                                                                     sinfo["max_row_error"], sinfo["max_row_norm"]))
            rer = [0]
            pending = [None] * len(lanes)
            for sh_l, _, _, _ in lanes:                  # (the lanes certify a step when they come back to it, not inside search())
                sh_l.defer_certify = True

            def certify(pend):
                c_l, k_, s_, c_, sp = pend
                return flat.shadow_certify(q.data_ptr(), batch, topk, k_.data_ptr(), s_.data_ptr(), c_.data_ptr(), c_l, d_exclude=ex_ptr, stream=sp)

            def step_shadow(i):
                li = i % len(lanes)
                sh_i, c_l, sp, ts = lanes[li]
                if pending[li] is not None:
                    rer[0] += certify(pending[li])
                if doc_filter is not None:           # (as step(): the predicate is materialised inside the timed region)
                    flat.build_filter(doc_filter, fctx, d_out=excl.data_ptr(), stream=stream_ptr)
                if ts is None:
                    k_, s_, c_ = sh_i.search(q, topk, sp, d_exclude=ex_ptr, threshold=None)
                else:
                    with torch.cuda.stream(ts):
                        k_, s_, c_ = sh_i.search(q, topk, sp, d_exclude=ex_ptr, threshold=None)
                pending[li] = (c_l, k_, s_, c_, sp)

            def drain():
                for li in range(len(lanes)):
                    if pending[li] is not None:
                        rer[0] += certify(pending[li])
                        pending[li] = None
            step_shadow(0)
            drain()
            torch.cuda.synchronize()
            k1, s1, c1 = lanes[0][0].search(q, topk, lanes[0][2], d_exclude=ex_ptr, threshold=None)
            parity_rerun = certify((lanes[0][1], k1, s1, c1, lanes[0][2]))
            torch.cuda.synchronize()
            same_ids = int((ref32[0] == k1).all(1).sum().item())
            same_bits = int((ref32[1].view(torch.int32) == s1.view(torch.int32)).all(1).sum().item())
            close = int(((ref32[1] - s1).abs() <= 4e-6 * dim * 8).all(1).sum().item())
            log("shadow route vs fp32 route: %d / %d queries with identical key lists, %d with identical score bits (%d within fp32 rounding), %d re-run in fp32" % (
                same_ids, batch, same_bits, close, parity_rerun))
            for i in range(args.warmup * len(lanes)):
                step_shadow(i)
                if i == 0:
                    torch.cuda.synchronize()
            drain()
            torch.cuda.synchronize()
            for _, c_l, _, _ in lanes:
                c_l.profile(True)
                c_l.profile_read(reset=True)
            rer[0] = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step_shadow(i)
            drain()
            torch.cuda.synchronize()
            el_s = time.perf_counter() - t0
            spr = {"scan_ms": 0.0, "launches": 0, "bytes": 0.0, "flops": 0.0}
            for _, c_l, _, _ in lanes:
                p1 = c_l.profile_read(reset=True)
                c_l.profile(False)
                for key in spr:
                    spr[key] += p1[key]
            s_ms = el_s / args.steps * 1e3
            s_kernel = spr["scan_ms"] / max(spr["launches"], 1)
            s_basis = min(s_kernel, s_ms) if s_kernel > 0 else s_ms
            s_fl = spr["flops"] / max(spr["launches"], 1)
            s_by = spr["bytes"] / max(spr["launches"], 1)
            shadow_leg = {
                "qps": batch * args.steps / el_s, "ms_per_step": s_ms, "unit": "queries/s", "speedup_over_value": (batch * args.steps / el_s) / (batch * args.steps / elapsed),
                "what": "the same steps with zvec_hip_flat_set_shadow on: scan over an fp16 twin of the rows for k' rows per query, fp32 re-scoring, "
                        "per-query certificate from the measured rounding, fp32 re-run of uncertified queries inside the timed region",
                "preselect": args.shadow_preselect or "chosen by the index (from max(32, 3k): narrowed after clean certify steps, widened after re-runs)",
            "preselect_rows_at_end": flat.shadow_width(topk), "shadow_bytes": sinfo["bytes"], "max_row_rounding": sinfo["max_row_error"],
                "max_row_norm": sinfo["max_row_norm"], "rerun_queries_per_step": rer[0] / float(args.steps),
                "parity_vs_fp32_route": {"queries": batch, "identical_key_lists": same_ids, "identical_score_bits": same_bits,
                                         "scores_within_fp32_rounding": close, "rerun_in_fp32": parity_rerun},
                "kernel": "the flat scan over the fp16 shadow rows (zvk::scan8_kernel fp16 for wide batches, zvk::scan_kernel<1, M16> for a handful of queries)",
                "kernel_ms": s_kernel, "algorithmic_bytes": s_by, "algorithmic_flops": s_fl,
                "mfma_tflops": s_fl / (s_basis * 1e-3) / 1e12 if s_basis > 0 else None, "frac_of_f16_peak": s_fl / (s_basis * 1e-3) / 1e12 / MFMA_F16_PEAK_TF if s_basis > 0 else None,
                "hbm_gbs": s_by / (s_basis * 1e-3) / 1e9 if s_basis > 0 else None,
            }
            flat.set_shadow(False)
            for sh_l, _, _, _ in lanes:
                sh_l.defer_certify = False
            log("certified half-width scan: %.0f QPS (%.3f ms per step, scan %.3f ms, %.2f queries re-run per step)" % (
                shadow_leg["qps"], s_ms, s_kernel, shadow_leg["rerun_queries_per_step"]))
        except Exception as e:      # (the second measurement must never cost the line its first)
            log("certified half-width leg failed: %r" % (e,))
            shadow_leg = {"error": repr(e)}
            try:
                flat.set_shadow(False)
            except Exception:
                pass
    return {"value": batch * args.steps / elapsed, "ms_per_step": ms_per_step, "roofline": roof, "cpu_baseline": cpu, "certified_half_scan": shadow_leg}


def box_calibration(zvec_amd, device):
    """Per-box calibration in the same process as the timed region (zvec_hip_calibrate): what this box's HBM gives a pure streaming
    reader (non-temporal 16-byte loads over 30 GB of scratch, best of 3) and the shader clock held under that load — so that 215 k
    on one box and 205 k on another can be read as box or as code."""
    import ctypes as C
    try:
        import torch
        free, _ = torch.cuda.mem_get_info(device)
        nbytes = int(min(30e9, free * 0.8)) // 4096 * 4096
        mhz, gbs = C.c_double(0), C.c_double(0)
        rc = zvec_amd._lib.lib().zvec_hip_calibrate(device, None, nbytes, 3, C.byref(mhz), C.byref(gbs))
        if rc != 0:
            return {"note": "zvec_hip_calibrate rc %d" % rc}
        log("box calibration: streaming read %.0f GB/s over %.1f GB, shader clock %.0f MHz under that load" % (gbs.value, nbytes / 1e9, mhz.value))
        return {"clock_mhz": mhz.value, "stream_gbs": gbs.value,
                "note": "pure non-temporal streaming read of %.1f GB of scratch HBM in this process right after the timed region, best of 3; "
                        "clock = d(s_memtime)/d(s_memrealtime) x 100 MHz, median over work-groups, under that load" % (nbytes / 1e9)}
    except Exception as e:                               # noqa: BLE001 - calibration is a courtesy, never a reason to lose the line
        return {"note": "calibration failed: %r" % (e,)}


def cpu_quota_cpus():
    """CPUs the cgroup lets this process burn per wall second (cpu.max), or None"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def all_physical_cores_run(run, nq, threads_used, cores):
    """BASELINE.md §2 / tools/core/bench.cc:30-45: the CPU path at T = all physical cores of the node, next to T = 1 and T = the
    box's share.  Run whenever the process may be scheduled on at least that many CPUs; the cgroup's CPU quota (cpu.max), when
    there is one, still caps what those threads can burn per second — it is reported beside the figure, so the number reads as
    "T physical-core threads under a Q-CPU quota", not as the node's unrestricted throughput."""
    phys, usable = cores.get("physical") or 0, cores.get("usable") or 0
    if phys <= threads_used or usable < phys:
        return None
    try:
        dt = run(phys)
        return {"threads": phys, "value": nq / dt, "unit": "queries/s", "queries": int(nq), "cpu_quota_cpus": cpu_quota_cpus(),
                "note": "same leg at T = physical cores (%d); the cgroup quota above is what the threads can actually use" % phys}
    except Exception as e:                               # noqa: BLE001
        return {"threads": phys, "error": repr(e)}


def host_threads():
    """cores this process may actually use: affinity mask, cgroup quota, and the GPU box's per-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("ZVEC_BENCH_CPU_THREADS", "16"))))


def host_cores():
    """what the node has (BASELINE.md §2 asks for the physical core count next to the threads used): lscpu's sockets x cores per
    socket, its logical CPUs, and the CPUs this process may run on"""
    info = {"logical": os.cpu_count() or 0, "usable": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0}
    try:
        import subprocess
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        f = {}
        for line in txt.splitlines():
            if ":" in line:
                a, b = line.split(":", 1)
                f[a.strip()] = b.strip()
        info["physical"] = int(f.get("Socket(s)", "1")) * int(f.get("Core(s) per socket", "0"))
        info["model"] = f.get("Model name", "")
    except Exception:                                    # noqa: BLE001 - lscpu missing: the counts above still stand
        info["physical"] = 0
    return info


def cpu_baseline_ivf(torch, ivf, q, topk, nprobe, max_scan, args, dtype, gpu=None):
    """The reference's CPU path on the host cores of this box, searching THE SAME index (exported centroids / list order, rows
    read back from the HBM store) with the same queries, parallel across queries, one query per call and thread at a time
    (tools/core/bench.cc:145-245; the product calls boundary B with count = 1, index.cc:617).
    kind "reference": the reference's OWN IVFSearcher (oracle/_ref/libzvec_ref_core.so = its core library compiled in place),
    opened over the exported arrays — its IVFDumper writes every small segment, the 30 GB body is lent, not copied
    (oracle/ref_core_shim.cc zref_ivf_searcher_over_rows).  kind "port" (only when that library did not travel or the element
    size is not a multiple of 32 bytes): the oracle's restated loops with the reference's AVX-512 1x1 kernels."""
    from oracle import oracle as O
    from oracle import refcore as R
    o = O.get()
    t0 = time.time()
    cent, offs, rows = ivf.export()
    n, dim = rows.shape[0], cent.shape[1]
    vecs = np.empty((n, dim), np.float16 if dtype == "fp16" else np.float32)
    step = 1 << 20
    for s in range(0, n, step):
        e = min(n, s + step)
        vecs[s:e] = ivf.get_vectors_by_ids(np.arange(s, e, dtype=np.uint64))
    qh = q.cpu().numpy()
    nq = args.cpu_queries or qh.shape[0]
    qh = qh[:nq]
    threads = host_threads()
    cores = host_cores()
    log("cpu baseline: index copied to host in %.1fs; %d queries on %d threads (host: %s physical cores, %s logical, %s usable)" % (
        time.time() - t0, nq, threads, cores.get("physical"), cores.get("logical"), cores.get("usable")))
    ref = None
    if R.available() and (vecs.shape[1] * vecs.dtype.itemsize) % 32 == 0 and not args.cpu_port:
        t1 = time.time()
        nlist = cent.shape[0]
        ratio = float(np.float32(nprobe) / np.float32(max(nlist, 1)))          # ivf_searcher_context.h:70-78, see set_nprobe
        # brute_force_threshold doubles as the max_scan_count floor (ivf_searcher_context.h:70-78) and, when >= N, sends the whole
        # search to brute force (ivf_searcher.cc:183-185): N - 1 keeps the probe walk and never cuts it short
        params = {"proxima.ivf.searcher.scan_ratio": ratio, "proxima.ivf.searcher.brute_force_threshold": int(max(min(max_scan, n - 1), 1))}
        ref = R.Runner.ivf_over_rows("IVFSearcher", cent, offs, vecs, rows.astype(np.uint64), "SquaredEuclidean", params=params)
        log("cpu baseline: the reference's IVFSearcher opened over the exported index in %.1fs" % (time.time() - t1))
    best = None
    reps = 0
    t_all = time.time()
    if ref is not None:
        while reps < 3 and (time.time() - t_all) < 25.0:
            ok, os_, oc, dt = ref.search_mt(qh, topk, threads)
            best = dt if best is None else min(best, dt)
            reps += 1
        n1 = min(32, nq)
        _, _, _, d1 = ref.search_mt(qh[:n1], topk, 1)
        one = n1 / d1
        phys = all_physical_cores_run(lambda t: ref.search_mt(qh, topk, t)[3], nq, threads, cores)
        ref.close()
        kind, what = "reference", "the reference's own IVFSearcher::search_impl (libzvec_ref_core.so, -O2 -march=skylake-avx512)"
    else:
        used_ref = o.use_reference_kernels(True)
        while reps < 3 and (time.time() - t_all) < 25.0:
            t1 = time.perf_counter()
            ok, os_, _, oc, _ = o.ivf_search(cent, offs, vecs, qh, topk, nprobe, max_scan, keys=rows, threads=threads)
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
            reps += 1
        # single-thread figure as well (report T = 1 and T = all cores), on a 32-query sample
        n1 = min(32, nq)
        t1 = time.perf_counter()
        o.ivf_search(cent, offs, vecs, qh[:n1], topk, nprobe, max_scan, keys=rows, threads=1)
        one = n1 / (time.perf_counter() - t1)
        o.use_reference_kernels(False)
        phys = None
        kind, what = "port", "scan loops = oracle restatement, 1x1 distance kernel = %s" % (
            "reference ailego AVX-512 (oracle/_ref)" if used_ref else "oracle C (-O3 -mavx2)")
    parity = parity_vs_cpu(gpu, ok, os_, oc, nq) if gpu is not None else None
    return {"value": nq / best, "unit": "queries/s", "cores": threads, "kind": kind, "value_1_thread": one, "parity": parity,
            "all_physical_cores": phys, "cpu_quota_cpus": cpu_quota_cpus(),
            "host_physical_cores": cores.get("physical"), "host_logical_cpus": cores.get("logical"), "host_usable_cpus": cores.get("usable"),
            "sample": "%d queries of the timed batch, same IVF index (exported), one query per call, %d threads across queries "
                      "(the box's share of its host; capped by ZVEC_BENCH_CPU_THREADS), best of %d; %s" % (nq, threads, reps, what)}


if __name__ == "__main__":
    main()
