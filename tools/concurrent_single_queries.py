#!/usr/bin/env python3
"""Product-style load: T host threads, each with its own context, each issuing single-query searches (count = 1, host
pointers) against one shared IVF index — the way zvec's Collection.query() drives boundary B (index.cc:617).
Prints queries/s for T = 1, 2, 4, 8, 16.  Usage (GPU box): python tools/concurrent_single_queries.py [n] [nlist]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import zvec_amd  # noqa: E402
from bench import corpus_proj, gen_corpus, SEED  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dim, topk, nprobe = 768, 10, 38
dev = torch.device("cuda:0")
proj = corpus_proj(torch, dim, dev, 12)
base = gen_corpus(torch, n, dim, dev, SEED, proj, torch.float32)
queries = gen_corpus(torch, 4096, dim, dev, SEED + 1, proj, torch.float32)
ivf = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=(nprobe + 0.25) / nlist, brute_force_threshold=n - 1)
zvec_amd._lib.check(ivf.build_dev(base.data_ptr(), n, nlist, kmeans_iters=10, seed=SEED), "build")
ivf.total_count = n
del base
torch.cuda.synchronize()
qh = queries.cpu().numpy()
L = zvec_amd._lib.lib()


_ctxs = {}


def worker(t, per, out):
    ctx = _ctxs.get(t)            # one context per caller thread, kept across the warm-up and the timed run (index.cc:24-45)
    if ctx is None:
        ctx = _ctxs[t] = ivf.create_context()
    k = np.zeros((1, topk), np.uint64)
    s = np.zeros((1, topk), np.float32)
    c = np.zeros(1, np.uint32)
    for i in range(per):
        q = qh[(t * per + i) % qh.shape[0]]
        rc = L.zvec_hip_ivf_search(ivf._h, ctx._h, q.ctypes.data, 1, topk, 3.4028234663852886e38, nprobe, n - 1, None,
                                   k.ctypes.data, s.ctypes.data, c.ctypes.data)
        assert rc == 0 and c[0] == topk
    out[t] = per


from zvec_amd.batcher import MicroBatcher  # noqa: E402
import collections  # noqa: E402

_pool = collections.deque()
_pool_mu = threading.Lock()


def run_batch(qs, k_):
    with _pool_mu:
        ctx = _pool.pop() if _pool else None
    if ctx is None:
        ctx = ivf.create_context()
    m = qs.shape[0]
    k = np.zeros((m, k_), np.uint64)
    s = np.zeros((m, k_), np.float32)
    c = np.zeros(m, np.uint32)
    rc = L.zvec_hip_ivf_search(ivf._h, ctx._h, qs.ctypes.data, m, k_, 3.4028234663852886e38, nprobe, n - 1, None,
                               k.ctypes.data, s.ctypes.data, c.ctypes.data)
    with _pool_mu:
        _pool.append(ctx)
    assert rc == 0
    return k, s, c


mb = MicroBatcher(run_batch, dim, np.float32, max_batch=1024, window_us=3000, linger_us=int(os.environ.get("LINGER_US", "0")))


def worker_b(t, per, out):
    for i in range(per):
        keys, scores = mb.search(qh[(t * per + i) % qh.shape[0]], topk)
        assert keys.size == topk
    out[t] = per


for T in (16, 64, 256):
    per = 200
    out = [0] * T
    ths = [threading.Thread(target=worker_b, args=(t, 20, out)) for t in range(T)]
    [x.start() for x in ths]
    [x.join() for x in ths]
    ths = [threading.Thread(target=worker_b, args=(t, per, out)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in ths]
    [x.join() for x in ths]
    dt = time.perf_counter() - t0
    print("micro-batched, threads %3d: %8.0f single-query searches/s  (%.3f ms per search per thread)" % (T, T * per / dt, dt / per * 1e3), flush=True)

for T in (1, 2, 4, 8, 16, 32, 64):
    per = 1000 if T <= 16 else 300
    out = [0] * T
    ths = [threading.Thread(target=worker, args=(t, 50, out)) for t in range(T)]      # warm-up
    [x.start() for x in ths]
    [x.join() for x in ths]
    ths = [threading.Thread(target=worker, args=(t, per, out)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in ths]
    [x.join() for x in ths]
    dt = time.perf_counter() - t0
    print("threads %2d: %8.0f single-query searches/s  (%.3f ms per search per thread)" % (T, T * per / dt, dt / per * 1e3), flush=True)
