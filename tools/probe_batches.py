#!/usr/bin/env python3
"""Times zvec_hip_ivf_search (host pointers) at several batch sizes on a synthetic index.
Usage (GPU box): python tools/probe_batches.py [n] [dim] [nlist] [scan_ratio]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import zvec_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_024_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nlist = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ratio = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
base = torch.randn((n, dim), generator=g, device=dev)
ivf = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=ratio, brute_force_threshold=10)
zvec_amd._lib.check(ivf.build_dev(base.data_ptr(), n, nlist, kmeans_iters=4), "build")
ivf.total_count = n
nprobe, max_scan = ivf.probe_params()
qh = torch.randn((1024, dim), generator=g, device=dev).cpu().numpy()
L = zvec_amd._lib.lib()
ctx = ivf.create_context()
for b in (1, 8, 32, 64, 256, 1024):
    k = np.zeros((b, 10), np.uint64)
    s = np.zeros((b, 10), np.float32)
    c = np.zeros(b, np.uint32)
    q = np.ascontiguousarray(qh[:b])

    def call():
        rc = L.zvec_hip_ivf_search(ivf._h, ctx._h, q.ctypes.data, b, 10, 3.4028234663852886e38, nprobe, max_scan, None,
                                   k.ctypes.data, s.ctypes.data, c.ctypes.data)
        assert rc == 0
    for _ in range(3):
        call()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    print("batch %4d: %.3f ms per call, %.0f QPS (nprobe %d of %d lists)" % (b, dt * 1e3, b / dt, nprobe, nlist), flush=True)
