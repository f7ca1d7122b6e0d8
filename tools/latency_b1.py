"""single-query latency of the host-pointer IVF search (what the product's count = 1 calls cost): per-call wall time and
the GPU-side share (sum of kernel durations is read from a rocprofv3 run of this script)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zvec_amd as zv

rng = np.random.default_rng(5)
n, dim, nlist = 2_000_000, 768, 2048
base = rng.standard_normal((n, dim)).astype(np.float32)
ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean")
assert ivf.build(base, nlist, kmeans_iters=4) == 0
ivf.set_nprobe(32)
q = rng.standard_normal((64, dim)).astype(np.float32)
ctx = ivf.create_context()
ctx.set_topk(10)
from zvec_amd import _lib
from zvec_amd.index import _np_ptr, FLT_MAX
L = _lib.lib()
keys, scores, counts = np.zeros((1, 10), np.uint64), np.zeros((1, 10), np.float32), np.zeros(1, np.uint32)
nprobe, max_scan = ivf.probe_params()
for w in range(20):
    L.zvec_hip_ivf_search(ivf._h, ctx._h, _np_ptr(q[w % 64:w % 64 + 1]), 1, 10, FLT_MAX, nprobe, max_scan, None, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
t0 = time.perf_counter()
R = 2000
for i in range(R):
    L.zvec_hip_ivf_search(ivf._h, ctx._h, _np_ptr(q[i % 64:i % 64 + 1]), 1, 10, FLT_MAX, nprobe, max_scan, None, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
dt = (time.perf_counter() - t0) / R
print("ivf 2M x 768, nprobe %d: %.1f us per single-query call (%.0f calls/s)" % (nprobe, dt * 1e6, 1 / dt))
