"""IVF search time per batch size with the small-batch (direct) route off / on: run with the tuning library
(ZVEC_HIP_LIBRARY=zvec_amd/_variants/libzvec_hip_tune.so) and ZVEC_HIP_IVF_DIRECT_Q=0 | 1000"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zvec_amd as zv
from zvec_amd import _lib
from zvec_amd.index import _np_ptr, FLT_MAX

rng = np.random.default_rng(5)
n, dim, nlist = 2_000_000, 768, 2048
ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean")
base = rng.standard_normal((n, dim)).astype(np.float32)
assert ivf.build(base, nlist, kmeans_iters=4) == 0
_, offs, _ = ivf.export()
sizes = np.diff(offs.astype(np.int64))
print("list sizes: mean %.0f max %d, 32 largest %d rows" % (sizes.mean(), sizes.max(), np.sort(sizes)[-32:].sum()), flush=True)
del base
ivf.set_nprobe(32)
ctx = ivf.create_context()
L = _lib.lib()
nprobe, max_scan = ivf.probe_params()
q = rng.standard_normal((256, dim)).astype(np.float32)
out = []
for b in (1, 2, 4, 8, 16, 32, 64, 128):
    keys, scores, counts = np.zeros((b, 10), np.uint64), np.zeros((b, 10), np.float32), np.zeros(b, np.uint32)
    def call(o):
        L.zvec_hip_ivf_search(ivf._h, ctx._h, _np_ptr(q[o:o + b]), b, 10, FLT_MAX, nprobe, max_scan, None, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
    for w in range(5):
        call(0)
    R = 200 if b <= 16 else 60
    t0 = time.perf_counter()
    for i in range(R):
        call((i * b) % (256 - b + 1))
    out.append("b%d %.0fus" % (b, (time.perf_counter() - t0) / R * 1e6))
print("direct_q=%s : %s" % (os.environ.get("ZVEC_HIP_IVF_DIRECT_Q", "default"), "  ".join(out)), flush=True)
