"""single-query latency of the host-pointer flat search for several index sizes (the product's count = 1 calls)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zvec_amd as zv
from zvec_amd import _lib
from zvec_amd.index import _np_ptr, FLT_MAX

rng = np.random.default_rng(5)
dim = 768
L = _lib.lib()
q = rng.standard_normal((64, dim)).astype(np.float32)
keys, scores, counts = np.zeros((1, 10), np.uint64), np.zeros((1, 10), np.float32), np.zeros(1, np.uint32)
for n in (10_000, 100_000, 1_000_000):
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    for o in range(0, n, 100_000):
        m = min(100_000, n - o)
        st.add_batch(rng.standard_normal((m, dim)).astype(np.float32), np.arange(o, o + m, dtype=np.uint64))
    ctx = st.create_context()
    for w in range(10):
        L.zvec_hip_flat_search(st._h, ctx._h, _np_ptr(q[w:w + 1]), 1, 10, FLT_MAX, None, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
    R = 500
    t0 = time.perf_counter()
    for i in range(R):
        L.zvec_hip_flat_search(st._h, ctx._h, _np_ptr(q[i % 64:i % 64 + 1]), 1, 10, FLT_MAX, None, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
    dt = (time.perf_counter() - t0) / R
    print("flat %8d x %d: %.1f us per single-query call = %.2f TB/s of rows" % (n, dim, dt * 1e6, n * dim * 4 / dt / 1e12), flush=True)
