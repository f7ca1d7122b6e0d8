"""times a group-by search against the plain top-k search of the same batch through the C ABI (1M x 768 fp32, 256 queries,
host pointers), so that Python result objects are not in the numbers"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zvec_amd as zv
from zvec_amd import _lib
from zvec_amd.index import _np_ptr, FLT_MAX

rng = np.random.default_rng(3)
n, dim, nq = 1_000_000, 768, 256
st = zv.HipFlatStreamer(dim, "InnerProduct")
for o in range(0, n, 100_000):
    st.add_batch(rng.standard_normal((100_000, dim)).astype(np.float32), np.arange(o, o + 100_000, dtype=np.uint64))
q = rng.standard_normal((nq, dim)).astype(np.float32)
ctx = st.create_context()
L = _lib.lib()
k = 10
keys, scores, counts = np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32)


def plain():
    _lib.check(L.zvec_hip_flat_search(st._h, ctx._h, _np_ptr(q), nq, k, FLT_MAX, None, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts)), "search")


for ngroups, gnum, gk in ((1000, 10, 10), (100, 5, 20), (50000, 20, 3)):
    of = (np.arange(n) % ngroups).astype(np.uint32)
    groups, ngr = np.zeros((nq, gnum), np.uint32), np.zeros(nq, np.uint32)
    gkeys, gscores, gcounts = np.zeros((nq, gnum, gk), np.uint64), np.zeros((nq, gnum, gk), np.float32), np.zeros((nq, gnum), np.uint32)

    def grouped():
        _lib.check(L.zvec_hip_flat_search_grouped(st._h, ctx._h, _np_ptr(q), nq, _np_ptr(of), ngroups, gnum, gk, FLT_MAX, None, _np_ptr(groups),
                                                  _np_ptr(ngr), _np_ptr(gkeys), _np_ptr(gscores), _np_ptr(gcounts)), "grouped")

    for name, fn in (("plain top-10", plain), ("group-by %d x %d of %d groups" % (gnum, gk, ngroups), grouped)):
        fn()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        print("%-36s %.2f ms per 256-query call" % (name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
