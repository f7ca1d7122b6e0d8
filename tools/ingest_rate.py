"""documents per second through the single-row add entries (how the product ingests: one add_with_id_impl per document)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zvec_amd as zv
from zvec_amd import _lib
from zvec_amd.index import _np_ptr

dim = 768
rng = np.random.default_rng(1)
rows = rng.standard_normal((20000, dim)).astype(np.float32)
L = _lib.lib()
# the cost of the Python call itself (slicing, pointer conversion, ctypes): the same loop around a trivial entry
import ctypes as C
st0 = zv.HipFlatStreamer(dim, "SquaredEuclidean")
cnt = C.c_uint64(0)
keys0 = np.arange(20000, dtype=np.uint64)
t0 = time.perf_counter()
for i in range(20000):
    a, b = _np_ptr(rows[i:i + 1]), _np_ptr(keys0[i:i + 1])
    L.zvec_hip_flat_count(st0._h, C.byref(cnt))
py = (time.perf_counter() - t0) / 20000
print("python loop overhead           %.1f us per iteration" % (py * 1e6), flush=True)
for name in ("append (add_impl)", "put (add_with_id_impl)"):
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    st.reserve(40000)
    keys = np.arange(20000, dtype=np.uint64)
    ids = np.arange(20000, dtype=np.uint32)
    t0 = time.perf_counter()
    for i in range(20000):
        if name.startswith("append"):
            rc = L.zvec_hip_flat_append(st._h, _np_ptr(rows[i:i + 1]), 1, _np_ptr(keys[i:i + 1]))
        else:
            rc = L.zvec_hip_flat_put(st._h, _np_ptr(ids[i:i + 1]), 1, _np_ptr(rows[i:i + 1]), None)
        assert rc == 0
    dt = time.perf_counter() - t0
    print("%-26s %.1f us per document (%.0f documents/s)" % (name, dt / 20000 * 1e6, 20000 / dt), flush=True)
    ctx = st.create_context(); ctx.set_topk(1)
    assert st.search_impl(rows[777:778], 1, ctx) == 0 and ctx.result(0)[0].key() == 777
