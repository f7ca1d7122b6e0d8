"""does the HBM-bound IVF list scan need all 256 CUs?  The same 1024-query search on streams created with
hipExtStreamCreateWithCUMask for 256 / 240 / 224 / 192 / 128 CUs (bits dealt evenly over the 8 XCDs)."""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zvec_amd as zv

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
n, dim, nlist = 2_000_000, 768, 2048
g = torch.Generator(device=dev); g.manual_seed(1)
base = torch.randn((n, dim), generator=g, device=dev)
ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean")
assert ivf.build_dev(base.data_ptr(), n, nlist, kmeans_iters=3) == 0
del base
ivf.set_nprobe(32)
q = torch.randn((1024, dim), generator=g, device=dev)
nprobe, max_scan = ivf.probe_params()
keys = torch.empty((1024, 10), dtype=torch.int64, device=dev)
scores = torch.empty((1024, 10), dtype=torch.float32, device=dev)
counts = torch.empty((1024,), dtype=torch.int32, device=dev)
from zvec_amd import _lib
L = _lib.lib()
for ncu in (256, 240, 224, 192, 128):
    per_xcd = ncu // 8                      # CUs enabled in each XCD (32 per XCD)
    words = (C.c_uint32 * 8)()
    # the mask is a bit per CU in the runtime's CU numbering (XCDs interleaved): enable the first per_xcd of every 32
    bits = 0
    for cu in range(256):
        if (cu // 8) < per_xcd:             # cu % 8 = XCD, cu // 8 = index inside the XCD
            bits |= 1 << cu
    for w in range(8):
        words[w] = (bits >> (32 * w)) & 0xffffffff
    stream = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), 8, words)
    assert rc == 0, rc
    ctx = ivf.create_context()
    ctx.set_stream(stream.value)
    ctx.profile(True)
    def call():
        rc = L.zvec_hip_ivf_search_dev(ivf._h, ctx._h, C.c_void_p(q.data_ptr()), 1024, 10, C.c_float(3.4e38), nprobe, max_scan, None,
                                       C.c_void_p(keys.data_ptr()), C.c_void_p(scores.data_ptr()), C.c_void_p(counts.data_ptr()), stream)
        assert rc == 0, rc
    for w in range(3):
        call()
    hip.hipStreamSynchronize(stream)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for i in range(20):
        call()
    hip.hipStreamSynchronize(stream)
    dt = (time.perf_counter() - t0) / 20
    pr = ctx.profile_read(reset=True)
    print("%3d CUs: %.3f ms per search, list scan %.3f ms (%.2f TB/s)" % (ncu, dt * 1e3, pr["scan_ms"] / max(pr["launches"], 1),
          pr["bytes"] / max(pr["launches"], 1) / (pr["scan_ms"] / max(pr["launches"], 1) * 1e-3) / 1e12), flush=True)
