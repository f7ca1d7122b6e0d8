"""Where the end of a pair of base tiles goes in the 256 x 256 flat scan (diagnostic build: tools/build_variant.sh stamps256
-DZVK_A256_STAMPS): shader-clock cycles of wave 0 of work-group 0 per pair, by section.  usage: flat_stamps.py [fp16|fp32] [variant]  (fp32 rows take the 128 x 128 kernel: nothing is stamped there)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dt = sys.argv[1] if len(sys.argv) > 1 else "fp16"
variant = sys.argv[2] if len(sys.argv) > 2 else "stamps256"
os.environ.setdefault("ZVEC_HIP_LIBRARY", os.path.join(ROOT, "zvec_amd", "_variants", "libzvec_hip_%s.so" % variant))
import torch  # noqa: E402
import zvec_amd as zv  # noqa: E402

n, dim, nq, k = 1000000, 768, 256, 10
g = torch.Generator(device="cuda").manual_seed(1)
tdt = torch.float16 if dt == "fp16" else torch.float32
se = zv.HipFlatSearcher(dim, "SquaredEuclidean", dtype=dt)
for a in range(0, n, 250000):
    rows = torch.randn(250000, dim, device="cuda", generator=g).to(tdt)
    zv._lib.check(se.add_batch_dev(rows.data_ptr(), 250000), "append")
    torch.cuda.synchronize()
q = torch.randn(nq, dim, device="cuda", generator=g).to(tdt)
ctx = se.create_context()
keys = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
scores = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
counts = torch.zeros((nq,), dtype=torch.int32, device="cuda")
L = zv._lib.lib()
fn = L.zvec_hip_debug_a256_stamps
fn.restype = C.c_int
out = (C.c_double * 40)()


def run():
    zv._lib.check(se.search_dev(q.data_ptr(), nq, k, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx), "search")
    torch.cuda.synchronize()


run()
fn(out)                                                   # (warm-up launch: counters reset)
reps = 5
for _ in range(reps):
    run()
fn(out)
pairs = 16 * reps                                         # pairs of work-group 0 (1M rows: 245 chunks of 32 tiles)
names = ["matrix_loop", "refresh_and_barrier", "tests_and_appends", "second_barrier", "drain_and_clear", "closing_barrier", "postponed_staging"]
res = {"dtype": dt, "variant": variant, "pairs": pairs}
for i, nm in enumerate(names):
    res[nm] = round(out[i] / pairs, 1)
print(json.dumps(res))
