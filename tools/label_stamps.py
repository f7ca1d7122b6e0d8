"""Where a step of the fp16 labelling kernel on the 256 x 256 tile goes (diagnostic build: tools/build_variant.sh stamps
-DZVK_A256_STAMPS [-DZVK_A256_LOCKSTEP]): shader-clock cycles per step and phase of waves 0 / 4 of work-group 0, split into
barrier wait · fragment reads + DMA issue + counted wait · mid barrier · MFMA issue.  One JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variant = sys.argv[1] if len(sys.argv) > 1 else "stamps"
os.environ.setdefault("ZVEC_HIP_LIBRARY", os.path.join(ROOT, "zvec_amd", "_variants", "libzvec_hip_%s.so" % variant))
import torch  # noqa: E402
import zvec_amd as zv  # noqa: E402

n, nlist, dim = 1 << 18, 16384, 768
g = torch.Generator(device="cuda").manual_seed(1)
rows = torch.randn(n, dim, device="cuda", generator=g).half()
cent = torch.randn(nlist, dim, device="cuda", generator=g).half()
se = zv.HipIVFSearcher(dim, "SquaredEuclidean", dtype="fp16")
assert se.set_centroids(cent.cpu().numpy()) == 0
lab = torch.zeros(n, dtype=torch.int32, device="cuda")
L = zv._lib.lib()
fn = L.zvec_hip_debug_a256_stamps
fn.restype = C.c_int
out = (C.c_double * 40)()
assert se.label_dev(rows.data_ptr(), n, lab.data_ptr()) == 0
fn(out)                                                   # (warm-up launch: counters reset)
assert se.label_dev(rows.data_ptr(), n, lab.data_ptr()) == 0
fn(out)
items = (n // 256 + 255) // 256                           # work items of work-group 0 (grid = 256 CUs)
steps = items * (nlist // 256) * (dim * 2 // 128)
res = {"variant": variant, "steps_of_workgroup_0": steps}
names = ["barrier_wait", "reads_dma_wait", "mid_barrier", "mfma_issue"]
for grp in range(2):
    tot = 0.0
    for p in range(1, 5):
        for x in range(4):
            v = out[(grp * 5 + p) * 4 + x] / steps
            res["g%d_p%d_%s" % (grp, p, names[x])] = round(v, 1)
            tot += v
    res["g%d_cycles_per_step" % grp] = round(tot, 1)
print(json.dumps(res))
