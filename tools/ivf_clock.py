"""Start / end times of the IVF list scan's work-groups (diagnostic build: tools/build_variant.sh clk -DZVK_CLOCK_STAMP).
Two shapes: the whole 10M x 768 index at batch 1024 (arguments: full), or one rank's 1/8 share emulated as 1.25M rows in 512
lists probed 4 deep (default) — every list is then probed by ~8 queries of the batch, as in the sharded run.  Reads the stamps of
the LAST list-scan launch: how long the launch ramps up, when the work-groups finish, how long the tail is.  One JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ZVEC_HIP_LIBRARY", os.path.join(ROOT, "zvec_amd", "_variants", "libzvec_hip_clk.so"))
import torch  # noqa: E402
import zvec_amd  # noqa: E402
from bench import corpus_proj, gen_corpus, SEED  # noqa: E402

full = len(sys.argv) > 1 and sys.argv[1] == "full"
n, nlist, nprobe = (10_000_000, 4096, 36) if full else (1_250_000, 512, 4)
dim, nq, k = 768, 1024, 10
dev = torch.device("cuda", 0)
proj = corpus_proj(torch, dim, dev, 12)
base = gen_corpus(torch, n, dim, dev, SEED, proj, torch.float32)
q = gen_corpus(torch, nq, dim, dev, SEED + 1, proj, torch.float32)
ivf = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean")
zvec_amd._lib.check(ivf.build_dev(base.data_ptr(), n, nlist, kmeans_iters=10, seed=SEED), "build")
del base
ctx = ivf.create_context()
ok = torch.empty((nq, k), dtype=torch.int64, device=dev)
os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev)
ctx.profile(True)
for _ in range(12):
    zvec_amd._lib.check(ivf.search_dev(q.data_ptr(), nq, k, nprobe, n - 1, ok.data_ptr(), os_.data_ptr(), oc.data_ptr(), ctx), "search")
torch.cuda.synchronize()
pr = ctx.profile_read()
print("list scan by HIP events: %.3f ms per launch" % (pr["scan_ms"] / max(pr["launches"], 1)), file=sys.stderr, flush=True)
L = zvec_amd._lib.lib()
fn = L.zvec_hip_debug_flat_clock
fn.restype = C.c_int
out = (C.c_double * 16)()
fn(out)
mhz, life, last_start, first_end, med_end, last_end, nwg = [float(x) for x in out[:7]]
print(json.dumps({"shape": "full" if full else "one eighth", "scan_ms_by_events": pr["scan_ms"] / max(pr["launches"], 1),
                  "bytes_per_launch": pr["bytes"] / max(pr["launches"], 1), "in_kernel_clock_mhz": mhz, "workgroups": int(nwg),
                  "workgroup_lifetime_ms_median": life, "lifetime_ms_p10": float(out[15]), "latest_start_ms": last_start,
                  "earliest_end_ms": first_end, "median_end_ms": med_end, "latest_end_ms": last_end,
                  "end_ms_p10_25_40_60_75_90_95_99": [round(float(x), 4) for x in out[7:15]]}))
