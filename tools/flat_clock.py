"""In-kernel clock of the wide flat kernel (diagnostic build: tools/build_variant.sh clk -DZVK_CLOCK_STAMP).
Runs the flat1m search back to back for >= 2 s on random data, then reads the work-groups' stamps of the last launch:
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ZVEC_HIP_LIBRARY"] = os.path.join(ROOT, "zvec_amd", "_variants", "libzvec_hip_clk.so")
DT = sys.argv[1] if len(sys.argv) > 1 else "fp32"          # fp16: the 256 x 256 tile (scan256_f16_kernel), L2 metric
import torch  # noqa: E402
import zvec_amd  # noqa: E402

n, dim, nq, k = 1_000_000, 768, 256, 10
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
g = torch.Generator(device=dev)
g.manual_seed(1)
tdt = torch.float16 if DT == "fp16" else torch.float32
base = torch.randn((n, dim), generator=g, device=dev).to(tdt)
q = torch.randn((nq, dim), generator=g, device=dev).to(tdt)
flat = zvec_amd.HipFlatSearcher(dim, "InnerProduct" if DT == "fp32" else "SquaredEuclidean", dtype=DT)
sp = s.cuda_stream
zvec_amd._lib.check(flat.add_batch_dev(base.data_ptr(), n, stream=sp), "append")
ctx = flat.create_context()
ctx.set_stream(sp)
ok = torch.empty((nq, k), dtype=torch.int64, device=dev)
os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
oc = torch.empty((nq,), dtype=torch.int32, device=dev)
t0 = time.time()
steps = 0
while time.time() - t0 < 3.0:
    for _ in range(20):
        flat.search_dev(q.data_ptr(), nq, k, ok.data_ptr(), os_.data_ptr(), oc.data_ptr(), ctx, stream=sp)
    torch.cuda.synchronize()
    steps += 20
L = zvec_amd._lib.lib()
fn = L.zvec_hip_debug_flat_clock
fn.restype = C.c_int
out = (C.c_double * 16)()
fn(out)
mhz, life, last_start, first_end, med_end, last_end, nwg = [float(x) for x in out[:7]]
pcts = [round(float(x), 4) for x in out[7:15]]
flops = 2.0 * nq * n * dim
print(json.dumps({"in_kernel_clock_mhz": mhz, "workgroups": int(nwg), "workgroup_lifetime_ms_median": life,
                  "latest_start_ms": last_start, "earliest_end_ms": first_end, "median_end_ms": med_end, "latest_end_ms": last_end,
                  "end_ms_p10_25_40_60_75_90_95_99": pcts, "launches": steps, "fp32_mfma_peak_at_that_clock_tflops": 1024 * 64 * mhz * 1e6 / 1e12,
                  "tflops_over_latest_end": flops / (last_end * 1e-3) / 1e12 if last_end else None}))
