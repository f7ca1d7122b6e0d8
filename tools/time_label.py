"""Times the IVF build's labelling step (zvec_hip_ivf_label_dev -> assign_kernel) on synthetic rows: rows x nlist x dim,
fp32 and fp16.  Prints TFLOP/s and the fraction of the dense MFMA peak (157.3 fp32 / 2516 f16, MI355X_MICROARCH.md)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zvec_amd as zv

def run(dtype, n, nlist, dim, reps=3):
    tdt = torch.float16 if dtype == "fp16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(1)
    rows = torch.randn(n, dim, device="cuda", generator=g).to(tdt)
    cent = torch.randn(nlist, dim, device="cuda", generator=g).to(tdt)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", dtype=dtype)
    assert se.set_centroids(cent.cpu().numpy()) == 0
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert se.label_dev(rows.data_ptr(), n, lab.data_ptr()) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert se.label_dev(rows.data_ptr(), n, lab.data_ptr()) == 0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = 2.0 * n * nlist * dim
    peak = 2516.0 if dtype == "fp16" else 157.3
    # spot check against torch on a slice
    m = min(n, 4096)
    d = (rows[:m].float() ** 2).sum(1, keepdim=True) + (cent.float() ** 2).sum(1)[None] - 2 * rows[:m].float() @ cent.float().T
    agree = (d.argmin(1).int() == lab[:m]).float().mean().item()
    print("%s rows %d x nlist %d x d %d: %.3f s  %.1f TFLOP/s  %.3f of peak  (argmin agreement with torch on %d rows: %.4f)"
          % (dtype, n, nlist, dim, dt, flop / dt / 1e12, flop / dt / 1e12 / peak, m, agree), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fp16":          # the one f16 figure (variant A/B runs)
        run("fp16", 1 << 21, 16384, 768)
        sys.exit(0)
    run("fp32", 1 << 20, 4096, 768)
    run("fp16", 1 << 21, 16384, 768)
    run("fp16", 1 << 20, 4096, 768)
    run("fp32", 1 << 18, 1000, 100)
