"""big-index check on one MI355X (288 GB HBM): a flat index of N x 768 fp32 (default 48M rows = 147 GB) appended from device
chunks, searched for planted rows near the start, the 2^32-byte / 2^35-byte / 2^37-byte marks and the very end (exact
hits, score 0), through the wide-batch kernel, the small-batch kernel and the by-ids path — 64-bit addressing everywhere."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zvec_amd as zv

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 48_000_000
dim, chunk = 768, 1_000_000
dev = torch.device("cuda:0")
st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
assert st.reserve(n) == 0 if hasattr(st, "reserve") else True
g = torch.Generator(device=dev)
plant = sorted(set([0, 1, 12345, (1 << 32) // (dim * 4) + 7, (1 << 35) // (dim * 4) + 3, (1 << 37) // (dim * 4) + 11, n // 2 + 1, n - 2, n - 1]))
plant = [p for p in plant if p < n]
planted = {}
t0 = time.time()
for o in range(0, n, chunk):
    m = min(chunk, n - o)
    g.manual_seed(1000 + o // chunk)
    x = torch.randn((m, dim), generator=g, device=dev, dtype=torch.float32)
    for p in plant:
        if o <= p < o + m:
            planted[p] = x[p - o].cpu().numpy().copy()
    torch.cuda.synchronize()
    assert st.add_batch_dev(x.data_ptr(), m) == 0
    if (o // chunk) % 8 == 0:
        print("appended %d rows (%.0f s)" % (o + m, time.time() - t0), flush=True)
torch.cuda.synchronize()
print("index: %d rows, %.1f GB of rows in HBM, %.0f s" % (st.count(), st.count() * dim * 4 / 1e9, time.time() - t0), flush=True)
q = np.stack([planted[p] for p in plant]).astype(np.float32)
for label, reps in (("small batch (%d queries)" % len(plant), 1), ("wide batch (%d queries)" % (len(plant) * 32), 32)):
    qq = np.tile(q, (reps, 1))
    ctx = st.create_context()
    ctx.set_topk(5)
    t1 = time.time()
    assert st.search_impl(qq, qq.shape[0], ctx) == 0
    dt = time.time() - t1
    for i in range(qq.shape[0]):
        d = ctx.result(i)[0]
        assert d.key() == plant[i % len(plant)] and d.score() == 0.0, (label, i, d.key(), d.score(), plant[i % len(plant)])
    print("%s: every planted row found at score 0 (%.2f s = %.2f TB/s of rows)" % (label, dt, st.count() * dim * 4 / dt / 1e12), flush=True)
pk = st.create_context()
pk.set_topk(3)
assert st.search_bf_by_p_keys_impl(q[-1:], [[n - 1, n - 2, 5]], 1, pk) == 0
assert pk.result(0)[0].key() == n - 1 and pk.result(0)[0].score() == 0.0
row = st.get_vector_by_id(n - 1)
assert np.array_equal(row, planted[n - 1])
print("by-ids search and get_vector at the last row: ok", flush=True)

# ---- the same at IVF: a streamed build of N2 x 768 fp32 (default 32M rows = 98 GB), nlist 8192 --------------------------
del st
torch.cuda.empty_cache()
n2 = int(float(sys.argv[2])) if len(sys.argv) > 2 else 32_000_000
nlist = 8192
ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean")


def chunk_rows(o):
    m = min(chunk, n2 - o)
    g.manual_seed(5000 + o // chunk)
    return torch.randn((m, dim), generator=g, device=dev, dtype=torch.float32)


t0 = time.time()
S = min(n2, 256 * nlist)
sample = torch.cat([chunk_rows(o)[: S // ((n2 + chunk - 1) // chunk) + 1] for o in range(0, n2, chunk)])[:S].contiguous()
assert ivf.train_dev(sample.data_ptr(), sample.shape[0], nlist, kmeans_iters=4) == 0
del sample
labels = torch.empty(n2, dtype=torch.int32, device=dev)
for o in range(0, n2, chunk):
    x = chunk_rows(o)
    assert ivf.label_dev(x.data_ptr(), x.shape[0], labels[o:o + x.shape[0]].data_ptr()) == 0
lab_h = labels.cpu().numpy().astype(np.uint32)
sizes = np.bincount(lab_h, minlength=nlist).astype(np.uint32)
assert ivf.begin_lists(sizes) == 0
plant2 = [p for p in sorted(set([0, 777, n2 // 3, n2 // 2 + 5, n2 - 1])) if p < n2]
planted2 = {}
for o in range(0, n2, chunk):
    x = chunk_rows(o)
    for p in plant2:
        if o <= p < o + x.shape[0]:
            planted2[p] = x[p - o].cpu().numpy().copy()
    torch.cuda.synchronize()
    assert ivf.add_dev(x.data_ptr(), x.shape[0], lab_h[o:o + x.shape[0]], o) == 0
assert ivf.end_lists() == 0
cnt, nl = ivf.info()
print("ivf: %d rows in %d lists (largest %d rows), %.1f GB, built in %.0f s" % (cnt, nl, sizes.max(), cnt * dim * 4 / 1e9, time.time() - t0), flush=True)
q2 = np.stack([planted2[p] for p in plant2]).astype(np.float32)
ivf.set_nprobe(4)
for label, reps in (("single queries", 0), ("batch of %d" % (len(plant2) * 64), 64)):
    ctx = ivf.create_context()
    ctx.set_topk(5)
    if reps == 0:
        for i, p in enumerate(plant2):
            assert ivf.search_impl(q2[i:i + 1], 1, ctx) == 0
            d = ctx.result(0)[0]
            assert d.key() == p and d.score() == 0.0, (label, p, d.key(), d.score())
    else:
        qq = np.tile(q2, (reps, 1))
        assert ivf.search_impl(qq, qq.shape[0], ctx) == 0
        for i in range(qq.shape[0]):
            d = ctx.result(i)[0]
            assert d.key() == plant2[i % len(plant2)] and d.score() == 0.0, (label, i, d.key(), d.score())
    print("ivf %s: every planted row found at score 0" % label, flush=True)
