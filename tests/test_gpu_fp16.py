"""fp16 rows (IndexMeta::DT_FP16 — BASELINE configs[3] stores the corpus in fp16): HalfFloatConverter output on the
base side, HalfFloatReformer output on the query side, fp32 accumulation.  The GPU multiplies halves on the f16
matrix cores (products exact in fp32) and accumulates in fp32; the oracle restates the reference's AVX-512
(no FP16 ISA) kernels bit for bit (distance_matrix_accum_fp16.i:554-594)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import tie_tolerant_compare, kmeans_lists, exact_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


@pytest.mark.parametrize("n,dim,nq,k", [(1000, 16, 5, 3), (3000, 100, 40, 10), (5000, 768, 33, 10), (2000, 65, 130, 20)])
def test_flat_fp16_integer_bit_exact(zv, oracle, n, dim, nq, k):
    rng = np.random.default_rng(n + dim)
    hi = 64 if dim <= 128 else 16
    base = rng.integers(0, hi, (n, dim)).astype(np.float16)       # integers are exact in fp16
    q = rng.integers(0, hi, (nq, dim)).astype(np.float16)
    for metric, name in ((O.METRIC_L2, "SquaredEuclidean"), (O.METRIC_IP, "InnerProduct")):
        se = zv.HipFlatSearcher(dim, name, dtype="fp16")
        assert se.load(base) == 0
        ctx = se.create_context()
        ctx.set_topk(k)
        assert se.search_impl(q, nq, ctx) == 0
        ok, os_, _, oc = oracle.flat_search(base, q, k, metric)
        tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="fp16 %s" % name)
    got = se.get_vector_by_id(n - 1)
    assert got.dtype == np.float16 and np.array_equal(got, base[n - 1])


def test_flat_fp16_gaussian_tolerance(zv, oracle):
    rng = np.random.default_rng(5)
    n, dim, nq, k = 4000, 768, 24, 10
    base = rng.standard_normal((n, dim)).astype(np.float16)        # HalfFloatConverter: RNE, == numpy astype
    q = rng.standard_normal((nq, dim)).astype(np.float16)
    qn = (q.astype(np.float64) ** 2).sum(1)
    bn = (base.astype(np.float64) ** 2).sum(1).max()
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean", dtype="fp16")
    assert se.load(base) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=2e-6, atol=1e-6, select_band=4e-6 * (qn + bn),
                         what="fp16 L2 gaussian")
    se = zv.HipFlatSearcher(dim, "InnerProduct", dtype="fp16")
    assert se.load(base) == 0
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_IP)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=4e-6, scale=np.sqrt(qn * bn), what="fp16 IP gaussian")


def test_ivf_fp16_same_index_on_both_sides(zv, oracle):
    rng = np.random.default_rng(8)
    n, dim, nlist, nq, k = 8000, 64, 50, 90, 10
    base32 = rng.integers(0, 40, (n, dim)).astype(np.float32)
    q = rng.integers(0, 40, (nq, dim)).astype(np.float16)
    cent, offs, order = kmeans_lists(rng, base32, nlist)
    cent = np.round(cent).astype(np.float16)
    vecs, keys = base32[order].astype(np.float16), order.astype(np.uint64)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.2, brute_force_threshold=100, dtype="fp16")
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, osc = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys)
    cd = np.sort(exact_l2(cent.astype(np.float32), q.astype(np.float32)), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, nprobe])[0]
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], what="ivf fp16")
    scanned, _ = se.last_stats(ctx, nq)
    assert np.array_equal(scanned[sel], osc[sel])
    c2, o2, rows = se.export()
    assert c2.dtype == np.float16 and np.array_equal(c2, cent) and np.array_equal(o2, offs)


def test_ivf_fp16_gpu_build(zv, oracle):
    rng = np.random.default_rng(9)
    n, dim, nlist, nq, k = 20000, 96, 48, 60, 10
    base = (rng.standard_normal((n, 8)) @ rng.standard_normal((8, dim))).astype(np.float16)
    q = (rng.standard_normal((nq, 8)) @ rng.standard_normal((8, dim))).astype(np.float16)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=8 / 48., brute_force_threshold=n - 1, dtype="fp16")
    assert se.build(base, nlist, kmeans_iters=6, sample_per_list=128) == 0
    cent, offs, rows = se.export()
    assert cent.dtype == np.float16 and offs[-1] == n
    vecs = base[rows.astype(np.int64)]
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=rows)
    qn = (q.astype(np.float64) ** 2).sum(1)
    bn = (base.astype(np.float64) ** 2).sum(1).max()
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=2e-6, atol=1e-5,
                         select_band=4e-6 * (qn + bn), what="ivf fp16 built")
    fk, _, _, _ = oracle.flat_search(base, q, k, threads=4)
    recall = np.mean([len(set(ctx.keys[i].tolist()) & set(fk[i].tolist())) / k for i in range(nq)])
    assert recall > 0.9, recall


@pytest.mark.parametrize("n,dim,nq,k", [(3000, 63, 40, 10), (9000, 200, 130, 7)])
def test_fp16_cosine_rows(zv, oracle, n, dim, nq, k):
    """fp16 cosine rows = d normalised halves + the fp32 norm in two half slots (cosine_converter.cc:112-134,205-212);
    distance = 1 - ip over the first d halves (CosineDistanceMatrix<Float16,1,1>, cosine_distance_matrix.h:32-50)."""
    rng = np.random.default_rng(n)
    base = oracle.cosine_transform16(rng.standard_normal((n, dim)).astype(np.float32))
    q = oracle.cosine_transform16(rng.standard_normal((nq, dim)).astype(np.float32))
    assert base.shape == (n, dim + 2) and base.dtype == np.float16
    se = zv.HipFlatSearcher(dim + 2, "Cosine", dtype="fp16")
    assert se.load(base) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_COSINE)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, atol=2e-6, what="cosine fp16")
    for pos in (0, 17, n - 1):            # the norm's two half slots come back bit for bit
        assert np.array_equal(se.get_vector_by_id(pos).view(np.uint16), base[pos].view(np.uint16))
    with pytest.raises(RuntimeError):
        zv.HipFlatSearcher(2, "Cosine", dtype="fp16")     # no room for a dimension besides the norm slots


@pytest.mark.parametrize("dim", [1, 7, 16, 17, 31, 32, 33, 48, 100, 768, 769])
def test_device_reformers_bit_exact(zv, oracle, dim):
    """CosineReformer::transform / HalfFloatReformer on the GPU against the oracle's restatement of the host code
    (cosine_reformer.cc:66-112, Norm2 in the AVX-512 order, FloatHelper::ToFP16): identical bits."""
    import torch
    rng = np.random.default_rng(dim)
    nq = 37
    q = (rng.standard_normal((nq, dim)) * rng.uniform(0.1, 30, (nq, 1))).astype(np.float32)
    q[3] = 0.0                                              # zero vector: norm 0, left unnormalised
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(q).to(dev)
    ctx = zv.IndexContext()
    out32 = torch.zeros((nq, dim + 1), dtype=torch.float32, device=dev)
    ctx.reform_queries_dev(d_in.data_ptr(), nq, dim, out32.data_ptr(), cosine=True, out_dtype="fp32")
    out16 = torch.zeros((nq, dim + 2), dtype=torch.float16, device=dev)
    ctx.reform_queries_dev(d_in.data_ptr(), nq, dim, out16.data_ptr(), cosine=True, out_dtype="fp16")
    half = torch.zeros((nq, dim), dtype=torch.float16, device=dev)
    ctx.reform_queries_dev(d_in.data_ptr(), nq, dim, half.data_ptr(), cosine=False, out_dtype="fp16")
    ctx.synchronize()
    torch.cuda.synchronize()
    want32 = oracle.cosine_transform(q)
    assert np.array_equal(out32.cpu().numpy().view(np.uint32), want32.view(np.uint32))
    assert np.array_equal(out16.cpu().numpy().view(np.uint16), oracle.cosine_transform16(q).view(np.uint16))
    with np.errstate(over="ignore"):
        assert np.array_equal(half.cpu().numpy().view(np.uint16), q.astype(np.float16).view(np.uint16))


def test_ivf_fp16_cosine(zv, oracle):
    """IVF over fp16 cosine rows (d halves + the fp32 norm in two half slots): coarse assign and list scan use 1 - ip over
    the first d halves on both sides; same index on both sides."""
    rng = np.random.default_rng(91)
    n, dim, nlist, nq, k = 5000, 40, 24, 50, 10
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    base = oracle.cosine_transform16(raw)
    q = oracle.cosine_transform16(rng.standard_normal((nq, dim)).astype(np.float32))
    lab = rng.integers(0, nlist, n)
    order = np.argsort(lab, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(lab, minlength=nlist))]).astype(np.uint64)
    cent = oracle.cosine_transform16(np.stack([raw[lab == l].mean(0) for l in range(nlist)]).astype(np.float32))
    vecs, keys = base[order], order.astype(np.uint64)
    se = zv.HipIVFSearcher(dim + 2, "Cosine", scan_ratio=0.3, brute_force_threshold=100, dtype="fp16")
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, metric=O.METRIC_COSINE, keys=keys)
    c32, q32 = cent[:, :dim].astype(np.float32), q[:, :dim].astype(np.float32)
    cs = np.sort(1.0 - q32 @ c32.T, 1)
    sel = np.nonzero(cs[:, nprobe] - cs[:, nprobe - 1] > 1e-4)[0]
    assert len(sel) > nq // 2
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], atol=3e-6, what="ivf fp16 cosine")
    assert np.array_equal(se.get_vector_by_id(7).view(np.uint16), vecs[7].view(np.uint16))
