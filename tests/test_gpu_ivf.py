"""GPU parity of the IVF-Flat path through the C ABI — run with -m gpu.
Both sides search THE SAME index (same centroids, same list order): zvec_hip_ivf_load of a host-built
structure, or the oracle reading back zvec_hip_ivf_export after a GPU build."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import tie_tolerant_compare, kmeans_lists, exact_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


def _ramp(n, dim):
    return np.repeat(np.arange(n, dtype=np.float32)[:, None], dim, 1)


def test_ivf_simple_reference_expectations(zv, golden_dir):
    """ivf_searcher_test.cc:200-321 (TestSimple): 33 rows, 1 centroid, scan_ratio 1, bf_threshold 1."""
    g = json.load(open(os.path.join(golden_dir, "scan_known_answers.json")))["ivf_simple"]
    n, dim = g["n"], g["dim"]
    base = _ramp(n, dim)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=g["scan_ratio"],
                           brute_force_threshold=g["brute_force_threshold"])
    assert se.load(base.mean(0, keepdims=True), np.array([0, n], np.uint64), base) == 0
    ctx = se.create_context()
    q = np.full((1, dim), g["single_query"], np.float32)
    qb = _ramp(33, dim)
    for fn in (se.search_bf_impl, se.search_impl):
        ctx.set_topk(33)
        assert fn(q, 1, ctx) == 0
        r = ctx.result(0)
        assert len(r) == 33
        for i in range(33):
            assert r[i].key() == 32 - i and r[i].score() == float(i * i * dim)
        ctx.set_topk(1)
        assert fn(qb, 33, ctx) == 0
        for qi in range(33):
            assert len(ctx.result(qi)) == 1
            assert ctx.result(qi)[0].key() == qi and ctx.result(qi)[0].score() == 0.0
    for i in (0, 5, 32):
        assert np.array_equal(se.get_vector_by_id(i), base[i])


def test_ivf_errors(zv):
    se = zv.HipIVFSearcher(8)
    ctx = se.create_context()
    ctx.set_topk(3)
    se.brute_force_threshold = 0
    se.total_count = 10
    assert se.search_impl(np.zeros((1, 8), np.float32), 1, ctx) == zv.IndexError_.NoIndexLoaded
    ctx.set_topk(0)
    assert se.search_impl(np.zeros((1, 8), np.float32), 1, ctx) == zv.IndexError_.InvalidArgument


@pytest.mark.parametrize("n,dim,nlist,nq,k,ratio", [(600, 8, 12, 5, 7, 0.34), (5000, 64, 50, 70, 10, 0.1),
                                                     (20000, 128, 128, 130, 10, 1 / 32.), (3000, 768, 40, 33, 10, 0.2),
                                                     (4000, 32, 64, 200, 50, 0.25)])
def test_ivf_integer_data_bit_exact(zv, oracle, n, dim, nlist, nq, k, ratio):
    rng = np.random.default_rng(n + dim)
    hi = 64 if dim <= 128 else 16
    base = rng.integers(0, hi, (n, dim)).astype(np.float32)
    q = rng.integers(0, hi, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent)                     # integer centroids: coarse distances exact as well
    vecs = base[order]
    keys = order.astype(np.uint64)            # key = original row
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=ratio, brute_force_threshold=100)
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, osc = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys)
    # coarse ties (equal centroid distances) make the probe set itself ambiguous in the reference
    # (unstable std::sort, heap.h:173-175): only compare queries whose nprobe-th and (nprobe+1)-th coarse
    # distances differ
    cd = np.sort(exact_l2(cent, q), 1)
    clean = np.ones(nq, bool) if nprobe >= nlist else cd[:, nprobe - 1] != cd[:, nprobe]
    assert clean.sum() >= nq * 0.8
    sel = np.nonzero(clean)[0]
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], what="ivf int")
    scanned, probes = se.last_stats(ctx, nq)
    assert np.array_equal(scanned[sel], osc[sel])          # IndexContext::Stats parity (scan volume)
    # brute force over the lists == exact flat top-k
    assert se.search_bf_impl(q, nq, ctx) == 0
    fk, fs, _, fc = oracle.flat_search(base, q, k)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, fk, fs, fc, what="ivf bf")


def test_ivf_max_scan_count_rule(zv, oracle):
    """probing stops once the running scanned count reaches max_scan_count (ivf_searcher.cc:223-237)."""
    rng = np.random.default_rng(23)
    n, dim, nlist, nq, k = 6000, 16, 60, 90, 10
    base = rng.integers(0, 100, (n, dim)).astype(np.float32)
    q = rng.integers(0, 100, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent)
    vecs, keys = base[order], order.astype(np.uint64)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.05, brute_force_threshold=100)
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    assert nprobe == 3 and max_scan == 300
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, osc, opr = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys, want_probes=True)
    cd = np.sort(exact_l2(cent, q), 1)
    clean = (np.diff(cd[:, :nprobe + 1], axis=1) != 0).all(1)
    sel = np.nonzero(clean)[0]
    scanned, probes = se.last_stats(ctx, nq)
    assert np.array_equal(scanned[sel], osc[sel])
    assert np.array_equal(probes[sel], (opr[sel] != 0xffffffff).sum(1))
    assert (probes < nprobe).any()            # the rule actually cut some probe lists short
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], what="max_scan")


def test_ivf_filter_and_threshold(zv, oracle):
    rng = np.random.default_rng(29)
    n, dim, nlist, nq, k = 5000, 24, 32, 40, 10
    base = rng.integers(0, 80, (n, dim)).astype(np.float32)
    q = rng.integers(0, 80, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent)
    vecs, keys = base[order], order.astype(np.uint64)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=1.0, brute_force_threshold=100)
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    ctx.set_filter(lambda key: key % 3 == 0)                # true = exclude (index_filter.h:48-50)
    assert se.search_impl(q, nq, ctx) == 0
    mask = (keys % 3 == 0)
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys,
                                          exclude_bits=O.pack_bits(mask))
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="ivf filter")
    assert (ctx.keys[ctx.counts[:, None] > np.arange(k)[None, :]] % 3 != 0).all()
    ctx.reset_filter()
    ctx.set_threshold(float(np.median(os_)))
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys, threshold=ctx.threshold())
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="ivf rnn")
    assert (ctx.counts < k).any()


def test_ivf_gpu_build_then_same_index_on_oracle(zv, oracle):
    """k-means build on the GPU; the oracle searches the exported structure."""
    rng = np.random.default_rng(31)
    n, dim, nlist, nq, k = 30000, 64, 64, 100, 10
    means = rng.standard_normal((256, dim)) * 3
    base = (means[rng.integers(0, 256, n)] + rng.standard_normal((n, dim))).astype(np.float32)
    q = (base[rng.choice(n, nq, replace=False)] + 0.1 * rng.standard_normal((nq, dim))).astype(np.float32)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=8 / 64., brute_force_threshold=n - 1)
    assert se.build(base, nlist, kmeans_iters=8, sample_per_list=128) == 0
    cent, offs, rows = se.export()
    assert offs[-1] == n and sorted(rows.tolist()) == list(range(n))
    sizes = np.diff(offs.astype(np.int64))
    assert sizes.min() > 0 and sizes.max() < 8 * n / nlist        # k-means did balance the lists
    for p in (0, 77, n - 1):
        assert np.array_equal(se.get_vector_by_id(p), base[rows[p]])
    vecs = base[rows.astype(np.int64)]
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, osc = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=rows)
    qn = (q.astype(np.float64) ** 2).sum(1)
    bn = (base.astype(np.float64) ** 2).sum(1).max()
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=2e-6, atol=1e-6,
                         select_band=4e-6 * (qn + bn), what="ivf built")
    # recall@10 against the exact flat answer
    fk, _, _, _ = oracle.flat_search(base, q, k)
    recall = np.mean([len(set(ctx.keys[i].tolist()) & set(fk[i].tolist())) / k for i in range(nq)])
    assert recall >= 0.95, recall


@pytest.mark.parametrize("metric_name,metric", [("InnerProduct", O.METRIC_IP), ("Cosine", O.METRIC_COSINE)])
def test_ivf_ip_and_cosine(zv, oracle, metric_name, metric):
    """IVF with the other two metrics of the path: coarse assign and list scan both use the index metric
    (MinusInnerProduct: inner_product_matrix_fp32.cc:870; Cosine: 1 - ip on normalised rows carrying their
    norm as an extra float, cosine_distance_matrix.h:32-50 / cosine_converter.cc:112-127)."""
    rng = np.random.default_rng(43)
    n, dim, nlist, nq, k = 6000, 48, 40, 64, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    if metric == O.METRIC_COSINE:
        base, q = oracle.cosine_transform(base), oracle.cosine_transform(q)
    ed = base.shape[1]
    lab = rng.integers(0, nlist, n)
    order = np.argsort(lab, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(lab, minlength=nlist))]).astype(np.uint64)
    cent = np.stack([base[lab == l].mean(0) for l in range(nlist)]).astype(np.float32)
    if metric == O.METRIC_COSINE:
        cent = oracle.cosine_transform(cent[:, :dim])
    vecs, keys = base[order], order.astype(np.uint64)
    se = zv.HipIVFSearcher(ed, metric_name, scan_ratio=0.3, brute_force_threshold=100)
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, metric=metric, keys=keys)
    # the probe set is decided by coarse scores; compare the queries whose nprobe-th / (nprobe+1)-th coarse
    # scores are separated by more than the fp32 band
    cs = np.sort(np.array([[oracle.dist(metric, c, qq) for c in cent] for qq in q]), 1)
    sel = np.nonzero(cs[:, nprobe] - cs[:, nprobe - 1] > 1e-4)[0]
    assert len(sel) > nq // 2
    scale = 1.0 if metric == O.METRIC_COSINE else float(np.abs(os_).max())
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], rtol=4e-6, scale=scale,
                         what="ivf " + metric_name)
    # the same queries a few at a time: the small-batch (wave per row) route, same answers
    small = se.create_context()
    small.set_topk(k)
    for a0 in range(0, 24, 6):
        assert se.search_impl(q[a0:a0 + 6], 6, small) == 0
        ss = sel[(sel >= a0) & (sel < a0 + 6)]
        if len(ss):
            tie_tolerant_compare(small.keys[ss - a0], small.scores[ss - a0], small.counts[ss - a0], ok[ss], os_[ss], oc[ss], rtol=4e-6,
                                 scale=scale, what="ivf small batches " + metric_name)
    for p in (0, n - 1):
        assert np.array_equal(se.get_vector_by_id(p), vecs[p])


def test_ivf_large_topk_fallback(zv, oracle):
    """topk beyond the LDS-resident lists of the scan kernel (k > ~470): position expansion + direct scoring.
    Same index on both sides, integer data => bit-exact scores; includes a filter and a sharded rank."""
    rng = np.random.default_rng(77)
    n, dim, nlist, nq, k = 6000, 24, 12, 9, 700
    base = rng.integers(-9, 10, (n, dim)).astype(np.float32)
    q = rng.integers(-9, 10, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    vecs, keys = base[order], order.astype(np.uint64)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.5, brute_force_threshold=100)
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys)
    cd = np.sort(exact_l2(cent, q), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, min(nprobe, nlist - 1)])[0] if nprobe < nlist else np.arange(nq)
    assert len(sel) > 0
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], what="ivf large k")
    # with a filter over list-order positions
    drop = rng.random(n) < 0.5
    ex = O.pack_bits(drop)
    ctx.set_exclude_bitset(ex)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys, exclude_bits=ex)
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], what="ivf large k filtered")
    dropped = set(keys[drop].tolist())
    assert not any(int(x) in dropped for qi in sel for x in ctx.keys[qi, : ctx.counts[qi]])


@pytest.mark.parametrize("dtype,column_major,dim", [(np.float32, False, 24), (np.float32, True, 20), (np.float16, True, 18),
                                                     (np.float16, False, 7)])
def test_ivf_load_from_reference_segments(zv, oracle, dtype, column_major, dim):
    """zvec_hip_ivf_load_segments: the dumped-index segment payloads (tests/ivf_format.py restates the reference's writer)
    give the same device index as loading the same lists from arrays: identical export, vectors and search results."""
    from tests.ivf_format import dump_ivf_segments
    rng = np.random.default_rng(dim)
    nlist = 9
    sizes = [0, 1, 31, 32, 33, 64, 100, 257, 5]                      # empty list, partial / exact / multi-block lists
    n = sum(sizes)
    base = rng.integers(-9, 10, (n, dim)).astype(dtype)
    keys = rng.permutation(10 * n)[:n].astype(np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    cent = np.stack([base[int(offs[l]):int(offs[l + 1])].astype(np.float32).mean(0) if sizes[l] else np.full(dim, 50.0)
                     for l in range(nlist)]).astype(dtype)
    lists = [(base[int(offs[l]):int(offs[l + 1])], keys[int(offs[l]):int(offs[l + 1])]) for l in range(nlist)]
    seg = dump_ivf_segments(lists, dim, dtype, column_major)
    dt = "fp16" if dtype == np.float16 else "fp32"
    a = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.5, brute_force_threshold=10, dtype=dt)
    assert a.load_segments(seg["ivf.inverted_header"], seg["ivf.inverted_meta"], seg["ivf.inverted_body"], seg["hc.keys"], cent) == 0
    b = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.5, brute_force_threshold=10, dtype=dt)
    assert b.load(cent, offs, base, keys) == 0
    ca, oa, ra = a.export()
    cb, ob, rb = b.export()
    assert np.array_equal(ca, cb) and np.array_equal(oa, ob) and np.array_equal(ra, rb)
    for pos in (0, 1, 31, 32, 33, 95, 200, n - 1):
        assert np.array_equal(a.get_vector_by_id(pos).view(np.uint8), base[pos].view(np.uint8))
    q = rng.integers(-9, 10, (17, dim)).astype(dtype)
    res = []
    for se in (a, b):
        ctx = se.create_context()
        ctx.set_topk(12)
        assert se.search_impl(q, 17, ctx) == 0
        res.append((ctx.keys.copy(), ctx.scores.copy(), ctx.counts.copy()))
    assert all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))
    # rejected: truncated body, wrong dimension, inconsistent list table
    assert a.load_segments(seg["ivf.inverted_header"], seg["ivf.inverted_meta"], seg["ivf.inverted_body"][:-64], seg["hc.keys"], cent) \
        == zv.IndexError_.InvalidArgument
    wrong = zv.HipIVFSearcher(dim + 1, "SquaredEuclidean", dtype=dt)
    assert wrong.load_segments(seg["ivf.inverted_header"], seg["ivf.inverted_meta"], seg["ivf.inverted_body"], seg["hc.keys"],
                               np.zeros((nlist, dim + 1), dtype)) == zv.IndexError_.Mismatch
    m = seg["ivf.inverted_meta"]                       # the metas of two non-empty lists swapped: id offsets no longer back to back
    assert a.load_segments(seg["ivf.inverted_header"], m[:80] + m[120:160] + m[80:120] + m[160:],
                           seg["ivf.inverted_body"], seg["hc.keys"], cent) == zv.IndexError_.InvalidArgument


def test_same_value_rows_all_ties(zv):
    """ivf_searcher_test.cc:3301-3420 (TestSameValue): every stored vector is the same point, so every score ties.
    Counts and scores are determined, the ids only as a set of distinct stored keys; the GPU k-means build must
    survive a corpus with one distinct point (all but one cluster empty)."""
    n, dim = 33, 8
    base = np.zeros((n, dim), np.float32)
    keys = np.arange(100, 100 + n, dtype=np.uint64)
    q1 = np.full((1, dim), 32.0, np.float32)
    qb = np.repeat(np.arange(33, dtype=np.float32)[:, None], dim, 1)
    flat = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert flat.load(base, keys) == 0
    ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=1.0, brute_force_threshold=1)
    assert ivf.build(base, 2, keys=keys, kmeans_iters=5) == 0
    assert ivf.info() == (n, 2)
    for se, fns in ((flat, (flat.search_impl,)), (ivf, (ivf.search_impl, ivf.search_bf_impl))):
        for fn in fns:
            ctx = se.create_context()
            for k in (33, 10, 40):
                ctx.set_topk(k)
                assert fn(q1, 1, ctx) == 0
                r = ctx.result(0)
                assert len(r) == min(k, n)
                assert all(d.score() == 32.0 * 32.0 * dim for d in r)
                got = [d.key() for d in r]
                assert len(set(got)) == len(got) and set(got) <= set(keys.tolist())
            ctx.set_topk(1)
            assert fn(qb, 33, ctx) == 0
            for qi in range(33):
                assert len(ctx.result(qi)) == 1 and ctx.result(qi)[0].score() == float(qi * qi * dim)


@pytest.mark.parametrize("nq,k", [(1, 10), (3, 64), (2, 65)])
def test_small_batch_equal_scores_keep_stream_order(zv, oracle, nq, k):
    """The small-batch route with heavy ties on a stream of several thousand candidates: equal scores come back in the
    order the reference scans them — probe rank, then place in the list (ivf_searcher.cc:223-260 walks the lists in coarse
    order and pushes rows in storage order) — whether the scoring blocks pre-select (k <= 64: pkeys_topk_kernel + one
    merge) or the whole score matrix is selected from in two steps (k = 65)."""
    rng = np.random.default_rng(811 + k)
    n, dim, nlist = 9000, 16, 24
    points = rng.integers(0, 6, (5, dim)).astype(np.float32)           # five distinct stored points => ~1800-way ties
    base = points[rng.integers(0, 5, n)]
    lab = rng.integers(0, nlist, n)
    order = np.argsort(lab, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(lab, minlength=nlist))]).astype(np.uint64)
    cent = rng.normal(3, 2, (nlist, dim)).astype(np.float32)           # (any centroids: the probe order only has to be untied)
    vecs, keys = base[order], order.astype(np.uint64) * 3 + 1
    q = rng.integers(0, 6, (nq, dim)).astype(np.float32)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.5, brute_force_threshold=10)
    assert se.load(cent, offs, vecs, keys) == 0
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    _, _, _, _, _, opr = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys, want_probes=True)
    for qi in range(nq):
        lists = [int(l) for l in opr[qi] if l != 0xffffffff]
        stream = np.concatenate([np.arange(offs[l], offs[l + 1]) for l in lists]).astype(np.int64)
        assert len(stream) > 2048                                       # several runs / many scoring blocks
        sc = ((vecs[stream] - q[qi]) ** 2).sum(1)
        first = stream[np.argsort(sc, kind="stable")[:k]]
        assert ctx.counts[qi] == k
        assert np.array_equal(ctx.scores[qi], np.sort(sc, kind="stable")[:k])
        assert np.array_equal(ctx.keys[qi], keys[first]), "tie order of query %d" % qi


def test_boundary_a_parameter_mapping_probes_exactly_nprobe(zv):
    """SURVEY H3 / patches/boundary_a.diff: nlist = 4096 honoured (no clamp to 1024), nprobe = 32 handed over as
    scan_ratio = 32/4096 with brute_force_threshold = N - 1 => every query probes exactly 32 lists and scans exactly the
    rows of those lists (BASELINE configs[2]'s probe shape), through the operator-level search_impl."""
    rng = np.random.default_rng(64)
    n, dim, nlist, nprobe, nq, k = 200_000, 32, 4096, 32, 64, 10
    base = (rng.standard_normal((n, 6)) @ rng.standard_normal((6, dim))).astype(np.float32)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean")
    assert se.build(base, nlist, kmeans_iters=3, sample_per_list=32) == 0
    assert se.info() == (n, nlist)
    se.set_nprobe(nprobe)
    assert se.probe_params() == (nprobe, n - 1)
    q = base[rng.choice(n, nq, replace=False)]
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    scanned, probes = se.last_stats(ctx, nq)
    assert (probes == nprobe).all()
    cent, offs, rows = se.export()
    sizes = np.diff(offs.astype(np.int64))
    d = ((q[:, None, :].astype(np.float64) - cent[None, :, :].astype(np.float64)) ** 2).sum(-1)
    near = np.argsort(d, axis=1, kind="stable")[:, :nprobe]
    want = sizes[near].sum(1)
    # (a coarse near-tie at rank 32 may swap one list: compare wherever the 32nd and 33rd centroid are well separated)
    ds = np.sort(d, axis=1)
    clear = (ds[:, nprobe] - ds[:, nprobe - 1]) > 1e-4 * ds[:, nprobe]
    assert clear.sum() > nq // 2 and (scanned[clear] == want[clear]).all()
    assert (ctx.scores[:, 0] == 0).all()                       # self-queries: found in their own list


@pytest.mark.parametrize("dtype,column_major", [(np.float32, False), (np.float32, True), (np.float16, True)])
def test_open_dumped_index_files(zv, oracle, dtype, column_major):
    """next-2 end to end: a dumped index FILE image -> container framing (zvec_hip_container_segments) -> segment
    payloads -> the segment loaders -> HBM; for IVF the centroid rows come out of the nested flat index file inside
    "ivf.centroid" (column-major, permuted).  Files written by the restated packer / dumpers (tests/ivf_format.py:
    layout parity unpinned); the opened indexes must answer exactly like ones loaded from plain arrays."""
    from tests.ivf_format import flat_index_file, ivf_index_file
    rng = np.random.default_rng(17)
    n, dim, nlist, nq, k = 3000 + 7, 40, 37, 25, 6
    base = rng.integers(-8, 9, (n, dim)).astype(dtype)
    keys = rng.permutation(5 * n)[:n].astype(np.uint64)
    q = rng.integers(-8, 9, (nq, dim)).astype(dtype)
    dt = "fp16" if dtype == np.float16 else "fp32"
    # flat
    fse = zv.open_flat_file(flat_index_file(base, keys, column_major, "InnerProduct"))
    assert fse.count() == n
    ref = zv.HipFlatSearcher(dim, "InnerProduct", dtype=dt)
    assert ref.load(base, keys) == 0
    c1, c2 = fse.create_context(), ref.create_context()
    c1.set_topk(k), c2.set_topk(k)
    assert fse.search_impl(q, nq, c1) == 0 and ref.search_impl(q, nq, c2) == 0
    assert np.array_equal(c1.keys, c2.keys) and np.array_equal(c1.scores, c2.scores)
    # ivf
    cent, offs, order = kmeans_lists(rng, base.astype(np.float32), nlist)
    cent = np.round(cent).astype(dtype)
    lists = [(base[order[int(offs[l]):int(offs[l + 1])]], keys[order[int(offs[l]):int(offs[l + 1])]]) for l in range(nlist)]
    image = ivf_index_file(cent, lists, column_major, centroid_column_major=True, centroid_perm=rng.permutation(nlist))
    ise = zv.open_ivf_file(image)
    assert ise.info() == (n, nlist)
    c2, o2, _ = ise.export()
    assert np.array_equal(c2.view(np.uint8), cent.view(np.uint8)) and np.array_equal(o2, offs)
    ise.scan_ratio, ise.brute_force_threshold = 6 / nlist, 10
    rse = zv.HipIVFSearcher(dim, "SquaredEuclidean", dtype=dt, scan_ratio=6 / nlist, brute_force_threshold=10)
    assert rse.load(cent, offs, base[order], keys[order]) == 0
    c1, c2 = ise.create_context(), rse.create_context()
    c1.set_topk(k), c2.set_topk(k)
    assert ise.search_impl(q, nq, c1) == 0 and rse.search_impl(q, nq, c2) == 0
    assert np.array_equal(c1.keys, c2.keys) and np.array_equal(c1.scores, c2.scores)


def test_open_reference_dumped_golden_files(zv):
    """next-2 pinned: index FILES written by the reference's OWN FlatBuilder / IVFDumper / MemoryDumper (golden fixtures,
    tests/golden/ref_index_files.npz) -> container parser -> segment loaders -> HBM; every opened index must hold exactly
    the rows / keys / lists the reference was given and answer like an index loaded from those arrays."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_index_files.npz"))
    rng = np.random.default_rng(3)
    for name in [str(x) for x in z["cases"]]:
        meta = z[name + "_meta"]
        f16 = bool(meta[0])
        dt = np.float16 if f16 else np.float32
        base = z[name + "_base"].view(dt) if f16 else z[name + "_base"]
        keys = z[name + "_keys"]
        n, dim = int(meta[1]), int(meta[2])
        q = rng.integers(-8, 9, (9, dim)).astype(dt)
        image = z[name + "_image"].tobytes()
        if name.startswith("flat"):
            se = zv.open_flat_file(image)
            assert se.count() == n
            for pos in (0, n // 2, n - 1):
                assert np.array_equal(se.get_vector_by_id(pos).view(np.uint8), base[pos].view(np.uint8))
            ref = zv.HipFlatSearcher(dim, "InnerProduct", dtype="fp16" if f16 else "fp32")
            assert ref.load(base, keys) == 0
            c1, c2 = se.create_context(), ref.create_context()
            c1.set_topk(5), c2.set_topk(5)
            assert se.search_impl(q, 9, c1) == 0 and ref.search_impl(q, 9, c2) == 0
            assert np.array_equal(c1.keys, c2.keys) and np.array_equal(c1.scores, c2.scores), name
        else:
            cent = z[name + "_cent"].view(dt) if f16 else z[name + "_cent"]
            offs = z[name + "_offs"]
            se = zv.open_ivf_file(image)
            assert se.info() == (n, int(meta[5]))
            c2, o2, _ = se.export()
            assert np.array_equal(c2.view(np.uint8), cent.view(np.uint8)) and np.array_equal(o2, offs), name
            pos = rng.integers(0, n, 12)
            assert np.array_equal(se.get_vectors_by_ids(pos).view(np.uint8), base[pos].view(np.uint8))
            se.scan_ratio, se.brute_force_threshold = 1.0, 0           # probe every list
            ref = zv.HipIVFSearcher(dim, "InnerProduct", dtype="fp16" if f16 else "fp32", scan_ratio=1.0, brute_force_threshold=0)
            assert ref.load(cent, offs, base, keys) == 0
            c1, c2 = se.create_context(), ref.create_context()
            c1.set_topk(5), c2.set_topk(5)
            assert se.search_impl(q, 9, c1) == 0 and ref.search_impl(q, 9, c2) == 0
            assert np.array_equal(c1.keys, c2.keys) and np.array_equal(c1.scores, c2.scores), name


def test_search_with_empty_centroid_reference_expectations(zv):
    """ivf_searcher_test.cc:1569-1673 (TestSearchWithEmptyCentroid): 10 documents holding only 5 distinct vectors, built
    into 9 lists (so some lists are empty and some centroids coincide), scan_ratio 1.0, brute_force_threshold 1; the query
    (999, ...) must find key 4 or 9 first (brute force, topk 1) and {4, 9}, {4, 9}, {3, 8} (knn, topk 3).  Here the index
    is built by the GPU builder — k-means on 10 points with duplicates — as the product's builder slot would."""
    dim, n = 256, 10
    base = np.repeat((np.arange(n) % 5).astype(np.float32)[:, None], dim, axis=1)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=1.0, brute_force_threshold=1)
    assert se.build(base, 9, keys=np.arange(n, dtype=np.uint64)) == 0
    cnt, nlist = se.info()
    assert cnt == n and nlist == 9
    q = np.full((1, dim), 999.0, np.float32)
    ctx = se.create_context()
    ctx.set_topk(1)
    assert se.search_bf_impl(q, 1, ctx) == 0
    assert len(ctx.result(0)) == 1 and ctx.result(0)[0].key() in (4, 9)
    ctx.set_topk(3)
    assert se.search_impl(q, 1, ctx) == 0
    r = ctx.result(0)
    assert len(r) == 3
    assert r[0].key() in (4, 9) and r[1].key() in (4, 9) and r[0].key() != r[1].key() and r[2].key() in (3, 8)
    # the same through a batch large enough for the list-major route
    qq = np.repeat(q, 20, axis=0)
    assert se.search_impl(qq, 20, ctx) == 0
    for i in range(20):
        r = ctx.result(i)
        assert {r[0].key(), r[1].key()} == {4, 9} and r[2].key() in (3, 8)


def test_context_update_overrides_probe_parameters(zv, oracle):
    """IndexContext::update(params) (ivf_searcher_context.h:61-79): scan_ratio / brute_force_threshold set on ONE context
    change what that context probes — nprobe = max(round(nlist * ratio), 1), max_scan_count = max(bft, ceil(N * ratio)) —
    other contexts keep the searcher's defaults; a ratio <= 0 is refused."""
    rng = np.random.default_rng(31)
    n, dim, nlist, nq, k = 6000, 32, 50, 12, 5
    base = rng.integers(0, 40, (n, dim)).astype(np.float32)
    q = rng.integers(0, 40, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.1, brute_force_threshold=10)
    assert se.load(cent, offs, base[order], order.astype(np.uint64)) == 0
    a, b = se.create_context(), se.create_context()
    a.set_topk(k)
    b.set_topk(k)
    assert b.update({"proxima.ivf.searcher.scan_ratio": 0.0}) == zv.IndexError_.InvalidArgument
    assert b.update({"proxima.ivf.searcher.scan_ratio": 0.5, "proxima.ivf.searcher.brute_force_threshold": n - 1}) == 0
    assert se.search_impl(q, nq, a) == 0 and se.search_impl(q, nq, b) == 0
    _, pa = se.last_stats(a, nq)
    _, pb = se.last_stats(b, nq)
    assert (pb == 25).all()                                  # round(50 * 0.5) lists, max_scan_count = N - 1 never cuts
    assert (pa <= 5).all() and (pa >= 1).all()               # round(50 * 0.1) = 5 lists at most (max_scan_count = ceil(N * 0.1))
    for ctx in (a, b):
        nprobe, max_scan = se.probe_params(ctx)
        ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, base[order], q, k, nprobe, max_scan, keys=order.astype(np.uint64))
        tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="ctx.update")


def test_handle_reuse_after_a_separate_coarse_space(zv, oracle):
    """zvec_hip_ivf_set_coarse_space puts the centroid store into a space of its own (MIPS-trained inner-product indexes,
    ivf_centroid_index.cc:273-297).  While it is in force the entries that assume centroids in the rows' space refuse
    (label / begin_lists / centroid read-back: Unsupported) instead of reading rows of the wrong width; a reload, a new set of
    centroids or a build on the SAME handle leaves the coarse space and behaves like a fresh handle."""
    import ctypes as C
    from zvec_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n, dim, nlist, k, cdim = 4000, 32, 16, 5, 36
    base = rng.integers(-20, 20, (n, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent).astype(np.float32)
    rows = np.ascontiguousarray(base[order])
    q = np.ascontiguousarray(base[rng.integers(0, n, 12)] + 1)
    h = C.c_void_p()
    assert L.zvec_hip_ivf_create(dim, 0, 1, 0, C.byref(h)) == 0                       # inner product
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    offs64 = np.ascontiguousarray(offs, np.uint64)
    assert L.zvec_hip_ivf_load(h, p(cent), nlist, offs64.ctypes.data_as(C.POINTER(C.c_uint64)), p(rows), None) == 0
    keys = np.zeros((12, k), np.uint64); sc = np.zeros((12, k), np.float32); cn = np.zeros(12, np.uint32)
    u32p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))

    def search():
        return L.zvec_hip_ivf_search(h, None, p(q), 12, k, C.c_float(3.4e38), 4, 1 << 30, None, p(keys), p(sc), p(cn))
    assert search() == 0
    first = keys.copy()
    # a coarse space of another width: plain searches now need the reformed queries
    ccent = np.ascontiguousarray(rng.standard_normal((nlist, cdim)).astype(np.float32))
    assert L.zvec_hip_ivf_set_coarse_space(h, cdim, 0, p(ccent), nlist) == 0
    assert search() == -31
    cq = np.ascontiguousarray(rng.standard_normal((12, cdim)).astype(np.float32))
    assert L.zvec_hip_ivf_search_coarse(h, None, p(q), p(cq), 12, k, C.c_float(3.4e38), 4, 1 << 30, None, p(keys), p(sc), p(cn)) == 0
    out = np.zeros((nlist, dim), np.float32)
    nl = C.c_uint32()
    assert L.zvec_hip_ivf_get_centroids(h, p(out), C.byref(nl)) == -12                # would overflow [nlist][dim]
    assert L.zvec_hip_ivf_get_centroids(h, None, C.byref(nl)) == 0 and nl.value == nlist
    assert L.zvec_hip_ivf_export(h, p(out), None, None) == -12
    sizes = np.diff(offs64).astype(np.uint32)
    assert L.zvec_hip_ivf_begin_lists(h, u32p(sizes)) == -12
    import torch
    d_rows = torch.from_numpy(rows).cuda()
    d_lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert L.zvec_hip_ivf_label_dev(h, C.c_void_p(d_rows.data_ptr()), n, C.cast(C.c_void_p(d_lab.data_ptr()), C.POINTER(C.c_uint32)), None) == -12
    # reload on the same handle: back in the rows' own space, everything works again
    assert L.zvec_hip_ivf_load(h, p(cent), nlist, offs64.ctypes.data_as(C.POINTER(C.c_uint64)), p(rows), None) == 0
    assert search() == 0 and np.array_equal(keys, first)
    assert L.zvec_hip_ivf_get_centroids(h, p(out), C.byref(nl)) == 0 and np.array_equal(out, cent)
    assert L.zvec_hip_ivf_label_dev(h, C.c_void_p(d_rows.data_ptr()), n, C.cast(C.c_void_p(d_lab.data_ptr()), C.POINTER(C.c_uint32)), None) == 0
    torch.cuda.synchronize()
    # set_coarse_space again, then new centroids in the rows' space: leaves it too
    assert L.zvec_hip_ivf_set_coarse_space(h, cdim, 0, p(ccent), nlist) == 0
    assert L.zvec_hip_ivf_set_centroids(h, p(cent), nlist) == 0
    assert L.zvec_hip_ivf_get_centroids(h, p(out), C.byref(nl)) == 0 and np.array_equal(out, cent)
    assert L.zvec_hip_ivf_destroy(h) == 0
