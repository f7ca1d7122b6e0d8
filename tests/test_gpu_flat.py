"""GPU parity of the flat scan, through the C ABI (include/zvec_hip.h) — run with -m gpu.

Oracle = oracle/zvec_oracle.c (restated reference loops, pinned in test_oracle_cpu.py).
Bars: integer-valued data => bit-exact scores and ids (outside exact boundary ties);
      real-valued data  => fp32 tolerance stated per metric below.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu

# fp32 tolerances.  IP / cosine: an fp32 FMA chain on the matrix core vs the reference's
# 32-lane SIMD partial sums: |delta| <= 4e-6 * |q||b|.  L2: candidates are SELECTED with |q|^2+|b|^2-2q.b
# (selection band 4e-6 * (|q|^2+|b|^2): only there may ids differ from the reference's) and the winners are
# re-scored directly as sum (q-b)^2, so the REPORTED score agrees within 2e-6 * score (+1e-6 absolute).
L2_BAND = 4e-6
L2_SCORE_RTOL = 2e-6
IP_RTOL = 4e-6


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


def _ramp(n, dim):
    return np.repeat(np.arange(n, dtype=np.float32)[:, None], dim, 1)


def _search(idx, q, k, ctx=None, **kw):
    ctx = ctx or idx.create_context()
    ctx.set_topk(k)
    for name, v in kw.items():
        getattr(ctx, name)(v)
    assert idx.search_impl(q, q.shape[0], ctx) == 0
    return ctx.keys, ctx.scores, ctx.counts, ctx


def test_flat_linear_search_reference_expectations(zv, golden_dir):
    """flat_streamer_test.cc:104-178 (TestLinearSearch), through HipFlatStreamer."""
    g = json.load(open(os.path.join(golden_dir, "scan_known_answers.json")))["flat_linear"]
    n, dim = g["n"], g["dim"]
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    base = _ramp(n, dim)
    for i in range(0, 10):                       # add_impl one by one, then the rest in bulk
        assert st.add_impl(i, base[i]) == 0
    assert st.add_batch(base[10:], np.arange(10, n, dtype=np.uint64)) == 0
    assert st.count() == n
    ctx = st.create_context()
    ctx.set_topk(3)
    q = _ramp(n, dim)
    assert st.search_impl(q, n, ctx) == 0                       # batch of 1000 queries
    assert np.array_equal(ctx.keys[:, 0], np.arange(n, dtype=np.uint64))
    assert not ctx.scores[:, 0].any()
    assert st.search_impl(q + np.float32(0.1), n, ctx) == 0
    last = n - 1
    for i in range(n):
        r = ctx.result(i)
        assert len(r) == 3
        assert r[0].key() == i
        assert r[1].key() == (i - 1 if i == last else i + 1)
        assert r[2].key() == (2 if i == 0 else (i - 2 if i == last else i - 1))
    ctx.set_topk(100)
    assert st.search_bf_impl(np.full((1, dim), 10.1, np.float32), 1, ctx) == 0
    res = ctx.result(0)
    assert len(res) == 100
    for rank, key in g["query_10p1_top100_ranks"].items():
        assert res[int(rank)].key() == key
    # get_vector_by_id (provider->get_vector in TestAddVector :69-101)
    for i in (0, 1, 127, 128, 999):
        assert np.array_equal(st.get_vector_by_id(i), base[i])


def test_flat_filter_reference_expectations(zv, golden_dir):
    """flat_streamer_test.cc:731-801 (TestFilter) + filter-all of flat_searcher_test.cpp:93-209."""
    g = json.load(open(os.path.join(golden_dir, "scan_known_answers.json")))["flat_filter"]
    st = zv.HipFlatStreamer(g["dim"], "SquaredEuclidean")
    assert st.add_batch(_ramp(g["n"], g["dim"])) == 0
    q = np.full((1, g["dim"]), g["query"], np.float32)
    ctx = st.create_context()
    ctx.set_topk(g["topk"])
    assert st.search_impl(q, 1, ctx) == 0
    assert [d.key() for d in ctx.result(0)[:3]] == g["top3_nofilter"] and len(ctx.result(0)) == 10
    ctx.set_filter(lambda key: key in (100, 101))
    assert st.search_impl(q, 1, ctx) == 0
    assert [d.key() for d in ctx.result(0)[:3]] == g["top3_filtered"] and len(ctx.result(0)) == 10
    assert st.search_bf_impl(q, 1, ctx) == 0
    assert [d.key() for d in ctx.result(0)[:3]] == g["top3_filtered"]
    ctx.set_filter(lambda key: True)
    assert st.search_impl(q, 1, ctx) == 0
    assert len(ctx.result(0)) == 0


def test_error_codes(zv):
    st = zv.HipFlatStreamer(16, "SquaredEuclidean")
    ctx = st.create_context()
    q = np.zeros((1, 16), np.float32)
    assert st.search_impl(q, 1, ctx) == zv.IndexError_.InvalidArgument      # topk not set (flat_searcher.cc:194)
    ctx.set_topk(5)
    assert st.search_impl(q, 1, ctx) == 0 and len(ctx.result(0)) == 0       # empty index -> empty result
    assert st.add_batch(np.zeros((2, 15), np.float32)) == zv.IndexError_.InvalidArgument
    assert st.get_vector_by_id(0) is None


@pytest.mark.parametrize("n,dim,nq,k", [(1, 8, 1, 1), (5, 3, 2, 10), (127, 16, 3, 7), (128, 32, 33, 1),
                                        (129, 33, 64, 10), (1000, 128, 65, 10), (4097, 96, 129, 32),
                                        (3000, 768, 7, 10), (2500, 100, 130, 100), (700, 24, 5, 300)])
def test_flat_integer_data_bit_exact(zv, oracle, n, dim, nq, k):
    """integer-valued vectors => every distance is exact in fp32 => scores and ids must be identical."""
    rng = np.random.default_rng(n * 7 + dim)
    hi = 256 if dim <= 128 else 16                      # keep |q|^2+|b|^2 < 2^24
    base = rng.integers(0, hi, (n, dim)).astype(np.float32)
    q = rng.integers(0, hi, (nq, dim)).astype(np.float32)
    keys = rng.permutation(10 * n)[:n].astype(np.uint64)
    for metric, name in ((O.METRIC_L2, "SquaredEuclidean"), (O.METRIC_IP, "InnerProduct")):
        se = zv.HipFlatSearcher(dim, name)
        assert se.load(base, keys) == 0
        gk, gs, gc, _ = _search(se, q, k)
        ok, os_, _, oc = oracle.flat_search(base, q, k, metric, keys=keys)
        tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="%s n=%d d=%d" % (name, n, dim))


@pytest.mark.parametrize("n,dim,nq,k", [(2000, 128, 40, 10), (5000, 768, 16, 10), (1500, 20, 200, 5)])
def test_flat_gaussian_tolerance(zv, oracle, n, dim, nq, k):
    rng = np.random.default_rng(11)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    qn = (q.astype(np.float64) ** 2).sum(1)
    bn = (base.astype(np.float64) ** 2).sum(1).max()
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base) == 0
    gk, gs, gc, _ = _search(se, q, k)
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, rtol=L2_SCORE_RTOL, atol=1e-6, select_band=L2_BAND * (qn + bn),
                         what="L2 gaussian")
    se = zv.HipFlatSearcher(dim, "InnerProduct")
    assert se.load(base) == 0
    gk, gs, gc, _ = _search(se, q, k)
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_IP)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, rtol=IP_RTOL, scale=np.sqrt(qn * bn), what="IP gaussian")
    # cosine: rows = CosineConverter output (normalised + norm), query = CosineReformer output
    cb, cq = oracle.cosine_transform(base), oracle.cosine_transform(q)
    se = zv.HipFlatSearcher(dim + 1, "Cosine")
    assert se.load(cb) == 0
    gk, gs, gc, _ = _search(se, cq, k)
    ok, os_, _, oc = oracle.flat_search(cb, cq, k, O.METRIC_COSINE)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, atol=IP_RTOL, what="cosine gaussian")
    assert np.array_equal(se.get_vector_by_id(3), cb[3])          # the stored norm column comes back


def test_flat_threshold_and_bitmap(zv, oracle):
    rng = np.random.default_rng(13)
    n, dim, nq, k = 3000, 32, 50, 20
    base = rng.integers(0, 64, (n, dim)).astype(np.float32)
    q = rng.integers(0, 64, (nq, dim)).astype(np.float32)
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base) == 0
    for p in (0.5, 0.9, 0.999):
        mask = rng.random(n) < p                                    # True = excluded
        words = O.pack_bits(mask)
        gk, gs, gc, _ = _search(se, q, k, set_exclude_bitset=words)
        ok, os_, _, oc = oracle.flat_search(base, q, k, exclude_bits=words)
        tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="bitmap p=%g" % p)
        assert not mask[gk[gc[:, None] > np.arange(k)[None, :]].astype(np.int64)].any()
    thr = float(np.median(oracle.flat_search(base, q, k)[1][:, k // 2]))
    gk, gs, gc, _ = _search(se, q, k, set_threshold=thr)
    ok, os_, _, oc = oracle.flat_search(base, q, k, threshold=thr)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="rnn threshold")
    assert (gc < k).any() and (gc > 0).any()


def test_flat_incremental_append_and_growth(zv, oracle):
    rng = np.random.default_rng(17)
    dim, k = 48, 10
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    chunks = [rng.integers(0, 100, (m, dim)).astype(np.float32) for m in (1, 63, 64, 300, 1000, 7)]
    q = rng.integers(0, 100, (9, dim)).astype(np.float32)
    have = np.zeros((0, dim), np.float32)
    for c in chunks:
        assert st.add_batch(c) == 0
        have = np.concatenate([have, c])
        gk, gs, gc, _ = _search(st, q, k)
        ok, os_, _, oc = oracle.flat_search(have, q, k)
        tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="append n=%d" % len(have))


def test_merge_topk_matches_oracle(zv, oracle):
    from zvec_amd.index import merge_topk, IndexContext
    rng = np.random.default_rng(19)
    nparts, nq, k = 8, 40, 10
    scores = np.sort(rng.integers(0, 1000, (nparts, nq, k)).astype(np.float32), -1)
    keys = rng.integers(0, 10 ** 9, (nparts, nq, k)).astype(np.uint64)
    counts = rng.integers(0, k + 1, (nparts, nq)).astype(np.uint32)
    ctx = IndexContext(0)
    gk, gs, gc = merge_topk(ctx, keys, scores, counts, k)
    ok, os_, oc = oracle.merge_topk(keys, scores, counts, k)
    assert np.array_equal(gc, oc)
    for q in range(nq):
        assert np.array_equal(gs[q, :oc[q]], os_[q, :oc[q]])
        assert np.array_equal(gk[q, :oc[q]], ok[q, :oc[q]])


def test_search_bf_by_p_keys(zv, oracle):
    """FlatStreamer::search_bf_by_p_keys_impl (flat_streamer.cc:346-389): only the listed keys compete;
    the oracle answer is a flat scan with every other position excluded."""
    rng = np.random.default_rng(37)
    n, dim, nq, k = 4000, 40, 12, 10
    base = rng.integers(0, 90, (n, dim)).astype(np.float32)
    q = rng.integers(0, 90, (nq, dim)).astype(np.float32)
    keys = (rng.permutation(5 * n)[:n] + 7).astype(np.uint64)
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    assert st.add_batch(base, keys) == 0
    ctx = st.create_context()
    ctx.set_topk(k)
    p_keys = []
    for i in range(nq):
        m = int(rng.integers(0, 300))
        sel = rng.choice(n, m, replace=False)
        lst = keys[sel].tolist() + [10 ** 12 + i]           # plus one unknown key: skipped
        p_keys.append(lst)
    assert st.search_bf_by_p_keys_impl(q, p_keys, nq, ctx) == 0
    key2pos = {int(kk): i for i, kk in enumerate(keys)}
    for i in range(nq):
        mask = np.ones(n, bool)
        mask[[key2pos[kk] for kk in p_keys[i][:-1]]] = False
        ok, os_, _, oc = oracle.flat_search(base, q[i:i + 1], k, keys=keys, exclude_bits=O.pack_bits(mask))
        tie_tolerant_compare(ctx.keys[i:i + 1], ctx.scores[i:i + 1], ctx.counts[i:i + 1], ok, os_, oc, what="p_keys q%d" % i)
    # with a filter on top (TestFilter's p_keys leg, flat_streamer_test.cc:752-760)
    ctx.set_filter(lambda key: key % 2 == 0)
    assert st.search_bf_by_p_keys_impl(q, p_keys, nq, ctx) == 0
    for i in range(nq):
        assert all(d.key() % 2 == 1 for d in ctx.result(i))


def test_concurrent_contexts(zv, oracle):
    """search_* are const and called concurrently from many threads, each with its own context
    (SURVEY §8(b) threading; flat_streamer_test.cc:509-729 multi-thread search tests)."""
    import threading
    rng = np.random.default_rng(53)
    n, dim, k = 20000, 64, 10
    base = rng.integers(0, 120, (n, dim)).astype(np.float32)
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base) == 0
    qs = [rng.integers(0, 120, (17 + 3 * t, dim)).astype(np.float32) for t in range(6)]
    expect = [oracle.flat_search(base, q, k, threads=2) for q in qs]
    errs = []

    def worker(t):
        try:
            ctx = se.create_context()
            ctx.set_topk(k)
            for _ in range(5):
                assert se.search_impl(qs[t], qs[t].shape[0], ctx) == 0
                ok, os_, _, oc = expect[t]
                tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="thread %d" % t)
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs


def test_add_between_searches_keeps_earlier_rows(zv, oracle):
    """streamer semantics: rows appended after a search are visible to the next one, earlier positions and
    keys are stable across capacity growth (flat_streamer_test.cc TestAddAndSearch)."""
    rng = np.random.default_rng(59)
    dim, k = 32, 5
    st = zv.HipFlatStreamer(dim, "InnerProduct")
    ctx = st.create_context()
    ctx.set_topk(k)
    have = np.zeros((0, dim), np.float32)
    q = rng.integers(-20, 20, (8, dim)).astype(np.float32)
    for step in range(12):
        add = rng.integers(-20, 20, (int(rng.integers(1, 400)), dim)).astype(np.float32)
        assert st.add_batch(add) == 0
        have = np.concatenate([have, add])
        assert st.search_impl(q, 8, ctx) == 0
        ok, os_, _, oc = oracle.flat_search(have, q, k, O.METRIC_IP)
        tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="step %d" % step)
        assert np.array_equal(st.get_vector_by_id(0), have[0])


@pytest.mark.parametrize("keep", [0.3, 0.02, 0.0])
def test_sparse_keep_set_is_compacted_and_exact(zv, oracle, keep):
    """bitmap-gated scan with a minority keep-set (>= 64k rows): the kept rows are compacted on the GPU and
    scanned densely; the answer must equal the gated scan's (oracle with the same exclusion bits)."""
    rng = np.random.default_rng(61)
    n, dim, nq, k = 100_000, 48, 24, 10
    base = rng.integers(0, 100, (n, dim)).astype(np.float32)
    q = rng.integers(0, 100, (nq, dim)).astype(np.float32)
    keys = (rng.permutation(3 * n)[:n]).astype(np.uint64)
    excl = rng.random(n) >= keep
    words = O.pack_bits(excl)
    for name, metric in (("SquaredEuclidean", O.METRIC_L2), ("InnerProduct", O.METRIC_IP)):
        se = zv.HipFlatSearcher(dim, name)
        assert se.load(base, keys) == 0
        ctx = se.create_context()
        ctx.set_topk(k)
        ctx.set_exclude_bitset(words)
        assert se.search_impl(q, nq, ctx) == 0
        ok, os_, _, oc = oracle.flat_search(base, q, k, metric, keys=keys, exclude_bits=words, threads=4)
        tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="compacted keep=%g %s" % (keep, name))


@pytest.mark.parametrize("k", [600, 2000])
def test_flat_large_topk(zv, oracle, k):
    """topk beyond the LDS-resident lists: dense-score path + whole-row selection (the reference's heap has no
    limit; its tests use topk up to the corpus size)."""
    rng = np.random.default_rng(67)
    n, dim, nq = 5000, 24, 6
    base = rng.integers(0, 200, (n, dim)).astype(np.float32)
    q = rng.integers(0, 200, (nq, dim)).astype(np.float32)
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k)
    assert np.array_equal(ctx.counts, oc)
    assert np.array_equal(ctx.scores, os_)              # integer data: exact scores, rank by rank
    for i in range(nq):                                 # ids equal wherever the score is unique in the list
        s_ = os_[i]
        uniq = np.concatenate([[True], np.diff(s_) != 0]) & np.concatenate([np.diff(s_) != 0, [True]])
        assert np.array_equal(ctx.keys[i][uniq], ok[i][uniq])


@pytest.mark.parametrize("dtype,column_major,n,dim", [(np.float32, True, 100, 12), (np.float32, False, 70, 9), (np.float16, True, 64, 10),
                                                        (np.float16, True, 31, 5)])
def test_flat_load_features_segment(zv, dtype, column_major, n, dim):
    """the features segment of a dumped flat index (flat_builder.cc:186-276): full 32-row blocks transposed for a
    column-major index, a row-major remainder, padding to 32 bytes — restated writer, layout parity unpinned."""
    rng = np.random.default_rng(n)
    base = rng.integers(-9, 10, (n, dim)).astype(dtype)
    keys = rng.permutation(5 * n)[:n].astype(np.uint64)
    blob = bytearray()
    for b0 in range(0, n, 32):
        blk = base[b0:b0 + 32]
        blob += (np.ascontiguousarray(blk.T) if (column_major and blk.shape[0] == 32) else blk).tobytes()
    blob += b"\0" * ((len(blob) + 31) // 32 * 32 - len(blob))
    dt = "fp16" if dtype == np.float16 else "fp32"
    a = zv.HipFlatSearcher(dim, "SquaredEuclidean", dtype=dt)
    assert a.load_features(bytes(blob), n, column_major=column_major, keys=keys) == 0
    b = zv.HipFlatSearcher(dim, "SquaredEuclidean", dtype=dt)
    assert b.load(base, keys) == 0
    for pos in (0, 1, 30, n // 2, n - 1):
        assert np.array_equal(a.get_vector_by_id(pos).view(np.uint8), base[pos].view(np.uint8))
    q = rng.integers(-9, 10, (5, dim)).astype(dtype)
    ra, rb = _search(a, q, 7), _search(b, q, 7)
    assert all(np.array_equal(x, y) for x, y in zip(ra[:3], rb[:3]))
    assert a.load_features(bytes(blob)[: n * dim * base.itemsize - 1], n, column_major=column_major) == zv.IndexError_.InvalidArgument


def test_fetch_vector_results_carry_stored_rows(zv):
    """IndexContext::set_fetch_vector (index_context.h:139, index.cc:635-647): documents come back with their vectors;
    flat (fp32, cosine rows with the norm column) and IVF, through one gather launch per search."""
    rng = np.random.default_rng(4)
    n, dim, nq, k = 3000, 19, 6, 5
    base = rng.integers(-9, 10, (n, dim)).astype(np.float32)
    keys = rng.permutation(7 * n)[:n].astype(np.uint64)
    key2row = {int(kk): i for i, kk in enumerate(keys)}
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base, keys) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    q = base[:nq] + 0.25
    assert se.search_impl(q, nq, ctx) == 0 and ctx.result(0)[0].vector() is None
    ctx.set_fetch_vector(True)
    assert se.search_impl(q, nq, ctx) == 0
    for qi in range(nq):
        assert len(ctx.result(qi)) == k
        for doc in ctx.result(qi):
            assert np.array_equal(doc.vector(), base[key2row[doc.key()]])
    assert np.array_equal(se.get_vectors_by_ids([5, 0, n - 1, 5]), base[[5, 0, n - 1, 5]])
    with pytest.raises(zv._lib.ZvecHipError):
        se.get_vectors_by_ids([n])
    ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.5, brute_force_threshold=10)
    assert ivf.build(base, 8, keys=keys, kmeans_iters=3) == 0
    ictx = ivf.create_context()
    ictx.set_topk(k)
    ictx.set_fetch_vector(True)
    assert ivf.search_impl(q, nq, ictx) == 0
    for qi in range(nq):
        for doc in ictx.result(qi):
            assert np.array_equal(doc.vector(), base[key2row[doc.key()]])


@pytest.mark.parametrize("metric", ["SquaredEuclidean", "InnerProduct"])
def test_seeded_bounds_survive_large_common_offset(zv, oracle, metric):
    """Seeded admission bounds (the main scan starts every query's bound at the k-th score of a 4096-row prefix scan):
    with a large common offset the norms dwarf the distances, so the selection score of one row differs between the
    prefix scan and the main scan (different tile shapes) by far more than 1e-6 of the SCORE.  All true neighbours sit
    in the prefix (rows 0..k-1): a bound that is only score-relative drops them and the query comes back short.
    Batch of 128 over a streamed base => the wide 128x128 path."""
    rng = np.random.default_rng(31)
    n, dim, nq, k = 300_000, 64, 128, 10
    base = (1000.0 + rng.integers(0, 8, (n, dim))).astype(np.float32)
    c = (1000.0 + rng.integers(0, 8, dim)).astype(np.float32)
    if metric == "InnerProduct":
        c += 8.0                                           # the largest inner products: rows closest to the biggest vector
    for i in range(k):                                     # the k true neighbours, all inside the 4096-row prefix
        base[i] = c
        base[i, i % dim] += 1.0 if metric == "InnerProduct" else (i % 2) * 1.0
    q = np.repeat(c[None, :], nq, 0)
    q[np.arange(nq), rng.integers(0, dim, nq)] += 1.0      # 128 different queries around c
    flat = zv.HipFlatSearcher(dim, metric)
    assert flat.load(base) == 0
    ctx = flat.create_context()
    ctx.set_topk(k)
    assert flat.search_impl(q, nq, ctx) == 0
    assert (ctx.counts == k).all(), ctx.counts.min()
    assert all(set(ctx.keys[i].tolist()) == set(range(k)) for i in range(nq))
    m = O.METRIC_IP if metric == "InnerProduct" else O.METRIC_L2
    ok, os_, _, oc = oracle.flat_search(base, q[:8], k, m, threads=8)
    qn = (q[:8].astype(np.float64) ** 2).sum(1)
    band = 4e-6 * (qn + (base[:k].astype(np.float64) ** 2).sum(1).max())
    if metric == "InnerProduct":
        tie_tolerant_compare(ctx.keys[:8], ctx.scores[:8], ctx.counts[:8], ok, os_, oc, rtol=4e-6, scale=float(qn.max()), what="offset ip")
    else:
        tie_tolerant_compare(ctx.keys[:8], ctx.scores[:8], ctx.counts[:8], ok, os_, oc, rtol=2e-6, atol=1e-6, select_band=band, what="offset l2")


def _column_blob(base):
    """FlatBuilder::write_column_index (flat_builder.cc:188-276): full 32-row blocks transposed, row-major remainder"""
    blob = bytearray()
    for b0 in range(0, base.shape[0], 32):
        blk = base[b0:b0 + 32]
        blob += (np.ascontiguousarray(blk.T) if blk.shape[0] == 32 else blk).tobytes()
    return bytes(blob)


@pytest.mark.parametrize("metric", ["SquaredEuclidean", "InnerProduct"])
@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_flat_vs_column_major_oracle(zv, oracle, metric, dtype):
    """the reference's dense path — a column-major index (d <= 512) scanned in 32 x K tiles by the
    M x N block kernels (flat_searcher_context.h:682-752, TransposeQueries flat_utility.h:106-158) — restated in the
    oracle on top of its pinned block kernels (zo_flat_search_column_t).  The HIP path loads the same column-major
    features segment and must agree: bit-exact on integer data (every kernel is exact there), within the stated fp32
    tolerance on real data; batches of 1..45 so that every query-group width 32/16/8/4/2/1 and the left-over rows
    (n % 32 != 0) are exercised, with and without a filter."""
    npdt = np.float16 if dtype == "fp16" else np.float32
    m = O.METRIC_IP if metric == "InnerProduct" else O.METRIC_L2
    rng = np.random.default_rng(77)
    for n, dim, k in ((1000 + 13, 64, 10), (95, 300, 7), (4096 + 31, 512, 10)):
        keys = rng.permutation(4 * n)[:n].astype(np.uint64)
        for exact in (True, False):
            base = (rng.integers(-8, 9, (n, dim)) if exact else rng.standard_normal((n, dim))).astype(npdt)
            se = zv.HipFlatSearcher(dim, metric, dtype=dtype)
            assert se.load_features(_column_blob(base), n, column_major=True, keys=keys) == 0
            ctx = se.create_context()
            ctx.set_topk(k)
            for nq in (1, 2, 5, 32, 45):
                q = (rng.integers(-8, 9, (nq, dim)) if exact else rng.standard_normal((nq, dim))).astype(npdt)
                for filt in (False, True):
                    bits = None
                    if filt:
                        mask = rng.random(n) < 0.4
                        mask[32:64] = True                   # one whole 32-row block filtered out (block_mask == 0 skip)
                        bits = O.pack_bits(mask)
                        ctx.set_exclude_bitset(bits)
                    else:
                        ctx.reset_filter()
                    assert se.search_impl(q, nq, ctx) == 0
                    ok, os_, _, oc = oracle.flat_search_column(base, q, k, m, keys=keys, exclude_bits=bits)
                    what = "column %s %s n=%d nq=%d exact=%s filt=%s" % (metric, dtype, n, nq, exact, filt)
                    if exact:
                        tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what=what)
                    else:
                        qn = (q.astype(np.float64) ** 2).sum(1)
                        bn = float((base.astype(np.float64) ** 2).sum(1).max())
                        if m == O.METRIC_IP:
                            tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=4e-6,
                                                 scale=np.sqrt(qn * bn), what=what)
                        else:
                            tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=2e-6, atol=1e-6,
                                                 select_band=4e-6 * (qn + bn), what=what)


@pytest.mark.parametrize("metric,dtype", [("SquaredEuclidean", "fp32"), ("InnerProduct", "fp32"), ("Cosine", "fp32"),
                                          ("SquaredEuclidean", "fp16")])
def test_batch_distance_one_to_many(zv, oracle, metric, dtype):
    """IndexMetric::batch_distance — one query against scattered rows.  BaseDistance::ComputeBatch
    (distance_batch.h:29-49) is a loop of the 1x1 kernel for L2 / IP, which the oracle restates exactly; the cosine
    variant sums in another order (inner_product_distance_batch_impl.h): restated as oracle.cosine_batch (pinned bit for
    bit to the reference's ComputeBatch, tests/test_oracle_cpu.py) and compared here within the same stated tolerance."""
    rng = np.random.default_rng(91)
    n, d = 5000, 96
    npdt = np.float16 if dtype == "fp16" else np.float32
    m = {"SquaredEuclidean": O.METRIC_L2, "InnerProduct": O.METRIC_IP, "Cosine": O.METRIC_COSINE}[metric]
    raw = rng.standard_normal((n, d)).astype(np.float32)
    qraw = rng.standard_normal(d).astype(np.float32)
    if metric == "Cosine":
        base, q = oracle.cosine_transform(raw), oracle.cosine_transform(qraw)[0]
    else:
        base, q = raw.astype(npdt), qraw.astype(npdt)
    dim = base.shape[1]
    se = zv.HipFlatSearcher(dim, metric, dtype=dtype)
    assert se.load(base) == 0
    pos = rng.choice(n, 777, replace=False).astype(np.uint32)
    pos[5] = n + 3                                           # out of range -> +inf
    got = se.batch_distance(q, pos)
    assert np.isinf(got[5])
    fn = oracle.dist16 if dtype == "fp16" else oracle.dist
    if metric == "Cosine":
        want = np.full(pos.size, np.inf, np.float32)
        inr = pos < n
        want[inr] = oracle.cosine_batch(base[pos[inr]], q)
    else:
        want = np.array([fn(m, base[p], q) if p < n else np.inf for p in pos], np.float32)
    ok = np.isfinite(want)
    qn, bn = float((q[:d].astype(np.float64) ** 2).sum()), (base[pos[ok], :d].astype(np.float64) ** 2).sum(1)
    tol = 2e-6 * np.abs(want[ok]) + 1e-6 if m == O.METRIC_L2 else 4e-6 * np.sqrt(qn * bn)
    assert np.all(np.abs(got[ok] - want[ok]) <= tol)
    # integer data: bit-exact
    ib = rng.integers(-9, 10, (300, 24)).astype(np.float32)
    iq = rng.integers(-9, 10, 24).astype(np.float32)
    s2 = zv.HipFlatSearcher(24, "SquaredEuclidean")
    assert s2.load(ib) == 0
    p2 = rng.permutation(300).astype(np.uint32)
    assert np.array_equal(s2.batch_distance(iq, p2), np.array([oracle.dist(O.METRIC_L2, ib[p], iq) for p in p2], np.float32))


def test_boundary_a_score_conventions(zv, oracle):
    """tests/core/interface/index_interface_test.cc:1002-1100 (IndexInterface.Score): two documents (3,4,5) / (1,20,3),
    query (1,2,3), tolerance 1e-2 — InnerProduct reports +dot, Cosine 1 - cos, L2 the squared distance.  Boundary B
    (this library) returns the metric kernels' scores; the steps above it are applied here as the product applies them:
    CosineConverter / CosineReformer on rows and query (oracle.cosine_transform), metric->normalize on the scores
    (zvec_amd.index.normalize_score)."""
    from zvec_amd.index import normalize_score
    docs = {2345: np.array([3.0, 4.0, 5.0], np.float32), 5432: np.array([1.0, 20.0, 3.0], np.float32)}
    q = np.array([[1.0, 2.0, 3.0]], np.float32)
    keys = np.array(list(docs), np.uint64)
    rows = np.stack([docs[int(k_)] for k_ in keys])
    ip = lambda a, b: float((a * b).sum())                                                     # noqa: E731
    want = {"InnerProduct": lambda v: ip(v, q[0]),
            "Cosine": lambda v: 1 - ip(v, q[0]) / (np.sqrt(ip(v, v)) * np.sqrt(ip(q[0], q[0]))),
            "SquaredEuclidean": lambda v: float(((v - q[0]) ** 2).sum())}
    for metric, fn in want.items():
        if metric == "Cosine":
            se = zv.HipFlatStreamer(4, metric)
            assert se.add_batch(oracle.cosine_transform(rows), keys) == 0
            qq = oracle.cosine_transform(q)
        else:
            se = zv.HipFlatStreamer(3, metric)
            assert se.add_batch(rows, keys) == 0
            qq = q
        ctx = se.create_context()
        ctx.set_topk(10)
        assert se.search_impl(qq, 1, ctx) == 0
        r = ctx.result(0)
        assert len(r) == 2
        for d in r:
            assert abs(normalize_score(metric, d.score()) - fn(docs[d.key()])) < 1e-2, (metric, d.key(), d.score())
        # best first: IP -> the larger dot product; cosine / L2 -> the smaller distance
        best = max(docs, key=lambda k_: fn(docs[k_])) if metric == "InnerProduct" else min(docs, key=lambda k_: fn(docs[k_]))
        assert r[0].key() == best


def _check_groups(ctx, qi, want, ids, full=None, rtol=0.0):
    """ctx.group_result(qi) against the oracle's [(group number, [(key, score, pos)])].  The reference orders groups with
    an unstable sort on the best score alone and its heaps keep the first seen of tied documents, so: the sequences of
    best scores are equal; a group whose best score is not shared with another listed group sits at the same place;
    the groups tied at the LAST place may be any of the groups with that best score (`full`: the oracle's answer with
    every group listed, to know them); inside a group the scores are equal and the documents strictly better than its
    last score are the same."""
    got = ctx.group_result(qi)
    assert len(got) == len(want), "number of groups of query %d" % qi
    wbest = [docs[0][1] if docs else None for _, docs in want]
    by_id = {ids[g]: docs for g, docs in (full if full is not None else want)}
    for i, g in enumerate(got):
        docs = by_id.get(g.group_id())
        assert docs is not None, "query %d: group %r is not a candidate" % (qi, g.group_id())
        if wbest[i] is not None and docs:
            assert abs(docs[0][1] - wbest[i]) <= rtol * abs(wbest[i]), "query %d place %d: best score" % (qi, i)
        gs = np.array([d.score() for d in g.docs()], np.float32)
        ws = np.array([d[1] for d in docs], np.float32)
        assert gs.shape == ws.shape and np.all(np.abs(gs - ws) <= rtol * np.abs(ws)), "query %d group %r scores" % (qi, g.group_id())
        if len(docs):
            last = ws[-1] * (1 - rtol)
            assert {d.key() for d in g.docs() if d.score() < last} <= {d[0] for d in docs}
            assert {d[0] for d in docs if d[1] < last} <= {d.key() for d in g.docs()}
    assert len({g.group_id() for g in got}) == len(got)


def test_group_by_reference_known_answers(zv, oracle):
    """flat_streamer_test.cc TestGroup (:929-1027) through the GPU path: group_num 5, group_topk 20, no set_topk; the
    p_keys leg must give the keys 10, 9, 8, 7, 6 as the first documents of its five groups."""
    n, dim = 5000, 16
    base = np.repeat((np.arange(n, dtype=np.float32) / np.float32(10.0))[:, None], dim, axis=1).astype(np.float32)
    q = np.full((1, dim), np.float32(n // 2) * np.float32(1.0) / np.float32(10) + np.float32(0.1), np.float32)
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    assert st.add_batch(base, np.arange(n, dtype=np.uint64)) == 0
    ctx = st.create_context()
    ctx.set_group_params(5, 20)
    assert st.search_impl(q, 1, ctx) == zv.IndexError_.InvalidArgument     # "Invalid group-by function"
    ctx.set_group_by(lambda key: "g_%d" % (key // 10 % 10))
    assert st.search_impl(q, 1, ctx) == 0
    res = ctx.group_result(0)
    assert len(res) == 5 and all(len(g.docs()) > 0 for g in res)
    want = oracle.flat_group_search(base, q, (np.arange(n) // 10) % 10, 5, 20)[0]
    # (rows i / 10 are not exactly representable: scores within the flat path's fp32 tolerance of the oracle's)
    _check_groups(ctx, 0, want, ["g_%d" % g for g in range(10)], rtol=2e-6)
    pk = st.create_context()
    pk.set_group_params(5, 20)
    pk.set_group_by(lambda key: "g_%d" % (key % 10))
    assert st.search_bf_by_p_keys_impl(q, [[4, 3, 2, 1, 5, 6, 7, 8, 9, 10]], 1, pk) == 0
    res = pk.group_result(0)
    assert len(res) == 5
    for i, g in enumerate(res):
        assert len(g.docs()) > 0 and g.docs()[0].key() == 10 - i and g.group_id() == "g_%d" % ((10 - i) % 10)


@pytest.mark.parametrize("metric,dtype", [("SquaredEuclidean", "fp32"), ("InnerProduct", "fp32"), ("SquaredEuclidean", "fp16")])
def test_group_by_matches_oracle(zv, oracle, metric, dtype):
    """group-by search on integer data (exact distances): groups, their order and their documents equal the restated
    reference loop — with a filter, a radius, groups smaller than group_topk, fewer groups than group_num for one query
    set, several queries per call, fetch_vector"""
    rng = np.random.default_rng(77)
    n, dim, nq = 6000, 48, 37
    npdt = np.float16 if dtype == "fp16" else np.float32
    base = rng.integers(-6, 7, (n, dim)).astype(npdt)
    q = rng.integers(-6, 7, (nq, dim)).astype(npdt)
    keys = (rng.permutation(3 * n)[:n] + 5).astype(np.uint64)
    ngroups = 41
    gid = lambda key: "grp%d" % (key % ngroups if key % 7 else 1000 + key % 3)     # 41 big groups + 3 small ones
    st = zv.HipFlatStreamer(dim, metric, dtype=dtype)
    assert st.add_batch(base, keys) == 0
    m = O.METRIC_L2 if metric == "SquaredEuclidean" else O.METRIC_IP
    names, number_of = [], {}
    of = np.zeros(n, np.int64)
    for i, k in enumerate(keys):
        g = gid(int(k))
        if g not in number_of:
            number_of[g] = len(names)
            names.append(g)
        of[i] = number_of[g]
    for gnum, gk, filt, thr in [(6, 9, None, None), (50, 3, lambda key: key % 3 == 0, None), (4, 70, None, "radius"), (1, 1, None, None)]:
        ctx = st.create_context()
        ctx.set_group_params(gnum, gk)
        ctx.set_group_by(gid)
        exb = None
        if filt is not None:
            ctx.set_filter(filt)
            exb = O.pack_bits(np.array([filt(int(k)) for k in keys]))
        radius = zv.index.FLT_MAX
        if thr is not None:
            probe = oracle.flat_search(base, q[:1], 300, m)[1][0]
            radius = float(probe[150])
            ctx.set_threshold(radius)
        assert st.search_impl(q, nq, ctx) == 0
        want = oracle.flat_group_search(base, q, of, gnum, gk, m, keys=keys, threshold=radius, exclude_bits=exb)
        full = oracle.flat_group_search(base, q, of, len(names), gk, m, keys=keys, threshold=radius, exclude_bits=exb)
        for qi in range(nq):
            _check_groups(ctx, qi, want[qi], names, full[qi])
    # p_keys + groups + fetch_vector
    ctx = st.create_context()
    ctx.set_group_params(3, 4)
    ctx.set_group_by(gid)
    ctx.set_fetch_vector(True)
    p_pos = [rng.choice(n, int(rng.integers(1, 200)), replace=False) for _ in range(nq)]
    assert st.search_bf_by_p_keys_impl(q, [[int(keys[p]) for p in pp] + [10 ** 12] for pp in p_pos], nq, ctx) == 0
    want = oracle.flat_group_search(base, q, of, 3, 4, m, keys=keys, candidates=p_pos)
    full = oracle.flat_group_search(base, q, of, len(names), 4, m, keys=keys, candidates=p_pos)
    key2pos = {int(k): i for i, k in enumerate(keys)}
    for qi in range(nq):
        _check_groups(ctx, qi, want[qi], names, full[qi])
        for g in ctx.group_result(qi):
            for d in g.docs():
                assert np.array_equal(d.vector(), base[key2pos[d.key()]])


def test_group_by_gaussian_tolerance_and_large_batch(zv, oracle):
    """Gaussian data, 300 queries (several query tiles of the dense pass): scores within the flat path's tolerance of
    the oracle's, group order equal wherever the best scores are separated by more than that tolerance"""
    rng = np.random.default_rng(5)
    n, dim, nq = 20000, 96, 300
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    assert st.add_batch(base, np.arange(n, dtype=np.uint64)) == 0
    ctx = st.create_context()
    ctx.set_group_params(8, 5)
    ctx.set_group_by(lambda key: key % 300)
    assert st.search_impl(q, nq, ctx) == 0
    of = np.arange(n) % 300
    for qi in list(range(0, nq, 23)):
        want = oracle.flat_group_search(base, q[qi:qi + 1], of, 8, 5)[0]
        got = ctx.group_result(qi)
        assert len(got) == len(want) == 8
        wb = np.array([docs[0][1] for _, docs in want], np.float32)
        gb = np.array([g.docs()[0].score() for g in got], np.float32)
        np.testing.assert_allclose(gb, wb, rtol=4e-6)
        for g, (wg, docs) in zip(got, want):
            if g.group_id() == wg:
                np.testing.assert_allclose([d.score() for d in g.docs()], [d[1] for d in docs], rtol=4e-6)
                assert [d.key() for d in g.docs()] == [d[0] for d in docs]


def test_group_by_consistent_with_plain_topk_at_size(zv):
    """size-independent property at 400k x 128, 128 queries, 997 groups: with every group listed and group_topk = k the
    grouped answer contains the plain top-k (same scores), the first group's first document is the plain top-1, every
    document sits in its group, groups come in ascending order of their best score, and no group is listed twice"""
    rng = np.random.default_rng(12)
    n, dim, nq, k, ng = 400000, 128, 128, 10, 997
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    st = zv.HipFlatStreamer(dim, "InnerProduct")
    assert st.add_batch(base, np.arange(n, dtype=np.uint64)) == 0
    plain = st.create_context()
    plain.set_topk(k)
    assert st.search_impl(q, nq, plain) == 0
    ctx = st.create_context()
    ctx.set_group_params(ng, k)
    ctx.set_group_by(lambda key: key % ng)
    assert st.search_impl(q, nq, ctx) == 0
    for qi in range(nq):
        groups = ctx.group_result(qi)
        assert len(groups) == ng and len({g.group_id() for g in groups}) == ng
        best = [g.docs()[0].score() for g in groups]
        assert all(best[i] <= best[i + 1] for i in range(ng - 1))
        docs = {d.key(): d.score() for g in groups for d in g.docs()}
        assert all(d.key() % ng == g.group_id() for g in groups for d in g.docs())
        top = plain.result(qi)
        assert groups[0].docs()[0].key() == top[0].key()
        for d in top:
            assert d.key() in docs and docs[d.key()] == d.score()


def test_add_with_id_holes_and_overwrites(zv, oracle):
    """IndexStreamer::add_with_id_impl as core_interface::Index::_dense_add drives it.  flat_streamer_test.cc
    TestAddAndSearchWithID (:1038-1117): the even ids first — the odd positions are holes no search may return — then the
    odd ids land on those holes; row i = (i, ..., i), query i + 0.1: the linear search must put key i first.  On top:
    overwriting a live row replaces what is searched, p_keys and group-by searches skip holes, a filter composes with
    the hole set, and the result always equals the oracle over the live rows."""
    dim, cnt = 16, 20000
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    ctx = st.create_context()
    ctx.set_topk(200)
    rows = np.repeat(np.arange(cnt, dtype=np.float32)[:, None], dim, axis=1)
    ev = np.arange(0, cnt, 2)
    assert st.add_with_id_batch(ev[:3000], rows[ev[:3000]]) == 0
    for i in ev[3000:3040]:                                  # one at a time, as the product calls it
        assert st.add_with_id_impl(int(i), rows[i]) == 0
    assert st.add_with_id_batch(ev[3040:], rows[ev[3040:]]) == 0
    assert st.count() == cnt - 1 and st.holes() == cnt // 2 - 1
    qs = np.arange(0, cnt, 100, dtype=np.float32)
    q = np.repeat(qs[:, None], dim, axis=1) + np.float32(0.1)
    assert st.search_impl(q, q.shape[0], ctx) == 0
    for j, i in enumerate(range(0, cnt, 100)):
        res = ctx.result(j)
        assert len(res) == 200 and res[0].key() == i and all(d.key() % 2 == 0 for d in res)   # no hole (odd position) comes back
    # (rows up to (20000, ...): the norms dwarf the distances, so the documented L2 selection band is wide here — the
    # reference's own test asks for the first key and 80 % recall only)
    band = 4e-6 * ((q.astype(np.float64) ** 2).sum(1) + float((rows[-1].astype(np.float64) ** 2).sum()))
    ok, os_, _, oc = oracle.flat_search(rows[ev], q, 200, keys=ev.astype(np.uint64))
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=2e-6, atol=1e-3, what="holes", select_band=band)
    # a filter on top of the holes; p_keys naming holes; group-by
    ctx.set_filter(lambda key: key % 4 == 0)
    assert st.search_impl(q[:5], 5, ctx) == 0
    assert all(d.key() % 4 == 2 for qi in range(5) for d in ctx.result(qi))
    ctx.reset_filter()
    pk = st.create_context()
    pk.set_topk(4)
    assert st.search_bf_by_p_keys_impl(q[:1], [[3, 2, 1, 4, 5, 6]], 1, pk) == 0
    assert [d.key() for d in pk.result(0)] == [2, 4, 6]
    g = st.create_context()
    g.set_group_params(3, 2)
    g.set_group_by(lambda key: key % 3)
    assert st.search_impl(q[:1], 1, g) == 0
    assert all(d.key() % 2 == 0 for grp in g.group_result(0) for d in grp.docs()) and len(g.group_result(0)) == 3
    # the odd ids land on the holes
    od = np.arange(1, cnt, 2)
    assert st.add_with_id_batch(od, rows[od]) == 0
    assert st.count() == cnt and st.holes() == 0
    assert st.search_bf_impl(q, q.shape[0], ctx) == 0
    for j, i in enumerate(range(0, cnt, 100)):
        res = ctx.result(j)
        assert len(res) == 200 and res[0].key() == i
    ok, os_, _, oc = oracle.flat_search(rows, q, 200)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, rtol=2e-6, atol=1e-3, what="filled", select_band=band)
    # overwrite a live row: the old vector is gone, the new one is found under the same key
    far = np.full((1, dim), 1.0e6, np.float32)
    assert st.add_with_id_impl(700, far[0]) == 0
    ctx.set_topk(3)
    assert st.search_impl(q[7:8], 1, ctx) == 0               # query 700.1
    assert [d.key() for d in ctx.result(0)] == [701, 699, 702]
    assert st.search_impl(far, 1, ctx) == 0
    assert ctx.result(0)[0].key() == 700 and ctx.result(0)[0].score() == 0.0
    assert np.array_equal(st.get_vector_by_id(700), far[0])


@pytest.mark.parametrize("metric,dtype", [("InnerProduct", "fp16"), ("Cosine", "fp32"), ("SquaredEuclidean", "fp32")])
def test_add_with_id_random_order_equals_oracle(zv, oracle, metric, dtype):
    """add_with_id in a random order with repeats (later writes win), chunks of mixed appends / gap fills / overwrites,
    fp16 and cosine rows included: after every chunk the search equals the oracle over the rows that are live"""
    rng = np.random.default_rng(404)
    n, d, nq, k = 3000, 40, 25, 12
    npdt = np.float16 if dtype == "fp16" else np.float32
    m = {"SquaredEuclidean": O.METRIC_L2, "InnerProduct": O.METRIC_IP, "Cosine": O.METRIC_COSINE}[metric]
    raw = rng.integers(-7, 8, (2 * n, d)).astype(np.float32)
    qraw = rng.integers(-7, 8, (nq, d)).astype(np.float32)
    if metric == "Cosine":
        pool, q = oracle.cosine_transform(raw + 0.25), oracle.cosine_transform(qraw + 0.25)
    else:
        pool, q = raw.astype(npdt), qraw.astype(npdt)
    st = zv.HipFlatStreamer(pool.shape[1], metric, dtype=dtype)
    ctx = st.create_context()
    ctx.set_topk(k)
    live = {}                                               # id -> row of `pool` currently stored
    order = rng.integers(0, n, 2 * n)                       # ids with repeats; write j stores pool[j]
    for c0 in range(0, 2 * n, 750):
        ids = order[c0:c0 + 750]
        assert st.add_with_id_batch(ids, pool[c0:c0 + 750]) == 0
        for j, i in enumerate(ids):
            live[int(i)] = c0 + j
        top = max(live) + 1
        assert st.count() == top and st.holes() == top - len(live)
        lid = np.array(sorted(live), np.int64)
        rows = pool[[live[int(i)] for i in lid]]
        assert st.search_impl(q, nq, ctx) == 0
        ok, os_, _, oc = oracle.flat_search(rows, q, k, m, keys=lid.astype(np.uint64))
        if metric == "Cosine":
            tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, atol=4e-6, what="put cosine chunk %d" % c0)
        else:
            tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="put chunk %d" % c0)
    i0 = int(lid[7])
    assert np.array_equal(st.get_vector_by_id(i0), pool[live[i0]])


def test_c_example_program_runs(zv):
    """examples/flat_search.c: the boundary from plain C — add-with-id per document, a small batch search; exit code 0 means
    every query found the document it had to"""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "flat_search")
        subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), "-o", exe, os.path.join(root, "examples", "flat_search.c"),
                               "-L" + os.path.join(root, "zvec_amd"), "-lzvec_hip", "-Wl,-rpath," + os.path.join(root, "zvec_amd")])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert "query 0: (100," in out.stdout
