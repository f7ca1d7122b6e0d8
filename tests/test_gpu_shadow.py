"""GPU tests of the half-width pre-selection of fp32 IVF indexes (zvec_hip_ivf_set_shadow, zvk_shadow.hip.h) — run with -m gpu.
The claim under test: a search through the fp16 shadow lists + fp32 re-scoring + certificate (+ the fp32 re-run of uncertified
queries) returns what the plain fp32 route returns — the same keys, for L2 the same score BITS — and therefore what the oracle
(IVFSearcher::search_impl restated, ivf_searcher.cc:183-250) returns."""
import ctypes as C

import numpy as np
import pytest

from tests.util import tie_tolerant_compare, kmeans_lists, exact_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


def _index(zv, rng, base, nlist, metric="SquaredEuclidean", ratio=0.25, round_centroids=False):
    cent, offs, order = kmeans_lists(rng, base, nlist)
    if round_centroids:
        cent = np.round(cent)
    vecs, keys = base[order], order.astype(np.uint64)
    se = zv.HipIVFSearcher(base.shape[1], metric, scan_ratio=ratio, brute_force_threshold=10)
    assert se.load(cent, offs, vecs, keys) == 0
    return se, cent, offs, vecs, keys


def _search(se, q, k, ctx=None, exclude=None):
    ctx = ctx or se.create_context()
    ctx.set_topk(k)
    if exclude is not None:
        ctx.set_filter(exclude)
    assert se.search_impl(q, len(q), ctx) == 0
    return ctx.keys.copy(), ctx.scores.copy(), ctx.counts.copy()


def _search_dev(zv, se, q, k):
    """device pointers: the search only enqueues, zvec_hip_ivf_shadow_certify finishes it; returns results + queries re-run"""
    import torch
    dq = torch.from_numpy(q).cuda()
    nq = len(q)
    keys = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    scores = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    counts = torch.zeros(nq, dtype=torch.int32, device="cuda")
    ctx = se.create_context()
    nprobe, max_scan = se.probe_params()
    torch.cuda.synchronize()
    rc = se.search_dev(dq.data_ptr(), nq, k, nprobe, max_scan, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx)
    assert rc == 0
    rerun = se.shadow_certify(dq.data_ptr(), nq, k, nprobe, max_scan, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx)
    again = se.shadow_certify(dq.data_ptr(), nq, k, nprobe, max_scan, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx)
    assert again == 0                       # nothing pending any more
    torch.cuda.synchronize()
    return keys.cpu().numpy().astype(np.uint64), scores.cpu().numpy(), counts.cpu().numpy().astype(np.uint32), rerun


@pytest.mark.parametrize("n,dim,nlist,nq,k", [(20000, 96, 64, 70, 10), (30000, 768, 48, 130, 10), (8000, 33, 32, 40, 1),
                                               (12000, 128, 40, 64, 32)])
def test_shadow_equals_the_fp32_route_l2(zv, n, dim, nlist, nq, k):
    rng = np.random.default_rng(n + dim)
    cl = rng.standard_normal((nlist * 2, dim)).astype(np.float32) * 2
    base = (cl[rng.integers(0, len(cl), n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    q = (cl[rng.integers(0, len(cl), nq)] + rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float32)
    se, *_ = _index(zv, rng, base, nlist)
    k0, s0, c0 = _search(se, q, k)
    se.set_shadow(True)
    info = se.shadow_info()
    assert info["enabled"] and info["bytes"] > 0 and 0 < info["max_row_error"] < 1e-2 * info["max_row_norm"]
    k1, s1, c1 = _search(se, q, k)
    assert np.array_equal(c0, c1) and np.array_equal(k0, k1)
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32))          # L2: both routes end in the same re-scoring kernel
    k2, s2, c2, rerun = _search_dev(zv, se, q, k)
    assert np.array_equal(k0, k2) and np.array_equal(s0.view(np.uint32), s2.view(np.uint32)) and np.array_equal(c0, c2)
    assert rerun < nq                                                      # well separated data: (nearly) everything is certified
    # a wider pre-selection certifies at least as much
    se.set_shadow(True, 64)
    k3, s3, c3, rerun64 = _search_dev(zv, se, q, k)
    assert np.array_equal(k0, k3) and np.array_equal(s0.view(np.uint32), s3.view(np.uint32)) and rerun64 <= rerun
    se.set_shadow(False)
    assert not se.shadow_info()["enabled"]
    k4, s4, _ = _search(se, q, k)
    assert np.array_equal(k0, k4) and np.array_equal(s0.view(np.uint32), s4.view(np.uint32))


def test_shadow_inner_product(zv):
    rng = np.random.default_rng(5)
    n, dim, nlist, nq, k = 20000, 128, 50, 90, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    se, *_ = _index(zv, rng, base, nlist, metric="InnerProduct")
    k0, s0, c0 = _search(se, q, k)
    se.set_shadow(True)
    k1, s1, c1, rerun = _search_dev(zv, se, q, k)
    assert np.array_equal(c0, c1)
    # the fp32 route reports the matrix-core sums, the shadow route the re-scored ones: same rows, scores within fp32 rounding
    # of the operand magnitudes (|q||b| ~ dim)
    tie_tolerant_compare(k1, s1, c1, k0, s0, c0, atol=4e-6 * dim, what="shadow ip")
    assert rerun < nq


def test_shadow_against_the_oracle_integer_data(zv, oracle):
    """small integers are exact in fp16: shadow scores ARE the true scores; ties at the k'-th place fail the (strict) certificate
    and are re-run — the result is bit-exact the oracle's either way"""
    rng = np.random.default_rng(11)
    n, dim, nlist, nq, k = 20000, 64, 100, 150, 10
    base = rng.integers(0, 32, (n, dim)).astype(np.float32)
    q = rng.integers(0, 32, (nq, dim)).astype(np.float32)
    se, cent, offs, vecs, keys = _index(zv, rng, base, nlist, ratio=0.1, round_centroids=True)
    se.set_shadow(True)
    assert se.shadow_info()["max_row_error"] == 0.0
    nprobe, max_scan = se.probe_params()
    gk, gs, gc, rerun = _search_dev(zv, se, q, k)
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys)
    cd = np.sort(exact_l2(cent, q), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, nprobe])[0]
    assert len(sel) >= nq * 0.8
    tie_tolerant_compare(gk[sel], gs[sel], gc[sel], ok[sel], os_[sel], oc[sel], what="shadow int")
    hk, hs, hc = _search(se, q, k)                       # host pointers: certified inside the call
    assert np.array_equal(hk, gk) and np.array_equal(hs, gs) and np.array_equal(hc, gc)


def test_uncertifiable_queries_are_rerun_in_fp32(zv):
    """hundreds of rows within the fp16 rounding of each other: the k' pre-selected rows cannot be told from the rest, every
    such query must be flagged and answered by the fp32 lists"""
    rng = np.random.default_rng(17)
    dim, nlist, k = 64, 16, 10
    centres = rng.standard_normal((nlist, dim)).astype(np.float32) * 4
    base = np.concatenate([c + 1e-5 * rng.standard_normal((300, dim)).astype(np.float32) for c in centres]).astype(np.float32)
    q = (centres[rng.integers(0, nlist, 40)] + 1e-3 * rng.standard_normal((40, dim))).astype(np.float32)
    se, *_ = _index(zv, rng, base, nlist, ratio=0.25)
    k0, s0, c0 = _search(se, q, k)
    se.set_shadow(True)
    k1, s1, c1, rerun = _search_dev(zv, se, q, k)
    assert rerun == len(q)
    assert np.array_equal(k0, k1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(c0, c1)
    k2, s2, c2 = _search(se, q, k)
    assert np.array_equal(k0, k2) and np.array_equal(s0.view(np.uint32), s2.view(np.uint32))
    # the governor: four certify steps in a row that re-ran more than half of their queries suspend the shadow route for the next
    # searches of the index — they read the fp32 lists directly (nothing to certify), same results
    for _ in range(2):
        assert _search_dev(zv, se, q, k)[3] == len(q)
    k3, s3, c3, rerun3 = _search_dev(zv, se, q, k)
    assert rerun3 == 0
    assert np.array_equal(k0, k3) and np.array_equal(s0.view(np.uint32), s3.view(np.uint32)) and np.array_equal(c0, c3)
    se.set_shadow(True)                                   # setting it again lifts the suspension
    assert _search_dev(zv, se, q, k)[3] == len(q)


def test_shadow_with_a_filter_and_short_lists(zv):
    rng = np.random.default_rng(23)
    n, dim, nlist, nq, k = 3000, 48, 64, 50, 10                # ~47 rows per list: fewer candidates than k' for narrow probes
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    se, *_ = _index(zv, rng, base, nlist, ratio=0.02)
    k0, s0, c0 = _search(se, q, k, exclude=lambda key: key % 3 == 0)
    se.set_shadow(True)
    k1, s1, c1 = _search(se, q, k, exclude=lambda key: key % 3 == 0)
    assert np.array_equal(c0, c1)
    for i in range(nq):
        assert np.array_equal(k0[i, :c0[i]], k1[i, :c1[i]]) and np.array_equal(s0[i, :c0[i]], s1[i, :c1[i]])
        assert not (k1[i, :c1[i]] % 3 == 0).any()


def test_shadow_refusals(zv):
    rng = np.random.default_rng(29)
    base = rng.standard_normal((2000, 32)).astype(np.float32)
    from zvec_amd import _lib
    lib = _lib.lib()
    se = zv.HipIVFSearcher(32, "SquaredEuclidean")
    assert lib.zvec_hip_ivf_set_shadow(se._h, 1, 0) == zv.IndexError_.NoIndexLoaded
    se, *_ = _index(zv, rng, base, 8)
    assert lib.zvec_hip_ivf_set_shadow(se._h, 1, 65) == zv.IndexError_.InvalidArgument
    big = base.copy()
    big[7, 3] = 1e5                                             # beyond the half range
    se2, *_ = _index(zv, rng, big, 8)
    assert lib.zvec_hip_ivf_set_shadow(se2._h, 1, 0) == zv.IndexError_.Unsupported
    assert not se2.shadow_info()["enabled"]
    cs, *_ = _index(zv, rng, np.concatenate([base, np.ones((2000, 1), np.float32)], 1), 8, metric="Cosine")
    assert lib.zvec_hip_ivf_set_shadow(cs._h, 1, 0) == zv.IndexError_.Unsupported
    n = C.c_uint32(7)
    assert lib.zvec_hip_ivf_shadow_certify(se._h, None, None, 1, 1, 1, 1, None, None, None, None, None, C.byref(n)) == zv.IndexError_.InvalidArgument


# ---- flat stores -------------------------------------------------------------------------------------------------------------------
def _flat_search(se, q, k, exclude=None):
    ctx = se.create_context()
    ctx.set_topk(k)
    if exclude is not None:
        ctx.set_filter(exclude)
    assert se.search_impl(q, len(q), ctx) == 0
    return ctx.keys.copy(), ctx.scores.copy(), ctx.counts.copy()


def _flat_search_dev(se, q, k):
    import torch
    dq = torch.from_numpy(q).cuda()
    nq = len(q)
    keys = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
    scores = torch.zeros((nq, k), dtype=torch.float32, device="cuda")
    counts = torch.zeros(nq, dtype=torch.int32, device="cuda")
    ctx = se.create_context()
    torch.cuda.synchronize()
    assert se.search_dev(dq.data_ptr(), nq, k, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx) == 0
    rerun = se.shadow_certify(dq.data_ptr(), nq, k, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx)
    assert se.shadow_certify(dq.data_ptr(), nq, k, keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx) == 0
    torch.cuda.synchronize()
    return keys.cpu().numpy().astype(np.uint64), scores.cpu().numpy(), counts.cpu().numpy().astype(np.uint32), rerun


@pytest.mark.parametrize("metric", ["SquaredEuclidean", "InnerProduct"])
@pytest.mark.parametrize("n,dim,nq,k", [(300000, 128, 256, 10), (150000, 96, 1, 10), (150000, 96, 8, 5), (40000, 64, 300, 32)])
def test_flat_shadow_equals_the_fp32_route(zv, metric, n, dim, nq, k):
    """wide batches (the 8-wave tile over fp16 rows), a handful of queries (the 16-row shape), a cache-resident base — every shape of
    flat_scan_prepared under the shadow rows returns what the fp32 rows return"""
    rng = np.random.default_rng(n + nq)
    cl = rng.standard_normal((200, dim)).astype(np.float32) * 2
    base = (cl[rng.integers(0, 200, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    q = (cl[rng.integers(0, 200, nq)] + rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float32)
    se = zv.HipFlatSearcher(dim, metric)
    assert se.load(base) == 0
    k0, s0, c0 = _flat_search(se, q, k)
    se.set_shadow(True)
    assert se.shadow_info()["enabled"]
    k1, s1, c1 = _flat_search(se, q, k)
    k2, s2, c2, rerun = _flat_search_dev(se, q, k)
    assert rerun < max(nq // 2, 1) + 1
    for kk, ss, cc in ((k1, s1, c1), (k2, s2, c2)):
        assert np.array_equal(cc, c0)
        if metric == "SquaredEuclidean":
            assert np.array_equal(kk, k0) and np.array_equal(ss.view(np.uint32), s0.view(np.uint32))
        else:
            tie_tolerant_compare(kk, ss, cc, k0, s0, c0, atol=4e-6 * dim * 8, what="flat shadow ip")
    # with a filter
    f0 = _flat_search(se, q, k, exclude=lambda key: key % 4 == 1)
    assert se.shadow_info()["enabled"]
    se.set_shadow(False)
    f1 = _flat_search(se, q, k, exclude=lambda key: key % 4 == 1)
    assert np.array_equal(f0[2], f1[2]) and not (f0[0][:, 0] % 4 == 1).any()
    if metric == "SquaredEuclidean":
        assert np.array_equal(f0[0], f1[0]) and np.array_equal(f0[1].view(np.uint32), f1[1].view(np.uint32))


def test_flat_shadow_integer_data_vs_the_oracle_and_mutation(zv, oracle):
    rng = np.random.default_rng(31)
    n, dim, nq, k = 100000, 64, 64, 10
    base = rng.integers(0, 32, (n, dim)).astype(np.float32)
    q = rng.integers(0, 32, (nq, dim)).astype(np.float32)
    se = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    assert se.add_batch(base[: n // 2], np.arange(n // 2, dtype=np.uint64)) == 0
    se.set_shadow(True)
    assert se.shadow_info()["max_row_error"] == 0.0
    gk, gs, gc, _ = _flat_search_dev(se, q, k)
    ok, os_, _, oc = oracle.flat_search(base[: n // 2], q, k)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="flat shadow int")
    # a mutation drops the twin: the store searches its own (now longer) rows
    assert se.add_batch(base[n // 2:], np.arange(n // 2, n, dtype=np.uint64)) == 0
    assert not se.shadow_info()["enabled"]
    gk, gs, gc = _flat_search(se, q, k)
    ok, os_, _, oc = oracle.flat_search(base, q, k)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="flat after growth")
    se.set_shadow(True)
    gk, gs, gc = _flat_search(se, q, k)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="flat shadow after growth")


@pytest.mark.parametrize("noise", [1e-3, 1e-2, 0.03, 0.06, 0.1, 0.3])
def test_certificate_at_the_rounding_scale(zv, oracle, noise):
    """rows whose mutual distances are small against their norms, down to the ORDER of the fp16 rounding: the fp16 ranking genuinely
    differs from the true one near the k'-th place, some queries certify and some do not.  The certificate speaks about the TRUE
    scores — sum((q - b)^2) in fp32, what the reference computes (euclidean_distance_matrix_fp32.cc:229-283) and what both routes
    report after re-scoring — so the yardstick here is the oracle, with NO selection band: in this regime the fp32 route itself may pick
    a different row inside its stated band of 4e-6 (|q|^2 + |b|^2) (it selects on |q|^2 + |b|^2 - 2 q.b), the shadow route
    may not."""
    rng = np.random.default_rng(int(noise * 1e6))
    dim, nlist, k, per = 96, 12, 10, 400
    centres = rng.standard_normal((nlist, dim)).astype(np.float32) * 6
    base = np.concatenate([c + noise * np.abs(c).mean() * rng.standard_normal((per, dim)).astype(np.float32) for c in centres]).astype(np.float32)
    q = (centres[rng.integers(0, nlist, 48)] + noise * 6 * rng.standard_normal((48, dim))).astype(np.float32)
    se, cent, offs, vecs, keys = _index(zv, rng, base, nlist, ratio=0.3)
    nprobe, max_scan = se.probe_params()
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys)
    cd = np.sort(exact_l2(cent, q), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, nprobe])[0]
    se.set_shadow(True)
    k1, s1, c1, rerun = _search_dev(zv, se, q, k)
    # (queries the certificate refuses are answered by the fp32 route, inside ITS band; a batch that certified everything has none)
    qn = (q.astype(np.float64) ** 2).sum(1)
    band = 4e-6 * (2 * qn.max() + 1)
    tie_tolerant_compare(k1[sel], s1[sel], c1[sel], ok[sel], os_[sel], oc[sel], rtol=2e-6, atol=1e-9, select_band=band if rerun else None,
                         what="ivf shadow vs oracle, noise %g" % noise)
    fl = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert fl.load(base) == 0
    fl.set_shadow(True)
    fk, fs, fc, frerun = _flat_search_dev(fl, q, k)
    gk, gs, _, gc = oracle.flat_search(base, q, k)
    tie_tolerant_compare(fk, fs, fc, gk, gs, gc, rtol=2e-6, atol=1e-9, select_band=band if frerun else None,
                         what="flat shadow vs oracle, noise %g" % noise)
    print("noise %g: ivf re-ran %d of 48, flat re-ran %d of 48" % (noise, rerun, frerun))


def test_queries_beyond_the_half_range_are_rerun(zv):
    """a query element beyond 65504 has no fp16 image: its shadow scores are infinite — the query must be flagged (never certified,
    not even by a short pre-selection) and answered by the fp32 rows"""
    rng = np.random.default_rng(41)
    n, dim, nlist, k = 6000, 32, 16, 5
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((20, dim)).astype(np.float32)
    q[3, 5] = 1.0e5
    q[11, 0] = -2.0e5
    se, *_ = _index(zv, rng, base, nlist, ratio=0.5)
    k0, s0, c0 = _search(se, q, k)
    se.set_shadow(True)
    k1, s1, c1, rerun = _search_dev(zv, se, q, k)
    assert rerun >= 2
    assert np.array_equal(k0, k1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(c0, c1)
    fl = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert fl.load(base) == 0
    f0 = _flat_search(fl, q, k)
    fl.set_shadow(True)
    f1 = _flat_search_dev(fl, q, k)
    assert f1[3] >= 2
    assert np.array_equal(f0[0], f1[0]) and np.array_equal(f0[1].view(np.uint32), f1[1].view(np.uint32))


def test_the_index_sizes_the_preselection_when_left_open(zv):
    """preselect = 0: k' starts at max(32, 3k); clean certify steps narrow it (down to 16 for k = 10), steps that re-ran more than 1/32
    of their queries widen it and pin the floor; an explicit width stays put.  Results never change."""
    rng = np.random.default_rng(43)
    n, dim, nlist, nq, k = 20000, 64, 32, 64, 10
    cl = rng.standard_normal((64, dim)).astype(np.float32) * 3
    base = (cl[rng.integers(0, 64, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    q = (cl[rng.integers(0, 64, nq)] + rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float32)
    se, *_ = _index(zv, rng, base, nlist)
    k0, s0, c0 = _search(se, q, k)
    assert se.shadow_width(k) == 0
    se.set_shadow(True)
    assert se.shadow_width(k) == 32 and se.shadow_width(20) == 60 and se.shadow_width(30) == 64
    widths, clean = [], True
    for _ in range(14):
        k1, s1, c1, rerun = _search_dev(zv, se, q, k)
        clean = clean and rerun == 0
        assert np.array_equal(k0, k1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(c0, c1)
        widths.append(se.shadow_width(k))
    if clean:
        assert widths[-1] == 16 and sorted(widths, reverse=True) == widths        # 32 -> 24 -> 16, never below
    se.set_shadow(True, 40)
    for _ in range(8):
        _search_dev(zv, se, q, k)
    assert se.shadow_width(k) == 40
    # near-duplicates: every step re-runs everything -> the width climbs to 64 and stays there
    centres = rng.standard_normal((8, dim)).astype(np.float32) * 4
    dup = np.concatenate([c + 1e-5 * rng.standard_normal((300, dim)).astype(np.float32) for c in centres]).astype(np.float32)
    dq = (centres[rng.integers(0, 8, 20)] + 1e-3 * rng.standard_normal((20, dim))).astype(np.float32)
    du, *_ = _index(zv, rng, dup, 8, ratio=0.5)
    d0 = _search(du, dq, k)
    du.set_shadow(True)
    for _ in range(3):
        d1 = _search_dev(zv, du, dq, k)
        assert d1[3] == len(dq) and np.array_equal(d0[0], d1[0]) and np.array_equal(d0[1].view(np.uint32), d1[1].view(np.uint32))
    assert du.shadow_width(k) == 56
