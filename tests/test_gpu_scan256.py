"""The wide fp16 flat scan on the 256 x 256 multi-phase tile (zvec_amd/csrc/zvk_scan256.hip.h) against the oracle
(FlatSearcher's batched search, flat_searcher.cc:143-189; fp16 contraction distance_matrix_accum_fp16.i:554-594).

The kernel is taken for >= 256 queries, k <= 11, unfiltered, fp16 rows with at least two k-steps (dim >= 65); on bases that
stay in the Infinity Cache only when option scan256 = 2 asks for it — which is how the small cases here reach it, WITHOUT the
bound-seeding pre-pass (n < 262144), so the lists fill from empty.  Integer-valued halves make every score exact: the lists
must equal the oracle's up to ties at the k-th place.  One larger case (streamed base, seeded bounds, several tiles per chunk)
is checked against the oracle on a sample of its queries and against the 128 x 128 kernel on all of them."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


@pytest.fixture()
def forced256():
    from zvec_amd import _lib
    L = _lib.lib()
    assert L.zvec_hip_set_option(b"scan256", 2) == 0
    yield L
    assert L.zvec_hip_set_option(b"scan256", 1) == 0


def _search(zv, base, q, k, name):
    se = zv.HipFlatSearcher(base.shape[1], name, dtype="fp16")
    assert se.load(base) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, len(q), ctx) == 0
    return ctx.keys.copy(), ctx.scores.copy(), ctx.counts.copy()


# (rows, dim, queries, k): one pair of tiles and a ragged tail; an odd number of tiles; a ragged batch (the moved-back last
# query tile); more chunks than CUs' worth of pairs; k = 1 and the largest k the lists hold
CASES = [(200, 72, 256, 10), (1000, 128, 256, 10), (3000, 100, 300, 10), (40000, 96, 256, 1), (70001, 136, 511, 11),
         (9000, 768, 257, 5)]


@pytest.mark.parametrize("n,dim,nq,k", CASES)
def test_small_bases_lists_from_empty(zv, oracle, forced256, n, dim, nq, k):
    rng = np.random.default_rng(n + dim + nq)
    hi = 32 if dim <= 136 else 12
    base = rng.integers(0, hi, (n, dim)).astype(np.float16)
    q = rng.integers(0, hi, (nq, dim)).astype(np.float16)
    for metric, name in ((O.METRIC_L2, "SquaredEuclidean"), (O.METRIC_IP, "InnerProduct")):
        keys, scores, counts = _search(zv, base, q, k, name)
        ok, os_, _, oc = oracle.flat_search(base, q, k, metric)
        tie_tolerant_compare(keys, scores, counts, ok, os_, oc, what="scan256 %s %s" % (name, (n, dim, nq, k)))


def test_the_option_switches_kernels(zv, forced256):
    # the same search on both kernels; the profile names the launch by its duration only, so the check is on the answers and on
    # the option's own read-back
    import ctypes as C
    L = forced256
    v = C.c_int(-1)
    assert L.zvec_hip_get_option(b"scan256", C.byref(v)) == 0 and v.value == 2
    assert L.zvec_hip_set_option(b"scan256", 3) != 0
    rng = np.random.default_rng(3)
    base = rng.integers(0, 32, (5000, 128)).astype(np.float16)
    q = rng.integers(0, 32, (256, 128)).astype(np.float16)
    a = _search(zv, base, q, 10, "SquaredEuclidean")
    assert L.zvec_hip_set_option(b"scan256", 0) == 0
    b = _search(zv, base, q, 10, "SquaredEuclidean")
    assert L.zvec_hip_set_option(b"scan256", 2) == 0
    tie_tolerant_compare(a[0], a[1], a[2], b[0], b[1], b[2], what="scan256 against scan8")


def test_cosine_rows(zv, oracle, forced256):
    rng = np.random.default_rng(12)
    n, dim, nq, k = 6000, 128, 256, 10
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    base = (base / np.linalg.norm(base, axis=1, keepdims=True)).astype(np.float16)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float16)
    keys, scores, counts = _search(zv, base, q, k, "InnerProduct")
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_IP)
    tie_tolerant_compare(keys, scores, counts, ok, os_, oc, rtol=4e-6, scale=np.ones(nq), what="scan256 IP unit rows")


def test_streamed_base_with_seeded_bounds(zv, oracle):
    # 400k x 128 halves = 98 MiB: past the Infinity Cache, seeded (>= 64 x 4096 rows), the DEFAULT dispatch
    from zvec_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(44)
    n, dim, nq, k = 400000, 128, 384, 10
    base = rng.integers(0, 32, (n, dim)).astype(np.float16)
    q = rng.integers(0, 32, (nq, dim)).astype(np.float16)
    got = {}
    for name in ("SquaredEuclidean", "InnerProduct"):
        got[name] = _search(zv, base, q, k, name)
    sel = np.arange(0, nq, 8)
    for metric, name in ((O.METRIC_L2, "SquaredEuclidean"), (O.METRIC_IP, "InnerProduct")):
        ok, os_, _, oc = oracle.flat_search(base, q[sel], k, metric)
        g = got[name]
        tie_tolerant_compare(g[0][sel], g[1][sel], g[2][sel], ok, os_, oc, what="scan256 streamed %s" % name)
    assert L.zvec_hip_set_option(b"scan256", 0) == 0
    try:
        for name in ("SquaredEuclidean", "InnerProduct"):
            b = _search(zv, base, q, k, name)
            g = got[name]
            tie_tolerant_compare(g[0], g[1], g[2], b[0], b[1], b[2], what="scan256 against scan8, streamed %s" % name)
    finally:
        assert L.zvec_hip_set_option(b"scan256", 1) == 0


@pytest.mark.parametrize("seed", range(12))
def test_seeded_sweep(zv, oracle, forced256, seed):
    """random shapes on the 256 x 256 tile: ragged batches, odd tile counts, every k it holds, remapped keys, a radius"""
    rng = np.random.default_rng(9100 + seed)
    dim = int(rng.choice([65, 96, 128, 200, 384, 520, 768, 1000]))
    n = int(rng.choice([130, 257, 1000, 4097, 12000, 30000]))
    nq = int(rng.choice([256, 257, 300, 383, 512, 600]))
    k = int(rng.integers(1, 12))
    name, metric = [("SquaredEuclidean", O.METRIC_L2), ("InnerProduct", O.METRIC_IP)][int(rng.integers(0, 2))]
    hi = 12 if dim > 200 else 30
    base = rng.integers(-hi, hi + 1, (n, dim)).astype(np.float16)
    q = rng.integers(-hi, hi + 1, (nq, dim)).astype(np.float16)
    keys = rng.permutation(2 * n)[:n].astype(np.uint64)
    se = zv.HipFlatSearcher(dim, name, dtype="fp16")
    assert se.load(base, keys) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    thr = O.FLT_MAX
    if rng.random() < 0.4:
        thr = float(np.median(oracle.flat_search(base, q[:1], min(3 * k, n), metric, threads=8)[1]))
        ctx.set_threshold(thr)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k, metric, keys=keys, threshold=thr, threads=16)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc,
                         what="scan256 sweep seed=%d n=%d d=%d nq=%d k=%d thr=%g %s" % (seed, n, dim, nq, k, thr, name))
