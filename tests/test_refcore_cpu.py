"""The oracle restatement (oracle/zvec_oracle.c) pinned against the reference's OWN operator classes, run live.

oracle/_ref/libzvec_ref_core.so = the reference's whole core library compiled in place (oracle/Makefile `ref_core`): FlatBuilder /
FlatSearcher / FlatStreamer / IVFBuilder / IVFSearcher / IVFStreamer created by their registered names.  First the reference's own
unit-test expectations are reproduced through that harness (it drives the classes the way the reference's tests do), then the
restated scan loops — row-major and column-major flat batches, the IVF driver loop with the max_scan_count rule, filters, the RNN
threshold, group-by — are compared with the classes on seeded real-valued data: BIT-EXACT scores (the restatement models the
AVX-512 lane order of the 1x1 kernels and the M x N block kernels' chains), ids equal outside exact boundary ties."""
import os
import shutil
import struct
import tempfile

import numpy as np
import pytest

from oracle import oracle as O
from oracle import refcore as R
from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libzvec_ref_core.so not built (needs the reference checkout at build time)")


def arrays(lists, k):
    nq = len(lists)
    keys = np.zeros((nq, k), np.uint64)
    scores = np.full((nq, k), np.inf, np.float32)
    counts = np.zeros(nq, np.uint32)
    for i, l in enumerate(lists):
        counts[i] = len(l[0])
        keys[i, :counts[i]], scores[i, :counts[i]] = l[0], l[1]
    return keys, scores, counts


def bit_exact(oracle_out, lists, k, what):
    ok, os_, oc = oracle_out
    rk, rs, rc = arrays(lists, k)
    tie_tolerant_compare(ok, os_, oc, rk, rs, rc, what=what)       # atol = rtol = 0: scores bit for bit, ids outside exact ties


def read_ivf_file(image, dt):
    """centroids [nlist][dim] (centroid-id order), list offsets, of a dumped IVF file (ivf_index_format.h:26-47)"""
    from zvec_amd.index import container_segments, parse_index_meta
    seg = container_segments(image)
    ho, _ = seg["ivf.inverted_header"]
    _, total, _, nlist, _, _, _, _ = struct.unpack_from("<IIQIIIII", image, ho)
    mo, _ = seg["ivf.inverted_meta"]
    sizes = [struct.unpack_from("<QIII", image, mo + l * 40)[2] for l in range(nlist)]
    co, cs = seg["ivf.centroid"]
    nested = image[co:co + cs]
    cseg = container_segments(nested)
    cmeta = parse_index_meta(nested[cseg["IndexMeta"][0]:sum(cseg["IndexMeta"])])
    dim = cmeta["dimension"]
    ck = np.frombuffer(nested, np.uint64, cseg["flat.keys"][1] // 8, cseg["flat.keys"][0])
    feat = np.frombuffer(nested, dt, nlist * dim, cseg["flat.features"][0])
    rows = np.empty((nlist, dim), dt)
    for i in range(nlist):
        blk, r = divmod(i, 32)
        if cmeta["major_order"] == 2 and (blk + 1) * 32 <= nlist:
            rows[ck[i]] = feat[blk * 32 * dim:(blk + 1) * 32 * dim].reshape(dim, 32)[:, r]
        else:
            rows[ck[i]] = feat[i * dim:(i + 1) * dim]
    return rows, np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64), total


def test_reference_unit_test_expectations_through_the_real_classes():
    """flat_streamer_test.cc:104-178 (TestLinearSearch), ivf_searcher_test.cc:200-321 (TestSimple): the harness drives the classes as
    the reference's own tests do and gets the answers those tests assert."""
    dim = 16
    tmp = tempfile.mkdtemp(prefix="zref_")
    try:
        st = R.Runner.streamer("FlatStreamer", os.path.join(tmp, "lin"), dim, "SquaredEuclidean")
        base = np.repeat(np.arange(1000, dtype=np.float32)[:, None], dim, 1)
        assert st.add(np.arange(1000, dtype=np.uint64), base) == 0
        ctx = st.create_context()
        ctx.set_topk(100)
        rc, l = st.search_lists(ctx, np.full((1, dim), 10.1, np.float32))
        assert rc == 0 and len(l[0][0]) == 100
        for rank, key in ((0, 10), (1, 11), (10, 5), (20, 0), (30, 30), (35, 35), (99, 99)):
            assert l[0][0][rank] == key
        ctx.close()
        st.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    base = np.repeat(np.arange(33, dtype=np.float32)[:, None], dim, 1)
    R.build("IVFBuilder", base, "SquaredEuclidean", "ts", params={"proxima.ivf.builder.centroid_count": "1", "proxima.ivf.builder.cluster_class": "KmeansCluster"})
    se = R.Runner.searcher("IVFSearcher", "ts", dim, params={"proxima.ivf.searcher.scan_ratio": 1.0, "proxima.ivf.searcher.brute_force_threshold": 1})
    ctx = se.create_context()
    ctx.set_topk(33)
    for mode in (1, 0):
        rc, l = se.search_lists(ctx, np.full((1, dim), 32.0, np.float32), mode)
        assert rc == 0
        assert np.array_equal(l[0][0], 32 - np.arange(33)) and np.array_equal(l[0][1], (np.arange(33.0) ** 2 * dim).astype(np.float32))
    ctx.set_topk(1)
    q = np.repeat(np.arange(33, dtype=np.float32)[:, None], dim, 1)
    for mode in (1, 0):
        rc, l = se.search_lists(ctx, q, mode)
        assert rc == 0 and all(len(x[0]) == 1 and x[0][0] == i and x[1][0] == 0 for i, x in enumerate(l))
    se.close()
    R.mem_remove("ts")


@pytest.mark.parametrize("metric,om", [("SquaredEuclidean", O.METRIC_L2), ("InnerProduct", O.METRIC_IP)])
@pytest.mark.parametrize("dt", [np.float32, np.float16])
@pytest.mark.parametrize("column_major", [False, True])
def test_flat_scan_loops_bit_exact_vs_flat_searcher(oracle, metric, om, dt, column_major):
    """FlatSearcher<32>::search_impl(count) (flat_searcher.cc:162-211 -> flat_searcher_context.h:420-1003): row-major loops and the
    column-major 32 x K tile path, batches that exercise every query-group size, filter and threshold."""
    rng = np.random.default_rng(5 + int(column_major))
    n, dim, k = 777, 40, 7
    base = rng.standard_normal((n, dim)).astype(dt)
    keys = rng.permutation(3 * n)[:n].astype(np.uint64)
    R.build("FlatBuilder", base, metric, "fs", keys=keys, column_major=column_major,
            params={"proxima.flat.column_major_order": bool(column_major)})
    se = R.Runner.searcher("FlatSearcher", "fs", dim, dt)
    ctx = se.create_context()
    ctx.set_topk(k)
    search = oracle.flat_search_column if column_major else oracle.flat_search
    for nq in (1, 3, 45):
        q = rng.standard_normal((nq, dim)).astype(dt)
        rc, lists = se.search_lists(ctx, q)
        assert rc == 0
        ok, os_, _, oc = search(base, q, k, om, keys=keys)
        bit_exact((ok, os_, oc), lists, k, "flat %s %s cm=%d nq=%d" % (metric, dt.__name__, column_major, nq))
    q = rng.standard_normal((9, dim)).astype(dt)
    keep = rng.random(n) < 0.5
    ex = np.zeros(3 * n, np.uint8)
    ex[keys[~keep]] = 1
    ctx.set_filter(ex)
    rc, lists = se.search_lists(ctx, q)
    ok, os_, _, oc = search(base, q, k, om, keys=keys, exclude_bits=O.pack_bits(~keep))
    bit_exact((ok, os_, oc), lists, k, "flat filter")
    ctx.set_filter(None)
    thr = float(np.median(os_[:, 2]))
    ctx.set_threshold(thr)
    rc, lists = se.search_lists(ctx, q)
    ok, os_, _, oc = search(base, q, k, om, keys=keys, threshold=thr)
    bit_exact((ok, os_, oc), lists, k, "flat threshold")
    ctx.close()
    se.close()
    R.mem_remove("fs")


@pytest.mark.parametrize("dt", [np.float32, np.float16])
def test_ivf_driver_loop_bit_exact_vs_ivf_searcher(oracle, dt):
    """IVFSearcher::search_impl / search_bf_impl (ivf_searcher.cc:106-250) over an index the reference's own IVFBuilder trained, built
    and dumped: coarse assign, probe walk with the max_scan_count rule (`total_scan_count >= max_scan_count` stops it), list scans in
    32-vector blocks, filter, brute force.  The restatement runs on the centroids / lists read back from the dumped file."""
    rng = np.random.default_rng(21)
    n, dim, nlist, k = 6000, 32, 40, 8
    means = rng.standard_normal((nlist, dim)).astype(np.float32) * 2
    base = (means[rng.integers(0, nlist, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(dt)
    keys = rng.permutation(2 * n)[:n].astype(np.uint64)
    R.build("IVFBuilder", base, "SquaredEuclidean", "iv", keys=keys, params={"proxima.ivf.builder.centroid_count": str(nlist), "proxima.ivf.builder.thread_count": 2})
    image = R.mem_get("iv").tobytes()
    cent, offs, total = read_ivf_file(image, dt)
    assert total == n and offs[-1] == n
    q = (means[rng.integers(0, nlist, 30)] + rng.standard_normal((30, dim)).astype(np.float32)).astype(dt)
    for ratio, bft in ((0.1, 100), (0.25, 3000), (1.0, 1)):
        se = R.Runner.searcher("IVFSearcher", "iv", dim, dt, params={"proxima.ivf.searcher.scan_ratio": ratio, "proxima.ivf.searcher.brute_force_threshold": bft})
        lkeys, lrows = se.walk()                               # provider walk = list order
        assert np.array_equal(np.sort(lkeys), np.sort(keys))
        ctx = se.create_context()
        ctx.set_topk(k)
        nprobe = max(int(round(np.float32(nlist) * np.float32(ratio))), 1)          # ivf_searcher_context.h:70-78
        max_scan = max(bft, int(np.ceil(np.float32(n) * np.float32(ratio))))
        rc, lists = se.search_lists(ctx, q)
        assert rc == 0
        ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, lrows, q, k, nprobe, max_scan, keys=lkeys)
        bit_exact((ok, os_, oc), lists, k, "ivf ratio %g bft %d" % (ratio, bft))
        rc, lists = se.search_lists(ctx, q, 1)
        ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, lrows, q, k, nprobe, max_scan, keys=lkeys, brute_force=True)
        bit_exact((ok, os_, oc), lists, k, "ivf bf")
        keep = rng.random(n) < 0.5                              # over list-order positions
        ex = np.zeros(2 * n, np.uint8)
        ex[lkeys[~keep]] = 1
        ctx.set_filter(ex)
        rc, lists = se.search_lists(ctx, q)
        ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, lrows, q, k, nprobe, max_scan, keys=lkeys, exclude_bits=O.pack_bits(~keep))
        bit_exact((ok, os_, oc), lists, k, "ivf filter ratio %g" % ratio)
        ctx.close()
        se.close()
    R.mem_remove("iv")


def test_group_by_vs_flat_searcher(oracle):
    """FlatSearcherContext::group_by_search_impl (flat_searcher_context.h:1005-1043).  The reference leaves the LOCAL ID in the group
    documents' key() there (emplace(id, dist), :1031); the restatement reports (key, score, position): compared through positions."""
    rng = np.random.default_rng(31)
    n, dim = 1500, 24
    base = rng.standard_normal((n, dim)).astype(np.float32)
    keys = rng.permutation(2 * n)[:n].astype(np.uint64)
    R.build("FlatBuilder", base, "SquaredEuclidean", "gb", keys=keys)
    se = R.Runner.searcher("FlatSearcher", "gb", dim)
    gof_key = rng.integers(0, 30, 2 * n).astype(np.uint32)
    ctx = se.create_context()
    ctx.set_group(gof_key, 4, 3)
    q = rng.standard_normal((5, dim)).astype(np.float32)
    assert se.search(ctx, q) == 0
    want = oracle.flat_group_search(base, q, gof_key[keys.astype(np.int64)], 4, 3, keys=keys)
    for qi in range(len(q)):
        got = ctx.groups(qi)
        assert [g[0] for g in got] == [g[0] for g in want[qi]], qi
        for (g, gk, gs), (_, docs) in zip(got, want[qi]):
            assert [int(x) for x in gk] == [d[2] for d in docs]                   # reference key() = position here
            assert np.array_equal(gs, np.array([d[1] for d in docs], np.float32))
    ctx.close()
    se.close()
    R.mem_remove("gb")


def test_flat_streamer_single_calls_and_add_with_id(oracle):
    """FlatStreamer (flat_streamer.cc:304-389, entity scan flat_streamer_entity.cc:212-316): count = 1 calls (its count > 1 loop is
    broken); add_with_id pads the gap with invalid rows that scans skip (flat_streamer_entity.cc:900-990)."""
    rng = np.random.default_rng(41)
    n, dim, k = 500, 16, 6
    base = rng.standard_normal((n, dim)).astype(np.float32)
    tmp = tempfile.mkdtemp(prefix="zref_")
    try:
        st = R.Runner.streamer("FlatStreamer", os.path.join(tmp, "s"), dim, "SquaredEuclidean")
        ids = np.arange(0, 2 * n, 2, dtype=np.uint64)                       # even ids: holes at the odd ones
        assert st.add(ids, base, with_id=True) == 0
        ctx = st.create_context()
        ctx.set_topk(k)
        q = rng.standard_normal((6, dim)).astype(np.float32)
        rc, lists = st.search_lists_single(ctx, q)
        assert rc == 0
        ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2, keys=ids)
        bit_exact((ok, os_, oc), lists, k, "streamer add_with_id")
        rc, lists = st.search_lists_single(ctx, q, 2, [ids[rng.choice(n, 20, replace=False)] for _ in range(len(q))])
        assert rc == 0 and all(len(l[0]) == k for l in lists)
        ctx.close()
        st.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_searcher_over_lent_rows_equals_searcher_over_the_dumped_file(oracle):
    """refcore.Runner.ivf_over_rows (bench.py's cpu_baseline leg): the reference's IVFDumper writes every small segment, the body is
    lent from the caller's rows.  Must answer exactly like the same structure dumped to a complete file (ref_format_shim) — and the
    lent body IS the dumped body byte for byte."""
    import ctypes as C
    from zvec_amd.index import container_segments
    rng = np.random.default_rng(51)
    n, dim, nlist, k = 5000, 32, 24, 9
    for dt in (np.float32, np.float16):
        base = rng.standard_normal((n, dim)).astype(dt)
        cent = base[rng.choice(n, nlist, replace=False)].copy()
        sizes = rng.multinomial(n, np.ones(nlist) / nlist)
        sizes[3] = 0
        sizes[-1] += n - sizes.sum()
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        keys = rng.permutation(3 * n)[:n].astype(np.uint64)
        params = {"proxima.ivf.searcher.scan_ratio": 0.2, "proxima.ivf.searcher.brute_force_threshold": 100}
        lent = R.Runner.ivf_over_rows("IVFSearcher", cent, offs, base, keys, "SquaredEuclidean", params=params)
        buf = np.zeros(base.nbytes + (4 << 20), np.uint8)
        sz = C.c_uint64(0)
        L = R.lib()
        L.zref_dump_ivf_index.restype = C.c_int
        rc = L.zref_dump_ivf_index(int(dt == np.float16), dim, 0, 0, b"SquaredEuclidean", cent.ctypes.data_as(C.c_void_p), nlist,
                                   offs.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p), keys.ctypes.data_as(C.c_void_p),
                                   buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.byref(sz))
        assert rc == 0
        image = buf[:sz.value].tobytes()
        bo, bs = container_segments(image)["ivf.inverted_body"]
        assert image[bo:bo + bs] == base.tobytes()                      # the premise of lending the rows
        R.mem_put("lent_cmp", image)
        filed = R.Runner.searcher("IVFSearcher", "lent_cmp", dim, dt, params=params)
        q = rng.standard_normal((20, dim)).astype(dt)
        c1, c2 = lent.create_context(), filed.create_context()
        c1.set_topk(k), c2.set_topk(k)
        for mode in (0, 1):
            r1, l1 = lent.search_lists(c1, q, mode)
            r2, l2 = filed.search_lists(c2, q, mode)
            assert r1 == 0 and r2 == 0
            for a, b in zip(l1, l2):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        kk, ss, cc, _ = lent.search_mt(q, k, 3)
        for i, b in enumerate(l2 if False else filed.search_lists(c2, q, 0)[1]):
            assert np.array_equal(kk[i, :cc[i]], b[0]) and np.array_equal(ss[i, :cc[i]], b[1])
        c1.close(), c2.close()
        lent.close(), filed.close()
        R.mem_remove("lent_cmp")
