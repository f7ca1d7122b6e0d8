"""The plugin's SEARCH classes (plugin/hip_plugin.cc) RUN inside the reference's own framework, beside the reference's own.

oracle/_ref/libzvec_ref_core.so   = the reference's whole core library compiled in place (oracle/Makefile `ref_core`; incl.
                                    ailego::BufferHandle from its own buffer_manager.cc against the image's real Arrow —
                                    nothing stubbed) + the by-name driver oracle/ref_core_shim.cc
plugin/build/libzvec_hip_plugin.so = plugin/hip_plugin.cc + hip_ivf_builder.cc linked to it and to zvec_amd/libzvec_hip.so,
                                    brought in through the reference's IndexPluginBroker::emplace (dlopen + static registrars)
Every case opens the SAME index file with the reference's class ("FlatSearcher", "IVFSearcher", "IVFStreamer", "FlatStreamer")
and with the plugin's ("HipFlatSearcher", "HipIVFSearcher", "HipIVFStreamer", "HipFlatStreamer") created by their registered names
through IndexFactory, calls the same IndexRunner virtuals (index_runner.h:476-585) with the same contexts settings and compares
the IndexDocumentLists: flat_searcher.cc:68-211, ivf_searcher.cc:43-250, ivf_streamer.cc:183-250, flat_streamer.cc:304-483.
A plugin class that mis-parses a segment (keys, features, column-major blocks, nested centroid index, list meta) fails here."""
import os
import shutil
import tempfile

import numpy as np
import pytest

from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def R():
    from oracle import refcore
    if not (os.path.exists(refcore.CORE) and os.path.exists(refcore.PLUGIN)):
        pytest.skip("oracle/_ref/libzvec_ref_core.so / libzvec_hip_plugin.so not built (needs the reference checkout at build time)")
    refcore.load_plugin()
    for kind, name in (("searcher", "HipFlatSearcher"), ("searcher", "HipIVFSearcher"), ("streamer", "HipIVFStreamer"),
                       ("streamer", "HipFlatStreamer"), ("builder", "HipIVFBuilder"), ("searcher", "FlatSearcher"),
                       ("searcher", "IVFSearcher")):
        assert refcore.has(kind, name), name
    return refcore


def lists_to_arrays(lists, k):
    nq = len(lists)
    keys = np.zeros((nq, k), np.uint64)
    scores = np.full((nq, k), np.inf, np.float32)
    counts = np.zeros(nq, np.uint32)
    for i, l in enumerate(lists):
        n = len(l[0])
        assert n <= k
        counts[i] = n
        keys[i, :n], scores[i, :n] = l[0], l[1]
    return keys, scores, counts


def compare(hip_lists, ref_lists, k, what, **tol):
    gk, gs, gc = lists_to_arrays(hip_lists, k)
    ok, os_, oc = lists_to_arrays(ref_lists, k)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what=what, **tol)


def golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_index_files.npz"))
    for name in [str(x) for x in z["cases"]]:
        meta = z[name + "_meta"]
        f16 = bool(meta[0])
        dt = np.float16 if f16 else np.float32
        base = z[name + "_base"].view(dt) if f16 else z[name + "_base"]
        yield name, dt, base, z[name + "_keys"], int(meta[2]), z[name + "_image"], z, meta


def test_flat_searcher_on_reference_dumped_files(R):
    """flat0..flat3 (fp32 / fp16, row- and column-major, explicit keys): HipFlatSearcher::load reads "flat.keys" + "flat.features"
    through the reference's storage; search_impl / search_bf_impl / filter / threshold / fetch_vector / by_p_keys / group-by."""
    rng = np.random.default_rng(11)
    for name, dt, base, keys, dim, image, _, _ in golden():
        if not name.startswith("flat"):
            continue
        R.mem_put("g_" + name, image)
        ref = R.Runner.searcher("FlatSearcher", "g_" + name, dim, dt)
        hip = R.Runner.searcher("HipFlatSearcher", "g_" + name, dim, dt)
        n = len(keys)
        assert hip.count() == ref.count() == n
        q = rng.integers(-8, 9, (9, dim)).astype(dt)
        k = 5
        rc_, hc = ref.create_context(), hip.create_context()
        rc_.set_topk(k), hc.set_topk(k)
        for mode in (0, 1):
            r1, l1 = ref.search_lists(rc_, q, mode)
            r2, l2 = hip.search_lists(hc, q, mode)
            assert r1 == 0 and r2 == 0, (name, mode, r1, r2)
            compare(l2, l1, k, "%s mode %d" % (name, mode))
        # IndexFilter (true = exclude) by key
        ex = np.zeros(int(keys.max()) + 1, np.uint8)
        ex[keys[rng.random(n) < 0.4]] = 1
        rc_.set_filter(ex), hc.set_filter(ex)
        _, l1 = ref.search_lists(rc_, q)
        _, l2 = hip.search_lists(hc, q)
        compare(l2, l1, k, name + " filter")
        for l in l2:
            assert not ex[l[0].astype(np.int64)].any()
        rc_.set_filter(None), hc.set_filter(None)
        # RNN threshold: lists end at score <= threshold
        thr = float(np.median(np.concatenate([l[1] for l in l1])))
        rc_.set_threshold(thr), hc.set_threshold(thr)
        _, l1 = ref.search_lists(rc_, q)
        _, l2 = hip.search_lists(hc, q)
        compare(l2, l1, k, name + " threshold")
        rc_.set_threshold(None), hc.set_threshold(None)
        # fetch_vector: every document carries its stored row
        hc.set_fetch_vector(True)
        assert hip.search(hc, q) == 0
        row_of = {int(kk): base[i] for i, kk in enumerate(keys)}
        for qi in range(len(q)):
            kk, _, _, vec, present = hc.result(qi, dt, dim)
            assert present == len(kk) == k
            for j in range(len(kk)):
                assert np.array_equal(vec[j].view(np.uint8), row_of[int(kk[j])].view(np.uint8)), (name, qi, j)
        hc.set_fetch_vector(False)
        # search_bf_by_p_keys_impl: only the listed keys are candidates; unknown keys are skipped
        # (the reference's FlatSearcher dereferences a null row on an unknown key — flat_searcher_provider.h:178 logs, then the
        # distance call faults; its FlatStreamer skips unknown keys, flat_streamer.cc:346-389, and so does the plugin: the
        # unknown keys go to the plugin only)
        pk = [rng.choice(keys, 12, replace=False) for i in range(len(q))]
        r1, l1 = ref.search_lists(rc_, q, 2, pk)
        r2, l2 = hip.search_lists(hc, q, 2, [np.concatenate([p_, [np.uint64(10 ** 9 + i)]]) for i, p_ in enumerate(pk)])
        assert r1 == 0 and r2 == 0, (name, r1, r2)
        compare(l2, l1, k, name + " p_keys")
        # get_vector(key)
        for key in keys[[0, n // 2, n - 1]]:
            r2, v2 = hip.get_vector(int(key))
            assert r2 == 0 and np.array_equal(v2.view(np.uint8), row_of[int(key)].view(np.uint8))
        assert hip.get_vector(10 ** 9)[0] != 0
        # provider walk: storage order
        wk, wr = hip.walk()
        assert np.array_equal(wk, keys) and np.array_equal(wr.view(np.uint8), base.view(np.uint8))
        ref.close(), hip.close()
        R.mem_remove("g_" + name)


def test_flat_searcher_half_width_preselect_through_the_plugin(R):
    """proxima.hip.searcher.half_width_preselect = 1 on HipFlatSearcher (zvec_hip_flat_set_shadow after the load): the golden flat
    files — integer data, exact in fp16 — answered like the reference's FlatSearcher, filter included; fp16 files accept the
    parameter and keep searching their own rows"""
    rng = np.random.default_rng(12)
    half = {"proxima.hip.searcher.half_width_preselect": 1}
    for name, dt, base, keys, dim, image, _, _ in golden():
        if not name.startswith("flat"):
            continue
        R.mem_put("g_" + name, image)
        ref = R.Runner.searcher("FlatSearcher", "g_" + name, dim, dt)
        hip = R.Runner.searcher("HipFlatSearcher", "g_" + name, dim, dt, params=half)
        q = rng.integers(-8, 9, (40, dim)).astype(dt)
        k = 5
        rc_, hc = ref.create_context(), hip.create_context()
        rc_.set_topk(k), hc.set_topk(k)
        r1, l1 = ref.search_lists(rc_, q, 0)
        r2, l2 = hip.search_lists(hc, q, 0)
        assert r1 == 0 and r2 == 0, (name, r1, r2)
        compare(l2, l1, k, name + " half-width")
        ex = np.zeros(int(keys.max()) + 1, np.uint8)
        ex[keys[rng.random(len(keys)) < 0.4]] = 1
        rc_.set_filter(ex), hc.set_filter(ex)
        _, l1 = ref.search_lists(rc_, q)
        _, l2 = hip.search_lists(hc, q)
        compare(l2, l1, k, name + " half-width filter")
        ref.close(), hip.close()
        R.mem_remove("g_" + name)


def test_flat_searcher_group_by_vs_reference(R):
    """group_by_search_impl (flat_searcher.cc:178-179,205-206 -> flat_searcher_context.h:1005-1043) on a reference-dumped file."""
    rng = np.random.default_rng(12)
    n, dim = 3000, 24
    base = rng.standard_normal((n, dim)).astype(np.float32)
    keys = rng.permutation(2 * n)[:n].astype(np.uint64)
    R.build("FlatBuilder", base, "SquaredEuclidean", "grp", keys=keys)
    ref = R.Runner.searcher("FlatSearcher", "grp", dim)
    hip = R.Runner.searcher("HipFlatSearcher", "grp", dim)
    gof = rng.integers(0, 40, 2 * n).astype(np.uint32)
    q = rng.standard_normal((6, dim)).astype(np.float32)
    rc_, hc = ref.create_context(), hip.create_context()
    for c in (rc_, hc):
        c.set_group(gof, 5, 3)
    assert ref.search(rc_, q) == 0 and hip.search(hc, q) == 0
    for qi in range(len(q)):
        g1, g2 = rc_.groups(qi), hc.groups(qi)
        assert [g[0] for g in g1] == [g[0] for g in g2], (qi, g1, g2)
        for a, b in zip(g1, g2):
            # the reference's FlatSearcher group path leaves the LOCAL ID in key() (topk_heap.emplace(id, dist),
            # flat_searcher_context.h:1031; the plain path maps ids to keys, :745-750): the plugin reports keys
            assert np.array_equal(keys[a[1].astype(np.int64)], b[1]), (qi, a, b)
            np.testing.assert_allclose(b[2], a[2], rtol=2e-6, atol=1e-6)
    ref.close(), hip.close()


def _centroid_gap_ok(cent, q, nprobe, ip):
    """queries whose nprobe-th and (nprobe+1)-th centroid scores differ (a tie there makes the probe SET arbitrary)"""
    c, qq = cent.astype(np.float64), q.astype(np.float64)
    s = -(qq @ c.T) if ip else ((qq[:, None, :] - c[None]) ** 2).sum(-1)
    s = np.sort(s, 1)
    return np.nonzero(s[:, nprobe - 1] != s[:, nprobe])[0] if nprobe < s.shape[1] else np.arange(len(q))


@pytest.mark.parametrize("cls_pair", [("IVFSearcher", "HipIVFSearcher", "searcher"), ("IVFStreamer", "HipIVFStreamer", "streamer")])
def test_ivf_classes_on_reference_dumped_files(R, cls_pair):
    """ivf0..ivf2 (fp32 / fp16, row- / column-major lists, empty and ragged lists, column-major nested centroid index): HipIVFCore::load
    walks IndexMeta, the nested "ivf.centroid" index, header / meta / body / keys through the reference's storage classes."""
    ref_cls, hip_cls, kind = cls_pair
    rng = np.random.default_rng(13)
    for name, dt, base, keys, dim, image, z, meta in golden():
        if not name.startswith("ivf"):
            continue
        cent = z[name + "_cent"].view(dt) if dt == np.float16 else z[name + "_cent"]
        nlist = int(meta[5])
        R.mem_put("g_" + name, image)
        for ratio, bft in ((1.0, 0), (0.45, 1)):
            params = {"proxima.ivf.searcher.scan_ratio": ratio, "proxima.ivf.searcher.brute_force_threshold": bft}
            if kind == "searcher":
                ref = R.Runner.searcher(ref_cls, "g_" + name, dim, dt, params=params)
                hip = R.Runner.searcher(hip_cls, "g_" + name, dim, dt, params=params)
            else:
                ref = R.Runner.streamer(ref_cls, "g_" + name, dim, "InnerProduct", dt, params=params, storage="MemoryReadStorage", create=False)
                hip = R.Runner.streamer(hip_cls, "g_" + name, dim, "InnerProduct", dt, params=params, storage="MemoryReadStorage", create=False)
            q = rng.integers(-8, 9, (24, dim)).astype(dt)
            nprobe = max(int(round(nlist * ratio)), 1)
            sel = _centroid_gap_ok(cent, q, nprobe, True)
            assert len(sel) >= 4
            q = q[sel]
            k = 6
            rc_, hc = ref.create_context(), hip.create_context()
            rc_.set_topk(k), hc.set_topk(k)
            for mode in (0, 1):
                r1, l1 = ref.search_lists(rc_, q, mode)
                r2, l2 = hip.search_lists(hc, q, mode)
                assert r1 == 0 and r2 == 0, (name, mode, r1, r2)
                if ratio == 1.0 or mode == 1:
                    compare(l2, l1, k, "%s %s ratio %g mode %d" % (hip_cls, name, ratio, mode))
                else:
                    # partial probes on tie-heavy integer data: max_scan_count cuts the probe walk (ivf_searcher.cc:217-247) at the
                    # same list on both sides, so the lists must still agree
                    compare(l2, l1, k, "%s %s ratio %g mode %d" % (hip_cls, name, ratio, mode))
            ex = np.zeros(int(keys.max()) + 1, np.uint8)
            ex[keys[rng.random(len(keys)) < 0.5]] = 1
            rc_.set_filter(ex), hc.set_filter(ex)
            _, l1 = ref.search_lists(rc_, q, 1)
            _, l2 = hip.search_lists(hc, q, 1)
            compare(l2, l1, k, "%s %s filter" % (hip_cls, name))
            rc_.set_filter(None), hc.set_filter(None)
            hc.set_fetch_vector(True)
            assert hip.search(hc, q, 1) == 0
            row_of = {int(kk): base[i] for i, kk in enumerate(keys)}
            for qi in range(len(q)):
                kk, _, _, vec, present = hc.result(qi, dt, dim)
                assert present == len(kk)
                for j in range(len(kk)):
                    assert np.array_equal(vec[j].view(np.uint8), row_of[int(kk[j])].view(np.uint8))
            hc.set_fetch_vector(False)
            wk, wr = hip.walk()
            assert np.array_equal(wk, keys) and np.array_equal(wr.view(np.uint8), base.view(np.uint8)), name
            ref.close(), hip.close()
        R.mem_remove("g_" + name)


@pytest.mark.parametrize("metric,dt", [("SquaredEuclidean", np.float32), ("SquaredEuclidean", np.float16), ("InnerProduct", np.float32),
                                       ("InnerProduct", np.float16)])
def test_ivf_built_by_the_reference_builder_searched_by_both(R, metric, dt):
    """An index TRAINED, BUILT and DUMPED by the reference's own IVFBuilder (its k-means, its labelling, its dumper) on real-valued
    data; partial probes with the max_scan_count rule live; one context handed from one index to another (magic re-bind,
    ivf_searcher.cc:198-202); context update(params) changing the scan ratio.  InnerProduct: the reference's builder trains through
    a MipsConverter (ivf_builder.cc:552-555), so the nested centroid index holds CONVERTED centroids (d + 4 dimensions, squared
    Euclidean) behind a MipsReformer: the plugin installs them as a coarse space of their own (zvec_hip_ivf_set_coarse_space) and
    reforms every query with the reference's reformer before the coarse pass, as IVFCentroidIndex::search does."""
    rng = np.random.default_rng(14)
    n, dim, nlist = 20000, 48, 64
    means = rng.standard_normal((nlist, dim)).astype(np.float32) * 2
    base = (means[rng.integers(0, nlist, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(dt)
    keys = (rng.permutation(3 * n)[:n]).astype(np.uint64)
    q = (means[rng.integers(0, nlist, 40)] + rng.standard_normal((40, dim)).astype(np.float32)).astype(dt)
    bp = {"proxima.ivf.builder.centroid_count": str(nlist), "proxima.ivf.builder.thread_count": 4}
    R.build("IVFBuilder", base, metric, "built_a", keys=keys, params=bp)
    R.build("IVFBuilder", base[: n // 2], metric, "built_b", keys=keys[: n // 2], params=bp)
    params = {"proxima.ivf.searcher.scan_ratio": 0.1, "proxima.ivf.searcher.brute_force_threshold": 100}
    tol = dict(rtol=4e-6, atol=1e-5) if dt == np.float32 else dict(rtol=4e-6, atol=1e-4)
    k = 10
    ref = R.Runner.searcher("IVFSearcher", "built_a", dim, dt, params=params)
    hip = R.Runner.searcher("HipIVFSearcher", "built_a", dim, dt, params=params)
    ref_b = R.Runner.searcher("IVFSearcher", "built_b", dim, dt, params=params)
    hip_b = R.Runner.searcher("HipIVFSearcher", "built_b", dim, dt, params=params)
    rc_, hc = ref.create_context(), hip.create_context()
    rc_.set_topk(k), hc.set_topk(k)
    band = 1e-4 if metric == "SquaredEuclidean" else None

    def both(a, b, qq, mode, what):
        r1, l1 = a.search_lists(rc_, qq, mode)
        r2, l2 = b.search_lists(hc, qq, mode)
        assert r1 == 0 and r2 == 0, (what, r1, r2)
        compare(l2, l1, k, what, select_band=band, **tol)

    both(ref, hip, q, 0, "built knn")
    both(ref, hip, q, 1, "built bf")
    both(ref, hip, q[:1], 0, "built knn single")          # the product's count = 1 call
    both(ref_b, hip_b, q, 0, "second index, contexts of the first")     # same contexts, other index instance
    both(ref, hip, q, 0, "back on the first")
    up = {"proxima.ivf.searcher.scan_ratio": 0.3}
    assert rc_.update(up) == 0 and hc.update(up) == 0
    both(ref, hip, q, 0, "scan_ratio 0.3")
    for r in (ref, hip, ref_b, hip_b):
        r.close()


@pytest.mark.parametrize("metric,dt", [("SquaredEuclidean", np.float32), ("InnerProduct", np.float32), ("SquaredEuclidean", np.float16)])
def test_ivf_half_width_preselect_through_the_plugin(R, metric, dt):
    """proxima.hip.searcher.half_width_preselect = 1: the plugin class pre-selects on an fp16 twin of the lists, re-scores in fp32,
    certifies and re-runs what it cannot certify (zvec_hip_ivf_set_shadow) — and must answer like the reference's IVFSearcher over
    an index the reference's own builder made, and like the plugin class without the parameter.  (fp16 index: the parameter is
    accepted and the index keeps searching its own lists.)"""
    rng = np.random.default_rng(15)
    n, dim, nlist = 20000, 48, 64
    means = rng.standard_normal((nlist, dim)).astype(np.float32) * 2
    base = (means[rng.integers(0, nlist, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(dt)
    keys = (rng.permutation(3 * n)[:n]).astype(np.uint64)
    q = (means[rng.integers(0, nlist, 64)] + rng.standard_normal((64, dim)).astype(np.float32)).astype(dt)
    R.build("IVFBuilder", base, metric, "built_hw", keys=keys, params={"proxima.ivf.builder.centroid_count": str(nlist),
                                                                        "proxima.ivf.builder.thread_count": 4})
    params = {"proxima.ivf.searcher.scan_ratio": 0.1, "proxima.ivf.searcher.brute_force_threshold": 100}
    half = dict(params, **{"proxima.hip.searcher.half_width_preselect": 1})
    tol = dict(rtol=4e-6, atol=1e-5) if dt == np.float32 else dict(rtol=4e-6, atol=1e-4)
    k = 10
    ref = R.Runner.searcher("IVFSearcher", "built_hw", dim, dt, params=params)
    hip = R.Runner.searcher("HipIVFSearcher", "built_hw", dim, dt, params=params)
    hw = R.Runner.searcher("HipIVFSearcher", "built_hw", dim, dt, params=half)
    rc_, hc, wc = ref.create_context(), hip.create_context(), hw.create_context()
    for c in (rc_, hc, wc):
        c.set_topk(k)
    band = 1e-4 if metric == "SquaredEuclidean" else None
    r1, l1 = ref.search_lists(rc_, q, 0)
    r2, l2 = hip.search_lists(hc, q, 0)
    r3, l3 = hw.search_lists(wc, q, 0)
    assert r1 == 0 and r2 == 0 and r3 == 0
    compare(l3, l1, k, "half-width vs reference", select_band=band, **tol)
    compare(l3, l2, k, "half-width vs fp32 route", select_band=band, **tol)
    if metric == "SquaredEuclidean":
        a, b = lists_to_arrays(l3, k), lists_to_arrays(l2, k)                     # same keys, same score bits
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
    r4, l4 = hw.search_lists(wc, q[:1], 0)                 # the product's count = 1 call: the small-batch route, fp32 lists
    r5, l5 = ref.search_lists(rc_, q[:1], 0)
    assert r4 == 0 and r5 == 0
    compare(l4, l5, k, "half-width single", select_band=band, **tol)
    for r in (ref, hip, hw):
        r.close()


def test_flat_streamer_wraps_the_reference_streamer(R):
    """HipFlatStreamer: add_impl / add_with_id_impl persist through the wrapped reference FlatStreamer and mirror into HBM;
    searches run on the GPU; close + re-open reads the persisted rows back (flat_streamer.cc:236-483)."""
    rng = np.random.default_rng(15)
    n, dim = 3000, 32
    base = rng.standard_normal((n, dim)).astype(np.float32)
    keys = np.arange(n, dtype=np.uint64)
    tmp = tempfile.mkdtemp(prefix="zvec_streamer_")
    try:
        ref = R.Runner.streamer("FlatStreamer", os.path.join(tmp, "ref"), dim, "SquaredEuclidean")
        hip = R.Runner.streamer("HipFlatStreamer", os.path.join(tmp, "hip"), dim, "SquaredEuclidean")
        assert ref.add(keys[:2000], base[:2000]) == 0 and hip.add(keys[:2000], base[:2000]) == 0
        # add_with_id: the product's add path (index.cc:505-537): ids at the count, then ids BEYOND it — the gap is padded with
        # invalid rows that no scan may return (flat_streamer_entity.cc:900-990)
        assert ref.add(keys[2000:], base[2000:], with_id=True) == 0 and hip.add(keys[2000:], base[2000:], with_id=True) == 0
        far = rng.standard_normal((40, dim)).astype(np.float32)
        far_ids = np.arange(n + 75, n + 115, dtype=np.uint64)
        assert ref.add(far_ids, far, with_id=True) == 0 and hip.add(far_ids, far, with_id=True) == 0
        q = rng.standard_normal((7, dim)).astype(np.float32)
        k = 8
        rc_, hc = ref.create_context(), hip.create_context()
        rc_.set_topk(k), hc.set_topk(k)
        tol = dict(rtol=4e-6, atol=1e-5, select_band=1e-4)

        def both(what, mode=0, pk=None):
            r1, l1 = ref.search_lists_single(rc_, q, mode, pk)      # count = 1 calls: the reference's batched streamer loop is broken
            r2, l2 = hip.search_lists(hc, q, mode, pk)
            r3, l3 = hip.search_lists_single(hc, q, mode, pk)
            assert r3 == 0
            compare(l3, l1, k, what + " (single calls)", **tol)
            assert r1 == 0 and r2 == 0, (what, r1, r2)
            compare(l2, l1, k, what, **tol)

        both("streamer knn")
        both("streamer bf", 1)
        both("streamer p_keys", 2, [rng.choice(keys, 30, replace=False) for _ in range(len(q))])
        ex = (rng.random(n) < 0.3).astype(np.uint8)
        rc_.set_filter(ex), hc.set_filter(ex)
        both("streamer filter")
        rc_.set_filter(None), hc.set_filter(None)
        r2, v2 = hip.get_vector(1234)
        assert r2 == 0 and np.array_equal(v2, base[1234])
        assert hip.flush() == 0 and ref.flush() == 0
        rc_.close(), hc.close()
        assert hip.close() == 0 and ref.close() == 0
        # re-open: the rows come back from the persisted block chain
        hip = R.Runner.streamer("HipFlatStreamer", os.path.join(tmp, "hip"), dim, "SquaredEuclidean", create=False)
        ref = R.Runner.streamer("FlatStreamer", os.path.join(tmp, "ref"), dim, "SquaredEuclidean", create=False)
        rc_, hc = ref.create_context(), hip.create_context()
        rc_.set_topk(k), hc.set_topk(k)
        both("re-opened streamer knn")                       # (block runs read straight from the storage: HipFlatStreamer::bulk_open)
        q[:] = far[:len(q)] + 0.01                           # the rows behind the gap are found, the gap's padding is not
        both("re-opened streamer, rows behind the add_with_id gap")
        r2, v2 = hip.get_vector(int(far_ids[3]))
        assert r2 == 0 and np.array_equal(v2, far[3])
        more = rng.standard_normal((100, dim)).astype(np.float32)
        mk = np.arange(n + 200, n + 300, dtype=np.uint64)
        assert ref.add(mk, more) == 0 and hip.add(mk, more) == 0
        both("re-opened streamer after more adds")
        rc_.close(), hc.close()
        hip.close(), ref.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("conv,metric,builder,ref_cls,hip_cls", [
    ("CosineFp32Converter", "Cosine", "FlatBuilder", "FlatSearcher", "HipFlatSearcher"),
    ("HalfFloatConverter", "SquaredEuclidean", "FlatBuilder", "FlatSearcher", "HipFlatSearcher"),
    ("CosineFp32Converter", "Cosine", "IVFBuilder", "IVFSearcher", "HipIVFSearcher"),
    ("CosineFp16Converter", "Cosine", "IVFBuilder", "IVFSearcher", "HipIVFSearcher"),
])
def test_boundary_a_search_batch_sequence(R, conv, metric, builder, ref_cls, hip_cls):
    """next-4: Index::SearchBatch of patches/boundary_a.diff, run step for step around the plugin's operators with the REFERENCE's own
    converter, reformer and metric (oracle/ref_core_shim.cc zref_search_sequence): rows converted by the converter the product picks
    (index.cc:111-183), raw queries through reformer->transform, ONE search_impl(count) on the GPU, per query metric->normalize and
    reformer->normalize — against count calls of Index::_dense_search's sequence (index.cc:596-652, count = 1) on the plugin and on
    the reference's CPU operator.  (Found by running it: CosineReformer has no batched transform, cosine_reformer.cc:146-150 — the
    patch falls back to per-query transforms.)"""
    rng = np.random.default_rng(16)
    n, dim, nq, k = 8000, 40, 37, 9
    means = rng.standard_normal((32, dim)).astype(np.float32) * 2
    base = (means[rng.integers(0, 32, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    q = (means[rng.integers(0, 32, nq)] + rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float32)
    keys = rng.permutation(2 * n)[:n].astype(np.uint64)
    ivf = builder == "IVFBuilder"
    bparams = {"proxima.ivf.builder.centroid_count": "32", "proxima.ivf.builder.thread_count": 2} if ivf else None
    R.build_converted(builder, conv, base, metric, "bnd_a", keys=keys, params=bparams)
    from zvec_amd.index import container_segments, parse_index_meta
    image = R.mem_get("bnd_a").tobytes()
    seg = container_segments(image)
    m = parse_index_meta(image[seg["IndexMeta"][0]:sum(seg["IndexMeta"])])
    sdt = np.float16 if m["data_type"] == 1 else np.float32
    sp = {"proxima.ivf.searcher.scan_ratio": 0.25, "proxima.ivf.searcher.brute_force_threshold": 10} if ivf else None
    ref = R.Runner.searcher(ref_cls, "bnd_a", m["dimension"], sdt, params=sp)
    hip = R.Runner.searcher(hip_cls, "bnd_a", m["dimension"], sdt, params=sp)
    rc_, hc = ref.create_context(), hip.create_context()
    want = ref.search_sequence(rc_, q, k, batched=False)            # the product today, on the reference's CPU operator
    got1 = hip.search_sequence(hc, q, k, batched=False)             # the same single calls on the GPU operator
    gotb = hip.search_sequence(hc, q, k, batched=True)              # Index::SearchBatch on the GPU operator
    half = sdt == np.float16
    tol = dict(rtol=4e-6, atol=2e-6) if not half else dict(rtol=4e-6, atol=2e-4)
    band = None if metric == "Cosine" else 1e-3
    if metric == "Cosine":
        tol["scale"] = 1.0
    tie_tolerant_compare(got1[0], got1[1], got1[2], want[0], want[1], want[2], what="singles: plugin vs reference", select_band=band, **tol)
    tie_tolerant_compare(gotb[0], gotb[1], gotb[2], want[0], want[1], want[2], what="SearchBatch: plugin vs reference singles", select_band=band, **tol)
    tie_tolerant_compare(gotb[0], gotb[1], gotb[2], got1[0], got1[1], got1[2], what="SearchBatch vs singles on the plugin", select_band=band, **tol)
    for x in (rc_, hc):
        x.close()
    ref.close(), hip.close()
    R.mem_remove("bnd_a")


@pytest.mark.parametrize("cls,kind", [("HipIVFSearcher", "ivf"), ("HipFlatSearcher", "flat")])
def test_many_threads_single_queries_through_the_plugins_micro_batcher(R, cls, kind):
    """zvec's real call pattern (index.cc:24-45,605-619; tools/core/bench.cc:145-245): T = 64 threads, each with a context of its
    own, each calling search_impl(count = 1) on the PLUGIN class — with proxima.hip.searcher.batch_window_us set, so the calls ride
    shared batches (include/zvec_hip_operator.hpp MicroBatcher, instantiated by plugin/hip_plugin.cc) — and, for comparison, with the
    batcher off.  Every query's list must equal the one the reference's own CPU class returns for a single call."""
    rng = np.random.default_rng(77)
    n, dim, nlist, k, nq = 30000, 64, 96, 10, 640
    means = rng.standard_normal((nlist, dim)).astype(np.float32) * 2
    base = (means[rng.integers(0, nlist, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    keys = (rng.permutation(2 * n)[:n]).astype(np.uint64)
    q = (means[rng.integers(0, nlist, nq)] + rng.standard_normal((nq, dim)).astype(np.float32)).astype(np.float32)
    tol = dict(rtol=4e-6, atol=1e-5)
    batch = {"proxima.hip.searcher.batch_window_us": 3000, "proxima.hip.searcher.max_batch": 48,
             "proxima.hip.searcher.batch_linger_us": 200}
    if kind == "ivf":
        bp = {"proxima.ivf.builder.centroid_count": str(nlist), "proxima.ivf.builder.thread_count": 4}
        R.build("IVFBuilder", base, "SquaredEuclidean", "mb_ivf", keys=keys, params=bp)
        params = {"proxima.ivf.searcher.scan_ratio": 0.15, "proxima.ivf.searcher.brute_force_threshold": 100}
        ref = R.Runner.searcher("IVFSearcher", "mb_ivf", dim, np.float32, params=params)
        target = "mb_ivf"
    else:
        R.build("FlatBuilder", base, "SquaredEuclidean", "mb_flat", keys=keys)
        params = {}
        ref = R.Runner.searcher("FlatSearcher", "mb_flat", dim, np.float32, params=params)
        target = "mb_flat"
    rk, rs, rc_, _ = ref.search_mt(q, k, 8)                      # the reference's singles
    for what, extra, threads in (("batched T=64", batch, 64), ("unbatched T=64", {}, 64), ("batched T=3", batch, 3)):
        hip = R.Runner.searcher(cls, target, dim, np.float32, params=dict(params, **extra))
        gk, gs, gc, sec = hip.search_mt(q, k, threads)
        tie_tolerant_compare(gk, gs, gc, rk, rs, rc_, what="%s %s" % (cls, what), select_band=1e-4, **tol)
        # a context with a filter / a radius bypasses the batcher and still answers alone
        ctx = hip.create_context()
        ctx.set_topk(k)
        ctx.set_threshold(float(np.median(rs[:, 3])))
        r2, lists = hip.search_lists(ctx, q[:5], 0)
        assert r2 == 0
        rctx = ref.create_context()
        rctx.set_topk(k)
        rctx.set_threshold(float(np.median(rs[:, 3])))
        r1, rlists = ref.search_lists(rctx, q[:5], 0)
        assert r1 == 0
        compare(lists, rlists, k, "%s %s radius" % (cls, what), select_band=1e-4, **tol)
        hip.close()
    ref.close()
