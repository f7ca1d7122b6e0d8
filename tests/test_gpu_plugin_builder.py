"""The plugin's "HipIVFBuilder" (plugin/hip_ivf_builder.cc) RUN inside the reference's own framework.

oracle/_ref/libzvec_ref_core.so (the reference's whole core library compiled in place, oracle/Makefile `ref_core`; test
infrastructure) + plugin/build/libzvec_hip_plugin.so (plugin/*.cc linked to it and to the product library, loaded through the
reference's IndexPluginBroker).  The driver (oracle/ref_core_shim.cc zref_build) creates the builder by its REGISTERED name through
IndexFactory::CreateBuilder, trains / builds from a holder over the caller's rows and dumps into the reference's MemoryDumper.  Here
the dumped FILE is opened by the product's loaders and searched: it must answer like an index built directly through the C ABI with
the same parameters, and like the oracle over the exported structure; the reference's own IVFSearcher must open it too.  Skipped
when the libraries were not built (no reference checkout at build time)."""
import os

import numpy as np
import pytest

from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_plugin_builder_dumps_an_index_the_loaders_open(oracle, dtype):
    import zvec_amd as zv
    from oracle import oracle as O
    from oracle import refcore as R
    if not (os.path.exists(R.CORE) and os.path.exists(R.PLUGIN)):
        pytest.skip("oracle/_ref/libzvec_ref_core.so / libzvec_hip_plugin.so not built (needs the reference checkout at build time)")
    R.load_plugin()
    rng = np.random.default_rng(2026)
    n, dim, nlist, nq, k = 6000, 40, 24, 50, 10
    npdt = np.float16 if dtype == "fp16" else np.float32
    base = rng.integers(-8, 9, (n, dim)).astype(npdt)
    keys = (rng.permutation(4 * n)[:n] + 11).astype(np.uint64)
    q = rng.integers(-8, 9, (nq, dim)).astype(npdt)
    R.build("HipIVFBuilder", base, "InnerProduct", "plugin_built", keys=keys,
            params={"proxima.ivf.builder.centroid_count": str(nlist), "proxima.hip.builder.kmeans_iters": 8})
    image = R.mem_get("plugin_built").tobytes()
    # the file is a reference index file: container -> segments -> IndexMeta naming the builder and the searcher defaults
    from zvec_amd.index import container_segments, parse_index_meta
    seg = container_segments(image)
    for sid in ("ivf.inverted_header", "ivf.inverted_meta", "ivf.inverted_body", "hc.keys", "ivf.centroid", "IndexMeta"):
        assert sid in seg, sid
    meta = parse_index_meta(image[seg["IndexMeta"][0]:sum(seg["IndexMeta"])])
    assert meta["dimension"] == dim and meta["data_type"] == (1 if dtype == "fp16" else 2)   # IndexMeta::DT_FP16 = 1, DT_FP32 = 2
    assert meta["builder_name"] == "HipIVFBuilder" and meta["searcher_name"] == "IVFSearcher"
    # opened by the loaders: same lists as a direct GPU build with the same parameters (same seed, sample rule, iterations)
    se = zv.open_ivf_file(image)
    cnt, nl = se.info()
    assert cnt == n and nl == nlist
    direct = zv.HipIVFSearcher(dim, "InnerProduct", dtype=dtype)
    assert direct.build(base, nlist, keys=keys, kmeans_iters=8) == 0
    c1, o1, r1 = se.export()
    c2, o2, r2 = direct.export()
    assert np.array_equal(c1, c2) and np.array_equal(o1, o2)
    for s_ in (se, direct):
        s_.scan_ratio, s_.brute_force_threshold = 0.25, 0
    ctx, ctx2 = se.create_context(), direct.create_context()
    ctx.set_topk(k)
    ctx2.set_topk(k)
    assert se.search_impl(q, nq, ctx) == 0 and direct.search_impl(q, nq, ctx2) == 0
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ctx2.keys, ctx2.scores, ctx2.counts, what="plugin-built vs direct")
    # and the oracle over the structure the file holds: list-order rows through the export's row ids (load order = list order)
    nprobe, max_scan = se.probe_params()
    cent, offs, _ = se.export()
    vecs, lkeys = direct_rows_in_list_order(direct, base, keys)
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, metric=O.METRIC_IP, keys=lkeys)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="plugin-built vs oracle")
    # the reference's OWN IVFSearcher opens the plugin-built file (every list probed) and answers like the GPU over it
    ref = R.Runner.searcher("IVFSearcher", "plugin_built", dim, npdt,
                            params={"proxima.ivf.searcher.scan_ratio": 1.0, "proxima.ivf.searcher.brute_force_threshold": 0})
    rctx = ref.create_context()
    rctx.set_topk(k)
    rc, lists = ref.search_lists(rctx, q)
    assert rc == 0
    se.scan_ratio = 1.0
    assert se.search_impl(q, nq, ctx) == 0
    rk = np.zeros((nq, k), np.uint64)
    rs = np.full((nq, k), np.inf, np.float32)
    rn = np.zeros(nq, np.uint32)
    for i, l in enumerate(lists):
        rn[i] = len(l[0])
        rk[i, :rn[i]], rs[i, :rn[i]] = l[0], l[1]
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, rk, rs, rn, what="plugin-built file: GPU vs the reference's IVFSearcher")
    ref.close()
    # searcher defaults as IVFBuilder::dump writes them (ivf_builder.cc:418-426)
    want_ratio = max(-0.004 * np.log(n) + 0.0751, 0.0001)
    assert abs(float(meta["searcher_params"]["proxima.ivf.searcher.scan_ratio"]) - want_ratio) < 1e-6


def direct_rows_in_list_order(direct, base, keys):
    """rows and keys of a directly built index in list order (export's row ids = original row of each position)"""
    _, _, rows = direct.export()
    rows = rows.astype(np.int64)
    return base[rows], keys[rows]


def test_plugin_builder_defaults_reach_the_reference_trainers_quality():
    """"HipIVFBuilder" with its DEFAULT parameters (20 Lloyd rounds over every row, as the reference's trainer) on the cluster-quality
    corpus of tests/golden/kmeans_quality.json; the dumped file is searched by the reference's OWN IVFSearcher at nprobe 4 of 256:
    recall@10 must not fall below the lowest of the reference builder's five runs (-0.01)."""
    import json
    from oracle import refcore as R
    from tests.test_gpu_build import _kmeans_quality_corpus
    if not (os.path.exists(R.CORE) and os.path.exists(R.PLUGIN)):
        pytest.skip("oracle/_ref libraries not built (needs the reference checkout at build time)")
    R.load_plugin()
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kmeans_quality.json")))
    c = fx["corpus"]
    base, q = _kmeans_quality_corpus(fx)
    R.build("HipIVFBuilder", base, "SquaredEuclidean", "plugin_quality", params={"proxima.ivf.builder.centroid_count": str(c["nlist"])})
    se = R.Runner.searcher("IVFSearcher", "plugin_quality", c["dim"],
                           params={"proxima.ivf.searcher.scan_ratio": c["nprobe"] / c["nlist"], "proxima.ivf.searcher.brute_force_threshold": c["n"] - 1})
    kk, _, cc, _ = se.search_mt(q, c["k"], 8)
    se.close()
    R.mem_remove("plugin_quality")
    d = (q.astype(np.float64) ** 2).sum(1)[:, None] + (base.astype(np.float64) ** 2).sum(1)[None] - 2 * q.astype(np.float64) @ base.astype(np.float64).T
    gt = np.argsort(d, 1, kind="stable")[:, :c["k"]]
    rec = float(np.mean([len(set(kk[i, :cc[i]].tolist()) & set(gt[i].tolist())) / c["k"] for i in range(len(q))]))
    low = min(r["recall_at_10"] for r in fx["runs"])
    print("plugin-built index searched by the reference's IVFSearcher: recall@10 %.4f (reference builder %.4f .. %.4f)"
          % (rec, low, max(r["recall_at_10"] for r in fx["runs"])))
    assert rec >= low - 0.01
