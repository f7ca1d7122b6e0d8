"""The plugin's "HipIVFBuilder" (plugin/hip_ivf_builder.cc) RUN inside the reference's own framework.

oracle/_ref/libzvec_ref_plugin.so (built in the build container by oracle/Makefile `ref_plugin`; test infrastructure) holds the
plugin translation unit linked with the reference's framework sources compiled in place — IndexFactory, IndexMeta,
MultiPassIndexHolder, IVFDumper, FlatBuilder, MemoryDumper + IndexPacker — and the product library for the GPU work.  The
driver (oracle/ref_plugin_shim.cc) creates the builder by its REGISTERED name, trains / builds from a holder and dumps into a
MemoryDumper.  Here the dumped FILE is opened by the product's loaders and searched: it must answer like an index built
directly through the C ABI with the same parameters, and like the oracle over the exported structure.  Only the
"InnerProduct" metric is linked into that library (see oracle/Makefile).  Skipped when the library was not built (no
reference checkout at build time)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu
_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libzvec_ref_plugin.so")


@pytest.mark.skipif(not os.path.exists(_SO), reason="oracle/_ref/libzvec_ref_plugin.so not built (needs the reference checkout at build time)")
@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_plugin_builder_dumps_an_index_the_loaders_open(oracle, dtype):
    import zvec_amd as zv
    from oracle import oracle as O
    lib = C.CDLL(_SO)
    fn = lib.zref_plugin_ivf_build_and_dump
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                   C.POINTER(C.c_uint64)]
    rng = np.random.default_rng(2026)
    n, dim, nlist, nq, k = 6000, 40, 24, 50, 10
    npdt = np.float16 if dtype == "fp16" else np.float32
    base = rng.integers(-8, 9, (n, dim)).astype(npdt)
    keys = (rng.permutation(4 * n)[:n] + 11).astype(np.uint64)
    q = rng.integers(-8, 9, (nq, dim)).astype(npdt)
    out = np.zeros(n * dim * 8 + (1 << 20), np.uint8)
    size = C.c_uint64(0)
    rc = fn(int(dtype == "fp16"), dim, b"InnerProduct", base.ctypes.data, keys.ctypes.data, n, nlist, 8, out.ctypes.data, out.size, C.byref(size))
    assert rc == 0, rc
    image = out[:size.value].tobytes()
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
    # searcher defaults as IVFBuilder::dump writes them (ivf_builder.cc:418-426)
    want_ratio = max(-0.004 * np.log(n) + 0.0751, 0.0001)
    assert abs(float(meta["searcher_params"]["proxima.ivf.searcher.scan_ratio"]) - want_ratio) < 1e-6


def direct_rows_in_list_order(direct, base, keys):
    """rows and keys of a directly built index in list order (export's row ids = original row of each position)"""
    _, _, rows = direct.export()
    rows = rows.astype(np.int64)
    return base[rows], keys[rows]
