"""GPU parity of the predicate materialisation (zvec_hip_*_build_filter) — run with -m gpu.
Oracle = oracle/roaring.py: the composite filter of doc_filter.cc:74-87 evaluated with numpy set operations over
the same keys; the roaring byte streams come from that module's writer (format parity unpinned, see its header)."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import roaring as R
from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


def _flat(zv, n, dim=8, keys=None, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(-8, 8, (n, dim)).astype(np.float32)
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base, keys) == 0
    return se, base


def _mixed_ids(rng, space):
    """ids that produce every container type: sparse arrays, a dense (> 4096) chunk, long runs"""
    sparse = rng.choice(space, min(space // 50 + 1, 5000), replace=False)
    dense0 = (space // 3) & ~0xFFFF
    dense = dense0 + rng.choice(65536, 30000, replace=False) if space > dense0 + 65536 else np.array([], np.int64)
    run0 = (space // 2) & ~0xFFFF
    runs = np.concatenate([np.arange(run0 + 100, run0 + 9000), np.arange(run0 + 20000, run0 + 20010)]) if space > run0 + 65536 else np.array([], np.int64)
    ids = np.unique(np.concatenate([sparse, dense, runs]).astype(np.uint64))
    return ids[ids < space]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 300_000])
@pytest.mark.parametrize("run_opt", [False, True])
def test_flat_filter_bits_match_oracle(zv, n, run_opt):
    rng = np.random.default_rng(n + run_opt)
    keys = rng.permutation(max(2 * n, 400_000))[:n].astype(np.uint64)      # ids need not equal positions
    se, _ = _flat(zv, n, keys=keys)
    space = int(keys.max()) + 70000
    deleted = _mixed_ids(rng, space)
    matched = _mixed_ids(rng, space)
    forward = rng.random(space // 2) < 0.7                                     # shorter than the id space on purpose
    for kw in (dict(deleted=deleted), dict(invert=matched), dict(forward=forward),
               dict(deleted=deleted, invert=matched, forward=forward)):
        f = zv.DocFilter(delete=R.serialize32(kw["deleted"], run_opt) if "deleted" in kw else None,
                         invert=R.serialize32(kw["invert"], run_opt) if "invert" in kw else None,
                         forward=kw.get("forward"))
        got = se.build_filter(f)
        want = R.mask_to_words(R.doc_filter_mask(keys, **kw))
        assert np.array_equal(got, want), "terms %s" % sorted(kw)


def test_filter_delete_kinds_64bit_and_file_image(zv):
    rng = np.random.default_rng(5)
    n = 5000
    keys = np.concatenate([rng.choice(1 << 20, n // 2, replace=False),
                           (np.uint64(3) << np.uint64(32)) + rng.choice(1 << 20, n - n // 2, replace=False).astype(np.uint64)]).astype(np.uint64)
    se, _ = _flat(zv, n, keys=keys)
    deleted = np.unique(np.concatenate([rng.choice(keys, 700, replace=False),
                                        rng.choice(1 << 20, 3000).astype(np.uint64)]))
    want64 = R.mask_to_words(R.doc_filter_mask(keys, deleted=deleted, deleted_is32=False))
    payload64 = R.serialize64map(deleted, run_optimize=True)
    assert np.array_equal(se.build_filter(zv.DocFilter(delete=payload64, kind="roaring64map")), want64)
    assert np.array_equal(se.build_filter(zv.DocFilter(delete=R.file_image(payload64, False), kind="file")), want64)
    # a 32-bit delete store behind 64-bit ids: probed with (uint32_t)id
    del32 = np.unique(deleted & np.uint64(0xFFFFFFFF))
    want32 = R.mask_to_words(R.doc_filter_mask(keys, deleted=del32, deleted_is32=True))
    assert np.array_equal(se.build_filter(zv.DocFilter(delete=R.file_image(R.serialize32(del32), True), kind="file")), want32)
    # empty bitmaps: nothing deleted / nothing matched
    assert not se.build_filter(zv.DocFilter(delete=R.serialize32([]))).any()
    allx = se.build_filter(zv.DocFilter(invert=R.serialize32([])))
    assert np.array_equal(allx, R.mask_to_words(np.ones(n, bool)))


def test_filter_rejects_malformed_streams(zv):
    se, _ = _flat(zv, 100)
    good = R.serialize32(range(0, 200000, 3))
    for bad in (b"", b"\x01\x02\x03", good[:-1], good[:20], b"\xff" * 64):
        with pytest.raises(zv._lib.ZvecHipError) as e:
            se.build_filter(zv.DocFilter(delete=bad))
        assert e.value.code == zv.IndexError_.InvalidArgument
    img = bytearray(R.file_image(good, True))
    img[70] ^= 1                                           # payload no longer matches the header's crc32c
    with pytest.raises(zv._lib.ZvecHipError) as e:
        se.build_filter(zv.DocFilter(delete=bytes(img), kind="file"))
    assert e.value.code == zv.IndexError_.Mismatch
    img = bytearray(R.file_image(good, True))
    img[0] ^= 1                                            # magic
    with pytest.raises(zv._lib.ZvecHipError) as e:
        se.build_filter(zv.DocFilter(delete=bytes(img), kind="file"))
    assert e.value.code == zv.IndexError_.Mismatch


def test_flat_search_with_doc_filter_matches_oracle(zv, oracle):
    rng = np.random.default_rng(9)
    n, dim, nq, k = 70_000, 24, 48, 10                    # >= 65536 rows: the sparse keep-set path is exercised too
    keys = rng.permutation(4 * n)[:n].astype(np.uint64)
    base = rng.integers(-6, 6, (n, dim)).astype(np.float32)
    q = rng.integers(-6, 6, (nq, dim)).astype(np.float32)
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base, keys) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    for keep in (0.5, 0.05):
        matched = np.unique(rng.choice(keys, int(keep * n), replace=False))
        deleted = np.unique(rng.choice(keys, n // 20, replace=False))
        ctx.set_doc_filter(zv.DocFilter(delete=R.serialize32(deleted), invert=R.serialize32(matched, True)))
        assert se.search_impl(q, nq, ctx) == 0
        ex = R.mask_to_words(R.doc_filter_mask(keys, deleted=deleted, invert=matched))
        ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2, keys=keys, exclude_bits=ex)
        tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="doc filter keep=%g" % keep)
        kept = set(np.asarray(keys)[~R.doc_filter_mask(keys, deleted=deleted, invert=matched)].tolist())
        assert all(int(x) in kept for qi in range(nq) for x in ctx.keys[qi, : ctx.counts[qi]])


def test_ivf_filter_is_over_list_order_positions(zv, oracle):
    rng = np.random.default_rng(13)
    n, dim, nlist, nq, k = 6000, 16, 24, 20, 8
    base = rng.standard_normal((n, dim)).astype(np.float32)
    keys = rng.permutation(10 * n)[:n].astype(np.uint64)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.5, brute_force_threshold=10)
    assert se.build(base, nlist, keys=keys, kmeans_iters=4) == 0
    lkeys = se.keys_in_list_order()
    deleted = np.unique(rng.choice(keys, n // 3, replace=False))
    f = zv.DocFilter(delete=R.serialize32(deleted, True))
    got = se.build_filter(f)
    assert np.array_equal(got, R.mask_to_words(R.doc_filter_mask(lkeys, deleted=deleted)))
    ctx = se.create_context()
    ctx.set_topk(k)
    ctx.set_doc_filter(f)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    assert se.search_bf_impl(q, nq, ctx) == 0
    dset = set(deleted.tolist())
    for qi in range(nq):
        assert ctx.counts[qi] == k
        assert not any(int(x) in dset for x in ctx.keys[qi, :k])


def test_filter_build_throughput_10m(zv, capsys):
    """10M positions (BASELINE configs[4] size), keep 10 %: device-resident output, timed after a warm-up.
    Not a parity case — it records what DESIGN.md quotes for the kernel (printed with -s)."""
    import time
    import torch
    n, dim = 10_000_000, 4
    dev = torch.device("cuda:0")
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    vec = torch.zeros((n, dim), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    assert se.add_batch_dev(vec.data_ptr(), n) == 0            # keys = positions
    rng = np.random.default_rng(3)
    matched = np.unique(rng.choice(n, n // 10, replace=False))
    deleted = np.unique(rng.choice(n, n // 100, replace=False))
    f = zv.DocFilter(delete=R.serialize32(deleted), invert=R.serialize32(matched))
    out = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    ctx = se.create_context()
    se.build_filter(f, ctx, d_out=out.data_ptr())
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        se.build_filter(f, ctx, d_out=out.data_ptr())
    ms = (time.perf_counter() - t0) / reps * 1e3
    want = R.mask_to_words(R.doc_filter_mask(np.arange(n, dtype=np.uint64), deleted=deleted, invert=matched))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    with capsys.disabled():
        print("\n[filter build] 10M positions, delete %d B + invert %d B of roaring: %.3f ms per build (parse + upload + kernel)"
              % (len(f.delete), len(f.invert), ms))


@pytest.mark.parametrize("metric_name,metric", [("SquaredEuclidean", O.METRIC_L2), ("InnerProduct", O.METRIC_IP)])
@pytest.mark.parametrize("keep", [0.8, 0.4, 0.02])
def test_wide_batch_sparse_filter_gathers_kept_rows(zv, oracle, metric_name, metric, keep):
    """> 64 queries over >= 65536 rows with a minority kept: the wide kernel fetches the kept rows from their stored
    positions (no compaction copy).  Integer data: scores and ids bit-exact against the oracle's filtered scan."""
    rng = np.random.default_rng(int(keep * 1000) + metric)
    n, dim, nq, k = 66_000, 40, 150, 10
    base = rng.integers(-5, 6, (n, dim)).astype(np.float32)
    q = rng.integers(-5, 6, (nq, dim)).astype(np.float32)
    keys = rng.permutation(3 * n)[:n].astype(np.uint64)
    se = zv.HipFlatSearcher(dim, metric_name)
    assert se.load(base, keys) == 0
    drop = rng.random(n) >= keep
    ex = O.pack_bits(drop)
    ctx = se.create_context()
    ctx.set_topk(k)
    ctx.set_exclude_bitset(ex)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k, metric, keys=keys, exclude_bits=ex)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="gather keep=%g %s" % (keep, metric_name))
    kept = set(keys[~drop].tolist())
    assert all(int(x) in kept for qi in range(nq) for x in ctx.keys[qi, : ctx.counts[qi]])


@pytest.mark.parametrize("kept", [0, 1, 5, 127, 129])
def test_wide_batch_filter_with_a_handful_of_kept_rows(zv, oracle, kept):
    """edge of the gather path: fewer kept rows than one tile / than topk, and none at all"""
    rng = np.random.default_rng(kept)
    n, dim, nq, k = 66_000, 16, 130, 10
    base = rng.integers(-5, 6, (n, dim)).astype(np.float32)
    q = rng.integers(-5, 6, (nq, dim)).astype(np.float32)
    drop = np.ones(n, bool)
    drop[rng.choice(n, kept, replace=False)] = False
    ex = O.pack_bits(drop)
    se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert se.load(base) == 0
    ctx = se.create_context()
    ctx.set_topk(k)
    ctx.set_exclude_bitset(ex)
    assert se.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2, exclude_bits=ex)
    assert np.array_equal(ctx.counts, oc) and int(oc.max(initial=0)) == min(k, kept)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="gather kept=%d" % kept)
