"""Sharded IVF on the GPU, single process: the index is cut into `world` shards by inverted list
(zvec_hip_ivf_keep_shard: the byte-balanced list -> shard map bench.py uses across ranks), every shard answers the whole
batch, the candidate lists are merged with zvec_hip_merge_topk_dev — the result must equal the
unsharded search and the oracle.  Covers everything of the N>1 path except the RCCL transport
(which tests/test_dist_cpu.py covers over gloo)."""
import numpy as np
import pytest

from tests.util import tie_tolerant_compare, kmeans_lists

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_search_equals_unsharded(oracle, world):
    import zvec_amd
    from zvec_amd.index import merge_topk
    rng = np.random.default_rng(100 + world)
    n, dim, nlist, nq, k = 20000, 64, 96, 150, 10
    base = rng.integers(0, 64, (n, dim)).astype(np.float32)
    q = rng.integers(0, 64, (nq, dim)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent)
    vecs, keys = base[order], order.astype(np.uint64)

    def make(shard, nshards):
        se = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=12 / 96., brute_force_threshold=100)
        assert se.set_shard(shard, nshards) == 0
        assert se.load(cent, offs, vecs, keys) == 0
        se.total_count = n                      # probe parameters come from the WHOLE index
        return se

    full = make(0, 1)
    nprobe, max_scan = full.probe_params()
    ctx = full.create_context()
    ctx.set_topk(k)
    assert full.search_impl(q, nq, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, max_scan, keys=keys)
    parts_k, parts_s, parts_c = [], [], []
    total_rows = 0
    for r in range(world):
        sh = make(r, world)
        total_rows += sh.info()[0]
        c = sh.create_context()
        c.set_topk(k)
        assert sh.search_impl(q, nq, c) == 0
        parts_k.append(c.keys.copy()); parts_s.append(c.scores.copy()); parts_c.append(c.counts.copy())
    assert total_rows == n                      # shards partition the rows
    mk, ms, mc = merge_topk(ctx, np.stack(parts_k), np.stack(parts_s), np.stack(parts_c), k)
    tie_tolerant_compare(mk, ms, mc, ctx.keys, ctx.scores, ctx.counts, what="sharded vs unsharded")
    from tests.util import exact_l2
    cd = np.sort(exact_l2(cent, q), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, nprobe])[0]
    tie_tolerant_compare(mk[sel], ms[sel], mc[sel], ok[sel], os_[sel], oc[sel], what="sharded vs oracle")


def test_sharded_build_is_deterministic_across_shards():
    """every rank builds from the same data + seed and keeps its own lists: the union must be the
    whole index with identical centroids (bench.py relies on this for N>1)."""
    import zvec_amd
    rng = np.random.default_rng(7)
    n, dim, nlist = 12000, 32, 48
    base = (rng.standard_normal((n, 6)) @ rng.standard_normal((6, dim))).astype(np.float32)
    cents, rows_all, sizes_all, owners = [], [], [], []
    for r in range(3):
        se = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean")
        assert se.set_shard(r, 3) == 0
        assert se.build(base, nlist, kmeans_iters=5, sample_per_list=64, seed=5) == 0
        c, offs, rows = se.export()
        cents.append(c)
        rows_all.append(rows)
        sizes_all.append(np.diff(offs.astype(np.int64)))
        owners.append(se.list_owners())
    assert np.array_equal(cents[0], cents[1]) and np.array_equal(cents[1], cents[2])
    # every rank derives the same byte-balanced list -> shard map from the global list sizes
    from tests.util import lpt_owner
    sizes_global = sizes_all[0] + sizes_all[1] + sizes_all[2]
    want = lpt_owner(sizes_global, 3)
    for r in range(3):
        assert np.array_equal(owners[r], want)
        assert np.array_equal(sizes_all[r], np.where(want == r, sizes_global, 0))
    tiles = np.array([((sizes_all[r] + 127) // 128).sum() for r in range(3)])
    assert tiles.max() - tiles.min() <= max(2, ((sizes_global + 127) // 128).max())
    allrows = np.sort(np.concatenate(rows_all))
    assert np.array_equal(allrows, np.arange(n, dtype=np.uint64))


def test_concurrent_contexts_on_one_index(oracle):
    """The reference hands one context to each searching thread (index.cc:24-45).  Here: one zvec_hip_ctx_t per thread
    (own stream + workspace) on shared flat and IVF handles; ctypes drops the GIL during the calls, so the searches
    really overlap.  Every thread must get exactly what a serial search returns."""
    import threading
    import zvec_amd as zv
    rng = np.random.default_rng(21)
    n, dim, nq, k, nlist = 30000, 32, 64, 10, 40
    base = rng.integers(-9, 10, (n, dim)).astype(np.float32)
    flat = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert flat.load(base) == 0
    ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.2, brute_force_threshold=10)
    assert ivf.build(base, nlist, kmeans_iters=3) == 0
    queries = [rng.integers(-9, 10, (nq, dim)).astype(np.float32) for _ in range(4)]
    want = []
    for q in queries:
        c1, c2 = flat.create_context(), ivf.create_context()
        c1.set_topk(k), c2.set_topk(k)
        assert flat.search_impl(q, nq, c1) == 0 and ivf.search_impl(q, nq, c2) == 0
        want.append((c1.keys.copy(), c1.scores.copy(), c2.keys.copy(), c2.scores.copy()))
    errors = []

    def worker(t):
        try:
            c1, c2 = flat.create_context(), ivf.create_context()
            c1.set_topk(k), c2.set_topk(k)
            for _ in range(25):
                assert flat.search_impl(queries[t], nq, c1) == 0 and ivf.search_impl(queries[t], nq, c2) == 0
                assert np.array_equal(c1.keys, want[t][0]) and np.array_equal(c1.scores, want[t][1])
                assert np.array_equal(c2.keys, want[t][2]) and np.array_equal(c2.scores, want[t][3])
        except Exception as e:   # noqa: BLE001 - reported below
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_shared_context_across_indexes(oracle):
    """ivf_searcher_test.cc:2700-2830 (TestSharedContext): a context created by one index is handed to other index
    instances (different sizes, dimensions and types), from several threads, in random order.  The context here is an
    index-agnostic workspace (stream + buffers), so every search must return what a private context returns."""
    import threading
    import zvec_amd as zv
    rng = np.random.default_rng(77)
    specs = [("flat", 900, 8), ("flat", 5000, 40), ("ivf", 7000, 24)]
    idx, data = [], []
    for kind, n, dim in specs:
        base = rng.integers(-9, 10, (n, dim)).astype(np.float32)
        if kind == "flat":
            se = zv.HipFlatSearcher(dim, "SquaredEuclidean")
            assert se.load(base) == 0
        else:
            se = zv.HipIVFSearcher(dim, "SquaredEuclidean", scan_ratio=0.3, brute_force_threshold=10)
            assert se.build(base, 16, kmeans_iters=3) == 0
        q = rng.integers(-9, 10, (20, dim)).astype(np.float32)
        own = se.create_context()
        own.set_topk(7)
        assert se.search_impl(q, 20, own) == 0
        idx.append(se)
        data.append((q, own.keys.copy(), own.scores.copy()))
    errors = []

    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            ctx = idx[seed % 3].create_context()          # created by one index ...
            ctx.set_topk(7)
            for _ in range(40):
                j = int(r.integers(0, 3))                 # ... used with any of them
                q, wk, ws = data[j]
                assert idx[j].search_impl(q, 20, ctx) == 0
                assert np.array_equal(ctx.keys, wk) and np.array_equal(ctx.scores, ws)
        except Exception as e:   # noqa: BLE001 - reported below
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_concurrent_add_and_search(oracle):
    """flat_streamer_test.cc TestConcurrentAddAndSearch: the streamer is searched while another thread keeps adding
    (growth reallocations included: the store starts empty and grows 1.5x at a time).  Every search must see a
    consistent prefix of the rows: the planted nearest neighbour (added first) is always rank 0 with score 0, every
    returned key exists, and at the end the index equals one built in a single call."""
    import threading
    import zvec_amd as zv
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    dim, batches, per = 24, 60, 700
    base = rng.integers(-9, 10, (batches * per, dim)).astype(np.float32)
    base[0] = 40.0                                           # far from everything else: a unique nearest neighbour
    q = np.repeat(base[:1], 8, 0)
    st = zv.HipFlatStreamer(dim, "SquaredEuclidean")
    assert st.add_batch(base[:per]) == 0
    errors, done = [], threading.Event()

    def adder():
        try:
            for b in range(1, batches):
                assert st.add_batch(base[b * per:(b + 1) * per]) == 0
        except Exception as e:   # noqa: BLE001
            errors.append(("add", repr(e)))
        finally:
            done.set()

    def searcher(seed):
        try:
            ctx = st.create_context()
            ctx.set_topk(5)
            n_seen = 0
            while not done.is_set() or n_seen == 0:
                assert st.search_impl(q, 8, ctx) == 0
                hi = st.count()
                assert (ctx.counts == 5).all()
                assert (ctx.keys[:, 0] == 0).all() and (ctx.scores[:, 0] == 0).all()
                assert int(ctx.keys.max()) < hi
                n_seen += 1
        except Exception as e:   # noqa: BLE001
            errors.append(("search", seed, repr(e)))

    ts = [threading.Thread(target=adder)] + [threading.Thread(target=searcher, args=(s,)) for s in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    assert st.count() == batches * per
    qq = rng.integers(-9, 10, (30, dim)).astype(np.float32)
    ctx = st.create_context()
    ctx.set_topk(10)
    assert st.search_impl(qq, 30, ctx) == 0
    ok, os_, _, oc = oracle.flat_search(base, qq, 10, O.METRIC_L2)
    tie_tolerant_compare(ctx.keys, ctx.scores, ctx.counts, ok, os_, oc, what="after concurrent adds")


@pytest.mark.parametrize("ndev", [1, 2, 5])
def test_in_process_shards_flat_equals_one_index(oracle, ndev):
    """zvec_hip_shards_* (one handle, G device shards, worker thread per shard, peer-copied candidate lists, GPU merge) —
    all shards on device 0 here; the fan-out / merge must reproduce the single-index answer (and the oracle's), with
    explicit keys, several appends (rows dealt in G pieces each time), a global exclude bitset and an RNN radius."""
    import zvec_amd
    from oracle import oracle as O
    rng = np.random.default_rng(500 + ndev)
    dim, nq, k = 48, 70, 10
    parts = [rng.integers(-9, 10, (m, dim)).astype(np.float32) for m in (3001, 17, 4999)]
    base = np.concatenate(parts)
    n = base.shape[0]
    keys = (np.arange(n, dtype=np.uint64) * 7 + 3)
    q = rng.integers(-9, 10, (nq, dim)).astype(np.float32)
    sh = zvec_amd.HipShardedIndex("flat", dim, "SquaredEuclidean", devices=[0] * ndev)
    o = 0
    for p in parts:
        assert sh.append(p, keys[o:o + p.shape[0]]) == 0
        o += p.shape[0]
    total, per = sh.counts()
    assert total == n and per.sum() == n and (per > 0).all()
    gk, gs, gc = sh.search(q, k)
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2, keys=keys)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="in-process flat shards")
    mask = rng.random(n) < 0.5
    gk, gs, gc = sh.search(q, k, exclude=O.pack_bits(mask), threshold=float(np.median(os_[:, -1])))
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2, keys=keys, exclude_bits=O.pack_bits(mask),
                                        threshold=float(np.median(os_[:, -1])))
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="in-process flat shards, filter + radius")
    # keys NULL -> global storage positions
    sh2 = zvec_amd.HipShardedIndex("flat", dim, "SquaredEuclidean", devices=[0] * ndev)
    for p in parts:
        assert sh2.append(p) == 0
    gk, gs, gc = sh2.search(q, k)
    ok, os_, _, oc = oracle.flat_search(base, q, k, O.METRIC_L2)
    tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="in-process flat shards, position keys")


@pytest.mark.parametrize("ndev", [1, 3])
def test_in_process_shards_ivf_equals_one_index(oracle, ndev):
    import zvec_amd
    from oracle import oracle as O
    rng = np.random.default_rng(600 + ndev)
    n, dim, nlist, nq, k, nprobe = 30000, 40, 64, 90, 10, 8
    base = rng.integers(-8, 9, (n, dim)).astype(np.float32)
    q = rng.integers(-8, 9, (nq, dim)).astype(np.float32)
    # (1) load of a given structure: compare with the oracle on that structure
    cent, offs, order = kmeans_lists(rng, base, nlist)
    cent = np.round(cent)
    vecs, keys = base[order], order.astype(np.uint64)
    sh = zvec_amd.HipShardedIndex("ivf", dim, "SquaredEuclidean", devices=[0] * ndev)
    assert sh.load(cent, offs, vecs, keys) == 0
    total, per = sh.counts()
    assert total == n
    from tests.util import lpt_owner
    sizes = np.diff(offs.astype(np.int64))
    owner = lpt_owner(sizes, ndev)
    assert np.array_equal(per.astype(np.int64), np.array([sizes[owner == g].sum() for g in range(ndev)]))
    gk, gs, gc = sh.search(q, k, nprobe=nprobe, max_scan=n)
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, n, keys=keys)
    from tests.util import exact_l2
    cd = np.sort(exact_l2(cent, q), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, nprobe])[0]
    tie_tolerant_compare(gk[sel], gs[sel], gc[sel], ok[sel], os_[sel], oc[sel], what="in-process ivf shards vs oracle")
    mask = rng.random(n) < 0.3                                        # global exclude set over list-order positions
    gk, gs, gc = sh.search(q, k, nprobe=nprobe, max_scan=n, exclude=O.pack_bits(mask))
    ok, os_, _, oc, _ = oracle.ivf_search(cent, offs, vecs, q, k, nprobe, n, keys=keys, exclude_bits=O.pack_bits(mask))
    tie_tolerant_compare(gk[sel], gs[sel], gc[sel], ok[sel], os_[sel], oc[sel], what="in-process ivf shards, filter")
    # the coarse pass DEALT over the shards (peer copies of the probe-list slices, zvec_hip_shards_deal_coarse): the very same lists,
    # also for a batch smaller than the shard count and with the filter on
    plain = sh.search(q, k, nprobe=nprobe, max_scan=n)
    assert sh.deal_coarse(True) == 0
    for qq, ex in ((q, None), (q[:2], None), (q, O.pack_bits(mask))):
        a = sh.search(qq, k, nprobe=nprobe, max_scan=n, exclude=ex)
        assert sh.deal_coarse(False) == 0
        b = sh.search(qq, k, nprobe=nprobe, max_scan=n, exclude=ex)
        assert sh.deal_coarse(True) == 0
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert sh.deal_coarse(False) == 0
    assert all(np.array_equal(x, y) for x, y in zip(sh.search(q, k, nprobe=nprobe, max_scan=n), plain))
    # (2) sharded build == unsharded build (same sample / seed rule): same answers as one zvec_hip_ivf_build index
    one = zvec_amd.HipIVFSearcher(dim, "SquaredEuclidean")
    assert one.build(base, nlist, kmeans_iters=4, sample_per_list=64, seed=11) == 0
    ctx = one.create_context()
    ctx.set_topk(k)
    one.set_nprobe(nprobe)
    assert one.search_impl(q, nq, ctx) == 0
    sb = zvec_amd.HipShardedIndex("ivf", dim, "SquaredEuclidean", devices=[0] * ndev)
    assert sb.build(base, nlist, kmeans_iters=4, sample_per_list=64, seed=11) == 0
    gk, gs, gc = sb.search(q, k, nprobe=nprobe, max_scan=n - 1)
    tie_tolerant_compare(gk, gs, gc, ctx.keys, ctx.scores, ctx.counts, what="sharded build vs one-call build")


def test_gated_contexts_pipeline_batches_with_identical_results():
    """zvec_hip_gate_t: two contexts on two streams share a gate, consecutive batches alternate between them (what
    bench.py times): the dominant scan kernels run in call order, the rest overlaps.  Every batch must get exactly the
    answer of an ungated, single-context search — IVF and flat, from one host thread and from two."""
    import threading
    import torch
    import zvec_amd as zv
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    n, dim, nlist, nq, k, nprobe = 200_000, 64, 256, 128, 10, 12
    proj = torch.randn((8, dim), generator=g, device=dev)
    base = (torch.randn((n, 8), generator=g, device=dev) @ proj).contiguous()
    batches = [(torch.randn((nq, 8), generator=g, device=dev) @ proj).contiguous() for _ in range(6)]
    s0 = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(s0)
    ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean")
    assert ivf.build_dev(base.data_ptr(), n, nlist, kmeans_iters=3, sample_per_list=64, stream=s0.cuda_stream) == 0
    flat = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert flat.add_batch_dev(base.data_ptr(), n, stream=s0.cuda_stream) == 0
    torch.cuda.synchronize()

    def out():
        return (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                torch.empty((nq,), dtype=torch.int32, device=dev))

    def run(idx, ctx, q, o, sp):
        if idx is ivf:
            rc = idx.search_dev(q.data_ptr(), nq, k, nprobe, n, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), ctx, stream=sp)
        else:
            rc = idx.search_dev(q.data_ptr(), nq, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), ctx, stream=sp)
        assert rc == 0
    for idx in (ivf, flat):
        ref_ctx = idx.create_context()
        ref_ctx.set_stream(s0.cuda_stream)
        want = []
        for q in batches:
            o = out()
            run(idx, ref_ctx, q, o, s0.cuda_stream)
            torch.cuda.synchronize()
            want.append([t.clone() for t in o])
        gate = zv.Gate(0)
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        ctxs = [idx.create_context(), idx.create_context()]
        for c, s in zip(ctxs, streams):
            c.set_stream(s.cuda_stream)
            c.set_gate(gate)
        torch.cuda.synchronize()
        # one host thread, alternating lanes, several rounds without synchronising in between
        outs = [out() for _ in batches]
        for rnd in range(3):
            for i, q in enumerate(batches):
                run(idx, ctxs[i % 2], q, outs[i], streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(batches)):
            assert all(torch.equal(a, b) for a, b in zip(outs[i], want[i])), (idx is ivf, i)
        # two host threads, one lane each
        outs2 = [out() for _ in batches]
        errs = []

        def worker(lane):
            try:
                for rnd in range(3):
                    for i in range(lane, len(batches), 2):
                        run(idx, ctxs[lane], batches[i], outs2[i], streams[lane].cuda_stream)
            except Exception as e:   # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=worker, args=(l,)) for l in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        assert not errs, errs
        for i in range(len(batches)):
            assert all(torch.equal(a, b) for a, b in zip(outs2[i], want[i])), ("threads", idx is ivf, i)
        for c in ctxs:
            c.set_gate(None)


@pytest.mark.parametrize("ndev", [1, 3])
def test_in_process_shards_load_reference_dumped_files(ndev):
    """the segment-payload loaders over shards: golden index files dumped by the reference's own writers
    (tests/golden/ref_index_files.npz) -> container parser -> zvec_hip_shards_{flat_load_features, ivf_load_segments} ->
    the sharded handle answers exactly like a single-device index opened from the same file."""
    import os
    import zvec_amd as zv
    from zvec_amd.index import container_segments, parse_index_meta
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_index_files.npz"))
    rng = np.random.default_rng(8)
    for name in [str(x) for x in z["cases"]]:
        image = z[name + "_image"].tobytes()
        seg = container_segments(image)
        meta = parse_index_meta(image[seg["IndexMeta"][0]:sum(seg["IndexMeta"])])
        f16 = meta["data_type"] == 1
        dt, dim = (np.float16 if f16 else np.float32), meta["dimension"]
        q = rng.integers(-8, 9, (11, dim)).astype(dt)

        def pay(sid):
            return image[seg[sid][0]:seg[sid][0] + seg[sid][1]]
        if name.startswith("flat"):
            one = zv.open_flat_file(image)
            sh = zv.HipShardedIndex("flat", dim, "InnerProduct", devices=[0] * ndev, dtype="fp16" if f16 else "fp32")
            keys = np.frombuffer(pay("flat.keys"), np.uint64)
            assert sh.load_features(pay("flat.features"), keys.size, column_major=(meta["major_order"] == 2), keys=keys) == 0
            assert sh.counts()[0] == keys.size
            c = one.create_context()
            c.set_topk(5)
            assert one.search_impl(q, 11, c) == 0
            gk, gs, gc = sh.search(q, 5)
        else:
            one = zv.open_ivf_file(image)
            n, nlist = one.info()
            cent = one.export()[0]
            sh = zv.HipShardedIndex("ivf", dim, "InnerProduct", devices=[0] * ndev, dtype="fp16" if f16 else "fp32")
            assert sh.load_segments(pay("ivf.inverted_header"), pay("ivf.inverted_meta"), pay("ivf.inverted_body"), pay("hc.keys"), cent) == 0
            assert sh.counts()[0] == n
            one.scan_ratio, one.brute_force_threshold = 1.0, 0
            c = one.create_context()
            c.set_topk(5)
            assert one.search_impl(q, 11, c) == 0
            gk, gs, gc = sh.search(q, 5, nprobe=nlist, max_scan=n)
        tie_tolerant_compare(gk, gs, gc, c.keys, c.scores, c.counts, what="sharded load of " + name)


def test_in_process_flat_shards_by_ids_and_fetch(oracle):
    """search_bf_by_p_keys_impl and the fetch_vector gather over 3 shards fed by two appends; the oracle answer is a
    flat scan with every position that is not listed (or is excluded) masked out."""
    import zvec_amd as zv
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    n, dim, nq, k = 3001, 40, 9, 6
    base = rng.integers(-8, 9, (n, dim)).astype(np.float32)
    q = rng.integers(-8, 9, (nq, dim)).astype(np.float32)
    sh = zv.HipShardedIndex("flat", dim, "SquaredEuclidean", devices=[0, 0, 0])
    sh.append(base[:1700])
    sh.append(base[1700:])
    ids = [rng.choice(n + 50, size=int(rng.integers(0, 60)), replace=False) for _ in range(nq)]   # some unknown positions
    ids[3] = np.zeros(0, np.int64)
    exm = np.zeros(n, bool)
    exm[rng.choice(n, 400, replace=False)] = True
    gk, gs, gc = sh.search_by_ids(q, ids, k, exclude=O.pack_bits(exm))
    for i in range(nq):
        mask = np.ones(n, bool)
        mask[[int(x) for x in ids[i] if x < n]] = False
        ok, os_, _, oc = oracle.flat_search(base, q[i:i + 1], k, exclude_bits=O.pack_bits(mask | exm))
        tie_tolerant_compare(gk[i:i + 1], gs[i:i + 1], gc[i:i + 1], ok, os_, oc, what="sharded by_ids q%d" % i)
    pos = rng.choice(n, 77, replace=False)
    rc, rows = sh.get_vectors(pos)
    assert rc == 0 and np.array_equal(rows, base[pos])
    assert sh.get_vectors([n + 3])[0] != 0
