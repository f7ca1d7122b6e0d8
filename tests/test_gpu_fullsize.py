"""BASELINE.json full-size cases on the GPU, checked through size-independent properties, oracle spot checks on a few queries
and — where oracle/_ref/libzvec_ref_core.so travelled — against THE REFERENCE ITSELF at full size: its own FlatSearcher over the
same 1M rows for all 256 queries of configs[1], its own IVFSearcher over the exported 10M index for 256 of configs[2]'s queries and
over one rank's exported 12.5M-row fp16 share for 64 of configs[3]'s (flat_searcher.cc:162-211, ivf_searcher.cc:183-250; the body of
the exported index is lent to it, not copied: oracle/ref_core_shim.cc zref_ivf_searcher_over_rows).

  configs[0]  Flat L2, SIFT-1M-shaped (1M x 128, uint8-valued fp32), batch 1      -> bit-exact vs oracle
  configs[1]  Flat IP, 1M x 768 fp32, batch 256                                   -> self-query / order / oracle spot
  configs[2]  IVF-Flat nlist 4096, 10M x 768 fp32, batch 1024                     -> bf==flat, shard union, self-query
  configs[3]  IVF-Flat nlist 16384, 100M x 768 fp16 over 8 GPUs: ONE rank's share  -> shard 0 of 8 (12.5M rows, 19.2 GB)
              held on this GPU, built streamed (the corpus is never resident): byte balance, self-query, keys stay
              inside the shard, oracle spot check on the probed lists
  configs[4]  filtered scan, 10M x 768 + bitmap, batch 512                        -> gate respected, equals unfiltered∖mask
Data is generated on the GPU with torch (seeded), indexes are built through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import tie_tolerant_compare

pytestmark = pytest.mark.gpu


def _refcore():
    """the reference's own core library (oracle/_ref/libzvec_ref_core.so, compiled in place; it travels to the GPU box), or None"""
    from oracle import refcore
    return refcore if refcore.available() else None


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


def _stream():
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    return s.cuda_stream


def _search_dev(idx, ctx, q, k, stream, exclude=None, **kw):
    nq = q.shape[0]
    keys = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    counts = torch.empty((nq,), dtype=torch.int32, device=q.device)
    rc = idx.search_dev(q.data_ptr(), nq, k, *kw.get("args", ()), keys.data_ptr(), scores.data_ptr(), counts.data_ptr(), ctx,
                        d_exclude=exclude.data_ptr() if exclude is not None else None, stream=stream)
    assert rc == 0
    torch.cuda.synchronize()
    return keys.cpu().numpy().astype(np.uint64), scores.cpu().numpy(), counts.cpu().numpy().astype(np.uint32)


def test_config0_flat_l2_sift1m_shape_batch1_bit_exact(zv, oracle):
    stream = _stream()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(20260320)
    n, dim, k = 1_000_000, 128, 10
    base = torch.randint(0, 256, (n, dim), generator=g, device=dev).float()
    qs = torch.randint(0, 256, (4, dim), generator=g, device=dev).float()
    flat = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert flat.add_batch_dev(base.data_ptr(), n, stream=stream) == 0
    ctx = flat.create_context()
    ctx.set_stream(stream)
    hb = base.cpu().numpy()
    for i in range(4):                                    # batch = 1, as the config says
        gk, gs, gc = _search_dev(flat, ctx, qs[i:i + 1].contiguous(), k, stream)
        ok, os_, _, oc = oracle.flat_search(hb, qs[i:i + 1].cpu().numpy(), k, threads=8)
        tie_tolerant_compare(gk, gs, gc, ok, os_, oc, what="config0 q%d" % i)   # atol=rtol=0: bit exact


def test_config1_flat_ip_1m_768_batch256_properties(zv, oracle):
    stream = _stream()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(20260321)
    n, dim, nq, k = 1_000_000, 768, 256, 10
    base = torch.randn((n, dim), generator=g, device=dev)
    base /= base.norm(dim=1, keepdim=True)                # unit rows: the best IP match of a row is itself
    sel = torch.randint(0, n, (nq,), generator=g, device=dev)
    q = base[sel].contiguous()
    flat = zv.HipFlatSearcher(dim, "InnerProduct")
    assert flat.add_batch_dev(base.data_ptr(), n, stream=stream) == 0
    ctx = flat.create_context()
    ctx.set_stream(stream)
    gk, gs, gc = _search_dev(flat, ctx, q, k, stream)
    assert (gc == k).all()
    assert np.array_equal(gk[:, 0], sel.cpu().numpy().astype(np.uint64))         # self-query => itself first
    assert np.all(np.abs(gs[:, 0] + 1.0) < 1e-5)                                 # score = -<x,x> = -1
    assert np.all(np.diff(gs, axis=1) >= 0)                                      # ascending
    # permutation invariance of the batch (no cross-query leakage): reversed batch gives reversed rows
    rk, rs, rc = _search_dev(flat, ctx, q.flip(0).contiguous(), k, stream)
    assert np.array_equal(rk[::-1], gk) and np.array_equal(rs[::-1], gs)
    # oracle spot check on 3 queries of the batch
    hb = base.cpu().numpy()
    hq = q[:3].cpu().numpy()
    ok, os_, _, oc = oracle.flat_search(hb, hq, k, O.METRIC_IP, threads=8)
    tie_tolerant_compare(gk[:3], gs[:3], gc[:3], ok, os_, oc, rtol=4e-6, scale=1.0, what="config1 spot")
    # ALL 256 queries against the reference's own FlatBuilder + FlatSearcher over the same rows (one query per call, as the product
    # calls boundary B)
    R = _refcore()
    if R is not None:
        R.build("FlatBuilder", hb, "InnerProduct", "cfg1_flat")
        ref = R.Runner.searcher("FlatSearcher", "cfg1_flat", dim, np.float32)
        rk, rs, rc_, _ = ref.search_mt(q.cpu().numpy(), k, 16)
        ref.close()
        R.mem_remove("cfg1_flat")
        tie_tolerant_compare(gk, gs, gc, rk, rs, rc_, rtol=4e-6, scale=1.0, what="config1: all 256 queries vs the reference's FlatSearcher")


def test_flat_l2_1m_768_fp16_batch256_on_the_256_tile(zv, oracle):
    """bench.py --workload flat1m_fp16 at full size: 1M x 768 fp16 rows (HalfFloatConverter output), L2, 256 queries — the shape
    that takes scan256_f16_kernel with seeded bounds.  Size-independent properties, an oracle spot check, the 128 x 128 kernel on
    the same search, and 64 of the queries against the reference's own FlatBuilder + FlatSearcher over the same fp16 rows."""
    stream = _stream()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(20260325)
    n, dim, nq, k = 1_000_000, 768, 256, 10
    base = torch.randn((n, dim), generator=g, device=dev).half()
    sel = torch.randint(0, n, (nq,), generator=g, device=dev)
    q = base[sel].contiguous()                                                    # self-queries: distance 0 to their own row
    flat = zv.HipFlatSearcher(dim, "SquaredEuclidean", dtype="fp16")
    assert flat.add_batch_dev(base.data_ptr(), n, stream=stream) == 0
    ctx = flat.create_context()
    ctx.set_stream(stream)
    gk, gs, gc = _search_dev(flat, ctx, q, k, stream)
    assert (gc == k).all()
    assert np.array_equal(gk[:, 0], sel.cpu().numpy().astype(np.uint64)) and np.all(gs[:, 0] == 0.0)
    assert np.all(np.diff(gs, axis=1) >= 0)
    rk, rs, rc = _search_dev(flat, ctx, q.flip(0).contiguous(), k, stream)      # no cross-query leakage
    assert np.array_equal(rk[::-1], gk) and np.array_equal(rs[::-1], gs)
    from zvec_amd import _lib
    L = _lib.lib()
    assert L.zvec_hip_set_option(b"scan256", 0) == 0                             # the 128 x 128 kernel on the same search
    try:
        bk, bs, bc = _search_dev(flat, ctx, q, k, stream)
    finally:
        assert L.zvec_hip_set_option(b"scan256", 1) == 0
    tie_tolerant_compare(gk, gs, gc, bk, bs, bc, rtol=2e-6, atol=1e-4, what="fp16 1M: 256 x 256 tile vs 128 x 128 tile")
    hb = base.cpu().numpy()
    hq = q.cpu().numpy()
    ok, os_, _, oc = oracle.flat_search(hb, hq[:3], k, O.METRIC_L2, threads=8)
    tie_tolerant_compare(gk[:3], gs[:3], gc[:3], ok, os_, oc, rtol=2e-6, atol=1e-4, what="fp16 1M spot")
    R = _refcore()
    if R is not None:
        R.build("FlatBuilder", hb, "SquaredEuclidean", "f16_flat")
        ref = R.Runner.searcher("FlatSearcher", "f16_flat", dim, np.float16)
        rk, rs, rc_, _ = ref.search_mt(hq[:64], k, 16)
        ref.close()
        R.mem_remove("f16_flat")
        tie_tolerant_compare(gk[:64], gs[:64], gc[:64], rk, rs, rc_, rtol=2e-6, atol=1e-4,
                             what="fp16 1M: 64 queries vs the reference's FlatSearcher")


@pytest.fixture(scope="module")
def ten_million(zv):
    stream = _stream()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(20260322)
    n, dim, r = 10_000_000, 768, 12
    proj = torch.randn((r, dim), generator=g, device=dev)
    base = torch.empty((n, dim), device=dev)
    for o in range(0, n, 1 << 20):
        m = min(1 << 20, n - o)
        torch.mm(torch.randn((m, r), generator=g, device=dev), proj, out=base[o:o + m])
    torch.cuda.synchronize()
    return dict(stream=stream, dev=dev, g=g, n=n, dim=dim, base=base, proj=proj)


def test_config2_ivf_10m_768_batch1024_properties(zv, oracle, ten_million):
    t = ten_million
    stream, dev, n, dim, base = t["stream"], t["dev"], t["n"], t["dim"], t["base"]
    nq, k, nlist, nprobe = 1024, 10, 4096, 32
    sel = torch.randint(0, n, (nq,), generator=t["g"], device=dev)
    q = base[sel].contiguous()
    ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean")
    assert ivf.build_dev(base.data_ptr(), n, nlist, kmeans_iters=4, stream=stream) == 0
    cnt, nl = ivf.info()
    assert cnt == n and nl == nlist
    ctx = ivf.create_context()
    ctx.set_stream(stream)
    gk, gs, gc = _search_dev(ivf, ctx, q, k, stream, args=(nprobe, n))
    assert (gc == k).all() and np.all(np.diff(gs, axis=1) >= 0)
    # self-query: every row's own list is its nearest centroid's => found at distance 0 (duplicates aside)
    assert (gs[:, 0] == 0).all()
    assert (gk[:, 0] == sel.cpu().numpy().astype(np.uint64)).mean() > 0.999
    scanned, probes = ivf.last_stats(ctx, nq)
    assert (probes == nprobe).all()
    # the half-width pre-selection (zvec_hip_ivf_set_shadow) on all 1024 queries: fp16 shadow lists, fp32 re-scoring, certificate,
    # fp32 re-run of what it cannot certify == the fp32 route, keys and score BITS (which the rest of this test ties to the oracle
    # and to the reference's own IVFSearcher)
    ivf.set_shadow(True)
    sk = torch.empty((nq, k), dtype=torch.int64, device=dev)
    ss = torch.empty((nq, k), dtype=torch.float32, device=dev)
    sc = torch.empty((nq,), dtype=torch.int32, device=dev)
    assert ivf.search_dev(q.data_ptr(), nq, k, nprobe, n, sk.data_ptr(), ss.data_ptr(), sc.data_ptr(), ctx, stream=stream) == 0
    rerun = ivf.shadow_certify(q.data_ptr(), nq, k, nprobe, n, sk.data_ptr(), ss.data_ptr(), sc.data_ptr(), ctx, stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(sk.cpu().numpy().astype(np.uint64), gk) and np.array_equal(sc.cpu().numpy().astype(np.uint32), gc)
    assert np.array_equal(ss.cpu().numpy().view(np.uint32), gs.view(np.uint32))
    print("config2 through the shadow lists: %d of %d queries re-run in fp32" % (rerun, nq))
    ivf.set_shadow(False)
    del sk, ss, sc
    # shard union: lists l%2==0 / l%2==1 searched separately and merged == the unsharded answer
    parts = []
    for r in range(2):
        sh = zv.HipIVFSearcher(dim, "SquaredEuclidean")
        assert sh.set_shard(r, 2) == 0
        assert sh.build_dev(base.data_ptr(), n, nlist, kmeans_iters=4, stream=stream) == 0
        c = sh.create_context()
        c.set_stream(stream)
        parts.append(_search_dev(sh, c, q, k, stream, args=(nprobe, n)))
        del sh, c
    from zvec_amd.index import merge_topk
    mk, ms, mc = merge_topk(ctx, np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                            np.stack([p[2] for p in parts]), k)
    tie_tolerant_compare(mk, ms, mc, gk, gs, gc, what="10M shard union")
    # oracle spot check: 2 queries against the exported index (same centroids, same list order)
    cent, offs, rows = ivf.export()
    hq = q[:2].cpu().numpy()
    rows_t = torch.from_numpy(rows.astype(np.int64)).to(dev)
    # only the probed lists are needed on the host: take them from the oracle's own probe set
    _, _, _, _, _, pr = oracle.ivf_search(cent, np.zeros(nlist + 1, np.uint64), np.zeros((0, dim), np.float32), hq, k,
                                          nprobe, n, want_probes=True)
    need = np.unique(pr[pr != 0xffffffff])
    sizes = np.diff(offs.astype(np.int64))
    sub_offs = np.zeros(nlist + 1, np.int64)
    sub_offs[1:][need] = sizes[need]
    sub_offs = np.cumsum(sub_offs)
    idx = np.concatenate([np.arange(offs[l], offs[l + 1]) for l in need]).astype(np.int64)
    sub_vecs = base.index_select(0, rows_t[torch.from_numpy(idx).to(dev)]).cpu().numpy()
    ok, os_, _, oc, _ = oracle.ivf_search(cent, sub_offs.astype(np.uint64), sub_vecs, hq, k, nprobe, n, keys=rows[idx])
    qn = (hq.astype(np.float64) ** 2).sum(1)
    tie_tolerant_compare(gk[:2], gs[:2], gc[:2], ok, os_, oc, rtol=2e-6, atol=1e-6, select_band=4e-6 * (2 * qn.max() + 1),
                         what="10M oracle spot")
    # 256 of the 1024 queries against the reference's own IVFSearcher over the exported index (rows brought to the host in list
    # order; its IVFDumper writes the small segments, the 30.7 GB body is lent)
    R = _refcore()
    if R is not None:
        vecs = np.empty((n, dim), np.float32)
        for o in range(0, n, 1 << 20):
            e = min(n, o + (1 << 20))
            vecs[o:e] = base.index_select(0, rows_t[o:e]).cpu().numpy()
        ratio = float(np.float32(nprobe) / np.float32(nlist))
        ref = R.Runner.ivf_over_rows("IVFSearcher", cent, offs, vecs, rows.astype(np.uint64), "SquaredEuclidean",
                                     params={"proxima.ivf.searcher.scan_ratio": ratio, "proxima.ivf.searcher.brute_force_threshold": n - 1})
        m = 256
        rk, rs, rc_, _ = ref.search_mt(q[:m].cpu().numpy(), k, 16)
        ref.close()
        del vecs
        qn = (q[:m].cpu().numpy().astype(np.float64) ** 2).sum(1)
        tie_tolerant_compare(gk[:m], gs[:m], gc[:m], rk, rs, rc_, rtol=2e-6, atol=1e-6, select_band=4e-6 * (2 * qn.max() + 1),
                             what="config2: 256 queries vs the reference's IVFSearcher over the exported 10M index")


def test_config4_filtered_scan_10m_bitmap_batch512(zv, oracle, ten_million):
    t = ten_million
    stream, dev, n, dim, base = t["stream"], t["dev"], t["n"], t["dim"], t["base"]
    nq, k = 512, 10
    q = (torch.randn((nq, t["proj"].shape[0]), generator=t["g"], device=dev) @ t["proj"]).contiguous()
    flat = zv.HipFlatSearcher(dim, "SquaredEuclidean")
    assert flat.add_batch_dev(base.data_ptr(), n, stream=stream) == 0
    ctx = flat.create_context()
    ctx.set_stream(stream)
    uk, us, uc = _search_dev(flat, ctx, q, 4 * k, stream)                        # unfiltered top-40
    for p_keep in (0.5, 0.1):
        keep = torch.rand((n,), generator=t["g"], device=dev) < p_keep
        excl = (~keep).cpu().numpy()
        words = torch.from_numpy(O.pack_bits(excl).view(np.int64)).to(dev)
        fk, fs, fc = _search_dev(flat, ctx, q, k, stream, exclude=words)
        assert (fc == k).all()
        assert not excl[fk.astype(np.int64)].any()                              # gate respected
        # equals the unfiltered ranking with the excluded rows struck out (wherever 40 candidates suffice)
        for i in range(nq):
            allowed = [(s, kk) for s, kk in zip(us[i], uk[i]) if not excl[int(kk)]]
            if len(allowed) >= k:
                assert [kk for _, kk in allowed[:k]] == fk[i].tolist() or np.array_equal(
                    np.array([s for s, _ in allowed[:k]], np.float32), fs[i])
        if p_keep == 0.1:
            # ... and against the ORACLE (not the GPU's own unfiltered answer): the ~1M kept rows are brought to the host and the
            # restated CPU scan (flat_searcher_context.h:949-963: a filtered row is skipped before its distance) runs over exactly
            # them for the first queries of the batch — the gather variant of the wide kernel at full size
            kept_pos = torch.nonzero(keep).squeeze(1)
            assert kept_pos.numel() >= 900_000
            kept_rows = base.index_select(0, kept_pos).cpu().numpy()
            hq = q[:12].cpu().numpy()
            ok, os_, _, oc = oracle.flat_search(kept_rows, hq, k, O.METRIC_L2, keys=kept_pos.cpu().numpy().astype(np.uint64), threads=8)
            qn = (hq.astype(np.float64) ** 2).sum(1)
            bn = float((kept_rows[:4096].astype(np.float64) ** 2).sum(1).max())
            tie_tolerant_compare(fk[:12], fs[:12], fc[:12], ok, os_, oc, rtol=2e-6, atol=1e-6, select_band=4e-6 * (qn.max() + bn),
                                 what="10M filtered (keep 10 %): gather scan vs the oracle over the kept rows")
            del kept_rows


@pytest.mark.parametrize("nshards", [8, 1])
def test_config3_ivf_100m_768_fp16_one_rank_share(zv, oracle, nshards):
    """BASELINE configs[3] at one rank's real share: shard 0 of 8 of a 100M x 768 fp16, nlist-16384 index on ONE GPU — and
    (nshards = 1) the WHOLE index, 153.6 GB of rows, which the 288 GB of one MI355X hold.
    The corpus (153.6 GB) is generated chunk by chunk three times and never held: sample -> k-means, labels of all 100M
    rows (ivf_builder.h:253-274 on the GPU), then only the rows of the lists the byte-balanced map gives shard 0 are
    kept.  Arithmetic under test: fp16 rows, fp32 accumulation (euclidean_distance_matrix_fp16.cc:137,
    distance_matrix_accum_fp16.i:554-594); search loop ivf_searcher.cc:217-247."""
    from tests.util import lpt_owner
    stream = _stream()
    dev = torch.device("cuda", 0)
    n, dim, r, nlist, nq, k, nprobe = 100_000_000, 768, 12, 16384, 1024, 10, 64
    chunk = 1 << 21
    pg = torch.Generator(device=dev)
    pg.manual_seed(20260324)
    proj = torch.randn((r, dim), generator=pg, device=dev)

    def chunks():
        g = torch.Generator(device=dev)
        g.manual_seed(20260325)
        for o in range(0, n, chunk):
            m = min(chunk, n - o)
            x = torch.mm(torch.randn((m, r), generator=g, device=dev), proj)
            x += torch.randn((m, dim), generator=g, device=dev) * 0.02
            yield o, x.half()

    ivf = zv.HipIVFSearcher(dim, "SquaredEuclidean", dtype="fp16")
    assert ivf.set_shard(0, nshards) == 0
    S = 64 * nlist                                            # the first 1M rows (i.i.d. rows: any subset is a sample)
    sample = torch.cat([x for o, x in chunks() if o < S])[:S].contiguous()
    assert ivf.train_dev(sample.data_ptr(), S, nlist, kmeans_iters=4, seed=3, stream=stream) == 0
    del sample
    labels = torch.empty((n,), dtype=torch.int32, device=dev)
    for o, x in chunks():
        assert ivf.label_dev(x.data_ptr(), x.shape[0], labels[o:o + x.shape[0]].data_ptr(), stream=stream) == 0
    lab = labels.cpu().numpy().astype(np.uint32)
    del labels
    sizes = np.bincount(lab, minlength=nlist).astype(np.uint32)
    assert ivf.begin_lists(sizes) == 0
    for o, x in chunks():
        assert ivf.add_dev(x.data_ptr(), x.shape[0], lab[o:o + x.shape[0]], o, stream=stream) == 0
    assert ivf.end_lists() == 0
    # ---- the shard is what the byte-balanced map says, and the map is balanced ----
    owner, shard_rows = zv.shard_map(sizes, nshards)
    assert np.array_equal(owner, lpt_owner(sizes, nshards)) and np.array_equal(ivf.list_owners(), owner)
    cnt, nl = ivf.info()
    assert nl == nlist and cnt == int(shard_rows[0]) == int(sizes[owner == 0].sum())
    assert shard_rows.max() <= 1.001 * shard_rows.mean()
    if nshards == 8:
        assert 12_000_000 < cnt < 13_000_000                  # one eighth of 100M: 19.2 GB of fp16 rows in HBM
    else:
        assert cnt == n                                       # all of it: 153.6 GB
    cent, offs, rows = ivf.export()
    assert (lab[rows[::997].astype(np.int64)] == np.repeat(np.arange(nlist), np.diff(offs.astype(np.int64)))[::997]).all()
    # ---- self-query: stored rows of the shard come back first with distance exactly 0 ----
    rng = np.random.default_rng(5)
    pos = np.sort(rng.choice(cnt, nq, replace=False)).astype(np.uint64)
    hq = ivf.get_vectors_by_ids(pos)
    q = torch.from_numpy(hq).to(dev)
    ctx = ivf.create_context()
    ctx.set_stream(stream)
    gk, gs, gc = _search_dev(ivf, ctx, q, k, stream, args=(nprobe, n))
    assert (gc == k).all() and np.all(np.diff(gs, axis=1) >= 0)
    assert (gs[:, 0] == 0).all()
    assert (gk[:, 0] == rows[pos.astype(np.int64)]).mean() > 0.999           # (exact duplicates aside)
    # ---- every returned key lives in a list this shard owns ----
    assert (owner[lab[gk.astype(np.int64).reshape(-1)]] == 0).all()
    scanned, probes = ivf.last_stats(ctx, nq)
    assert (probes == nprobe).all()
    # ---- oracle spot check: 2 queries against the probed lists of the exported shard (same centroids, same order) ----
    pk, _, _, pc = oracle.flat_search(cent, hq[:2], nprobe)                   # IVFCentroidIndex::search: top-nprobe centroids
    need = np.unique(pk[:, :nprobe].astype(np.int64))
    lsz = np.diff(offs.astype(np.int64))
    sub_offs = np.zeros(nlist + 1, np.int64)
    sub_offs[1:][need] = lsz[need]
    sub_offs = np.cumsum(sub_offs)
    idx = np.concatenate([np.arange(offs[l], offs[l + 1]) for l in need]).astype(np.uint64)
    sub_vecs = ivf.get_vectors_by_ids(idx)
    ok, os_, _, oc, _ = oracle.ivf_search(cent, sub_offs.astype(np.uint64), sub_vecs, hq[:2], k, nprobe, n, keys=rows[idx.astype(np.int64)])
    qn = (hq[:2].astype(np.float64) ** 2).sum(1)
    tie_tolerant_compare(gk[:2], gs[:2], gc[:2], ok, os_, oc, rtol=2e-6, atol=1e-6, select_band=4e-6 * (2 * qn.max() + 1),
                         what="100M fp16 shard oracle spot")
    # one rank's share: 64 queries against the reference's own IVFSearcher over the exported shard — every centroid, the lists this
    # rank owns (the others are empty here, as they are on the rank), fp16 rows of 1536 bytes lent as the dumped body.  The probe
    # walk is never cut short (brute_force_threshold = rows - 1), so local and global list sizes give the same probe set.
    R = _refcore()
    if R is not None and nshards == 8:
        vecs = np.empty((cnt, dim), np.float16)
        for o in range(0, cnt, 1 << 20):
            e = min(cnt, o + (1 << 20))
            vecs[o:e] = ivf.get_vectors_by_ids(np.arange(o, e, dtype=np.uint64))
        ratio = float(np.float32(nprobe) / np.float32(nlist))
        ref = R.Runner.ivf_over_rows("IVFSearcher", cent, offs, vecs, rows.astype(np.uint64), "SquaredEuclidean",
                                     params={"proxima.ivf.searcher.scan_ratio": ratio, "proxima.ivf.searcher.brute_force_threshold": cnt - 1})
        m = 64
        rk, rs, rc_, _ = ref.search_mt(hq[:m], k, 16)
        ref.close()
        del vecs
        qn = (hq[:m].astype(np.float64) ** 2).sum(1)
        tie_tolerant_compare(gk[:m], gs[:m], gc[:m], rk, rs, rc_, rtol=2e-6, atol=1e-6, select_band=4e-6 * (2 * qn.max() + 1),
                             what="config3 share: 64 queries vs the reference's IVFSearcher over the exported shard")
