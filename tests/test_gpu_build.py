"""GPU IVF build against the oracle — run with -m gpu.

* labelling: every row of a GPU-built index sits in the list of its nearest centroid as the oracle's restatement of
  IVFBuilder::label finds it (ivf_builder.h:253-274: top-1 of the centroid index), exactly on integer data, up to the
  stated selection band on real data;
* packing: the list order equals the oracle's stable grouping (rows of a list in ascending row number);
* the streamed build (train / label / begin_lists / add / end_lists) equals the one-call build, for 1 and 3 shards,
  and a shard holds exactly the lists the byte-balanced map gives it.
k-means itself (OptKmeansCluster) is NOT compared: the reference's trainer is a different algorithm with its own
random initialisation (parity unpinned for the centroids; both sides then search the same exported centroids)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import lpt_owner

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zv():
    import zvec_amd
    return zvec_amd


def _labels_of_export(offs, rows, n):
    lab = np.empty(n, np.uint32)
    sizes = np.diff(offs.astype(np.int64))
    lab[rows.astype(np.int64)] = np.repeat(np.arange(sizes.size, dtype=np.uint32), sizes)
    return lab


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_labelling_and_packing_integer_data_exact(zv, oracle, dtype):
    rng = np.random.default_rng(11)
    n, dim, nlist = 20000, 48, 64
    npdt = np.float16 if dtype == "fp16" else np.float32
    base = rng.integers(-8, 9, (n, dim)).astype(npdt)
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean", dtype=dtype)
    assert se.build(base, nlist, kmeans_iters=3, sample_per_list=64, seed=3) == 0
    cent, offs, rows = se.export()
    got = _labels_of_export(offs, rows, n)
    want, woffs, worder = oracle.ivf_label_and_pack(cent, base)
    # centroids are means (not integers): L2 scores are selected on |q|^2+|b|^2-2qb, so a row may legitimately sit in
    # another list only if that centroid is within the selection band of the nearest one
    bad = np.nonzero(got != want)[0]
    for i in bad:
        d_got = oracle.dist16(O.METRIC_L2, cent[got[i]], base[i]) if dtype == "fp16" else oracle.dist(O.METRIC_L2, cent[got[i]], base[i])
        d_want = oracle.dist16(O.METRIC_L2, cent[want[i]], base[i]) if dtype == "fp16" else oracle.dist(O.METRIC_L2, cent[want[i]], base[i])
        norms = float((base[i].astype(np.float64) ** 2).sum() + (cent[want[i]].astype(np.float64) ** 2).sum())
        assert abs(d_got - d_want) <= 4e-6 * norms, (i, d_got, d_want)
    assert bad.size <= n // 1000
    # packing = stable grouping by label: inside every list the rows ascend
    assert np.array_equal(rows, np.argsort(got, kind="stable").astype(np.uint64))
    if bad.size == 0:
        assert np.array_equal(offs, woffs) and np.array_equal(rows, worder)


def test_labelling_exact_when_centroids_are_lattice_points(zv, oracle):
    """integer rows AND integer centroids (set_centroids): every score is exact, so labels / offsets / order must equal
    the oracle's bit for bit, ties included (first centroid in id order wins, heap.h:103-114)."""
    import torch
    rng = np.random.default_rng(12)
    n, dim, nlist = 30000, 32, 40
    base = rng.integers(-6, 7, (n, dim)).astype(np.float32)
    cent = rng.integers(-6, 7, (nlist, dim)).astype(np.float32)
    cent[7] = cent[3]                                    # duplicate centroid: list 7 must stay empty
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean")
    assert se.set_centroids(cent) == 0
    d_base = torch.from_numpy(base).cuda()
    d_lab = torch.empty(n, dtype=torch.int32, device="cuda")
    assert se.label_dev(d_base.data_ptr(), n, d_lab.data_ptr()) == 0
    got = d_lab.cpu().numpy().astype(np.uint32)
    want, woffs, worder = oracle.ivf_label_and_pack(cent, base)
    assert np.array_equal(got, want)
    assert (got != 7).all()
    assert se.begin_lists(np.bincount(got, minlength=nlist)) == 0
    for o in range(0, n, 7001):                          # ragged chunks
        m = min(7001, n - o)
        assert se.add_dev(d_base[o:o + m].data_ptr(), m, got[o:o + m], o) == 0
    assert se.end_lists() == 0
    c2, offs, rows = se.export()
    assert np.array_equal(c2, cent) and np.array_equal(offs, woffs) and np.array_equal(rows, worder)
    # and the index answers like the oracle on that very structure
    q = rng.integers(-6, 7, (50, dim)).astype(np.float32)
    se.scan_ratio, se.brute_force_threshold = 8 / 40., 10
    nprobe, max_scan = se.probe_params()
    ctx = se.create_context()
    ctx.set_topk(10)
    assert se.search_impl(q, 50, ctx) == 0
    ok, os_, _, oc, _ = oracle.ivf_search(cent, woffs, base[worder.astype(np.int64)], q, 10, nprobe, max_scan, keys=worder)
    from tests.util import tie_tolerant_compare, exact_l2
    cd = np.sort(exact_l2(cent, q), 1)
    sel = np.nonzero(cd[:, nprobe - 1] != cd[:, nprobe])[0]
    tie_tolerant_compare(ctx.keys[sel], ctx.scores[sel], ctx.counts[sel], ok[sel], os_[sel], oc[sel], what="streamed-built index")


@pytest.mark.parametrize("nshards", [1, 3])
def test_streamed_build_equals_one_call_build(zv, nshards):
    import torch
    rng = np.random.default_rng(13)
    n, dim, nlist, spl = 24000, 40, 48, 64
    base = (rng.standard_normal((n, 6)) @ rng.standard_normal((6, dim))).astype(np.float32)
    d_base = torch.from_numpy(base).cuda()
    S = min(n, spl * nlist)
    sample_ids = (np.arange(S, dtype=np.uint64) * np.uint64(n)) // np.uint64(S)       # the one-call build's strided sample
    d_sample = d_base[torch.from_numpy(sample_ids.astype(np.int64)).cuda()].contiguous()
    # labels once (any shard can do it: centroids are replicated)
    tr = zv.HipIVFSearcher(dim, "SquaredEuclidean")
    assert tr.train_dev(d_sample.data_ptr(), S, nlist, kmeans_iters=5, seed=5) == 0
    cent = tr.get_centroids()
    d_lab = torch.empty(n, dtype=torch.int32, device="cuda")
    for o in range(0, n, 5000):
        m = min(5000, n - o)
        assert tr.label_dev(d_base[o:o + m].data_ptr(), m, d_lab[o:o + m].data_ptr()) == 0
    labels = d_lab.cpu().numpy().astype(np.uint32)
    sizes = np.bincount(labels, minlength=nlist)
    owner = lpt_owner(sizes, nshards)
    for r in range(nshards):
        one = zv.HipIVFSearcher(dim, "SquaredEuclidean")
        assert one.set_shard(r, nshards) == 0
        assert one.build_dev(d_base.data_ptr(), n, nlist, kmeans_iters=5, sample_per_list=spl, seed=5) == 0
        st = zv.HipIVFSearcher(dim, "SquaredEuclidean")
        assert st.set_shard(r, nshards) == 0
        assert st.set_centroids(cent) == 0                      # (as a rank that received rank 0's centroids would)
        assert st.begin_lists(sizes) == 0
        assert st.end_lists() == zv.IndexError_.NoReady         # lists still short
        for o in range(0, n, 5000):
            m = min(5000, n - o)
            assert st.add_dev(d_base[o:o + m].data_ptr(), m, labels[o:o + m], o) == 0
        assert st.end_lists() == 0
        c1, o1, r1 = one.export()
        c2, o2, r2 = st.export()
        assert np.array_equal(c1, c2) and np.array_equal(o1, o2) and np.array_equal(r1, r2)
        assert np.array_equal(st.list_owners(), owner)
        assert np.array_equal(np.diff(o2.astype(np.int64)), np.where(owner == r, sizes, 0))
        pos = rng.integers(0, st.info()[0], 50)
        assert np.array_equal(one.get_vectors_by_ids(pos), st.get_vectors_by_ids(pos))
        assert np.array_equal(st.get_vectors_by_ids(pos), base[r2[pos].astype(np.int64)])


def test_streamed_build_errors(zv):
    import torch
    se = zv.HipIVFSearcher(8)
    d = torch.zeros((16, 8), device="cuda")
    lab = torch.zeros(16, dtype=torch.int32, device="cuda")
    assert se.label_dev(d.data_ptr(), 16, lab.data_ptr()) == zv.IndexError_.NoTrained
    assert se.begin_lists(np.array([16], np.uint32)) == zv.IndexError_.NoTrained
    assert se.add_dev(d.data_ptr(), 16, np.zeros(16, np.uint32), 0) == zv.IndexError_.NoReady
    assert se.end_lists() == zv.IndexError_.NoReady
    assert se.set_centroids(np.zeros((2, 8), np.float32)) == 0
    assert se.begin_lists(np.array([4, 0], np.uint32)) == 0
    assert se.add_dev(d.data_ptr(), 16, np.zeros(16, np.uint32), 0) == zv.IndexError_.InvalidArgument    # list 0 overflows
    assert se.add_dev(d.data_ptr(), 1, np.array([5], np.uint32), 0) == zv.IndexError_.InvalidArgument     # no such list


def _kmeans_quality_corpus(fx):
    c = fx["corpus"]
    rng = np.random.default_rng(c["seed"])
    means = rng.standard_normal((1024, c["dim"])).astype(np.float32) * 1.0
    base = (means[rng.integers(0, 1024, c["n"])] + rng.standard_normal((c["n"], c["dim"])).astype(np.float32)).astype(np.float32)
    q = (means[rng.integers(0, 1024, c["queries"])] + rng.standard_normal((c["queries"], c["dim"])).astype(np.float32)).astype(np.float32)
    return base, q


def test_kmeans_quality_vs_reference_trainer(zv):
    """The reference's trainer (IVFBuilder: StratifiedClusterTrainer + OptKmeansCluster, ivf_builder.cc:524-531) seeds from
    std::random_device, so its centroids are not reproducible bit for bit; its clustering QUALITY is.  Fixture
    tests/golden/kmeans_quality.json (made by tests/golden/make_kmeans_quality.py with the reference's builder compiled in place):
    five runs on a seeded 200k x 64 corpus of overlapping Gaussians — within-cluster sum of squares per row, list-size spread,
    recall@10 at nprobe 4 of 256 through the reference's own IVFSearcher.  The GPU build (zvec_hip_ivf_build: Lloyd on a strided
    sample + labelling of every row) of the same corpus must sit inside the reference's own seed-to-seed spread: SSE no worse than
    its worst run (+0.2 %), recall no lower than its lowest (-0.01), list-size spread no worse than its worst (+15 %)."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kmeans_quality.json")))
    c = fx["corpus"]
    base, q = _kmeans_quality_corpus(fx)
    n, dim, nlist, k, nprobe = c["n"], c["dim"], c["nlist"], c["k"], c["nprobe"]
    se = zv.HipIVFSearcher(dim, "SquaredEuclidean")
    # the reference's defaults: at most 20 Lloyd rounds (opt_kmeans_cluster.cc:115) over EVERY row (train_sample_count 0 =
    # no sampling, stratified_cluster_trainer.cc:150-185)
    assert se.build(base, nlist, kmeans_iters=20, sample_per_list=(n + nlist - 1) // nlist) == 0
    cent, offs, rows = se.export()
    labels = np.empty(n, np.int64)
    for l in range(nlist):
        labels[rows[int(offs[l]):int(offs[l + 1])].astype(np.int64)] = l
    diff = base.astype(np.float64) - cent.astype(np.float64)[labels]
    sse = float((diff ** 2).sum() / n)
    sizes = np.bincount(labels, minlength=nlist)
    spread = float(sizes.max() / sizes.mean())
    se.set_nprobe(nprobe, exact=True)
    ctx = se.create_context()
    ctx.set_topk(k)
    assert se.search_impl(q, len(q), ctx) == 0
    d = (q.astype(np.float64) ** 2).sum(1)[:, None] + (base.astype(np.float64) ** 2).sum(1)[None] - 2 * q.astype(np.float64) @ base.astype(np.float64).T
    gt = np.argsort(d, 1, kind="stable")[:, :k]
    rec = float(np.mean([len(set(ctx.keys[i, :ctx.counts[i]].tolist()) & set(gt[i].tolist())) / k for i in range(len(q))]))
    ref = fx["runs"]
    worst_sse = max(r["sse_per_row"] for r in ref)
    low_rec = min(r["recall_at_10"] for r in ref)
    worst_spread = max(r["size_max_over_mean"] for r in ref)
    print("GPU k-means: SSE/row %.3f (reference %.3f .. %.3f), spread %.2f (reference <= %.2f), recall@10 %.4f (reference %.4f .. %.4f), empty lists %d"
          % (sse, min(r["sse_per_row"] for r in ref), worst_sse, spread, worst_spread, rec, low_rec, max(r["recall_at_10"] for r in ref),
             int((sizes == 0).sum())))
    assert sse <= worst_sse * 1.002
    assert rec >= low_rec - 0.01
    assert spread <= worst_spread * 1.15
    assert (sizes == 0).sum() == 0


@pytest.mark.parametrize("metric", ["SquaredEuclidean", "InnerProduct"])
@pytest.mark.parametrize("n,dim,nlist", [(5000, 64, 300), (4097, 100, 513), (1024, 768, 256), (777 + 512, 40, 1000), (3000, 192, 192)])
def test_fp16_labelling_on_the_256_tile_is_exact_on_lattice_data(zv, metric, n, dim, nlist):
    """The fp16 labelling kernel on the 256 x 256 multi-phase tile (zvk_assign256.hip.h; taken for >= 512 rows and >= 192 centroids)
    against an exact integer arg-min: small-integer rows and centroids make every product and every fp32 sum exact, so the label must
    be THE nearest centroid, lowest id on ties (IVFBuilder::label, ivf_builder.h:253-274; heap.h:103-114) — ragged row counts (the
    last work item is moved back), odd numbers of 128-centroid tiles (the second tile of the last pair is missing), centroid counts
    that are no multiple of 16, dimensions that are no multiple of the 64-half k-step, duplicate centroids; and the same labels as
    the 128 x 128 one-barrier kernel it replaces (option "assign256" = 0)."""
    import ctypes as C
    import torch
    from zvec_amd import _lib
    rng = np.random.default_rng(n + nlist)
    base = rng.integers(-5, 6, (n, dim)).astype(np.float16)
    cent = rng.integers(-5, 6, (nlist, dim)).astype(np.float16)
    cent[nlist - 1] = cent[1]                            # duplicate centroid: the later id must never be chosen
    se = zv.HipIVFSearcher(dim, metric, dtype="fp16")
    assert se.set_centroids(cent) == 0
    d_base = torch.from_numpy(base).cuda()
    labs = []
    L = _lib.lib()
    for opt in (1, 0):
        assert L.zvec_hip_set_option(b"assign256", opt) == 0
        d_lab = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        assert se.label_dev(d_base.data_ptr(), n, d_lab.data_ptr()) == 0
        torch.cuda.synchronize()
        labs.append(d_lab.cpu().numpy().astype(np.int64))
    assert L.zvec_hip_set_option(b"assign256", 1) == 0
    b64, c64 = base.astype(np.int64), cent.astype(np.int64)
    ip = b64 @ c64.T
    sc = (b64 ** 2).sum(1)[:, None] + (c64 ** 2).sum(1)[None] - 2 * ip if metric == "SquaredEuclidean" else -ip
    want = sc.argmin(1)                                   # numpy's argmin returns the FIRST minimum
    assert np.array_equal(labs[0], want), (np.nonzero(labs[0] != want)[0][:8], labs[0][labs[0] != want][:8], want[labs[0] != want][:8])
    assert np.array_equal(labs[1], want)
    assert (labs[0] != nlist - 1).all()
