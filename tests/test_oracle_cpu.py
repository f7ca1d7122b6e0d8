"""Pins the CPU oracle (oracle/zvec_oracle.c) — no GPU needed.

 1. against the known-answer values of the reference's unit tests (tests/golden/kernel_known_answers.json)
 2. bit-for-bit against outputs of the reference's own kernels/heap on seeded inputs
    (tests/golden/ref_kernel_vectors.npz, made by tests/golden/make_ref_vectors.py from oracle/_ref)
 3. live against oracle/_ref when that library is present (it is wherever /root/reference was compiled)
 4. the scan loops against the reference's structured-data expectations (scan_known_answers.json)
"""
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as O


def _ulp_close(a, b, ulps=4):
    a32, b32 = np.float32(a), np.float32(b)
    if a32 == b32:
        return True
    return abs(float(a32) - float(b32)) <= ulps * float(np.spacing(np.float32(max(abs(a32), abs(b32)))))


def test_kernel_known_answers(oracle, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "kernel_known_answers.json")))["cases"]
    assert len(cases) >= 40
    for c in cases:
        a, b = np.array(c["a"], np.float32), np.array(c["b"], np.float32)
        kind = c["kind"]
        if kind == "euclidean":
            got = math.sqrt(oracle.dist(O.METRIC_L2, a, b))      # Distance::Euclidean = sqrt(squared)
        elif kind == "sqeuclidean":
            got = oracle.dist(O.METRIC_L2, a, b)
        elif kind == "inner_product":
            got = oracle.ip(a, b)
        elif kind == "minus_inner_product":
            got = oracle.dist(O.METRIC_IP, a, b)
        elif kind == "cosine":
            got = oracle.dist(O.METRIC_COSINE, oracle.cosine_transform(a)[0], oracle.cosine_transform(b)[0])
        else:
            raise AssertionError(kind)
        if c["tol"] == "4ulp":
            assert _ulp_close(got, c["expect"]), (c["src"], got, c["expect"])
        else:
            assert abs(got - c["expect"]) <= float(c["tol"]), (c["src"], got, c["expect"])


def test_bit_exact_vs_reference_vectors(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_kernel_vectors.npz"))
    off = 0
    for i, d in enumerate(z["dims"]):
        a, b = z["a"][off:off + d], z["b"][off:off + d]
        assert np.float32(oracle.dist(O.METRIC_L2, a, b)) == z["l2"][i], d
        assert np.float32(oracle.ip(a, b)) == z["ip"][i], d
        assert np.float32(oracle.dist(O.METRIC_IP, a, b)) == z["minus_ip"][i], d
        assert np.float32(oracle.norm2(a)) == z["norm2"][i], d
        v, _ = oracle.normalize_l2(a)
        assert np.array_equal(v, z["normalized_a"][off:off + d]), d
        if d >= 2:
            a1 = oracle.cosine_transform(a)[0]
            b1 = oracle.cosine_transform(b)[0]
            assert np.float32(oracle.dist(O.METRIC_COSINE, a1, b1)) == z["cosine"][i], d
        off += d
    off = 0
    for i, d in enumerate(z["dims"]):                       # fp16 rows
        a, b = z["h_a"][off:off + d].view(np.float16), z["h_b"][off:off + d].view(np.float16)
        assert np.float32(oracle.dist16(O.METRIC_L2, a, b)) == z["h_l2"][i], d
        assert np.float32(oracle.dist16(O.METRIC_IP, a, b)) == z["h_minus_ip"][i], d
        off += d
    with np.errstate(over="ignore"):                        # HalfFloatConverter == numpy round-to-nearest-even
        assert np.array_equal(z["tofp16_in"].astype(np.float16).view(np.uint16), z["tofp16_out"])
    so = io = 0
    for n, k, kept in z["heap_meta"]:
        s = z["heap_scores"][so:so + n]
        idx, sc = oracle.heap_replay(s, int(k))
        assert np.array_equal(idx, z["heap_index"][io:io + kept])
        assert np.array_equal(sc, z["heap_kept_scores"][io:io + kept])
        so += n
        io += kept


def test_block_kernels_bit_exact_vs_reference_vectors(oracle, golden_dir):
    """M x N block kernels: the column-major 32-row blocks x interleaved queries the CPU's dense
    path computes (euclidean_distance_matrix_fp32.cc:323-929, inner_product_matrix_fp32.cc:588-1179), outputs of the
    compiled reference for seeded blocks; also: a block equals the 1x1 kernel's VALUE only up to summation order,
    which is why both are restated."""
    z = np.load(os.path.join(golden_dir, "ref_kernel_vectors.npz"))
    om = oq = oo = 0
    for M, N, d in z["block_shapes"]:
        m = z["block_m"][om:om + d * M].reshape(d, M)
        q = z["block_q"][oq:oq + d * N].reshape(d, N)
        l2 = oracle.block_dist(O.METRIC_L2, m, q)
        ip = oracle.block_dist(O.METRIC_IP, m, q)
        assert np.array_equal(l2.ravel().view(np.uint32), z["block_l2"][oo:oo + M * N].view(np.uint32)), (M, N, d)
        assert np.array_equal(ip.ravel().view(np.uint32), z["block_minus_ip"][oo:oo + M * N].view(np.uint32)), (M, N, d)
        # layout check against plain arithmetic: out[j][i] = sum_k (m[k][i] - q[k][j])^2
        want = ((m.astype(np.float64)[:, None, :] - q.astype(np.float64)[:, :, None]) ** 2).sum(0)
        assert np.allclose(l2, want, rtol=1e-5, atol=1e-5)
        om += d * M
        oq += d * N
        oo += M * N


def test_fp16_block_kernels_bit_exact_vs_reference_vectors(oracle, golden_dir):
    """fp16 M x N blocks (euclidean_distance_matrix_fp16.cc, inner_product_matrix_fp16.cc): outputs of the compiled
    reference for seeded half blocks"""
    z = np.load(os.path.join(golden_dir, "ref_kernel_vectors.npz"))
    om = oq = oo = 0
    for M, N, d in z["hblock_shapes"]:
        m = z["hblock_m"][om:om + d * M].view(np.float16).reshape(d, M)
        q = z["hblock_q"][oq:oq + d * N].view(np.float16).reshape(d, N)
        assert np.array_equal(oracle.block_dist(O.METRIC_L2, m, q).ravel().view(np.uint32), z["hblock_l2"][oo:oo + M * N].view(np.uint32)), (M, N, d)
        assert np.array_equal(oracle.block_dist(O.METRIC_IP, m, q).ravel().view(np.uint32), z["hblock_minus_ip"][oo:oo + M * N].view(np.uint32)), (M, N, d)
        om += d * M
        oq += d * N
        oo += M * N


def test_small_m_block_kernels_bit_exact_vs_reference_vectors(oracle, golden_dir):
    """M = 2 and M = 4 blocks (ACCUM_FP32_2X1 / 2X2 / 4X1 / 4X2 / 4X4_AVX, distance_matrix_accum_fp32.i:496-680, and their
    fp16 twins): several k steps share one register, so a pair's sum is split into 4 resp. 2 partial chains — the
    column-major scan reaches them through its left-over rows (single_enqueue_nofilter<4>, <2>)."""
    z = np.load(os.path.join(golden_dir, "ref_kernel_vectors.npz"))
    for tag, dt in (("sblock", np.float32), ("hsblock", np.float16)):
        om = oq = oo = 0
        for M, N, d in z[tag + "_shapes"]:
            m = z[tag + "_m"][om:om + d * M].view(dt).reshape(d, M)
            q = z[tag + "_q"][oq:oq + d * N].view(dt).reshape(d, N)
            assert np.array_equal(oracle.block_dist(O.METRIC_L2, m, q).ravel().view(np.uint32), z[tag + "_l2"][oo:oo + M * N].view(np.uint32)), (tag, M, N, d)
            assert np.array_equal(oracle.block_dist(O.METRIC_IP, m, q).ravel().view(np.uint32), z[tag + "_minus_ip"][oo:oo + M * N].view(np.uint32)), (tag, M, N, d)
            om += d * M
            oq += d * N
            oo += M * N


def test_live_vs_compiled_reference(oracle):
    if oracle.ref is None:
        pytest.skip("oracle/_ref/libzvec_ref.so not present (or CPU lacks AVX-512)")
    rng = np.random.default_rng(7)
    for d in list(range(1, 70)) + [127, 128, 700, 768]:
        for _ in range(5):
            a = rng.standard_normal(d).astype(np.float32)
            b = rng.standard_normal(d).astype(np.float32)
            for m in (O.METRIC_L2, O.METRIC_IP):
                assert oracle.dist(m, a, b) == oracle.dist(m, a, b, use_ref=True), (d, m)
            assert oracle.norm2(a) == oracle.norm2(a, use_ref=True)
    for d in list(range(1, 70)) + [127, 128, 768, 769]:
        for _ in range(3):
            a = (rng.standard_normal(d) * 2).astype(np.float16)
            b = (rng.standard_normal(d) * 2).astype(np.float16)
            for m in (O.METRIC_L2, O.METRIC_IP):
                assert oracle.dist16(m, a, b) == oracle.dist16(m, a, b, use_ref=True), (d, m)
    for M in (2, 4, 8, 16, 32):                             # block kernels, every specialised width
        for N in (1, 2, 4, 8, 16, 32):
            if N > M:
                continue
            for d in (1, 7, 64, 333):
                mb = rng.standard_normal((d, M)).astype(np.float32)
                qb = rng.standard_normal((d, N)).astype(np.float32)
                for m in (O.METRIC_L2, O.METRIC_IP):
                    assert np.array_equal(oracle.block_dist(m, mb, qb).view(np.uint32),
                                          oracle.block_dist(m, mb, qb, use_ref=True).view(np.uint32)), (M, N, d, m)
                    hm, hq = mb.astype(np.float16), qb.astype(np.float16)
                    assert np.array_equal(oracle.block_dist(m, hm, hq).view(np.uint32),
                                          oracle.block_dist(m, hm, hq, use_ref=True).view(np.uint32)), ("fp16", M, N, d, m)
    for _ in range(100):
        n, k = int(rng.integers(1, 400)), int(rng.integers(1, 64))
        s = rng.integers(0, 9, n).astype(np.float32)
        thr = float(rng.integers(2, 9))
        for t in (O.FLT_MAX, thr):
            i1, s1 = oracle.heap_replay(s, k, t)
            i2, s2 = oracle.heap_replay(s, k, t, use_ref=True)
            assert np.array_equal(i1, i2) and np.array_equal(s1, s2)


def _ramp(n, dim):
    return np.repeat(np.arange(n, dtype=np.float32)[:, None], dim, 1)


def test_flat_linear_known_answers(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "scan_known_answers.json")))["flat_linear"]
    n, dim = g["n"], g["dim"]
    base = _ramp(n, dim)
    q = np.arange(0, n, 37, dtype=np.float32)
    keys, scores, _, cnt = oracle.flat_search(base, np.repeat(q[:, None], dim, 1), 3)
    assert np.array_equal(keys[:, 0], q.astype(np.uint64))
    keys, _, _, _ = oracle.flat_search(base, np.repeat((q + np.float32(0.1))[:, None], dim, 1), 3)
    for j, i in enumerate(q.astype(int)):
        last = n - 1
        assert keys[j, 0] == i
        assert keys[j, 1] == (i - 1 if i == last else i + 1)
        assert keys[j, 2] == (2 if i == 0 else (i - 2 if i == last else i - 1))
    keys, _, _, cnt = oracle.flat_search(base, np.full((1, dim), 10.1, np.float32), 100)
    assert cnt[0] == 100
    for rank, key in g["query_10p1_top100_ranks"].items():
        assert keys[0, int(rank)] == key


def test_flat_filter_known_answers(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "scan_known_answers.json")))["flat_filter"]
    base = _ramp(g["n"], g["dim"])
    q = np.full((1, g["dim"]), g["query"], np.float32)
    keys, _, _, cnt = oracle.flat_search(base, q, g["topk"])
    assert cnt[0] == 10 and keys[0, :3].tolist() == g["top3_nofilter"]
    mask = np.zeros(g["n"], bool)
    mask[g["excluded_keys"]] = True
    keys, _, _, cnt = oracle.flat_search(base, q, g["topk"], exclude_bits=O.pack_bits(mask))
    assert cnt[0] == 10 and keys[0, :3].tolist() == g["top3_filtered"]
    # filter everything => no result (flat_searcher_test.cpp:93-209)
    keys, _, _, cnt = oracle.flat_search(base, q, g["topk"], exclude_bits=O.pack_bits(np.ones(g["n"], bool)))
    assert cnt[0] == 0


def test_ivf_simple_known_answers(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "scan_known_answers.json")))["ivf_simple"]
    n, dim = g["n"], g["dim"]
    base = _ramp(n, dim)
    cent = base.mean(0, keepdims=True).astype(np.float32)
    offs = np.array([0, n], np.uint64)
    q = np.full((1, dim), g["single_query"], np.float32)
    for bf in (True, False):
        keys, scores, _, cnt, scanned = oracle.ivf_search(cent, offs, base, q, g["single_topk"], nprobe=1,
                                                          max_scan_count=n, brute_force=bf)
        assert cnt[0] == 33 and scanned[0] == 33
        assert keys[0].tolist() == [32 - i for i in range(33)]
        assert scores[0].tolist() == [float(i * i * dim) for i in range(33)]
        qb = _ramp(33, dim)
        keys, scores, _, cnt, _ = oracle.ivf_search(cent, offs, base, qb, 1, 1, n, brute_force=bf)
        assert keys[:, 0].tolist() == list(range(33)) and not scores.any()


def test_ivf_probe_rule_and_threshold(oracle):
    # driver loop of ivf_searcher.cc:217-237: probe in coarse order while scanned < max_scan_count
    rng = np.random.default_rng(3)
    from tests.util import kmeans_lists, exact_l2
    base = rng.integers(0, 50, (600, 8)).astype(np.float32)
    cent, offs, order = kmeans_lists(rng, base, 12)
    vecs = base[order]
    q = rng.integers(0, 50, (5, 8)).astype(np.float32)
    sizes = np.diff(offs.astype(np.int64))
    keys, scores, idx, cnt, scanned, probes = oracle.ivf_search(cent, offs, vecs, q, 7, nprobe=4, max_scan_count=120,
                                                                keys=order.astype(np.uint64), want_probes=True)
    for qi in range(5):
        tot = 0
        for r in range(4):
            if probes[qi, r] == 0xffffffff:
                assert tot >= 120
                break
            tot += sizes[probes[qi, r]]
        assert scanned[qi] == tot
    # RNN radius (index_document.h:250-261, ivf_searcher_context.h:184-208)
    k2, s2, _, c2, _ = oracle.ivf_search(cent, offs, vecs, q, 50, 12, 10 ** 6, keys=order.astype(np.uint64),
                                         threshold=300.0)
    assert all(s2[i, :c2[i]].max(initial=0) <= 300.0 for i in range(5))
    # brute force == exact top-k
    kb, sb, _, cb, _ = oracle.ivf_search(cent, offs, vecs, q, 5, 1, 1, keys=order.astype(np.uint64), brute_force=True)
    d = exact_l2(base, q)
    assert np.allclose(np.sort(d, 1)[:, :5], sb)


def test_merge_matches_concat_sort_truncate(oracle):
    rng = np.random.default_rng(5)
    nparts, nq, k = 4, 6, 5
    scores = np.sort(rng.integers(0, 30, (nparts, nq, k)).astype(np.float32), -1)
    keys = rng.integers(0, 10 ** 6, (nparts, nq, k)).astype(np.uint64)
    counts = rng.integers(0, k + 1, (nparts, nq)).astype(np.uint32)
    ok, os_, oc = oracle.merge_topk(keys, scores, counts, k)
    for q in range(nq):
        allv = [(scores[p, q, j], p * k + j, keys[p, q, j]) for p in range(nparts) for j in range(counts[p, q])]
        allv.sort(key=lambda t: (t[0], t[1]))
        exp = allv[:k]
        assert oc[q] == len(exp)
        assert [e[2] for e in exp] == ok[q, :oc[q]].tolist()


def test_multithreaded_matches_single(oracle):
    rng = np.random.default_rng(9)
    base = rng.standard_normal((500, 24)).astype(np.float32)
    q = rng.standard_normal((17, 24)).astype(np.float32)
    a = oracle.flat_search(base, q, 10, O.METRIC_IP)
    b = oracle.flat_search(base, q, 10, O.METRIC_IP, threads=4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_label_and_pack_restatement_on_exact_data(oracle):
    """IVFBuilder::label restated (oracle.ivf_label_and_pack): on integer data every distance is exact, so the label is
    the fp64 argmin with the FIRST centroid winning ties (heap.h:103-114), and the pack order is the stable grouping."""
    rng = np.random.default_rng(8)
    n, dim, nlist = 3000, 12, 20
    base = rng.integers(-5, 6, (n, dim)).astype(np.float32)
    cent = rng.integers(-5, 6, (nlist, dim)).astype(np.float32)
    cent[11] = cent[2]
    lab, offs, order = oracle.ivf_label_and_pack(cent, base)
    d = ((base[:, None, :].astype(np.float64) - cent[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(lab, d.argmin(1).astype(np.uint32))      # numpy argmin: first minimum, like the heap
    assert (lab != 11).all()
    assert np.array_equal(np.diff(offs.astype(np.int64)), np.bincount(lab, minlength=nlist))
    for l in range(nlist):
        seg = order[int(offs[l]):int(offs[l + 1])]
        assert (lab[seg.astype(np.int64)] == l).all() and (np.diff(seg.astype(np.int64)) > 0).all()


def _column_replay_with_reference_kernels(oracle, base, q, k, metric):
    """the column-major batch loop (flat_searcher_context.h:682-752) replayed in Python on top of the REFERENCE's own
    compiled M x N kernels and heap (oracle/_ref): transposed 32-row blocks x query groups of 32/16/.../1."""
    n, dim = base.shape
    nq = q.shape[0]
    groups, qi, left, K = [], 0, nq, 32
    while K >= 1:
        while left >= K:
            groups.append((qi, K))
            qi += K
            left -= K
        K >>= 1
    scores = np.zeros((nq, n), np.float32)
    full = n // 32 * 32
    for b0 in range(0, full, 32):
        blk = np.ascontiguousarray(base[b0:b0 + 32].T)                        # [dim][32]
        for g0, K in groups:
            out = oracle.block_dist(metric, blk, np.ascontiguousarray(q[g0:g0 + K].T), use_ref=True)   # [K][32]
            scores[g0:g0 + K, b0:b0 + 32] = out
    for r in range(full, n):
        for g0, K in groups:
            if K > 1:                                                         # the GROUP is the matrix, the row the query
                out = oracle.block_dist(metric, np.ascontiguousarray(q[g0:g0 + K].T), np.ascontiguousarray(base[r:r + 1].T), use_ref=True)
                scores[g0:g0 + K, r] = out[0]
            else:
                scores[g0, r] = oracle.dist(metric, base[r], q[g0], use_ref=True)
    res = []
    for i in range(nq):
        idx, sc = oracle.heap_replay(scores[i], k, use_ref=True)
        order = np.argsort(sc, kind="stable")
        res.append((idx[order], sc[order]))
    return res


@pytest.mark.parametrize("metric", [O.METRIC_L2, O.METRIC_IP])
def test_column_major_loop_pinned_to_reference_kernels(oracle, metric):
    """(a)4: the restated column-major dense path (zo_flat_search_column_t) against a replay of the same loop on the
    reference's OWN block kernels, 1x1 kernel and heap compiled in place — bit for bit, real-valued data, ragged sizes."""
    if oracle.ref is None:
        pytest.skip("oracle/_ref/libzvec_ref.so not present (or CPU lacks AVX-512)")
    rng = np.random.default_rng(19)
    for n, dim, nq, k in ((70, 20, 45, 7), (32, 9, 1, 5), (31, 33, 3, 4), (200, 64, 63, 10), (97, 128, 32, 10)):
        base = rng.standard_normal((n, dim)).astype(np.float32)
        q = rng.standard_normal((nq, dim)).astype(np.float32)
        ok, os_, oi, oc = oracle.flat_search_column(base, q, k, metric)
        want = _column_replay_with_reference_kernels(oracle, base, q, k, metric)
        for i in range(nq):
            widx, wsc = want[i]
            assert oc[i] == len(widx)
            assert np.array_equal(os_[i, :oc[i]].view(np.uint32), wsc.view(np.uint32)), (n, dim, nq, i)
            # ids: equal wherever scores are distinct (heap.sort is unstable among equal scores, heap.h:173-175)
            distinct = np.concatenate([[True], np.diff(wsc) != 0]) & np.concatenate([np.diff(wsc) != 0, [True]])
            assert np.array_equal(oi[i, :oc[i]][distinct], widx[distinct])


def test_column_major_equals_row_major_on_exact_data(oracle):
    """flat_searcher_test.cpp:93-209 (row == column == filtered) on integer data, where every kernel is exact: the two
    scan orders give identical lists; filter-all gives 0 results; a block whose 32 rows are all filtered is skipped."""
    rng = np.random.default_rng(23)
    for n, dim, nq, k in ((1000, 16, 70, 10), (65, 8, 2, 3), (20, 5, 33, 4)):
        base = rng.integers(-7, 8, (n, dim)).astype(np.float32)
        q = rng.integers(-7, 8, (nq, dim)).astype(np.float32)
        keys = (np.arange(n, dtype=np.uint64) * 3 + 11)
        for metric in (O.METRIC_L2, O.METRIC_IP):
            a = oracle.flat_search(base, q, k, metric, keys=keys)
            b = oracle.flat_search_column(base, q, k, metric, keys=keys)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
            mask = rng.random(n) < 0.6
            mask[:min(n, 64)] = True                                         # the first two blocks entirely filtered out
            bits = O.pack_bits(mask)
            a = oracle.flat_search(base, q, k, metric, keys=keys, exclude_bits=bits)
            b = oracle.flat_search_column(base, q, k, metric, keys=keys, exclude_bits=bits)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        none = oracle.flat_search_column(base, q, k, exclude_bits=O.pack_bits(np.ones(n, bool)))
        assert (none[3] == 0).all()


def _test_group_corpus():
    """flat_streamer_test.cc TestGroup (:929-1027): 5000 rows of dim 16, row i = i/10 in every component"""
    n, dim = 5000, 16
    base = np.repeat((np.arange(n, dtype=np.float32) / np.float32(10.0))[:, None], dim, axis=1).astype(np.float32)
    q = np.full((1, dim), np.float32(n // 2) * np.float32(1.0) / np.float32(10) + np.float32(0.1), np.float32)
    return base, q


def test_group_by_known_answers(oracle):
    """the oracle's group-by restatement against what flat_streamer_test.cc TestGroup asserts: the p_keys leg lists keys
    {4,3,2,1,5..10} with group = key % 10 and expects exactly group_num = 5 groups whose first documents are the keys
    10, 9, 8, 7, 6 (:1019-1036); the full-scan leg (group = key / 10 % 10, 5 groups x 20) expects non-empty groups."""
    base, q = _test_group_corpus()
    n = base.shape[0]
    res = oracle.flat_group_search(base, q, np.arange(n) % 10, 5, 20, O.METRIC_L2, candidates=[[4, 3, 2, 1, 5, 6, 7, 8, 9, 10]])[0]
    assert len(res) == 5
    for i, (g, docs) in enumerate(res):
        assert len(docs) > 0 and docs[0][0] == 10 - i and g == (10 - i) % 10
    res = oracle.flat_group_search(base, q, (np.arange(n) // 10) % 10, 5, 20, O.METRIC_L2)[0]
    assert len(res) == 5
    # the query sits at key 2501: its decade (group 0: keys 2500-2509) holds the best document, then the neighbours
    assert res[0][0] == 0 and res[0][1][0][0] == 2501
    assert sorted(g for g, _ in res) == [0, 1, 2, 8, 9]
    for g, docs in res:
        assert len(docs) == 20 and all((k // 10) % 10 == g for k, _, _ in docs)
        assert all(docs[j][1] <= docs[j + 1][1] for j in range(19))


def test_cosine_batch_distance_bit_exact_vs_reference(oracle, golden_dir):
    """IndexMetric::batch_distance of the Cosine metric (one-to-many inner product of ailego/math_batch, a different lane
    order than the 1x1 kernel): the restatement equals the golden vectors the reference's own ComputeBatch produced, fp32
    and fp16, every tail shape — and the live reference when oracle/_ref is present.  (SquaredEuclidean / InnerProduct
    batch_distance is a loop of the 1x1 kernels already pinned above.)"""
    z = np.load(os.path.join(golden_dir, "ref_kernel_vectors.npz"))
    for tag, dt in (("cosb", np.float32), ("hcosb", np.float16)):
        ro = qo = oo = 0
        for dim in z[tag + "_dims"]:
            rows = z[tag + "_rows"][ro:ro + 29 * dim].view(dt).reshape(29, dim)
            q = z[tag + "_q"][qo:qo + dim].view(dt)
            want = z[tag + "_out"][oo:oo + 29]
            got = oracle.cosine_batch(rows, q)
            assert np.array_equal(got, want), (tag, int(dim), np.nonzero(got != want)[0][:5])
            if oracle.ref is not None:
                assert np.array_equal(oracle.cosine_batch(rows, q, use_ref=True), want), (tag, int(dim))
            ro += 29 * dim
            qo += dim
            oo += 29
