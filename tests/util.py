"""Shared helpers of the parity tests (compare a GPU result list with the oracle's)."""
import numpy as np


def tie_tolerant_compare(g_keys, g_scores, g_counts, o_keys, o_scores, o_counts, atol=0.0, rtol=0.0,
                         scale=None, what="", select_band=None):
    """Scores must agree rank by rank within atol + rtol*scale; key SETS must agree except where the
    disagreeing keys sit within the tolerance of the k-th (boundary) score — exact ties / near ties,
    which the reference itself resolves arbitrarily (heap.h:103-114,173-175).  With atol=rtol=0 this is
    bit-exact equality of scores and of ids outside exact boundary ties.
    `select_band` (per query, absolute): L2 lists are SELECTED with norm-expansion scores (error ~ ulp of the
    norms) and then re-scored directly (error ~ ulp of the distance); ids may differ from the oracle's only
    for candidates within that wider selection band of the k-th score, while the reported scores must
    agree within the tight atol/rtol."""
    nq = len(o_counts)
    assert len(g_counts) == nq
    for q in range(nq):
        c = int(o_counts[q])
        assert int(g_counts[q]) == c, "%s query %d: count %d vs oracle %d" % (what, q, g_counts[q], c)
        if c == 0:
            continue
        gs, os_ = g_scores[q, :c].astype(np.float64), o_scores[q, :c].astype(np.float64)
        sc = (np.abs(os_) if scale is None else np.full(c, float(scale if np.isscalar(scale) else scale[q])))
        tol = atol + rtol * sc
        band = 0.0 if select_band is None else float(select_band if np.isscalar(select_band) else select_band[q])
        if band == 0.0:
            assert np.all(np.abs(gs - os_) <= tol), "%s query %d: scores differ\n gpu %r\n ora %r" % (what, q, gs, os_)
        else:   # ranks may shift inside the band: compare the score multisets with band slack
            assert np.all(np.abs(gs - os_) <= tol + 2 * band), "%s query %d: scores differ\n gpu %r\n ora %r" % (what, q, gs, os_)
        assert np.all(np.diff(gs) >= 0), "%s query %d: gpu scores not ascending" % (what, q)
        gk, ok = set(g_keys[q, :c].tolist()), set(o_keys[q, :c].tolist())
        assert len(gk) == c, "%s query %d: duplicate keys in the gpu list" % (what, q)
        if band:
            om = {int(k_): s_ for k_, s_ in zip(o_keys[q, :c], os_)}
            for k_, s_ in zip(g_keys[q, :c], gs):
                if int(k_) in om:   # same document => the refined score must be tight
                    t_ = atol + rtol * (abs(om[int(k_)]) if scale is None else float(scale if np.isscalar(scale) else scale[q]))
                    assert abs(s_ - om[int(k_)]) <= t_, "%s query %d key %d: %r vs %r" % (what, q, k_, s_, om[int(k_)])
        if gk != ok:
            bound = os_[c - 1]
            btol = float(np.max(tol)) + band
            for k in gk - ok:
                s = gs[list(g_keys[q, :c]).index(k)]
                assert s >= bound - 2 * btol, "%s query %d: key %d (score %r) not in oracle list and not a boundary tie (%r)" % (what, q, k, s, bound)
            for k in ok - gk:
                s = os_[list(o_keys[q, :c]).index(k)]
                assert s >= gs[c - 1] - 2 * btol, "%s query %d: oracle key %d (score %r) missing from gpu list" % (what, q, k, s)


def exact_l2(base, queries):
    b = base.astype(np.float64)
    q = queries.astype(np.float64)
    return ((q[:, None, :] - b[None, :, :]) ** 2).sum(-1)


def kmeans_lists(rng, base, nlist):
    """tiny host IVF structure for tests: random centroids = sampled rows, nearest assignment (fp64)."""
    n = base.shape[0]
    cent = base[rng.choice(n, nlist, replace=False)].astype(np.float32).copy()
    d = exact_l2(cent, base)            # [n][nlist]
    lab = d.argmin(1)
    order = np.argsort(lab, kind="stable")
    sizes = np.bincount(lab, minlength=nlist)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    return cent, offs, order


def lpt_owner(sizes, nshards, tile=128):
    """the list -> shard map of zvec_hip_ivf_shard_map restated: lists by (tiles desc, id asc), each to the shard
    with the fewest 128-row tiles so far (lowest shard on ties).  SURVEY §8(e): whole lists, balanced by bytes."""
    sizes = np.asarray(sizes, np.int64)
    owner = np.zeros(sizes.size, np.uint32)
    if nshards <= 1:
        return owner
    load = [0] * nshards
    for l in np.argsort(-sizes, kind="stable"):
        g = min(range(nshards), key=lambda j: (load[j], j))
        owner[l] = g
        load[g] += (int(sizes[l]) + tile - 1) // tile
    return owner
