"""ctypes door onto the CPU oracle (oracle/liboracle.so) and, when present, the compiled reference
kernels (oracle/_ref/libzvec_ref.so).

TEST INFRASTRUCTURE ONLY — may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (zvec_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 2
FLT_MAX = float(np.finfo(np.float32).max)

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "zvec_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libzvec_ref.so")
    if os.path.isdir("/root/reference/src/ailego/math") and (force or not os.path.exists(ref_so)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    # the reference's whole core library + the plugin as a loadable library (oracle/Makefile `ref_core`; make decides what is stale)
    if os.path.isdir("/root/reference/src/core/algorithm/ivf"):
        subprocess.check_call(["make", "-j8", "-C", _HERE, "ref_core"], stdout=subprocess.DEVNULL)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


class Oracle:
    def __init__(self):
        build()
        self.lib = L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        for name in ("zo_sqeuclid_f32", "zo_ip_f32", "zo_minus_ip_f32", "zo_cosine_f32"):
            f = getattr(L, name)
            f.restype = C.c_float
            f.argtypes = [_f32p, _f32p, C.c_size_t]
        L.zo_cosine_batch_f32.restype = C.c_float
        L.zo_cosine_batch_f32.argtypes = [_f32p, _f32p, C.c_size_t]
        L.zo_cosine_batch_f16.restype = C.c_float
        L.zo_cosine_batch_f16.argtypes = [C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_size_t]
        L.zo_norm2_f32.restype = C.c_float
        L.zo_norm2_f32.argtypes = [_f32p, C.c_size_t]
        L.zo_normalize_l2_f32.restype = None
        L.zo_normalize_l2_f32.argtypes = [_f32p, C.c_size_t, _f32p]
        L.zo_cosine_transform_f32.restype = None
        L.zo_cosine_transform_f32.argtypes = [_f32p, C.c_size_t, _f32p]
        L.zo_heap_replay.restype = C.c_size_t
        L.zo_heap_replay.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_float, _u32p, _f32p]
        L.zo_set_distance_override.restype = None
        L.zo_set_distance_override.argtypes = [C.c_int, C.c_void_p]
        L.zo_flat_search_mt.restype = C.c_int
        L.zo_flat_search_mt.argtypes = [_f32p, _u64p, C.c_uint64, C.c_uint32, C.c_int, _f32p,
                                        C.c_uint32, C.c_uint32, C.c_float, _u64p, _u64p, _f32p,
                                        _u32p, _u32p, C.c_int]
        L.zo_ivf_search.restype = C.c_int
        L.zo_ivf_search.argtypes = [_f32p, C.c_uint32, _u64p, _f32p, _u64p, C.c_uint32, C.c_int,
                                    _f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32,
                                    C.c_uint32, C.c_int, _u64p, _u64p, _f32p, _u32p, _u32p, _u32p,
                                    _u32p]
        L.zo_ivf_search_mt.restype = C.c_int
        L.zo_ivf_search_mt.argtypes = [_f32p, C.c_uint32, _u64p, _f32p, _u64p, C.c_uint32, C.c_int,
                                       _f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32,
                                       C.c_uint32, C.c_int, _u64p, _u64p, _f32p, _u32p, _u32p,
                                       _u32p, C.c_int]
        _u16p = C.POINTER(C.c_uint16)
        for name in ("zo_sqeuclid_f16", "zo_ip_f16", "zo_minus_ip_f16"):
            f = getattr(L, name)
            f.restype = C.c_float
            f.argtypes = [_u16p, _u16p, C.c_size_t]
        L.zo_set_distance_override_f16.restype = None
        L.zo_set_distance_override_f16.argtypes = [C.c_int, C.c_void_p]
        L.zo_flat_search_mt_t.restype = C.c_int
        L.zo_flat_search_mt_t.argtypes = [C.c_int, C.c_void_p, _u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p,
                                          C.c_uint32, C.c_uint32, C.c_float, _u64p, _u64p, _f32p, _u32p, _u32p, C.c_int]
        L.zo_ivf_search_mt_t.restype = C.c_int
        L.zo_ivf_search_mt_t.argtypes = [C.c_int, C.c_void_p, C.c_uint32, _u64p, C.c_void_p, _u64p, C.c_uint32, C.c_int,
                                         C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_int,
                                         _u64p, _u64p, _f32p, _u32p, _u32p, _u32p, C.c_int]
        L.zo_merge_topk.restype = C.c_int
        L.zo_merge_topk.argtypes = [_u64p, _f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u64p,
                                    _f32p, _u32p]
        self.ref = None
        ref_so = os.path.join(_HERE, "_ref", "libzvec_ref.so")
        if os.path.exists(ref_so) and _cpu_has("avx512f"):
            self.ref = R = C.CDLL(ref_so)
            for name in ("zref_sqeuclid_f32", "zref_ip_f32", "zref_minus_ip_f32",
                         "zref_cosine_f32"):
                f = getattr(R, name)
                f.restype = C.c_float
                f.argtypes = [_f32p, _f32p, C.c_size_t]
            R.zref_norm2_f32.restype = C.c_float
            R.zref_norm2_f32.argtypes = [_f32p, C.c_size_t]
            R.zref_normalize_l2_f32.restype = None
            R.zref_normalize_l2_f32.argtypes = [_f32p, C.c_size_t, _f32p]
            _u16p = C.POINTER(C.c_uint16)
            for name in ("zref_sqeuclid_f16", "zref_minus_ip_f16"):
                f = getattr(R, name)
                f.restype = C.c_float
                f.argtypes = [_u16p, _u16p, C.c_size_t]
            R.zref_to_fp16.restype = None
            R.zref_to_fp16.argtypes = [_f32p, C.c_size_t, _u16p]
            R.zref_heap_replay.restype = C.c_size_t
            R.zref_heap_replay.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_float, _u32p, _f32p]
            for name in ("zref_cosine_batch_f32", "zref_cosine_batch_f16"):
                f = getattr(R, name)
                f.restype = None
                f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, _f32p]

    # -- kernels -------------------------------------------------------------------------------
    def dist(self, metric, m, q, use_ref=False):
        m = np.ascontiguousarray(m, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        lib, pre = (self.ref, "zref_") if use_ref else (self.lib, "zo_")
        fn = {METRIC_L2: "sqeuclid_f32", METRIC_IP: "minus_ip_f32", METRIC_COSINE: "cosine_f32"}[metric]
        return float(getattr(lib, pre + fn)(_ptr(m, _f32p), _ptr(q, _f32p), m.size))

    def cosine_batch(self, rows, q, use_ref=False):
        """IndexMetric::batch_distance of the Cosine metric (cosine_metric.cc:202-212): ONE query against the rows, the
        one-to-many inner product of math_batch/ (NOT the 1x1 kernel's lane order).  rows [n][dim_with_norm] fp32 or fp16
        (converted rows: normalised vector + norm slot(s)), q likewise.  use_ref: the reference's own function (12 rows at a
        time + the remainder one by one, as it runs in the product)."""
        half = np.asarray(rows).dtype == np.float16
        dt = np.float16 if half else np.float32
        rows = np.ascontiguousarray(np.atleast_2d(rows), dt)
        q = np.ascontiguousarray(q, dt)
        n, dim = rows.shape
        out = np.zeros(n, np.float32)
        if use_ref:
            ptrs = (C.c_void_p * n)(*[rows.ctypes.data + i * rows.strides[0] for i in range(n)])
            fn = self.ref.zref_cosine_batch_f16 if half else self.ref.zref_cosine_batch_f32
            fn(C.cast(ptrs, C.c_void_p), C.c_void_p(q.ctypes.data), n, dim, _ptr(out, _f32p))
            return out
        for i in range(n):
            if half:
                u16 = C.POINTER(C.c_uint16)
                out[i] = self.lib.zo_cosine_batch_f16(rows[i].ctypes.data_as(u16), q.ctypes.data_as(u16), dim)
            else:
                out[i] = self.lib.zo_cosine_batch_f32(_ptr(rows[i], _f32p), _ptr(q, _f32p), dim)
        return out

    def dist16(self, metric, m, q, use_ref=False):
        """fp16 rows (numpy float16): SquaredEuclidean / MinusInnerProduct with fp32 accumulation."""
        m = np.ascontiguousarray(m, np.float16)
        q = np.ascontiguousarray(q, np.float16)
        _u16p = C.POINTER(C.c_uint16)
        lib, pre = (self.ref, "zref_") if use_ref else (self.lib, "zo_")
        fn = {METRIC_L2: "sqeuclid_f16", METRIC_IP: "minus_ip_f16"}[metric]
        return float(getattr(lib, pre + fn)(m.ctypes.data_as(_u16p), q.ctypes.data_as(_u16p), m.size))

    def to_fp16_ref(self, x):
        """FloatHelper::ToFP16 of the reference (needs _ref)."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.shape, np.float16)
        self.ref.zref_to_fp16(_ptr(x, _f32p), x.size, out.ctypes.data_as(C.POINTER(C.c_uint16)))
        return out

    def ip(self, m, q, use_ref=False):
        m = np.ascontiguousarray(m, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        lib, pre = (self.ref, "zref_") if use_ref else (self.lib, "zo_")
        return float(getattr(lib, pre + "ip_f32")(_ptr(m, _f32p), _ptr(q, _f32p), m.size))

    def norm2(self, m, use_ref=False):
        m = np.ascontiguousarray(m, np.float32)
        lib, pre = (self.ref, "zref_") if use_ref else (self.lib, "zo_")
        return float(getattr(lib, pre + "norm2_f32")(_ptr(m, _f32p), m.size))

    def normalize_l2(self, v, use_ref=False):
        v = np.array(v, np.float32, copy=True)
        n = C.c_float(0)
        lib, pre = (self.ref, "zref_") if use_ref else (self.lib, "zo_")
        getattr(lib, pre + "normalize_l2_f32")(_ptr(v, _f32p), v.size, C.byref(n))
        return v, float(n.value)

    def cosine_transform(self, x):
        """rows of d floats -> rows of d+1 floats (normalised + norm), CosineConverter/Reformer."""
        x = np.ascontiguousarray(np.atleast_2d(x), np.float32)
        out = np.empty((x.shape[0], x.shape[1] + 1), np.float32)
        for i in range(x.shape[0]):
            self.lib.zo_cosine_transform_f32(_ptr(x[i], _f32p), x.shape[1], _ptr(out[i], _f32p))
        return out

    def block_dist(self, metric, m_block, q_block, use_ref=False):
        """M x N block kernel: m_block [dim][M] (column-major block of M vectors), q_block [dim][N] (N interleaved
        queries) -> out [N][M].  metric L2 or IP (minus inner product).  M in {8, 16, 32}."""
        half = np.asarray(m_block).dtype == np.float16
        m = np.ascontiguousarray(m_block, np.float16 if half else np.float32)
        q = np.ascontiguousarray(q_block, np.float16 if half else np.float32)
        dim, M = m.shape
        N = q.shape[1]
        assert q.shape[0] == dim and M in (2, 4, 8, 16, 32)
        out = np.zeros((N, M), np.float32)
        name = "sqeuclid" if metric == METRIC_L2 else "minus_ip"
        if np.asarray(m_block).dtype == np.float16:                 # fp16 blocks
            m = np.ascontiguousarray(m_block, np.float16)
            q = np.ascontiguousarray(q_block, np.float16)
            fn = getattr(self.ref if use_ref else self.lib, ("zref_%s_block_f16" if use_ref else "zo_%s_block_f16") % name)
            fn.restype = C.c_int if use_ref else None
            rc = fn(M, N, m.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), C.c_size_t(dim), _ptr(out, _f32p))
            assert not use_ref or rc == 0
            return out
        if use_ref:
            fn = getattr(self.ref, "zref_%s_block_f32" % name)
            fn.restype = C.c_int
            rc = fn(M, N, _ptr(m, _f32p), _ptr(q, _f32p), C.c_size_t(dim), _ptr(out, _f32p))
            assert rc == 0
        else:
            fn = getattr(self.lib, "zo_%s_block_f32" % name)
            fn.restype = None
            fn(M, N, _ptr(m, _f32p), _ptr(q, _f32p), C.c_size_t(dim), _ptr(out, _f32p))
        return out

    def cosine_transform16(self, x):
        """fp32 rows of d floats -> fp16 rows of d+2 halves: CosineConverter with original fp32 / stored fp16
        (cosine_converter.cc:112-134): normalise in fp32, FloatHelper::ToFP16, then the fp32 norm's 4 bytes in the
        two trailing half slots."""
        c = self.cosine_transform(x)
        d = c.shape[1] - 1
        out = np.empty((c.shape[0], d + 2), np.float16)
        out[:, :d] = self.to_fp16_ref(c[:, :d]) if self.ref is not None else c[:, :d].astype(np.float16)
        out.view(np.uint16)[:, d:] = np.ascontiguousarray(c[:, d:]).view(np.uint16)
        return out

    def heap_replay(self, scores, limit, threshold=FLT_MAX, use_ref=False):
        s = np.ascontiguousarray(scores, np.float32)
        oi = np.zeros(max(limit, 1), np.uint32)
        os_ = np.zeros(max(limit, 1), np.float32)
        fn = self.ref.zref_heap_replay if use_ref else self.lib.zo_heap_replay
        n = fn(_ptr(s, _f32p), s.size, limit, threshold, _ptr(oi, _u32p), _ptr(os_, _f32p))
        return oi[:n].copy(), os_[:n].copy()

    def use_reference_kernels(self, on=True):
        """Route the scan loops' 1x1 distance through the reference's SIMD kernels (CPU baseline)."""
        names = {METRIC_L2: "zref_sqeuclid_f32", METRIC_IP: "zref_minus_ip_f32",
                 METRIC_COSINE: "zref_cosine_f32"}
        for m, nm in names.items():
            addr = C.cast(getattr(self.ref, nm), C.c_void_p) if (on and self.ref) else None
            self.lib.zo_set_distance_override(m, addr)
        for m, nm in {METRIC_L2: "zref_sqeuclid_f16", METRIC_IP: "zref_minus_ip_f16"}.items():
            addr = C.cast(getattr(self.ref, nm), C.c_void_p) if (on and self.ref) else None
            self.lib.zo_set_distance_override_f16(m, addr)
        return bool(on and self.ref)

    # -- scans ---------------------------------------------------------------------------------
    def flat_search(self, base, queries, topk, metric=METRIC_L2, keys=None, threshold=FLT_MAX,
                    exclude_bits=None, threads=1):
        half = np.asarray(base).dtype == np.float16          # fp16 rows => fp16 queries (HalfFloatReformer)
        dt = np.float16 if half else np.float32
        base = np.ascontiguousarray(base, dt)
        queries = np.ascontiguousarray(np.atleast_2d(queries), dt)
        n, dim = base.shape
        nq = queries.shape[0]
        keys = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        ex = None if exclude_bits is None else np.ascontiguousarray(exclude_bits, np.uint64)
        ok = np.zeros((nq, topk), np.uint64)
        os_ = np.full((nq, topk), np.inf, np.float32)
        oi = np.zeros((nq, topk), np.uint32)
        oc = np.zeros(nq, np.uint32)
        rc = self.lib.zo_flat_search_mt_t(int(half), C.c_void_p(base.ctypes.data), _ptr(keys, _u64p), n, dim, metric,
                                          C.c_void_p(queries.ctypes.data), nq, topk, threshold, _ptr(ex, _u64p),
                                          _ptr(ok, _u64p), _ptr(os_, _f32p), _ptr(oi, _u32p),
                                          _ptr(oc, _u32p), threads)
        if rc != 0:
            raise RuntimeError("zo_flat_search rc=%d" % rc)
        return ok, os_, oi, oc

    def flat_group_search(self, base, queries, group_of, group_num, group_topk, metric=METRIC_L2, keys=None, threshold=FLT_MAX,
                          exclude_bits=None, candidates=None):
        """group-by search restated (FlatStreamer::group_by_search_impl, flat_streamer.cc:391-437, result assembly
        topk_to_group_result, flat_streamer_context.h:135-180): every row not filtered out goes into the bounded heap
        (limit group_topk: the smallest scores, first seen wins a tie) of its group; the groups are ordered by their
        best score and the first group_num kept; documents beyond `threshold` are cut from the sorted lists.
        group_of: group number per storage position.  candidates: per query, the storage positions that compete, in
        scan order (group_by_search_p_keys_impl, :439-483); None = every row in storage order.  Built on flat_search
        with topk = all rows: the heap's kept set, in ascending (score, scan order).
        Returns per query a list of (group number, [(key, score, position), ...])."""
        base = np.asarray(base)
        queries = np.atleast_2d(queries)
        n = base.shape[0]
        out = []
        for qi in range(queries.shape[0]):
            if candidates is None:
                ok, os_, oi, oc = self.flat_search(base, queries[qi:qi + 1], max(n, 1), metric, keys=keys, exclude_bits=exclude_bits)
                order = [(float(os_[0, j]), int(oi[0, j]), int(ok[0, j])) for j in range(int(oc[0]))]
            else:
                cand = [int(p) for p in candidates[qi]]
                order = []
                if cand:
                    sub = np.ascontiguousarray(base[cand])
                    ok, os_, oi, oc = self.flat_search(sub, queries[qi:qi + 1], len(cand), metric)
                    for j in range(int(oc[0])):
                        p = cand[int(oi[0, j])]
                        order.append((float(os_[0, j]), p, int(keys[p]) if keys is not None else p))
            heaps, first = {}, []
            for sc, pos, key in order:                   # ascending (score, scan order): the first group_topk are the heap's
                g = int(group_of[pos])
                lst = heaps.get(g)
                if lst is None:
                    lst = heaps[g] = []
                    first.append((sc, g))                # best score of the group
                if len(lst) < group_topk:
                    lst.append((key, sc, pos))
            first.sort(key=lambda t: t[0])
            res = []
            for sc, g in first[:group_num]:
                res.append((g, [d for d in heaps[g] if not (d[1] > threshold)]))
            out.append(res)
        return out

    def flat_search_column(self, base, queries, topk, metric=METRIC_L2, keys=None, threshold=FLT_MAX, exclude_bits=None):
        """the column-major dense path (FlatSearcherContext::batch_search_column_*, flat_searcher_context.h:682-845):
        32-row transposed blocks x query groups of 32/16/8/4/2/1 through the M x N block kernels.  `base` row-major."""
        half = np.asarray(base).dtype == np.float16
        dt = np.float16 if half else np.float32
        base = np.ascontiguousarray(base, dt)
        queries = np.ascontiguousarray(np.atleast_2d(queries), dt)
        n, dim = base.shape
        nq = queries.shape[0]
        keys = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        ex = None if exclude_bits is None else np.ascontiguousarray(exclude_bits, np.uint64)
        ok = np.zeros((nq, topk), np.uint64)
        os_ = np.full((nq, topk), np.inf, np.float32)
        oi = np.zeros((nq, topk), np.uint32)
        oc = np.zeros(nq, np.uint32)
        fn = self.lib.zo_flat_search_column_t
        fn.restype = C.c_int
        rc = fn(int(half), C.c_void_p(base.ctypes.data), _ptr(keys, _u64p), C.c_uint64(n), C.c_uint32(dim), C.c_int(metric),
                C.c_void_p(queries.ctypes.data), C.c_uint32(nq), C.c_uint32(topk), C.c_float(threshold), _ptr(ex, _u64p),
                _ptr(ok, _u64p), _ptr(os_, _f32p), _ptr(oi, _u32p), _ptr(oc, _u32p))
        if rc != 0:
            raise RuntimeError("zo_flat_search_column rc=%d" % rc)
        return ok, os_, oi, oc

    def ivf_search(self, centroids, list_offsets, vecs, queries, topk, nprobe, max_scan_count,
                   metric=METRIC_L2, keys=None, threshold=FLT_MAX, brute_force=False,
                   exclude_bits=None, threads=1, want_probes=False):
        half = np.asarray(vecs).dtype == np.float16
        dt = np.float16 if half else np.float32
        centroids = np.ascontiguousarray(centroids, dt)
        vecs = np.ascontiguousarray(vecs, dt)
        queries = np.ascontiguousarray(np.atleast_2d(queries), dt)
        lo = np.ascontiguousarray(list_offsets, np.uint64)
        nlist, dim = centroids.shape
        nq = queries.shape[0]
        keys = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        ex = None if exclude_bits is None else np.ascontiguousarray(exclude_bits, np.uint64)
        ok = np.zeros((nq, topk), np.uint64)
        os_ = np.full((nq, topk), np.inf, np.float32)
        oi = np.zeros((nq, topk), np.uint32)
        oc = np.zeros(nq, np.uint32)
        osc = np.zeros(nq, np.uint32)
        npb = max(1, min(nprobe, nlist))
        if want_probes:
            assert not half, "probe listing is only wired for fp32"
            op = np.zeros((nq, npb), np.uint32)
            rc = self.lib.zo_ivf_search(_ptr(centroids, _f32p), nlist, _ptr(lo, _u64p),
                                        _ptr(vecs, _f32p), _ptr(keys, _u64p), dim, metric,
                                        _ptr(queries, _f32p), nq, topk, threshold, nprobe,
                                        max_scan_count, int(brute_force), _ptr(ex, _u64p),
                                        _ptr(ok, _u64p), _ptr(os_, _f32p), _ptr(oi, _u32p),
                                        _ptr(oc, _u32p), _ptr(osc, _u32p), _ptr(op, _u32p))
        else:
            op = None
            rc = self.lib.zo_ivf_search_mt_t(int(half), C.c_void_p(centroids.ctypes.data), nlist, _ptr(lo, _u64p),
                                             C.c_void_p(vecs.ctypes.data), _ptr(keys, _u64p), dim, metric,
                                             C.c_void_p(queries.ctypes.data), nq, topk, threshold, nprobe,
                                             max_scan_count, int(brute_force), _ptr(ex, _u64p),
                                             _ptr(ok, _u64p), _ptr(os_, _f32p), _ptr(oi, _u32p),
                                             _ptr(oc, _u32p), _ptr(osc, _u32p), threads)
        if rc != 0:
            raise RuntimeError("zo_ivf_search rc=%d" % rc)
        if want_probes:
            return ok, os_, oi, oc, osc, op
        return ok, os_, oi, oc, osc

    def ivf_label_and_pack(self, centroids, rows, metric=METRIC_L2, threads=1):
        """IVFBuilder::label + the dump order restated (test infrastructure).
        label: every row goes to the top-1 result of the centroid index for that row (ivf_builder.h:253-274 ->
        IVFCentroidIndex::search, ivf_centroid_index.cc:273-297, i.e. the flat scan over the centroids with topk = 1:
        ties keep the first centroid in id order, heap.h:103-114); pack: rows are written list by list, inside a list in
        the order they were labelled (ivf_builder.cc:607-650 appends to `labels_[cid]` under a lock per task of 10
        vectors and sorts nothing; the single-threaded order is ascending row number, which is what the dumper walks:
        ivf_builder.cc:652-729).  Returns (labels[n], list_offsets[nlist+1], order[n] = row of each list position)."""
        cent = np.ascontiguousarray(centroids)
        ok, _, oi, oc = self.flat_search(cent, rows, 1, metric, threads=threads)
        assert (oc == 1).all()
        labels = oi[:, 0].astype(np.uint32)
        nlist = cent.shape[0]
        sizes = np.bincount(labels, minlength=nlist)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        order = np.argsort(labels, kind="stable").astype(np.uint64)
        return labels, offs, order

    def merge_topk(self, keys, scores, counts, topk):
        """keys/scores [nparts][nq][topk], counts [nparts][nq]."""
        keys = np.ascontiguousarray(keys, np.uint64)
        scores = np.ascontiguousarray(scores, np.float32)
        counts = np.ascontiguousarray(counts, np.uint32)
        nparts, nq = counts.shape
        ok = np.zeros((nq, topk), np.uint64)
        os_ = np.full((nq, topk), np.inf, np.float32)
        oc = np.zeros(nq, np.uint32)
        rc = self.lib.zo_merge_topk(_ptr(keys, _u64p), _ptr(scores, _f32p), _ptr(counts, _u32p),
                                    nparts, nq, topk, _ptr(ok, _u64p), _ptr(os_, _f32p),
                                    _ptr(oc, _u32p))
        if rc != 0:
            raise RuntimeError("zo_merge_topk rc=%d" % rc)
        return ok, os_, oc


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


_ORACLE = None


def get():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE


def pack_bits(mask):
    """bool[n] (True = excluded) -> uint64 words, bit i of word i//64."""
    mask = np.asarray(mask, bool)
    n = mask.size
    words = np.zeros((n + 63) // 64, np.uint64)
    idx = np.nonzero(mask)[0]
    np.bitwise_or.at(words, idx // 64, np.uint64(1) << (idx % 64).astype(np.uint64))
    return words
