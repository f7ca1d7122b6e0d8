"""Python plumbing over the C ABI (include/zvec_hip.h), shaped like the reference's index operators.

The C++ host mirror of the reference interface lives in zvec_amd/csrc/host/ (hip_index.h); this module
is the same surface for Python callers (pytest, bench.py): `search_impl(query, count, ctx)` fills
`ctx.result(i)` with IndexDocument(key, score) lists exactly like
IndexSearcher::search_impl(query, qmeta, count, context) (src/include/zvec/core/framework/
index_runner.h:490-500) and returns 0 or a negative IndexError value (index_error.cc:20-71).
No exceptions cross `search_impl` for argument errors — same contract as the reference — but a
missing HIP library / device raises at construction: there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib

METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 2
DT_FP32, DT_FP16 = 0, 1


def _dtype_of(dtype):
    """IndexMeta::DataType of the rows: "fp32" (DT_FP32) or "fp16" (DT_FP16: what HalfFloatConverter writes and
    HalfFloatReformer turns queries into, src/core/quantizer/half_float_converter.cc / half_float_reformer.cc)."""
    if dtype in ("fp16", "float16", np.float16, DT_FP16) and dtype is not False and dtype != 0:
        return DT_FP16, np.float16
    return DT_FP32, np.float32
FLT_MAX = float(np.finfo(np.float32).max)

_METRIC_NAMES = {
    "SquaredEuclidean": METRIC_L2,   # src/core/metric/euclidean_metric.cc:743
    "InnerProduct": METRIC_IP,       # src/core/metric/inner_product_metric.cc:256
    "Cosine": METRIC_COSINE,         # src/core/metric/cosine_metric.cc:141
}


def metric_from_name(name):
    return _METRIC_NAMES[name]


class IndexError_:
    """Negative IndexError values used on this path (index_error.cc:20-71)."""
    Success = 0
    Runtime = -1
    Unsupported = -12
    OutOfRange = -17
    NoMemory = -19
    NoReady = -21
    NoExist = -22
    Mismatch = -24
    InvalidArgument = -31
    NoIndexLoaded = -204
    NoTrained = -205


class IndexDocument:
    """key + score (+ the stored vector when the context asked for it) — index_document.h:207-216 subset"""

    def __init__(self, key, score, vector=None):
        self._key = int(key)
        self._score = float(score)
        self._vector = vector

    def key(self):
        return self._key

    def score(self):
        return self._score

    def vector(self):
        """the stored row (fetch_vector contexts only, index.cc:635-647), else None"""
        return self._vector

    def __repr__(self):
        return "IndexDocument(key=%d, score=%r)" % (self._key, self._score)


class GroupIndexDocument:
    """group id + its documents (GroupIndexDocument, index_document.h:270-314)"""

    def __init__(self, group_id, docs):
        self._group_id, self._docs = group_id, docs

    def group_id(self):
        return self._group_id

    def docs(self):
        return self._docs


class DocFilter:
    """The reference's composite document filter (DocFilter::is_filtered, doc_filter.cc:74-87) as data:
        excluded(id) = deleted.contains(id) || !invert.contains(uint32(id)) || !forward[id]
    delete: bytes of a roaring bitmap — `kind` "roaring32" (roaring_bitmap_portable_serialize), "roaring64map"
    (Roaring64Map::write) or "file" (a delete-store file image incl. its 64-byte header); invert: bytes of a 32-bit
    portable roaring bitmap (ids that MATCH the inverted-index condition); forward: numpy bool array indexed by id
    (or packed LSB-first uint8 bits with `forward_len`).  Every term is optional."""

    KINDS = {"roaring32": _lib.ROARING_32, "roaring64map": _lib.ROARING_64MAP, "file": _lib.ROARING_FILE}

    def __init__(self, delete=None, kind="roaring32", invert=None, forward=None, forward_len=None):
        self.delete = None if delete is None else bytes(delete)
        self.kind = self.KINDS[kind]
        self.invert = None if invert is None else bytes(invert)
        if forward is None:
            self.forward, self.forward_len = None, 0
        elif forward_len is None:
            f = np.ascontiguousarray(forward, bool)
            self.forward, self.forward_len = np.packbits(f, bitorder="little"), int(f.size)
        else:
            self.forward, self.forward_len = np.ascontiguousarray(forward, np.uint8), int(forward_len)

    def _desc(self):
        d = _lib.DocFilterDesc()
        self._keep = []                      # buffers the descriptor points into
        if self.delete is not None:
            buf = C.create_string_buffer(self.delete, len(self.delete))
            self._keep.append(buf)
            d.delete_bitmap, d.delete_bytes, d.delete_kind = C.addressof(buf), len(self.delete), self.kind
        if self.invert is not None:
            buf = C.create_string_buffer(self.invert, len(self.invert))
            self._keep.append(buf)
            d.invert_bitmap, d.invert_bytes = C.addressof(buf), len(self.invert)
        if self.forward is not None:
            d.forward_bits, d.forward_len = self.forward.ctypes.data, self.forward_len
        return d


class IndexContext:
    """IndexContext subset used by the scan path (index_context.h:123-195): topk, filter, threshold,
    per-query result lists.  Owns one HIP stream + workspace (zvec_hip_ctx_t)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().zvec_hip_ctx_create(device, C.byref(self._h)), "zvec_hip_ctx_create")
        self._topk = 0
        self._threshold = FLT_MAX
        self._exclude = None      # numpy uint64 words (host) — materialised IndexFilter
        self._filter_fn = None
        self._doc_filter = None   # DocFilter: materialised by zvec_hip_*_build_filter
        self._fetch_vector = False
        self._scan_ratio, self._bf_threshold = None, None   # update(params) overrides of the IVF searcher's defaults
        self._group_num, self._group_topk, self._group_by = 0, 0, None
        self._group_results = []
        self._group_cache = None  # (fn, n) -> (group number of every position, group ids)
        self._results = []
        self.keys = None
        self.scores = None
        self.counts = None

    def __del__(self):
        try:
            if self._h:
                _lib.lib().zvec_hip_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- reference surface -------------------------------------------------------------------------
    def set_topk(self, topk):
        self._topk = int(topk)

    def topk(self):
        return self._topk

    def set_fetch_vector(self, enable):
        """IndexContext::set_fetch_vector (index_context.h:139): result documents carry their stored vectors."""
        self._fetch_vector = bool(enable)

    def fetch_vector(self):
        return self._fetch_vector

    def set_group_params(self, group_num, group_topk):
        """IndexContext::set_group_params (index_context.h:129; flat_streamer_context.h:187-191): group_num > 0 turns the
        next searches into group-by searches"""
        self._group_num, self._group_topk = int(group_num), int(group_topk)

    def set_group_by(self, fn):
        """IndexGroupBy: callable(key) -> group id (a string in the reference; any hashable here)"""
        self._group_by = fn
        self._group_cache = None

    def reset_group_by(self):
        self._group_by = None
        self._group_cache = None

    def group_by_search(self):
        return self._group_num > 0

    def group_result(self, index=0):
        """IndexGroupDocumentList of query `index`: groups best first, documents ascending by score"""
        return self._group_results[index]

    def _groups_for(self, keys_of_positions):
        """the group_by callback swept once over the stored keys (cached while the key list and the callback are the
        same): dense group numbers per position + the ids they stand for"""
        n = len(keys_of_positions)
        if self._group_cache is not None and self._group_cache[0] is self._group_by and self._group_cache[1] == n:
            return self._group_cache[2], self._group_cache[3]
        ids, number_of = [], {}
        of = np.empty(n, np.uint32)
        for i, k in enumerate(keys_of_positions):
            g = self._group_by(int(k))
            j = number_of.get(g)
            if j is None:
                j = number_of[g] = len(ids)
                ids.append(g)
            of[i] = j
        self._group_cache = (self._group_by, n, of, ids)
        return of, ids

    def _set_group_results(self, ids, groups, ngroups, keys, scores, counts, vectors_of=None):
        self.group_keys, self.group_scores, self.group_counts, self.group_numbers, self.group_ngroups = keys, scores, counts, groups, ngroups
        self._group_results = []
        for q in range(keys.shape[0]):
            lst = []
            for s in range(int(ngroups[q])):
                c = int(counts[q, s])
                vecs = vectors_of(keys[q, s, :c]) if (self._fetch_vector and vectors_of is not None and c) else None
                lst.append(GroupIndexDocument(ids[int(groups[q, s])],
                                              [IndexDocument(keys[q, s, j], scores[q, s, j], None if vecs is None else vecs[j]) for j in range(c)]))
            self._group_results.append(lst)

    def update(self, params):
        """IndexContext::update(params) (index_context.h:163-166): per-context overrides of the IVF search parameters,
        as IVFSearcherContext::update reads them (ivf_searcher_context.h:61-79) — "proxima.ivf.searcher.scan_ratio" and
        "proxima.ivf.searcher.brute_force_threshold"; a scan_ratio <= 0 is InvalidArgument; flat contexts ignore it."""
        ratio = params.get("proxima.ivf.searcher.scan_ratio", self._scan_ratio)
        if ratio is not None and float(ratio) <= 0.0:
            return IndexError_.InvalidArgument
        self._scan_ratio = None if ratio is None else float(ratio)
        bft = params.get("proxima.ivf.searcher.brute_force_threshold", self._bf_threshold)
        self._bf_threshold = None if bft is None else int(bft)
        return 0

    def set_threshold(self, val):
        self._threshold = float(val)

    def threshold(self):
        return self._threshold

    def set_filter(self, fn):
        """IndexFilter: callable(key) -> True means EXCLUDE (index_filter.h:48-50).  It is swept once
        over the index keys on the host into a bitset at the next search."""
        self._filter_fn = fn
        self._exclude = None

    def set_doc_filter(self, doc_filter):
        """the composite filter as data (DocFilter): materialised on the GPU at the next search, no host sweep."""
        self._doc_filter = doc_filter
        self._filter_fn = None
        self._exclude = None

    def set_exclude_bitset(self, words):
        """side channel: an already materialised predicate, 1 bit per storage position."""
        self._exclude = None if words is None else np.ascontiguousarray(words, np.uint64)
        self._filter_fn = None

    def reset_filter(self):
        self._filter_fn = None
        self._exclude = None
        self._doc_filter = None

    def result(self, index=0):
        return self._results[index]

    def results(self):
        return self._results

    # -- plumbing ----------------------------------------------------------------------------------
    def _exclude_for(self, keys_of_positions):
        if self._exclude is not None:
            return self._exclude
        if self._filter_fn is None:
            return None
        n = len(keys_of_positions)
        mask = np.fromiter((bool(self._filter_fn(int(k))) if k != 0xffffffffffffffff else False for k in keys_of_positions), bool, n)
        words = np.zeros((n + 63) // 64, np.uint64)
        idx = np.nonzero(mask)[0]
        np.bitwise_or.at(words, idx // 64, np.uint64(1) << (idx % 64).astype(np.uint64))
        return words

    def _set_results(self, keys, scores, counts, vectors_of=None):
        """vectors_of: callable(keys 1-D uint64) -> rows, used when fetch_vector is on"""
        self.keys, self.scores, self.counts = keys, scores, counts
        vecs = None
        if self._fetch_vector and vectors_of is not None:
            flat = np.concatenate([keys[q, :int(counts[q])] for q in range(keys.shape[0])]) if keys.shape[0] else np.zeros(0, np.uint64)
            vecs = vectors_of(flat)
        self._results = []
        o = 0
        for q in range(keys.shape[0]):
            c = int(counts[q])
            self._results.append([IndexDocument(keys[q, j], scores[q, j], None if vecs is None else vecs[o + j]) for j in range(c)])
            o += c

    def reform_queries_dev(self, d_in, count, dim, d_out, cosine=False, out_dtype="fp32", stream=None):
        """CosineReformer / HalfFloatReformer on raw fp32 queries already in HBM (device pointers)."""
        dt, _ = _dtype_of(out_dtype)
        _lib.check(_lib.lib().zvec_hip_reform_queries_dev(self._h, C.c_void_p(d_in), int(count), int(dim), int(bool(cosine)), dt,
                                                          C.c_void_p(d_out), C.c_void_p(stream or 0)), "zvec_hip_reform_queries_dev")

    def synchronize(self):
        _lib.check(_lib.lib().zvec_hip_ctx_synchronize(self._h), "zvec_hip_ctx_synchronize")

    def set_stream(self, stream_ptr):
        _lib.check(_lib.lib().zvec_hip_ctx_set_stream(self._h, C.c_void_p(stream_ptr)), "ctx_set_stream")

    def set_gate(self, gate):
        """share a Gate with other contexts: their dominant scan kernels then run one after the other while the rest of
        their searches overlaps (zvec_hip_ctx_set_gate); None detaches"""
        _lib.check(_lib.lib().zvec_hip_ctx_set_gate(self._h, gate._h if gate is not None else None), "zvec_hip_ctx_set_gate")
        self._gate = gate        # keeps the gate alive as long as the context uses it

    def profile(self, enable=True):
        _lib.check(_lib.lib().zvec_hip_ctx_profile(self._h, int(enable)), "zvec_hip_ctx_profile")

    def profile_read(self, reset=True):
        n = C.c_uint64(0)
        ms, b, f = C.c_double(0), C.c_double(0), C.c_double(0)
        _lib.check(_lib.lib().zvec_hip_ctx_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(b),
                                                        C.byref(f), int(reset)), "profile_read")
        return {"launches": int(n.value), "scan_ms": ms.value, "bytes": b.value, "flops": f.value}


def _np_ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class _FlatBase:
    """Common body of HipFlatStreamer / HipFlatSearcher: one HBM-resident blocked store."""

    def __init__(self, dim, metric=METRIC_L2, device=0, dtype="fp32"):
        if isinstance(metric, str):
            metric = metric_from_name(metric)
        self.dim = int(dim)
        self.metric = metric
        self.device = device
        self.dtype, self.np_dtype = _dtype_of(dtype)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().zvec_hip_flat_create(self.dim, self.dtype, metric, device, C.byref(self._h)),
                   "zvec_hip_flat_create")
        self._keys_host = []      # for IndexFilter sweeps

    def __del__(self):
        try:
            if self._h:
                _lib.lib().zvec_hip_flat_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def create_context(self):
        return IndexContext(self.device)

    def count(self):
        n = C.c_uint64(0)
        _lib.check(_lib.lib().zvec_hip_flat_count(self._h, C.byref(n)), "zvec_hip_flat_count")
        return int(n.value)

    def reserve(self, n):
        return _lib.lib().zvec_hip_flat_reserve(self._h, int(n))

    def add_batch(self, vecs, keys=None):
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        if vecs.ndim != 2 or vecs.shape[1] != self.dim:
            return IndexError_.InvalidArgument
        n0 = self.count()
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        rc = _lib.lib().zvec_hip_flat_append(self._h, _np_ptr(vecs), vecs.shape[0], _np_ptr(k))
        if rc == 0:
            self._keys_host.append(np.arange(n0, n0 + vecs.shape[0], dtype=np.uint64) if k is None else k.copy())
        return rc

    def add_batch_dev(self, d_ptr, n, d_keys_ptr=None, stream=None):
        """append rows already resident in HBM (device pointer, [n][dim] fp32 row-major)."""
        n0 = self.count()
        rc = _lib.lib().zvec_hip_flat_append_dev(self._h, C.c_void_p(d_ptr), int(n),
                                                 C.c_void_p(d_keys_ptr) if d_keys_ptr else None,
                                                 C.c_void_p(stream) if stream else None)
        if rc == 0 and d_keys_ptr is None:
            self._keys_host.append(("range", n0, n0 + int(n)))
        return rc

    def _all_keys(self):
        parts = []
        for p in self._keys_host:
            if isinstance(p, tuple):
                parts.append(np.arange(p[1], p[2], dtype=np.uint64))
            else:
                parts.append(p)
        return np.concatenate(parts) if parts else np.zeros(0, np.uint64)

    def get_vector_by_id(self, pos):
        out = np.zeros(self.dim, self.np_dtype)
        rc = _lib.lib().zvec_hip_flat_get_vector(self._h, int(pos), _np_ptr(out))
        return out if rc == 0 else None

    def get_vectors_by_ids(self, positions):
        pos = np.ascontiguousarray(positions, np.uint64)
        out = np.zeros((pos.size, self.dim), self.np_dtype)
        _lib.check(_lib.lib().zvec_hip_flat_get_vectors(self._h, _np_ptr(pos), pos.size, _np_ptr(out)), "zvec_hip_flat_get_vectors")
        return out

    def _vectors_of_keys(self, keys):
        """stored rows of the documents with these keys (fetch_vector); key -> position through a sorted view of the keys"""
        allk = self._all_keys()
        order = np.argsort(allk, kind="stable")
        pos = order[np.searchsorted(allk[order], keys)]
        return self.get_vectors_by_ids(pos)

    def build_filter(self, doc_filter, ctx=None, d_out=None, stream=None):
        """DocFilter -> exclude bitset over this index's storage positions, built on the GPU.  Returns numpy
        uint64 words, or fills the device buffer `d_out` ((count+63)//64 uint64) and returns None."""
        n = self.count()
        words = None if d_out is not None else np.zeros((n + 63) // 64, np.uint64)
        desc = doc_filter._desc()
        rc = _lib.lib().zvec_hip_flat_build_filter(self._h, ctx._h if ctx else None, C.byref(desc),
                                                   C.c_void_p(d_out) if d_out is not None else _np_ptr(words),
                                                   int(d_out is not None), C.c_void_p(stream or 0))
        _lib.check(rc, "zvec_hip_flat_build_filter")
        return words

    def search_impl(self, query, count, ctx):
        """IndexRunner::search_impl(query, qmeta, count, context); query: [count][dim] fp32."""
        if ctx is None or (ctx.topk() == 0 and not ctx.group_by_search()):
            return IndexError_.InvalidArgument      # flat_searcher.cc:194-198
        q = np.ascontiguousarray(query, self.np_dtype).reshape(-1)
        if q.size != int(count) * self.dim:
            return IndexError_.InvalidArgument
        if ctx.group_by_search():
            return self._group_search(q, count, ctx, None)      # flat_streamer.cc:323-324
        k = ctx.topk()
        keys = np.zeros((count, k), np.uint64)
        scores = np.zeros((count, k), np.float32)
        counts = np.zeros(count, np.uint32)
        if ctx._doc_filter is not None:
            ex = self.build_filter(ctx._doc_filter, ctx)
        else:
            ex = ctx._exclude_for(self._all_keys()) if (ctx._filter_fn or ctx._exclude is not None) else None
        rc = _lib.lib().zvec_hip_flat_search(self._h, ctx._h, _np_ptr(q), count, k, ctx.threshold(),
                                             _np_ptr(ex), _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
        if rc == 0:
            ctx._set_results(keys, scores, counts, self._vectors_of_keys)
        return rc

    # brute force == the flat scan itself (flat_streamer.cc:304-344)
    search_bf_impl = search_impl

    def _group_search(self, q, count, ctx, p_keys):
        """group_by_search_impl / group_by_search_p_keys_impl (flat_streamer.cc:391-483)"""
        if ctx._group_by is None:
            return IndexError_.InvalidArgument                  # "Invalid group-by function"
        allk = self._all_keys()
        of, ids = ctx._groups_for(allk)
        gnum, gk = ctx._group_num, ctx._group_topk
        groups = np.zeros((count, gnum), np.uint32)
        ngroups = np.zeros(count, np.uint32)
        keys = np.zeros((count, gnum, gk), np.uint64)
        scores = np.zeros((count, gnum, gk), np.float32)
        counts = np.zeros((count, gnum), np.uint32)
        L = _lib.lib()
        if p_keys is None:
            if ctx._doc_filter is not None:
                ex = self.build_filter(ctx._doc_filter, ctx)
            else:
                ex = ctx._exclude_for(allk) if (ctx._filter_fn or ctx._exclude is not None) else None
            rc = L.zvec_hip_flat_search_grouped(self._h, ctx._h, _np_ptr(q), count, _np_ptr(of), max(len(ids), 1), gnum, gk,
                                                ctx.threshold(), _np_ptr(ex), _np_ptr(groups), _np_ptr(ngroups), _np_ptr(keys),
                                                _np_ptr(scores), _np_ptr(counts))
        else:
            ids_l, offs = self._p_keys_positions(p_keys, ctx)
            rc = L.zvec_hip_flat_search_grouped_by_ids(self._h, ctx._h, _np_ptr(q), count, _np_ptr(ids_l), _np_ptr(offs), _np_ptr(of),
                                                       max(len(ids), 1), gnum, gk, ctx.threshold(), _np_ptr(ctx._exclude),
                                                       _np_ptr(groups), _np_ptr(ngroups), _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
        if rc == 0:
            ctx._set_group_results(ids, groups, ngroups, keys, scores, counts, self._vectors_of_keys)
        return rc

    def _p_keys_positions(self, p_keys, ctx):
        """primary keys -> storage positions (unknown keys and keys the context's filter rejects are dropped)"""
        allk = self._all_keys()
        if getattr(self, "_key2pos_n", -1) != len(allk):
            self._key2pos = {int(k): i for i, k in enumerate(allk)}
            self._key2pos_n = len(allk)
        ids, offs = [], [0]
        for keys in p_keys:
            for key in keys:
                pos = self._key2pos.get(int(key))
                if pos is not None and not (ctx._filter_fn and ctx._filter_fn(int(key))):
                    ids.append(pos)
            offs.append(len(ids))
        return np.asarray(ids if ids else [0], np.uint32), np.asarray(offs, np.uint32)

    def search_bf_by_p_keys_impl(self, query, p_keys, count, ctx):
        """FlatStreamer::search_bf_by_p_keys_impl (flat_streamer.cc:346-389): p_keys[q] = primary keys query q
        is compared with; unknown keys are skipped; the context's filter applies per key."""
        if ctx is None or (ctx.topk() == 0 and not ctx.group_by_search()):
            return IndexError_.InvalidArgument
        q = np.ascontiguousarray(query, self.np_dtype).reshape(-1)
        if q.size != int(count) * self.dim or len(p_keys) != count:
            return IndexError_.InvalidArgument
        if ctx.group_by_search():
            return self._group_search(q, count, ctx, p_keys)    # flat_streamer.cc:365-366
        ids, offs = self._p_keys_positions(p_keys, ctx)
        k = ctx.topk()
        keys_o = np.zeros((count, k), np.uint64)
        scores = np.zeros((count, k), np.float32)
        counts = np.zeros(count, np.uint32)
        rc = _lib.lib().zvec_hip_flat_search_by_ids(self._h, ctx._h, _np_ptr(q), count, _np_ptr(ids), _np_ptr(offs), k,
                                                    ctx.threshold(), _np_ptr(ctx._exclude), _np_ptr(keys_o),
                                                    _np_ptr(scores), _np_ptr(counts))
        if rc == 0:
            ctx._set_results(keys_o, scores, counts, self._vectors_of_keys)
        return rc

    def batch_distance(self, query, positions, ctx=None):
        """IndexMetric::batch_distance: one query against the listed storage positions, scores in that order"""
        q = np.ascontiguousarray(query, self.np_dtype).reshape(-1)
        pos = np.ascontiguousarray(positions, np.uint32)
        out = np.zeros(pos.size, np.float32)
        _lib.check(_lib.lib().zvec_hip_flat_batch_distance(self._h, ctx._h if ctx else None, _np_ptr(q), _np_ptr(pos), pos.size,
                                                           _np_ptr(out)), "zvec_hip_flat_batch_distance")
        return out

    def search_dev(self, d_queries, count, topk, d_out_keys, d_out_scores, d_out_counts, ctx,
                   threshold=FLT_MAX, d_exclude=None, stream=None):
        """device-pointer form (async): all arguments are raw device pointers (ints)."""
        return _lib.lib().zvec_hip_flat_search_dev(
            self._h, ctx._h, C.c_void_p(d_queries), count, topk, threshold,
            C.c_void_p(d_exclude) if d_exclude else None, C.c_void_p(d_out_keys),
            C.c_void_p(d_out_scores), C.c_void_p(d_out_counts), C.c_void_p(stream) if stream else None)


    def set_shadow(self, enable=True, preselect=0):
        """half-width pre-selection (zvec_hip_flat_set_shadow): an fp16 twin of the rows present now is scanned for `preselect` rows
        per query, those are re-scored in fp32 and the result is certified; any later mutation of the store drops the twin"""
        _lib.check(_lib.lib().zvec_hip_flat_set_shadow(self._h, int(bool(enable)), int(preselect)), "zvec_hip_flat_set_shadow")

    def shadow_info(self):
        on, nbytes, err, norm = C.c_int(0), C.c_uint64(0), C.c_float(0), C.c_float(0)
        _lib.check(_lib.lib().zvec_hip_flat_shadow_info(self._h, C.byref(on), C.byref(nbytes), C.byref(err), C.byref(norm)),
                   "zvec_hip_flat_shadow_info")
        return {"enabled": bool(on.value), "bytes": int(nbytes.value), "max_row_error": float(err.value), "max_row_norm": float(norm.value)}

    def shadow_width(self, topk):
        """rows the next search with this k pre-selects per query (zvec_hip_flat_shadow_width; 0: no twin)"""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().zvec_hip_flat_shadow_width(self._h, int(topk), C.byref(n)), "zvec_hip_flat_shadow_width")
        return int(n.value)

    def shadow_certify(self, d_queries, count, topk, d_out_keys, d_out_scores, d_out_counts, ctx, d_exclude=None, stream=None):
        """the second half of search_dev on a store with shadow rows: waits, re-runs the uncertified queries on the fp32 rows;
        returns how many were re-run"""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().zvec_hip_flat_shadow_certify(
            self._h, ctx._h, C.c_void_p(d_queries), count, topk, C.c_void_p(d_exclude) if d_exclude else None, C.c_void_p(d_out_keys),
            C.c_void_p(d_out_scores), C.c_void_p(d_out_counts), C.c_void_p(stream) if stream else None, C.byref(n)),
            "zvec_hip_flat_shadow_certify")
        return int(n.value)


class HipFlatStreamer(_FlatBase):
    """stands where "FlatStreamer" is registered (flat_streamer.cc:486-489): mutable, add + search."""

    def add_impl(self, key, vec, ctx=None):
        return self.add_batch(np.asarray(vec, self.np_dtype).reshape(1, -1), np.array([key], np.uint64))

    INVALID_KEY = np.uint64(0xffffffffffffffff)          # kInvalidKey (flat_index_format.h:29)

    def add_with_id_impl(self, doc_id, vec, ctx=None):
        """IndexStreamer::add_with_id_impl (index_runner.h:483-487): what core_interface::Index::_dense_add calls"""
        return self.add_with_id_batch([doc_id], np.asarray(vec, self.np_dtype).reshape(1, -1))

    def add_with_id_batch(self, ids, vecs):
        """FlatStreamerEntity::add_vector_with_id row by row (flat_streamer_entity.cc:900-990): row ids[i] lives at
        storage position ids[i] under key ids[i]; gaps are padded with holes no search returns, an id below the count
        overwrites in place."""
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        ids = np.ascontiguousarray(ids, np.uint32)
        if vecs.ndim != 2 or vecs.shape[1] != self.dim or ids.size != vecs.shape[0]:
            return IndexError_.InvalidArgument
        rc = _lib.lib().zvec_hip_flat_put(self._h, _np_ptr(ids), ids.size, _np_ptr(vecs), None)
        if rc == 0:
            keys = self._all_keys()
            n = self.count()
            if keys.size < n:
                keys = np.concatenate([keys, np.full(n - keys.size, self.INVALID_KEY, np.uint64)])
            keys[ids.astype(np.int64)] = ids.astype(np.uint64)
            self._keys_host = [keys]
        return rc

    def holes(self):
        c = C.c_uint64(0)
        _lib.check(_lib.lib().zvec_hip_flat_holes(self._h, C.byref(c)), "zvec_hip_flat_holes")
        return int(c.value)


class _FlatFeatures:
    def load_features(self, features, count, column_major=False, batch_size=32, keys=None):
        """the features segment of a dumped flat index (FlatBuilder write_row_index / write_column_index layouts)"""
        fb = bytes(features)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        n0 = self.count()
        rc = _lib.lib().zvec_hip_flat_load_features(self._h, fb, len(fb), int(count), int(bool(column_major)), int(batch_size),
                                                    _np_ptr(k))
        if rc == 0:
            self._keys_host.append(np.arange(n0, n0 + int(count), dtype=np.uint64) if k is None else k.copy())
        return rc


class HipFlatSearcher(_FlatBase, _FlatFeatures):
    """stands where "FlatSearcher" is registered (flat_searcher.cc:247-250): load once, search."""

    def load(self, vecs, keys=None):
        return self.add_batch(vecs, keys)


class HipIVFSearcher:
    """stands where "IVFSearcher"/"IVFStreamer" are registered (ivf_searcher.cc:183-250)."""

    def __init__(self, dim, metric=METRIC_L2, device=0, scan_ratio=0.1, brute_force_threshold=1000, dtype="fp32"):
        if isinstance(metric, str):
            metric = metric_from_name(metric)
        self.dim = int(dim)
        self.metric = metric
        self.device = device
        self.dtype, self.np_dtype = _dtype_of(dtype)
        # IVFSearcherContext defaults (ivf_searcher_context.h:211-213)
        self.scan_ratio = float(scan_ratio)
        self.brute_force_threshold = int(brute_force_threshold)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().zvec_hip_ivf_create(self.dim, self.dtype, metric, device, C.byref(self._h)),
                   "zvec_hip_ivf_create")
        self._list_keys = None    # keys in list order (filter sweeps)
        self._orig_keys = None    # keys per original row (build)
        self.total_count = 0

    def __del__(self):
        try:
            if self._h:
                _lib.lib().zvec_hip_ivf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def create_context(self):
        return IndexContext(self.device)

    def set_shard(self, shard, nshards):
        return _lib.lib().zvec_hip_ivf_keep_shard(self._h, shard, nshards)

    def load(self, centroids, list_offsets, vecs, keys=None):
        centroids = np.ascontiguousarray(centroids, self.np_dtype)
        lo = np.ascontiguousarray(list_offsets, np.uint64)
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        rc = _lib.lib().zvec_hip_ivf_load(self._h, _np_ptr(centroids), centroids.shape[0], _np_ptr(lo),
                                          _np_ptr(vecs), _np_ptr(k))
        if rc == 0:
            self.total_count = int(lo[-1])
            self._orig_keys = k
            self._list_keys = None
        return rc

    def load_segments(self, inverted_header, inverted_meta, inverted_body, keys, centroids):
        """IVFSearcher::load from the raw segment payloads of a dumped reference index (bytes objects: "ivf.inverted_header",
        "ivf.inverted_meta", "ivf.inverted_body", "hc.keys") + the centroid rows; see zvec_hip_ivf_load_segments."""
        cent = np.ascontiguousarray(centroids, self.np_dtype)
        hb, mb, bb, kb = bytes(inverted_header), bytes(inverted_meta), bytes(inverted_body), bytes(keys)
        rc = _lib.lib().zvec_hip_ivf_load_segments(self._h, hb, len(hb), mb, len(mb), bb, len(bb), kb, len(kb), _np_ptr(cent))
        if rc == 0:
            self.total_count = len(kb) // 8
            self._orig_keys = np.frombuffer(kb, np.uint64, self.total_count).copy()
            self._list_keys = None
        return rc

    def build(self, vecs, nlist, keys=None, kmeans_iters=10, sample_per_list=256, seed=20260320):
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        rc = _lib.lib().zvec_hip_ivf_build(self._h, _np_ptr(vecs), vecs.shape[0], _np_ptr(k), nlist,
                                           kmeans_iters, sample_per_list, seed)
        if rc == 0:
            self.total_count = vecs.shape[0]
            self._orig_keys = k
            self._list_keys = None
        return rc

    def build_dev(self, d_vecs, n, nlist, keys=None, kmeans_iters=10, sample_per_list=256, seed=20260320,
                  stream=None):
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        rc = _lib.lib().zvec_hip_ivf_build_dev(self._h, C.c_void_p(d_vecs), int(n), _np_ptr(k), nlist,
                                               kmeans_iters, sample_per_list, seed,
                                               C.c_void_p(stream) if stream else None)
        if rc == 0:
            self.total_count = int(n)
            self._orig_keys = k
            self._list_keys = None
        return rc

    # ---- streamed build (IVFBuilder train / label / dump as separate steps; see include/zvec_hip.h) ----
    def train_dev(self, d_sample, n_sample, nlist, kmeans_iters=10, seed=20260320, stream=None):
        return _lib.lib().zvec_hip_ivf_train_dev(self._h, C.c_void_p(d_sample), int(n_sample), int(nlist), int(kmeans_iters),
                                                 int(seed), C.c_void_p(stream) if stream else None)

    def set_centroids(self, centroids):
        cent = np.ascontiguousarray(centroids, self.np_dtype)
        return _lib.lib().zvec_hip_ivf_set_centroids(self._h, _np_ptr(cent), cent.shape[0])

    def get_centroids(self):
        nl = C.c_uint32(0)
        _lib.check(_lib.lib().zvec_hip_ivf_get_centroids(self._h, None, C.byref(nl)), "zvec_hip_ivf_get_centroids")
        cent = np.zeros((int(nl.value), self.dim), self.np_dtype)
        _lib.check(_lib.lib().zvec_hip_ivf_get_centroids(self._h, _np_ptr(cent), None), "zvec_hip_ivf_get_centroids")
        return cent

    def label_dev(self, d_rows, n, d_labels, stream=None):
        """d_labels: device uint32[n]; complete when the call returns."""
        return _lib.lib().zvec_hip_ivf_label_dev(self._h, C.c_void_p(d_rows), int(n), C.c_void_p(d_labels),
                                                 C.c_void_p(stream) if stream else None)

    def begin_lists(self, list_sizes):
        sz = np.ascontiguousarray(list_sizes, np.uint32)
        rc = _lib.lib().zvec_hip_ivf_begin_lists(self._h, _np_ptr(sz))
        if rc == 0:
            self.total_count = int(sz.astype(np.uint64).sum())
            self._orig_keys = None
            self._list_keys = None
            self._streamed_keys = False
        return rc

    def add_dev(self, d_rows, n, labels, first_row, keys=None, stream=None):
        lab = np.ascontiguousarray(labels, np.uint32)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        if k is not None:
            self._streamed_keys = True     # explicit keys arrive per chunk: keys_in_list_order() is not tracked for them
        return _lib.lib().zvec_hip_ivf_add_dev(self._h, C.c_void_p(d_rows), int(n), _np_ptr(lab), _np_ptr(k), int(first_row),
                                               C.c_void_p(stream) if stream else None)

    def end_lists(self):
        return _lib.lib().zvec_hip_ivf_end_lists(self._h)

    def list_owners(self):
        own = np.zeros(self.info()[1], np.uint32)
        _lib.check(_lib.lib().zvec_hip_ivf_list_owners(self._h, _np_ptr(own)), "zvec_hip_ivf_list_owners")
        return own

    def info(self):
        n = C.c_uint64(0)
        nl = C.c_uint32(0)
        _lib.check(_lib.lib().zvec_hip_ivf_info(self._h, C.byref(n), C.byref(nl)), "zvec_hip_ivf_info")
        return int(n.value), int(nl.value)

    def export(self):
        n, nlist = self.info()
        cent = np.zeros((nlist, self.dim), self.np_dtype)
        lo = np.zeros(nlist + 1, np.uint64)
        rows = np.zeros(n, np.uint64)
        _lib.check(_lib.lib().zvec_hip_ivf_export(self._h, _np_ptr(cent), _np_ptr(lo), _np_ptr(rows)),
                   "zvec_hip_ivf_export")
        return cent, lo, rows

    def get_vector_by_id(self, list_pos):
        out = np.zeros(self.dim, self.np_dtype)
        rc = _lib.lib().zvec_hip_ivf_get_vector(self._h, int(list_pos), _np_ptr(out))
        return out if rc == 0 else None

    def get_vectors_by_ids(self, list_positions):
        pos = np.ascontiguousarray(list_positions, np.uint64)
        out = np.zeros((pos.size, self.dim), self.np_dtype)
        _lib.check(_lib.lib().zvec_hip_ivf_get_vectors(self._h, _np_ptr(pos), pos.size, _np_ptr(out)), "zvec_hip_ivf_get_vectors")
        return out

    def _vectors_of_keys(self, keys):
        lk = self.keys_in_list_order()
        order = np.argsort(lk, kind="stable")
        return self.get_vectors_by_ids(order[np.searchsorted(lk[order], keys)])

    # IVFSearcherContext::update (ivf_searcher_context.h:61-79)
    def probe_params(self, ctx=None):
        """nprobe / max_scan_count from the searcher's defaults, or from what the context's update(params) set"""
        ratio = self.scan_ratio if ctx is None or ctx._scan_ratio is None else ctx._scan_ratio
        bft = self.brute_force_threshold if ctx is None or ctx._bf_threshold is None else ctx._bf_threshold
        return ivf_probe_params(self.info()[1], self.total_count, ratio, bft)

    def set_nprobe(self, nprobe, exact=True):
        """boundary A's `nprobe` as patches/boundary_a.diff hands it to the searcher context: scan_ratio = nprobe / nlist;
        exact=True also sets brute_force_threshold = N - 1, which makes every query probe exactly nprobe lists
        (max_scan_count then never cuts the probe loop short)."""
        self.scan_ratio = float(np.float32(nprobe) / np.float32(max(self.info()[1], 1)))
        if exact:
            self.brute_force_threshold = max(self.total_count - 1, 0)

    def keys_in_list_order(self):
        """key of every dense list-order position (what IVFEntity::get_keys reads, ivf_entity.cc:612)."""
        if self._list_keys is None:
            _, _, rows = self.export()
            self._list_keys = rows if self._orig_keys is None else self._orig_keys[rows.astype(np.int64)]
        return self._list_keys

    def build_filter(self, doc_filter, ctx=None, d_out=None, stream=None):
        """DocFilter -> exclude bitset over list-order positions (see _FlatBase.build_filter)."""
        n = self.info()[0]
        words = None if d_out is not None else np.zeros((n + 63) // 64, np.uint64)
        desc = doc_filter._desc()
        rc = _lib.lib().zvec_hip_ivf_build_filter(self._h, ctx._h if ctx else None, C.byref(desc),
                                                  C.c_void_p(d_out) if d_out is not None else _np_ptr(words),
                                                  int(d_out is not None), C.c_void_p(stream or 0))
        _lib.check(rc, "zvec_hip_ivf_build_filter")
        return words

    def _exclude(self, ctx):
        if ctx._doc_filter is not None:
            return self.build_filter(ctx._doc_filter, ctx)
        if ctx._exclude is not None:
            return ctx._exclude
        if ctx._filter_fn is None:
            return None
        return ctx._exclude_for(self.keys_in_list_order())

    def search_impl(self, query, count, ctx):
        if ctx is None or ctx.topk() == 0:
            return IndexError_.InvalidArgument      # ivf_searcher.cc:197-200
        bft = self.brute_force_threshold if ctx._bf_threshold is None else ctx._bf_threshold
        if self.total_count <= bft:
            return self.search_bf_impl(query, count, ctx)   # ivf_searcher.cc:188-190
        q = np.ascontiguousarray(query, self.np_dtype).reshape(-1)
        if q.size != int(count) * self.dim:
            return IndexError_.InvalidArgument
        k = ctx.topk()
        nprobe, max_scan = self.probe_params(ctx)
        keys = np.zeros((count, k), np.uint64)
        scores = np.zeros((count, k), np.float32)
        counts = np.zeros(count, np.uint32)
        ex = self._exclude(ctx)
        rc = _lib.lib().zvec_hip_ivf_search(self._h, ctx._h, _np_ptr(q), count, k, ctx.threshold(), nprobe,
                                            max_scan, _np_ptr(ex), _np_ptr(keys), _np_ptr(scores),
                                            _np_ptr(counts))
        if rc == 0:
            ctx._set_results(keys, scores, counts, self._vectors_of_keys)
        return rc

    def search_bf_impl(self, query, count, ctx):
        if ctx is None or ctx.topk() == 0:
            return IndexError_.InvalidArgument
        q = np.ascontiguousarray(query, self.np_dtype).reshape(-1)
        if q.size != int(count) * self.dim:
            return IndexError_.InvalidArgument
        k = ctx.topk()
        keys = np.zeros((count, k), np.uint64)
        scores = np.zeros((count, k), np.float32)
        counts = np.zeros(count, np.uint32)
        ex = self._exclude(ctx)
        rc = _lib.lib().zvec_hip_ivf_search_bf(self._h, ctx._h, _np_ptr(q), count, k, ctx.threshold(),
                                               _np_ptr(ex), _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
        if rc == 0:
            ctx._set_results(keys, scores, counts, self._vectors_of_keys)
        return rc

    def search_dev(self, d_queries, count, topk, nprobe, max_scan, d_out_keys, d_out_scores, d_out_counts,
                   ctx, threshold=FLT_MAX, d_exclude=None, stream=None):
        return _lib.lib().zvec_hip_ivf_search_dev(
            self._h, ctx._h, C.c_void_p(d_queries), count, topk, threshold, nprobe, max_scan,
            C.c_void_p(d_exclude) if d_exclude else None, C.c_void_p(d_out_keys), C.c_void_p(d_out_scores),
            C.c_void_p(d_out_counts), C.c_void_p(stream) if stream else None)

    def set_shadow(self, enable=True, preselect=0):
        """half-width pre-selection (zvec_hip_ivf_set_shadow): an fp16 twin of the fp32 lists is scanned for `preselect` rows per
        query, those are re-scored in fp32 and the result is certified; uncertified queries are re-run on the fp32 lists"""
        _lib.check(_lib.lib().zvec_hip_ivf_set_shadow(self._h, int(bool(enable)), int(preselect)), "zvec_hip_ivf_set_shadow")

    def shadow_info(self):
        on, nbytes, err, norm = C.c_int(0), C.c_uint64(0), C.c_float(0), C.c_float(0)
        _lib.check(_lib.lib().zvec_hip_ivf_shadow_info(self._h, C.byref(on), C.byref(nbytes), C.byref(err), C.byref(norm)),
                   "zvec_hip_ivf_shadow_info")
        return {"enabled": bool(on.value), "bytes": int(nbytes.value), "max_row_error": float(err.value), "max_row_norm": float(norm.value)}

    def shadow_width(self, topk):
        """rows the next search with this k pre-selects per query (zvec_hip_ivf_shadow_width; 0: no twin)"""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().zvec_hip_ivf_shadow_width(self._h, int(topk), C.byref(n)), "zvec_hip_ivf_shadow_width")
        return int(n.value)

    def shadow_certify(self, d_queries, count, topk, nprobe, max_scan, d_out_keys, d_out_scores, d_out_counts, ctx,
                       d_exclude=None, stream=None):
        """the second half of search_dev on an index with shadow lists: waits, re-runs the uncertified queries on the fp32 lists;
        returns how many were re-run"""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().zvec_hip_ivf_shadow_certify(
            self._h, ctx._h, C.c_void_p(d_queries), count, topk, nprobe, max_scan, C.c_void_p(d_exclude) if d_exclude else None,
            C.c_void_p(d_out_keys), C.c_void_p(d_out_scores), C.c_void_p(d_out_counts), C.c_void_p(stream) if stream else None,
            C.byref(n)), "zvec_hip_ivf_shadow_certify")
        return int(n.value)

    def coarse_dev(self, d_queries, count, nprobe, d_probe_idx, d_probe_cnt, ctx, stream=None):
        """the coarse pass alone (zvec_hip_ivf_coarse_dev): probe lists [count][min(nprobe, nlist)] u32 + [count] u32, device pointers"""
        return _lib.lib().zvec_hip_ivf_coarse_dev(self._h, ctx._h, C.c_void_p(d_queries), count, nprobe, C.c_void_p(d_probe_idx),
                                                  C.c_void_p(d_probe_cnt), C.c_void_p(stream) if stream else None)

    def search_probes_dev(self, d_queries, count, topk, nprobe, max_scan, d_probe_idx, d_probe_cnt, d_out_keys, d_out_scores,
                          d_out_counts, ctx, threshold=FLT_MAX, d_exclude=None, stream=None):
        """zvec_hip_ivf_search_dev without its coarse pass: plans from the given probe lists (zvec_hip_ivf_search_probes_dev)"""
        return _lib.lib().zvec_hip_ivf_search_probes_dev(
            self._h, ctx._h, C.c_void_p(d_queries), count, topk, threshold, nprobe, max_scan, C.c_void_p(d_probe_idx),
            C.c_void_p(d_probe_cnt), C.c_void_p(d_exclude) if d_exclude else None, C.c_void_p(d_out_keys), C.c_void_p(d_out_scores),
            C.c_void_p(d_out_counts), C.c_void_p(stream) if stream else None)

    def last_stats(self, ctx, count):
        scanned = np.zeros(count, np.uint32)
        probes = np.zeros(count, np.uint32)
        _lib.check(_lib.lib().zvec_hip_ivf_last_stats(self._h, ctx._h, count, _np_ptr(scanned), _np_ptr(probes)),
                   "zvec_hip_ivf_last_stats")
        return scanned, probes


# "IVFStreamer" (what the product instantiates, indexes/ivf_index.cc:38-39) is a read-only operator over a dumped index
# in the reference too (ivf_streamer.h:28-85): same class
HipIVFStreamer = HipIVFSearcher


def container_segments(image, checksum=False):
    """segment table of a dumped index FILE image (zvec_hip_container_segments): {id: (offset, size)} in file order"""
    image = bytes(image)
    n = C.c_uint32(0)
    L = _lib.lib()
    _lib.check(L.zvec_hip_container_segments(image, len(image), int(checksum), None, 0, C.byref(n)), "zvec_hip_container_segments")
    arr = (_lib.Segment * max(int(n.value), 1))()
    _lib.check(L.zvec_hip_container_segments(image, len(image), int(checksum), arr, n.value, C.byref(n)), "zvec_hip_container_segments")
    return {arr[i].id.decode(): (int(arr[i].offset), int(arr[i].size)) for i in range(n.value)}


def parse_index_meta(blob):
    """the "IndexMeta" segment: IndexMetaFormatHeader (src/core/framework/index_meta.cc:23-34) + its JSON attachment
    (metric name under "metric"."name", index_meta.cc:39-80)"""
    import json
    import struct
    hs, meta_type, major, dt, dim, unit, space, aoff, asz = struct.unpack_from("<9I", blob, 0)
    metric, att = None, {}
    if asz:                               # attachment_offset counts from the start of the blob (index_meta.cc:62-80)
        try:
            att = json.loads(bytes(blob[aoff:aoff + asz]).decode())
            metric = att.get("metric", {}).get("name")
        except (ValueError, UnicodeDecodeError):
            metric, att = None, {}
    out = {"major_order": major, "data_type": dt, "dimension": dim, "unit_size": unit, "metric": metric}
    for role in ("builder", "searcher", "streamer"):      # index_meta.cc:80-107
        out[role + "_name"] = att.get(role, {}).get("name")
        out[role + "_params"] = att.get(role, {}).get("params", {})
    return out


def open_flat_file(image, device=0, metric=None):
    """FlatSearcher::load from a dumped flat index FILE (container -> "IndexMeta" / "flat.keys" / "flat.features")"""
    image = bytes(image)
    seg = container_segments(image)
    meta = parse_index_meta(image[seg["IndexMeta"][0]:sum(seg["IndexMeta"])])
    dtype = "fp16" if meta["data_type"] == 1 else "fp32"
    se = HipFlatSearcher(meta["dimension"], metric or meta["metric"] or "SquaredEuclidean", device=device, dtype=dtype)
    ko, ks = seg["flat.keys"]
    fo, fs_ = seg["flat.features"]
    keys = np.frombuffer(image, np.uint64, ks // 8, ko)
    _lib.check(se.load_features(image[fo:fo + fs_], ks // 8, column_major=(meta["major_order"] == 2), keys=keys), "load_features")
    return se


def open_ivf_file(image, device=0, metric=None):
    """IVFSearcher::load from a dumped IVF index FILE: the ivf.* / hc.keys segments, and the centroid rows out of the
    NESTED flat index file that the "ivf.centroid" segment holds (ivf_searcher.cc:43-103)"""
    image = bytes(image)
    seg = container_segments(image)
    meta = parse_index_meta(image[seg["IndexMeta"][0]:sum(seg["IndexMeta"])])
    dtype = "fp16" if meta["data_type"] == 1 else "fp32"
    npdt = np.float16 if dtype == "fp16" else np.float32
    co, cs = seg["ivf.centroid"]
    nested = image[co:co + cs]
    cseg = container_segments(nested)
    cmeta = parse_index_meta(nested[cseg["IndexMeta"][0]:sum(cseg["IndexMeta"])])
    ck = np.frombuffer(nested, np.uint64, cseg["flat.keys"][1] // 8, cseg["flat.keys"][0])
    nlist, dim = ck.size, meta["dimension"]
    feat = np.frombuffer(nested, npdt, nlist * dim, cseg["flat.features"][0]).copy()
    rows = np.empty((nlist, dim), npdt)
    if cmeta["major_order"] == 2:                       # column-major centroid index: full 32-row blocks are transposed
        full = nlist // 32 * 32
        rows[:full] = feat[:full * dim].reshape(full // 32, dim, 32).transpose(0, 2, 1).reshape(full, dim)
        rows[full:] = feat[full * dim:].reshape(nlist - full, dim)
    else:
        rows[:] = feat.reshape(nlist, dim)
    cent = np.empty_like(rows)
    cent[ck.astype(np.int64)] = rows                    # row i of the centroid index holds centroid ck[i]
    se = HipIVFSearcher(dim, metric or meta["metric"] or "SquaredEuclidean", device=device, dtype=dtype)

    def payload(name):
        o, sz = seg[name]
        return image[o:o + sz]
    _lib.check(se.load_segments(payload("ivf.inverted_header"), payload("ivf.inverted_meta"), payload("ivf.inverted_body"),
                                payload("hc.keys"), cent), "load_segments")
    return se


class Gate:
    """zvec_hip_gate_t: contexts sharing it take turns on their dominant scan kernel (pipelining consecutive batches)"""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().zvec_hip_gate_create(int(device), C.byref(self._h)), "zvec_hip_gate_create")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().zvec_hip_gate_destroy(self._h)
                self._h = None
        except Exception:
            pass


class HipShardedIndex:
    """one index over several GPUs inside ONE process (zvec_hip_shards_*): the in-process counterpart of the
    one-rank-per-GPU sharding in zvec_amd/dist.py, for embedded callers like zvec itself.  kind: "flat" or "ivf";
    devices: HIP device ordinals, repeats allowed (several shards on one GPU)."""

    def __init__(self, kind, dim, metric=METRIC_L2, devices=(0,), dtype="fp32"):
        if isinstance(metric, str):
            metric = metric_from_name(metric)
        self.kind = {"flat": 0, "ivf": 1}[kind]
        self.dim = int(dim)
        self.dtype, self.np_dtype = _dtype_of(dtype)
        self.ndev = len(devices)
        dev = (C.c_int * self.ndev)(*[int(d) for d in devices])
        self._h = C.c_void_p()
        _lib.check(_lib.lib().zvec_hip_shards_create(self.dim, self.dtype, metric, self.kind, dev, self.ndev, C.byref(self._h)),
                   "zvec_hip_shards_create")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().zvec_hip_shards_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def counts(self):
        total = C.c_uint64(0)
        per = np.zeros(self.ndev, np.uint64)
        _lib.check(_lib.lib().zvec_hip_shards_count(self._h, C.byref(total), _np_ptr(per)), "zvec_hip_shards_count")
        return int(total.value), per

    def append(self, vecs, keys=None):
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        return _lib.lib().zvec_hip_shards_flat_append(self._h, _np_ptr(vecs), vecs.shape[0], _np_ptr(k))

    def build(self, vecs, nlist, keys=None, kmeans_iters=10, sample_per_list=256, seed=20260320):
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        return _lib.lib().zvec_hip_shards_ivf_build(self._h, _np_ptr(vecs), vecs.shape[0], _np_ptr(k), nlist, kmeans_iters,
                                                    sample_per_list, seed)

    def load(self, centroids, list_offsets, vecs, keys=None):
        centroids = np.ascontiguousarray(centroids, self.np_dtype)
        lo = np.ascontiguousarray(list_offsets, np.uint64)
        vecs = np.ascontiguousarray(vecs, self.np_dtype)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        return _lib.lib().zvec_hip_shards_ivf_load(self._h, _np_ptr(centroids), centroids.shape[0], _np_ptr(lo), _np_ptr(vecs),
                                                   _np_ptr(k))

    def load_segments(self, inverted_header, inverted_meta, inverted_body, keys, centroids):
        cent = np.ascontiguousarray(centroids, self.np_dtype)
        hb, mb, bb, kb = bytes(inverted_header), bytes(inverted_meta), bytes(inverted_body), bytes(keys)
        return _lib.lib().zvec_hip_shards_ivf_load_segments(self._h, hb, len(hb), mb, len(mb), bb, len(bb), kb, len(kb), _np_ptr(cent))

    def load_features(self, features, count, column_major=False, batch_size=32, keys=None):
        fb = bytes(features)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
        return _lib.lib().zvec_hip_shards_flat_load_features(self._h, fb, len(fb), int(count), int(bool(column_major)),
                                                             int(batch_size), _np_ptr(k))

    def search_by_ids(self, queries, ids_per_query, topk, threshold=FLT_MAX, exclude=None):
        """search_bf_by_p_keys_impl over the shards: ids are global storage positions."""
        q = np.ascontiguousarray(queries, self.np_dtype)
        count = q.shape[0]
        offs = np.zeros(count + 1, np.uint32)
        offs[1:] = np.cumsum([len(x) for x in ids_per_query])
        ids = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint64) for x in ids_per_query] + [np.zeros(1, np.uint64)]))
        ex = None if exclude is None else np.ascontiguousarray(exclude, np.uint64)
        keys = np.empty((count, topk), np.uint64)
        scores = np.empty((count, topk), np.float32)
        counts = np.empty(count, np.uint32)
        _lib.check(_lib.lib().zvec_hip_shards_flat_search_by_ids(self._h, _np_ptr(q), count, _np_ptr(ids), _np_ptr(offs), int(topk),
                                                                 float(threshold), _np_ptr(ex), _np_ptr(keys), _np_ptr(scores),
                                                                 _np_ptr(counts)), "zvec_hip_shards_flat_search_by_ids")
        return keys, scores, counts

    def get_vectors(self, positions):
        pos = np.ascontiguousarray(positions, np.uint64)
        out = np.empty((pos.size, self.dim), self.np_dtype)
        rc = _lib.lib().zvec_hip_shards_flat_get_vectors(self._h, _np_ptr(pos), pos.size, _np_ptr(out))
        return rc, out

    def deal_coarse(self, enable):
        """IVF: deal the coarse pass over the shards (zvec_hip_shards_deal_coarse; off by default)"""
        return _lib.lib().zvec_hip_shards_deal_coarse(self._h, int(bool(enable)))

    def search(self, queries, topk, nprobe=1, max_scan=0xffffffff, threshold=FLT_MAX, exclude=None):
        q = np.ascontiguousarray(queries, self.np_dtype)
        count = q.shape[0]
        keys = np.zeros((count, topk), np.uint64)
        scores = np.zeros((count, topk), np.float32)
        counts = np.zeros(count, np.uint32)
        ex = None if exclude is None else np.ascontiguousarray(exclude, np.uint64)
        rc = _lib.lib().zvec_hip_shards_search(self._h, _np_ptr(q), count, topk, threshold, nprobe, max_scan, _np_ptr(ex),
                                               _np_ptr(keys), _np_ptr(scores), _np_ptr(counts))
        _lib.check(rc, "zvec_hip_shards_search")
        return keys, scores, counts


def normalize_score(metric, score):
    """what core_interface::Index::_dense_search applies ABOVE boundary B (index.cc:624-630): metric->normalize() —
    InnerProductMetric::normalize negates the kernel's minus-inner-product back to +ip (inner_product_metric.cc:377-384);
    SquaredEuclidean and Cosine ("1 - ip" already) have no normalize step."""
    if isinstance(metric, str):
        metric = metric_from_name(metric)
    return -score if metric == METRIC_IP else score


def ivf_probe_params(nlist, n, scan_ratio, brute_force_threshold):
    """IVFSearcherContext::update (ivf_searcher_context.h:61-79) in the reference's float arithmetic:
    nprobe = max(round(nlist * scan_ratio), 1) (std::round: half away from zero),
    max_scan_count = max(brute_force_threshold, ceil(N * scan_ratio))."""
    nprobe = max(int(np.floor(np.float32(np.float32(nlist) * np.float32(scan_ratio)) + np.float32(0.5))), 1)
    max_scan = int(np.ceil(np.float32(np.float32(n) * np.float32(scan_ratio))))
    return nprobe, max(int(brute_force_threshold), max_scan)


def shard_map(list_sizes, nshards):
    """list -> shard map of zvec_hip_ivf_shard_map (pure host arithmetic): (owner[nlist], rows per shard)."""
    sz = np.ascontiguousarray(list_sizes, np.uint32)
    owner = np.zeros(sz.size, np.uint32)
    rows = np.zeros(nshards, np.uint64)
    _lib.check(_lib.lib().zvec_hip_ivf_shard_map(_np_ptr(sz), sz.size, int(nshards), _np_ptr(owner), _np_ptr(rows)),
               "zvec_hip_ivf_shard_map")
    return owner, rows


def merge_topk(ctx, keys, scores, counts, topk):
    """host form of the shard merge: inputs [nparts][count][topk] / [nparts][count]."""
    keys = np.ascontiguousarray(keys, np.uint64)
    scores = np.ascontiguousarray(scores, np.float32)
    counts = np.ascontiguousarray(counts, np.uint32)
    nparts, count = counts.shape
    ok = np.zeros((count, topk), np.uint64)
    os_ = np.zeros((count, topk), np.float32)
    oc = np.zeros(count, np.uint32)
    _lib.check(_lib.lib().zvec_hip_merge_topk(ctx._h, _np_ptr(keys), _np_ptr(scores), _np_ptr(counts), nparts,
                                              count, topk, _np_ptr(ok), _np_ptr(os_), _np_ptr(oc)),
               "zvec_hip_merge_topk")
    return ok, os_, oc
