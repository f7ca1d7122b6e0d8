"""TEST INFRASTRUCTURE ONLY — ctypes face of oracle/_ref/libzvec_ref_core.so (oracle/ref_core_shim.cc): the reference's
whole core library compiled in place, its operators driven BY REGISTERED NAME through the reference's own factories.  The
same calls drive the plugin's operators once `load_plugin()` has brought plugin/build/libzvec_hip_plugin.so in through the
reference's IndexPluginBroker.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CORE = os.path.join(HERE, "_ref", "libzvec_ref_core.so")
PLUGIN = os.path.join(os.path.dirname(HERE), "plugin", "build", "libzvec_hip_plugin.so")

_lib = None
_plugin_loaded = False


def available():
    return os.path.exists(CORE)


def lib():
    global _lib
    if _lib is None:
        # RTLD_GLOBAL: the plugin (dlopen'ed later by the reference's broker) resolves the framework's symbols here
        L = C.CDLL(CORE, mode=C.RTLD_GLOBAL)
        vp, u64, u32, i32, f32, cp = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_char_p
        P = C.POINTER
        sig = {
            "zref_core_abi": (i32, []),
            "zref_load_plugin": (i32, [cp, cp, u64]),
            "zref_has": (i32, [cp, cp]),
            "zref_mem_put": (i32, [cp, vp, u64]),
            "zref_mem_size": (i32, [cp, P(u64)]),
            "zref_mem_get": (i32, [cp, vp, u64]),
            "zref_mem_remove": (i32, [cp]),
            "zref_build": (i32, [cp, i32, u32, cp, i32, cp, vp, vp, u64, cp, cp, P(C.c_double)]),
            "zref_searcher_open": (vp, [cp, cp, cp, cp, P(i32)]),
            "zref_streamer_open": (vp, [cp, i32, u32, cp, cp, cp, cp, i32, P(i32)]),
            "zref_streamer_add": (i32, [vp, i32, u32, vp, vp, u64, i32]),
            "zref_streamer_flush": (i32, [vp]),
            "zref_streamer_dump": (i32, [vp, cp, cp]),
            "zref_runner_close": (i32, [vp]),
            "zref_runner_count": (u64, [vp]),
            "zref_runner_get_vector": (i32, [vp, u64, vp, u32]),
            "zref_runner_walk": (C.c_int64, [vp, vp, vp, u32, u64]),
            "zref_ctx_create": (vp, [vp]),
            "zref_ctx_destroy": (None, [vp]),
            "zref_ctx_set_topk": (None, [vp, u32]),
            "zref_ctx_set_threshold": (None, [vp, i32, f32]),
            "zref_ctx_set_fetch_vector": (None, [vp, i32]),
            "zref_ctx_update": (i32, [vp, cp]),
            "zref_ctx_set_filter": (None, [vp, vp, u64]),
            "zref_ctx_set_group": (None, [vp, vp, u64, u32, u32]),
            "zref_search": (i32, [vp, vp, i32, vp, i32, u32, u32, vp, vp]),
            "zref_ctx_result_size": (u32, [vp, u32]),
            "zref_ctx_result": (i32, [vp, u32, vp, vp, vp, vp, u32, P(u32)]),
            "zref_ctx_group_count": (u32, [vp, u32]),
            "zref_ctx_group": (i32, [vp, u32, u32, P(u32), vp, vp, u32]),
            "zref_ivf_searcher_over_rows": (vp, [cp, cp, i32, u32, cp, vp, u32, vp, vp, vp, P(i32)]),
            "zref_build_converted": (i32, [cp, cp, i32, u32, cp, cp, vp, vp, u64, cp]),
            "zref_search_sequence": (i32, [vp, vp, i32, u32, vp, u32, i32, i32, u32, vp, vp, vp]),
            "zref_search_mt": (i32, [vp, i32, vp, i32, u32, u32, u32, cp, u32, vp, vp, vp, P(C.c_double)]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        assert L.zref_core_abi() == 2
        _lib = L
    return _lib


def load_plugin(path=PLUGIN):
    """IndexPluginBroker::emplace(path): dlopen runs the plugin's static registrars (index_factory.h:237-250)."""
    global _plugin_loaded
    if _plugin_loaded:
        return
    # ONE HIP runtime per process: the plugin links zvec_amd/libzvec_hip.so, which must already be loaded the way the package loads
    # it (after torch, whose wheel bundles its own libamdhip64 — see zvec_amd/_lib.py); a C++ host has no such concern
    from zvec_amd import _lib as _product
    _product.lib()
    err = C.create_string_buffer(512)
    rc = lib().zref_load_plugin(path.encode(), err, 512)
    if rc != 0:
        raise RuntimeError("plugin %s: %s" % (path, err.value.decode()))
    _plugin_loaded = True


def has(kind, name):
    return lib().zref_has(kind.encode(), name.encode()) == 1


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _js(params):
    return json.dumps(params).encode() if params else b""


def _dt(a):
    return 1 if a.dtype == np.float16 else 0


def mem_put(name, image):
    image = np.ascontiguousarray(np.frombuffer(image, np.uint8) if not isinstance(image, np.ndarray) else image)
    rc = lib().zref_mem_put(name.encode(), _p(image), image.nbytes)
    assert rc == 0, rc


def mem_get(name):
    n = C.c_uint64()
    assert lib().zref_mem_size(name.encode(), C.byref(n)) == 0
    out = np.empty(n.value, np.uint8)
    assert lib().zref_mem_get(name.encode(), _p(out), out.nbytes) == 0
    return out


def mem_remove(name):
    lib().zref_mem_remove(name.encode())


def build(cls, rows, metric, target, keys=None, params=None, column_major=False, dumper="MemoryDumper"):
    """IndexFactory::CreateBuilder(cls) -> init / train / build / dump; returns the train+build seconds."""
    rows = np.ascontiguousarray(rows)
    keys = None if keys is None else np.ascontiguousarray(keys, np.uint64)
    sec = C.c_double()
    rc = lib().zref_build(cls.encode(), _dt(rows), rows.shape[1], metric.encode(), int(column_major), _js(params), _p(rows),
                          _p(keys), rows.shape[0], dumper.encode(), target.encode(), C.byref(sec))
    if rc != 0:
        raise RuntimeError("%s build: rc %d" % (cls, rc))
    return sec.value


def build_converted(builder, converter, rows, metric, target, keys=None, params=None):
    """Index::Add/Train/Dump for a converted index: the converter (e.g. "CosineFp32Converter", "HalfFloatConverter") transforms the
    rows and names the reformer in its meta (index.cc:111-183), the builder builds and dumps the converted rows."""
    rows = np.ascontiguousarray(rows)
    keys = None if keys is None else np.ascontiguousarray(keys, np.uint64)
    rc = lib().zref_build_converted(builder.encode(), converter.encode(), _dt(rows), rows.shape[1], metric.encode(), _js(params), _p(rows),
                                    _p(keys), rows.shape[0], target.encode())
    if rc != 0:
        raise RuntimeError("%s + %s: rc %d" % (converter, builder, rc))


class Context:
    def __init__(self, runner):
        self.runner = runner
        self.h = lib().zref_ctx_create(runner.h)
        assert self.h, "create_context failed"

    def close(self):
        if self.h:
            lib().zref_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_topk(self, k):
        lib().zref_ctx_set_topk(self.h, k)

    def set_threshold(self, v):
        lib().zref_ctx_set_threshold(self.h, 0 if v is None else 1, 0.0 if v is None else float(v))

    def set_fetch_vector(self, on):
        lib().zref_ctx_set_fetch_vector(self.h, int(on))

    def update(self, params):
        return lib().zref_ctx_update(self.h, _js(params))

    def set_filter(self, exclude_by_key):
        if exclude_by_key is None:
            lib().zref_ctx_set_filter(self.h, None, 0)
        else:
            ex = np.ascontiguousarray(exclude_by_key, np.uint8)
            lib().zref_ctx_set_filter(self.h, _p(ex), ex.size)

    def set_group(self, group_of_key, group_num, group_topk):
        if group_of_key is None:
            lib().zref_ctx_set_group(self.h, None, 0, group_num, group_topk)
        else:
            g = np.ascontiguousarray(group_of_key, np.uint32)
            lib().zref_ctx_set_group(self.h, _p(g), g.size, group_num, group_topk)

    def result(self, qi, elem_dtype=None, dim=0):
        n = lib().zref_ctx_result_size(self.h, qi)
        keys, scores, index = np.empty(n, np.uint64), np.empty(n, np.float32), np.empty(n, np.uint32)
        vec, present = None, C.c_uint32(0)
        if elem_dtype is not None:
            vec = np.zeros((n, dim), elem_dtype)
        lib().zref_ctx_result(self.h, qi, _p(keys), _p(scores), _p(index), _p(vec), 0 if vec is None else vec.strides[0] if n else 0,
                              C.byref(present))
        if vec is not None:
            return keys, scores, index, vec, present.value
        return keys, scores, index

    def groups(self, qi, cap=4096):
        out = []
        for s in range(lib().zref_ctx_group_count(self.h, qi)):
            g = C.c_uint32()
            keys, scores = np.empty(cap, np.uint64), np.empty(cap, np.float32)
            n = lib().zref_ctx_group(self.h, qi, s, C.byref(g), _p(keys), _p(scores), cap)
            assert n >= 0
            out.append((g.value, keys[:n].copy(), scores[:n].copy()))
        return out


class Runner:
    """A searcher or a streamer created by its registered name."""

    def __init__(self, h, dim, dtype):
        self.h, self.dim, self.dtype = h, dim, dtype

    @classmethod
    def searcher(cls, name, target, dim, dtype=np.float32, params=None, storage="MemoryReadStorage"):
        rc = C.c_int()
        h = lib().zref_searcher_open(name.encode(), _js(params), storage.encode(), target.encode(), C.byref(rc))
        if not h:
            raise RuntimeError("%s load: rc %d" % (name, rc.value))
        return cls(h, dim, np.dtype(dtype))

    @classmethod
    def streamer(cls, name, target, dim, metric, dtype=np.float32, params=None, storage="MMapFileStorage", create=True):
        rc = C.c_int()
        h = lib().zref_streamer_open(name.encode(), 1 if np.dtype(dtype) == np.float16 else 0, dim, metric.encode(), _js(params),
                                     storage.encode(), target.encode(), int(create), C.byref(rc))
        if not h:
            raise RuntimeError("%s open: rc %d" % (name, rc.value))
        return cls(h, dim, np.dtype(dtype))

    @classmethod
    def ivf_over_rows(cls, name, centroids, list_offsets, rows, keys, metric, params=None):
        """An IVF searcher over an index given as arrays (centroids, list offsets, rows and keys in list order): the reference's own
        IVFDumper writes every small segment, the body is LENT from `rows` (no 30 GB file).  fp32 / fp16 rows, element size % 32 == 0."""
        centroids = np.ascontiguousarray(centroids, rows.dtype)
        lo = np.ascontiguousarray(list_offsets, np.uint64)
        keys = np.ascontiguousarray(keys, np.uint64)
        assert rows.flags.c_contiguous
        rc = C.c_int()
        h = lib().zref_ivf_searcher_over_rows(name.encode(), _js(params), _dt(rows), rows.shape[1], metric.encode(), _p(centroids),
                                              centroids.shape[0], _p(lo), _p(rows), _p(keys), C.byref(rc))
        if not h:
            raise RuntimeError("%s over rows: rc %d" % (name, rc.value))
        r = cls(h, rows.shape[1], rows.dtype)
        r._keep = (centroids, lo, rows, keys)           # the storage lends these
        return r

    def close(self):
        if self.h:
            rc = lib().zref_runner_close(self.h)
            self.h = None
            return rc
        return 0

    def __del__(self):
        self.close()

    def create_context(self):
        return Context(self)

    def add(self, keys, rows, with_id=False):
        rows = np.ascontiguousarray(rows, self.dtype)
        keys = np.ascontiguousarray(keys, np.uint64)
        return lib().zref_streamer_add(self.h, _dt(rows), self.dim, _p(keys), _p(rows), rows.shape[0], int(with_id))

    def flush(self):
        return lib().zref_streamer_flush(self.h)

    def dump(self, target, dumper="MemoryDumper"):
        return lib().zref_streamer_dump(self.h, dumper.encode(), target.encode())

    def count(self):
        return lib().zref_runner_count(self.h)

    def get_vector(self, key):
        out = np.empty(self.dim, self.dtype)
        rc = lib().zref_runner_get_vector(self.h, key, _p(out), out.nbytes)
        return rc, out

    def walk(self, elem_cols=None):
        cols = elem_cols or self.dim
        n = self.count()
        keys, rows = np.empty(n, np.uint64), np.empty((n, cols), self.dtype)
        got = lib().zref_runner_walk(self.h, _p(keys), _p(rows), cols * self.dtype.itemsize, n)
        assert got == n, (got, n)
        return keys, rows

    def search(self, ctx, q, mode=0, p_keys=None):
        """mode 0 search_impl, 1 search_bf_impl, 2 search_bf_by_p_keys_impl; q [count][cols] in the index's dtype."""
        q = np.ascontiguousarray(q, self.dtype)
        count = q.shape[0]
        pk = po = None
        if mode == 2:
            po = np.zeros(count + 1, np.uint32)
            po[1:] = np.cumsum([len(x) for x in p_keys])
            pk = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint64) for x in p_keys]) if po[-1] else np.zeros(1, np.uint64))
        return lib().zref_search(self.h, ctx.h, mode, _p(q), _dt(q), q.shape[1], count, _p(pk), _p(po))

    def search_lists(self, ctx, q, mode=0, p_keys=None):
        rc = self.search(ctx, q, mode, p_keys)
        if rc != 0:
            return rc, None
        return 0, [ctx.result(i) for i in range(len(q))]

    def search_lists_single(self, ctx, q, mode=0, p_keys=None):
        """one call per query (count = 1): how the product calls boundary B (index.cc:617) — and the only correct way to batch the
        reference's FlatStreamer, whose count > 1 loop reuses one heap across queries (flat_streamer.cc:330-342)."""
        out = []
        for i in range(len(q)):
            rc = self.search(ctx, q[i:i + 1], mode, None if p_keys is None else p_keys[i:i + 1])
            if rc != 0:
                return rc, None
            out.append(ctx.result(0))
        return 0, out

    def search_sequence(self, ctx, raw_queries, topk, batched, linear=False):
        """boundary A's sequence around this runner with the index's own reformer / metric: batched=False = Index::_dense_search per
        query (index.cc:596-652), batched=True = Index::SearchBatch of patches/boundary_a.diff.  raw_queries: unconverted rows."""
        q = np.ascontiguousarray(raw_queries)
        count = q.shape[0]
        keys = np.zeros((count, topk), np.uint64)
        scores = np.full((count, topk), np.inf, np.float32)
        counts = np.zeros(count, np.uint32)
        rc = lib().zref_search_sequence(self.h, ctx.h, _dt(q), q.shape[1], _p(q), count, int(batched), int(linear), topk, _p(keys), _p(scores),
                                        _p(counts))
        if rc != 0:
            raise RuntimeError("search_sequence rc %d" % rc)
        return keys, scores, counts

    def search_mt(self, q, topk, threads, mode=0, ctx_params=None):
        """Queries dealt over `threads` threads, one query per call (tools/core/bench.cc:145-245); returns arrays + wall seconds."""
        q = np.ascontiguousarray(q, self.dtype)
        count = q.shape[0]
        keys = np.zeros((count, topk), np.uint64)
        scores = np.full((count, topk), np.inf, np.float32)
        counts = np.zeros(count, np.uint32)
        sec = C.c_double()
        rc = lib().zref_search_mt(self.h, mode, _p(q), _dt(q), q.shape[1], count, topk, _js(ctx_params), threads, _p(keys), _p(scores),
                                  _p(counts), C.byref(sec))
        if rc != 0:
            raise RuntimeError("search_mt rc %d" % rc)
        return keys, scores, counts, sec.value
