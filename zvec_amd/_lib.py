"""ctypes binding of include/zvec_hip.h (the C ABI of the gfx950 scan core).

There is no CPU fallback: if the shared library cannot be loaded, or no HIP device is present when
an index is created, the call raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os

from . import build as _build

_f32p = C.c_void_p
_u64p = C.c_void_p
_u32p = C.c_void_p
_h = C.c_void_p

class DocFilterDesc(C.Structure):
    """zvec_hip_doc_filter_t (include/zvec_hip.h): the composite document filter's three optional terms."""
    _fields_ = [("delete_bitmap", C.c_void_p), ("delete_bytes", C.c_uint64), ("delete_kind", C.c_int32),
                ("reserved_", C.c_int32), ("invert_bitmap", C.c_void_p), ("invert_bytes", C.c_uint64),
                ("forward_bits", C.c_void_p), ("forward_len", C.c_uint64)]


class Segment(C.Structure):
    """zvec_hip_segment_t: one segment of a dumped index file"""
    _fields_ = [("id", C.c_char * 64), ("offset", C.c_uint64), ("size", C.c_uint64), ("padding", C.c_uint64),
                ("crc", C.c_uint32), ("reserved_", C.c_uint32)]


ROARING_NONE, ROARING_32, ROARING_64MAP, ROARING_FILE = 0, 1, 2, 3

# every symbol include/zvec_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "zvec_hip_abi_version": (C.c_int, []),
    "zvec_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "zvec_hip_error_string": (C.c_char_p, [C.c_int]),
    "zvec_hip_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "zvec_hip_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "zvec_hip_calibrate": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "zvec_hip_host_alloc": (C.c_int, [C.c_uint64, C.POINTER(C.c_void_p)]),
    "zvec_hip_host_free": (C.c_int, [C.c_void_p]),
    "zvec_hip_ctx_create": (C.c_int, [C.c_int, C.POINTER(_h)]),
    "zvec_hip_ctx_destroy": (C.c_int, [_h]),
    "zvec_hip_ctx_synchronize": (C.c_int, [_h]),
    "zvec_hip_ctx_set_stream": (C.c_int, [_h, C.c_void_p]),
    "zvec_hip_gate_create": (C.c_int, [C.c_int, C.POINTER(_h)]),
    "zvec_hip_gate_destroy": (C.c_int, [_h]),
    "zvec_hip_ctx_set_gate": (C.c_int, [_h, _h]),
    "zvec_hip_flat_create": (C.c_int, [C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(_h)]),
    "zvec_hip_flat_destroy": (C.c_int, [_h]),
    "zvec_hip_flat_reserve": (C.c_int, [_h, C.c_uint64]),
    "zvec_hip_flat_append": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u64p]),
    "zvec_hip_flat_put": (C.c_int, [_h, _u32p, C.c_uint64, C.c_void_p, _u64p]),
    "zvec_hip_flat_holes": (C.c_int, [_h, _u64p]),
    "zvec_hip_flat_append_dev": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u64p, C.c_void_p]),
    "zvec_hip_flat_count": (C.c_int, [_h, C.POINTER(C.c_uint64)]),
    "zvec_hip_flat_get_vector": (C.c_int, [_h, C.c_uint64, C.c_void_p]),
    "zvec_hip_flat_get_vectors": (C.c_int, [_h, _u64p, C.c_uint64, C.c_void_p]),
    "zvec_hip_flat_search": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, _u64p,
                                       _u64p, _f32p, _u32p]),
    "zvec_hip_flat_search_by_ids": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, _u32p, _u32p, C.c_uint32, C.c_float,
                                              _u64p, _u64p, _f32p, _u32p]),
    "zvec_hip_flat_search_dev": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float,
                                           _u64p, _u64p, _f32p, _u32p, C.c_void_p]),
    "zvec_hip_flat_search_grouped": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                               _u64p, _u32p, _u32p, _u64p, _f32p, _u32p]),
    "zvec_hip_flat_search_grouped_by_ids": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, C.c_uint32,
                                                      C.c_uint32, C.c_float, _u64p, _u32p, _u32p, _u64p, _f32p, _u32p]),
    "zvec_hip_flat_batch_distance": (C.c_int, [_h, _h, C.c_void_p, _u32p, C.c_uint32, _f32p]),
    "zvec_hip_ivf_create": (C.c_int, [C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(_h)]),
    "zvec_hip_ivf_destroy": (C.c_int, [_h]),
    "zvec_hip_ivf_load": (C.c_int, [_h, C.c_void_p, C.c_uint32, _u64p, C.c_void_p, _u64p]),
    "zvec_hip_flat_load_features": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, _u64p]),
    "zvec_hip_flat_load_blocks": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]),
    "zvec_hip_ivf_coarse_dev": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zvec_hip_ivf_search_probes_dev": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zvec_hip_ivf_set_coarse_space": (C.c_int, [_h, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32]),
    "zvec_hip_ivf_search_coarse": (C.c_int, [_h, _h, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32,
                                             _u64p, _u64p, _f32p, _u32p]),
    "zvec_hip_ivf_load_segments": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_uint64, C.c_void_p]),
    "zvec_hip_ivf_build_dev": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u64p, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint64, C.c_void_p]),
    "zvec_hip_ivf_build": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u64p, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_uint64]),
    "zvec_hip_ivf_info": (C.c_int, [_h, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "zvec_hip_ivf_export": (C.c_int, [_h, C.c_void_p, _u64p, _u64p]),
    "zvec_hip_ivf_get_vector": (C.c_int, [_h, C.c_uint64, C.c_void_p]),
    "zvec_hip_ivf_get_vectors": (C.c_int, [_h, _u64p, C.c_uint64, C.c_void_p]),
    "zvec_hip_ivf_search": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float,
                                      C.c_uint32, C.c_uint32, _u64p, _u64p, _f32p, _u32p]),
    "zvec_hip_ivf_search_dev": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float,
                                          C.c_uint32, C.c_uint32, _u64p, _u64p, _f32p, _u32p,
                                          C.c_void_p]),
    "zvec_hip_flat_set_shadow": (C.c_int, [_h, C.c_int, C.c_uint32]),
    "zvec_hip_flat_shadow_info": (C.c_int, [_h, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "zvec_hip_flat_shadow_certify": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, _u64p, _u64p, _f32p, _u32p, C.c_void_p,
                                               C.POINTER(C.c_uint32)]),
    "zvec_hip_ivf_shadow_width": (C.c_int, [_h, C.c_uint32, C.POINTER(C.c_uint32)]),
    "zvec_hip_flat_shadow_width": (C.c_int, [_h, C.c_uint32, C.POINTER(C.c_uint32)]),
    "zvec_hip_ivf_set_shadow": (C.c_int, [_h, C.c_int, C.c_uint32]),
    "zvec_hip_ivf_shadow_info": (C.c_int, [_h, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "zvec_hip_ivf_shadow_certify": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _u64p, _f32p,
                                              _u32p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "zvec_hip_ivf_search_bf": (C.c_int, [_h, _h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, _u64p,
                                         _u64p, _f32p, _u32p]),
    "zvec_hip_ivf_keep_shard": (C.c_int, [_h, C.c_uint32, C.c_uint32]),
    "zvec_hip_ivf_shard_map": (C.c_int, [_u32p, C.c_uint32, C.c_uint32, _u32p, _u64p]),
    "zvec_hip_ivf_list_owners": (C.c_int, [_h, _u32p]),
    "zvec_hip_ivf_train_dev": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]),
    "zvec_hip_ivf_set_centroids": (C.c_int, [_h, C.c_void_p, C.c_uint32]),
    "zvec_hip_ivf_get_centroids": (C.c_int, [_h, C.c_void_p, C.POINTER(C.c_uint32)]),
    "zvec_hip_ivf_label_dev": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u32p, C.c_void_p]),
    "zvec_hip_ivf_begin_lists": (C.c_int, [_h, _u32p]),
    "zvec_hip_ivf_add_dev": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u32p, _u64p, C.c_uint64, C.c_void_p]),
    "zvec_hip_ivf_end_lists": (C.c_int, [_h]),
    "zvec_hip_ivf_last_stats": (C.c_int, [_h, _h, C.c_uint32, _u32p, _u32p]),
    "zvec_hip_merge_topk": (C.c_int, [_h, _u64p, _f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                      _u64p, _f32p, _u32p]),
    "zvec_hip_merge_topk_dev": (C.c_int, [_h, _u64p, _f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          _u64p, _f32p, _u32p, C.c_void_p]),
    "zvec_hip_packed_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32]),
    "zvec_hip_merge_topk_packed_dev": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 _u64p, _f32p, _u32p, C.c_void_p]),
    "zvec_hip_shards_create": (C.c_int, [C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_uint32, C.POINTER(_h)]),
    "zvec_hip_shards_destroy": (C.c_int, [_h]),
    "zvec_hip_shards_count": (C.c_int, [_h, C.POINTER(C.c_uint64), _u64p]),
    "zvec_hip_shards_flat_append": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u64p]),
    "zvec_hip_shards_ivf_build": (C.c_int, [_h, C.c_void_p, C.c_uint64, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]),
    "zvec_hip_shards_ivf_load": (C.c_int, [_h, C.c_void_p, C.c_uint32, _u64p, C.c_void_p, _u64p]),
    "zvec_hip_shards_ivf_load_segments": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                    C.c_void_p, C.c_uint64, C.c_void_p]),
    "zvec_hip_shards_flat_load_features": (C.c_int, [_h, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, _u64p]),
    "zvec_hip_shards_flat_search_by_ids": (C.c_int, [_h, C.c_void_p, C.c_uint32, _u64p, _u32p, C.c_uint32, C.c_float, _u64p,
                                                     _u64p, _f32p, _u32p]),
    "zvec_hip_shards_flat_get_vectors": (C.c_int, [_h, _u64p, C.c_uint64, C.c_void_p]),
    "zvec_hip_shards_deal_coarse": (C.c_int, [_h, C.c_int]),
    "zvec_hip_shards_search": (C.c_int, [_h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, _u64p,
                                         _u64p, _f32p, _u32p]),
    "zvec_hip_flat_build_filter": (C.c_int, [_h, _h, C.POINTER(DocFilterDesc), _u64p, C.c_int, C.c_void_p]),
    "zvec_hip_ivf_build_filter": (C.c_int, [_h, _h, C.POINTER(DocFilterDesc), _u64p, C.c_int, C.c_void_p]),
    "zvec_hip_reform_queries_dev": (C.c_int, [_h, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "zvec_hip_container_segments": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Segment), C.c_uint32, C.POINTER(C.c_uint32)]),
    "zvec_hip_crc32c": (C.c_uint32, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "zvec_hip_ctx_profile": (C.c_int, [_h, C.c_int]),
    "zvec_hip_ctx_profile_read": (C.c_int, [_h, C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
}

_LIB = None


class ZvecHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().zvec_hip_error_string(code)
        super().__init__("%s failed: %d (%s)" % (where, code, msg.decode() if msg else "?"))


def library_path():
    return _build.OUT


def lib():
    """Load (building first if the sources are newer) the HIP shared library. Raises if absent."""
    global _LIB
    if _LIB is None:
        # ONE HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (same SONAME as the system one this
        # library links).  Loaded after torch, this library binds to torch's copy and both share streams and
        # allocations; loaded BEFORE torch, the process would end up with two runtimes (torch then intermittently sees
        # "No HIP GPUs").  So when torch is installed it is imported first.  Not needed for C/C++ callers.
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 - torch absent or unusable: the library then owns the only HIP runtime
            pass
        # ZVEC_HIP_LIBRARY: another build of the same library (kernel A/B experiments on one GPU box)
        path = os.environ.get("ZVEC_HIP_LIBRARY") or _build.build()
        if not os.path.exists(path):
            raise RuntimeError("zvec_amd: %s missing — the HIP extension is required (no CPU fallback)" % path)
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(code, where):
    if code != 0:
        raise ZvecHipError(code, where)
