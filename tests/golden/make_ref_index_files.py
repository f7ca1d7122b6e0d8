"""Generates tests/golden/ref_index_files.npz: index FILE images dumped by the REFERENCE's own writers.

Needs oracle/_ref/libzvec_ref_core.so = the reference's core library (FlatBuilder<32>, IVFDumper, MemoryDumper + IndexPacker,
IndexMeta serialisation, ...) compiled in place by `make -C oracle ref_core`; the doors used here are oracle/ref_format_shim.cc
(no stand-ins, nothing copied).  Each case stores its inputs and the byte image the reference wrote; the images pin the product's
container parser (zvec_hip_container_segments) and segment loaders (zvec_hip_flat_load_features,
zvec_hip_ivf_load_segments) — SURVEY §8(f) next-2 — and the restated test-side writers in tests/ivf_format.py.
Run:  python tests/golden/make_ref_index_files.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzvec_ref_core.so"))
rng = np.random.default_rng(20260321)
buf = np.zeros(8 << 20, np.uint8)
out = {}


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def dump_flat(name, dtype, n, dim, column_major, explicit_keys):
    base = rng.integers(-8, 9, (n, dim)).astype(dtype)
    keys = (rng.permutation(5 * n)[:n].astype(np.uint64) if explicit_keys else np.arange(n, dtype=np.uint64))
    sz = C.c_uint64(0)
    rc = L.zref_dump_flat_index(int(dtype == np.float16), dim, int(column_major), b"InnerProduct", ptr(base), ptr(keys), C.c_uint64(n),
                                ptr(buf), C.c_uint64(buf.size), C.byref(sz))
    assert rc == 0, (name, rc)
    out[name + "_base"] = base.view(np.uint16) if dtype == np.float16 else base
    out[name + "_keys"] = keys
    out[name + "_meta"] = np.array([int(dtype == np.float16), n, dim, int(column_major)], np.int64)
    out[name + "_image"] = buf[:sz.value].copy()


def dump_ivf(name, dtype, sizes, dim, column_major, centroid_column_major):
    nlist, n = len(sizes), int(sum(sizes))
    base = rng.integers(-8, 9, (n, dim)).astype(dtype)
    keys = rng.permutation(7 * n)[:n].astype(np.uint64)
    cent = rng.integers(-8, 9, (nlist, dim)).astype(dtype)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    sz = C.c_uint64(0)
    rc = L.zref_dump_ivf_index(int(dtype == np.float16), dim, int(column_major), int(centroid_column_major), b"InnerProduct", ptr(cent),
                               nlist, ptr(offs), ptr(base), ptr(keys), ptr(buf), C.c_uint64(buf.size), C.byref(sz))
    assert rc == 0, (name, rc)
    f16 = dtype == np.float16
    out[name + "_base"] = base.view(np.uint16) if f16 else base
    out[name + "_cent"] = cent.view(np.uint16) if f16 else cent
    out[name + "_keys"] = keys
    out[name + "_offs"] = offs
    out[name + "_meta"] = np.array([int(f16), n, dim, int(column_major), int(centroid_column_major), nlist], np.int64)
    out[name + "_image"] = buf[:sz.value].copy()


dump_flat("flat0", np.float32, 70, 12, False, True)
dump_flat("flat1", np.float32, 100, 20, True, True)
dump_flat("flat2", np.float16, 64, 10, True, False)
dump_flat("flat3", np.float16, 31, 7, False, True)
dump_ivf("ivf0", np.float32, [10, 0, 35, 15, 64, 1, 33], 24, False, False)
dump_ivf("ivf1", np.float32, [40, 32, 7, 0, 96], 16, True, True)
dump_ivf("ivf2", np.float16, [33, 64, 5, 70], 10, True, False)
dump_ivf("ivf3", np.float32, [20, 0, 33, 0, 0], 8, False, False)      # TRAILING empty lists: their meta stays zeroed (ivf_dumper.cc:284-291)
out["cases"] = np.array(["flat0", "flat1", "flat2", "flat3", "ivf0", "ivf1", "ivf2", "ivf3"])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_index_files.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
