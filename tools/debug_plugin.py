"""diagnostics for the plugin-in-framework tests (run on the GPU box)"""
import sys, os, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import refcore as R
import zvec_amd as zv
from zvec_amd.index import container_segments
R.load_plugin()
for dt in (np.float16,):
    rng = np.random.default_rng(14)
    n, dim, nlist = 20000, 48, 64
    means = rng.standard_normal((nlist, dim)).astype(np.float32) * 2
    base = (means[rng.integers(0, nlist, n)] + rng.standard_normal((n, dim)).astype(np.float32)).astype(dt)
    keys = (rng.permutation(3 * n)[:n]).astype(np.uint64)
    bp = {"proxima.ivf.builder.centroid_count": str(nlist), "proxima.ivf.builder.thread_count": 4}
    R.build("IVFBuilder", base, "SquaredEuclidean", "built_a", keys=keys, params=bp)
    R.build("IVFBuilder", base[: n // 2], "SquaredEuclidean", "built_b", keys=keys[: n // 2], params=bp)
    for nm in ("built_a", "built_b"):
        img = R.mem_get(nm).tobytes()
        seg = container_segments(img)
        ho, hs = seg["ivf.inverted_header"]
        print(nm, "header", struct.unpack_from("<IIQIIIII", img, ho))
        mo, ms = seg["ivf.inverted_meta"]
        seen = 0
        for l in range(nlist):
            off, bc, vc, ido = struct.unpack_from("<QIII", img, mo + l * 40)
            if ido != seen or vc == 0:
                print("  list", l, off, bc, vc, ido, "seen", seen)
            seen += vc
        try:
            se = zv.open_ivf_file(img)
            print(nm, "python loader ok", se.info())
        except Exception as e:
            print(nm, "python loader:", repr(e))
        try:
            hip = R.Runner.searcher("HipIVFSearcher", nm, dim, dt, params={"proxima.ivf.searcher.scan_ratio": 0.1})
            print(nm, "plugin ok", hip.count())
        except Exception as e:
            print(nm, "plugin:", repr(e))
