"""Open time of a persisted mutable flat index: the reference's FlatStreamer writes N x d rows into an
MMapFileStorage file; HipFlatStreamer::open brings them into HBM — block runs read straight from the storage segments
(bulk_open: one strided copy + one pack launch per segment) — against the reference streamer's own provider walk."""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import refcore as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
R.load_plugin()
tmp = tempfile.mkdtemp(prefix="zvec_open_", dir=os.environ.get("ZVEC_TMP", "/tmp"))
try:
    path = os.path.join(tmp, "idx")
    rng = np.random.default_rng(1)
    st = R.Runner.streamer("FlatStreamer", path, dim, "SquaredEuclidean")
    t = time.time()
    step = 50_000
    probe = None
    for o in range(0, n, step):
        rows = rng.standard_normal((min(step, n - o), dim)).astype(np.float32)
        if probe is None:
            probe = rows[:4].copy()
        assert st.add(np.arange(o, o + len(rows), dtype=np.uint64), rows) == 0
    assert st.flush() == 0 and st.close() == 0
    print("reference FlatStreamer wrote %d x %d rows (%.2f GB) in %.1fs" % (n, dim, n * dim * 4 / 1e9, time.time() - t), flush=True)
    t = time.time()
    ref = R.Runner.streamer("FlatStreamer", path, dim, "SquaredEuclidean", create=False)
    t_ref_open = time.time() - t
    t = time.time()
    keys, rows = ref.walk()
    t_walk = time.time() - t
    ref.close()
    del rows
    t = time.time()
    hip = R.Runner.streamer("HipFlatStreamer", path, dim, "SquaredEuclidean", create=False)
    t_hip = time.time() - t
    assert hip.count() == n
    ctx = hip.create_context()
    ctx.set_topk(3)
    rc, lists = hip.search_lists(ctx, probe)
    assert rc == 0 and [int(l[0][0]) for l in lists] == [0, 1, 2, 3] and all(l[1][0] == 0 for l in lists)
    print("open: reference FlatStreamer::open %.2fs (+ provider walk of every row %.2fs); HipFlatStreamer::open (reference open + bulk "
          "load into HBM) %.2fs  -> bulk load ~%.2fs = %.2f GB/s" % (t_ref_open, t_walk, t_hip, t_hip - t_ref_open,
                                                                      n * dim * 4 / 1e9 / max(t_hip - t_ref_open, 1e-9)), flush=True)
    ctx.close()
    hip.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
