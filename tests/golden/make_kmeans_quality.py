"""Generates tests/golden/kmeans_quality.json: what the REFERENCE's own trainer achieves on a seeded corpus.

The reference's IVFBuilder (ivf_builder.cc:212-403: StratifiedClusterTrainer + OptKmeansCluster, chosen at ivf_builder.cc:524-531)
seeds its k-means from std::random_device (src/ailego/algorithm/kmeans.h:96), so centroids cannot be compared bit for bit; what can
be compared is the QUALITY of the clustering.  This script builds the same seeded corpus five times with the reference's builder
(oracle/_ref/libzvec_ref_core.so = its core library compiled in place) and records, per run: the within-cluster sum of squares per
row (SSE / n, fp64), the list-size spread (max / mean), and recall@10 at nprobe 4 of 256 lists over 500 held-out queries searched by
the reference's own IVFSearcher.  tests/test_gpu_build.py::test_kmeans_quality_vs_reference_trainer requires the GPU build
(zvec_hip_ivf_build) of the SAME corpus to sit inside the reference's spread.
Run:  python tests/golden/make_kmeans_quality.py
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refcore as R  # noqa: E402

N, DIM, NLIST, NQ, K, NPROBE, SEED = 200_000, 64, 256, 500, 10, 4, 20260404


def corpus():
    """mixture of 1024 Gaussians (4 per list on average), unit noise, means ~ N(0, 1): overlapping clusters, so the partition quality shows in recall: the shape of bench.py's corpus"""
    rng = np.random.default_rng(SEED)
    means = rng.standard_normal((1024, DIM)).astype(np.float32) * 1.0
    base = (means[rng.integers(0, 1024, N)] + rng.standard_normal((N, DIM)).astype(np.float32)).astype(np.float32)
    q = (means[rng.integers(0, 1024, NQ)] + rng.standard_normal((NQ, DIM)).astype(np.float32)).astype(np.float32)
    return base, q


def exact_topk(base, q):
    d = (q.astype(np.float64) ** 2).sum(1)[:, None] + (base.astype(np.float64) ** 2).sum(1)[None] - 2 * q.astype(np.float64) @ base.astype(np.float64).T
    return np.argsort(d, 1, kind="stable")[:, :K]


def quality(cent, labels, base):
    diff = base.astype(np.float64) - cent.astype(np.float64)[labels]
    sizes = np.bincount(labels, minlength=len(cent))
    return float((diff ** 2).sum() / len(base)), float(sizes.max() / sizes.mean()), int((sizes == 0).sum())


def main():
    from zvec_amd.index import container_segments
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_refcore_cpu import read_ivf_file
    base, q = corpus()
    gt = exact_topk(base, q)
    runs = []
    for r in range(5):
        sec = R.build("IVFBuilder", base, "SquaredEuclidean", "kq", params={"proxima.ivf.builder.centroid_count": str(NLIST), "proxima.ivf.builder.thread_count": 8})
        image = R.mem_get("kq").tobytes()
        cent, offs, total = read_ivf_file(image, np.float32)
        se = R.Runner.searcher("IVFSearcher", "kq", DIM, params={"proxima.ivf.searcher.scan_ratio": NPROBE / NLIST, "proxima.ivf.searcher.brute_force_threshold": N - 1})
        keys, _ = se.walk()                                      # list order: position -> original row (key = row number)
        labels = np.empty(N, np.int64)
        for l in range(NLIST):
            labels[keys[int(offs[l]):int(offs[l + 1])].astype(np.int64)] = l
        sse, spread, empty = quality(cent, labels, base)
        kk, _, cc, _ = se.search_mt(q, K, 8)
        rec = float(np.mean([len(set(kk[i, :cc[i]].tolist()) & set(gt[i].tolist())) / K for i in range(NQ)]))
        se.close()
        runs.append({"sse_per_row": sse, "size_max_over_mean": spread, "empty_lists": empty, "recall_at_10": rec, "train_build_seconds": sec})
        print(runs[-1], flush=True)
    out = {"corpus": {"n": N, "dim": DIM, "nlist": NLIST, "queries": NQ, "k": K, "nprobe": NPROBE, "seed": SEED,
                      "what": "mixture of 1024 unit Gaussians, means ~ N(0, I); see corpus() in tests/golden/make_kmeans_quality.py"},
           "reference": "IVFBuilder (StratifiedClusterTrainer + OptKmeansCluster) of /root/reference compiled in place; 5 runs (random_device seeds)",
           "runs": runs}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kmeans_quality.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
