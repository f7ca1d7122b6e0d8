"""Generates tests/golden/ref_kernel_vectors.npz from the REFERENCE's own kernels.

Needs oracle/_ref/libzvec_ref.so, i.e. /root/reference compiled in place by oracle/Makefile (the
reference's ailego math kernels and ailego::Heap, -march=skylake-avx512 so its run-time dispatch is
AVX512 > AVX > SSE).  Output = seeded inputs + the reference's outputs; it lets the oracle be pinned
where neither /root/reference nor _ref exists.  Run:  python tests/golden/make_ref_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

o = O.get()
assert o.ref is not None, "oracle/_ref/libzvec_ref.so missing (run `make -C oracle`)"
rng = np.random.default_rng(20260320)
dims = list(range(1, 41)) + [63, 64, 65, 100, 127, 128, 129, 255, 256, 768, 769]
out = {}
A, B, L2, IP, MIP, COS, NRM, NV = [], [], [], [], [], [], [], []
for d in dims:
    a = rng.standard_normal(d).astype(np.float32)
    b = (rng.standard_normal(d) * 2 + 0.5).astype(np.float32)
    A.append(a)
    B.append(b)
    L2.append(o.dist(O.METRIC_L2, a, b, use_ref=True))
    MIP.append(o.dist(O.METRIC_IP, a, b, use_ref=True))
    IP.append(o.ip(a, b, use_ref=True))
    NRM.append(o.norm2(a, use_ref=True))
    v, n = o.normalize_l2(a, use_ref=True)
    NV.append(v)
    if d >= 2:
        a1 = np.append(*o.normalize_l2(a, use_ref=True)).astype(np.float32)
        b1 = np.append(*o.normalize_l2(b, use_ref=True)).astype(np.float32)
        COS.append(o.dist(O.METRIC_COSINE, a1, b1, use_ref=True))
    else:
        COS.append(np.nan)
out["dims"] = np.array(dims, np.int32)
out["a"] = np.concatenate(A)
out["b"] = np.concatenate(B)
out["normalized_a"] = np.concatenate(NV)
for k, v in (("l2", L2), ("ip", IP), ("minus_ip", MIP), ("cosine", COS), ("norm2", NRM)):
    out[k] = np.array(v, np.float32)
# heap replays: sequences with many ties; expected = the reference heap's memory image
hs, hl, hi, hv, hoff = [], [], [], [], [0]
for t in range(64):
    n = int(rng.integers(1, 300))
    k = int(rng.integers(1, 50))
    s = rng.integers(0, 20, n).astype(np.float32)
    idx, sc = o.heap_replay(s, k, use_ref=True)
    hs.append(s)
    hl.append((n, k, len(idx)))
    hi.append(idx)
    hv.append(sc)
# fp16 rows: SquaredEuclideanDistanceMatrix<Float16,1,1> / MinusInnerProductMatrix<Float16,1,1>, FloatHelper::ToFP16
H_A, H_B, H_L2, H_MIP = [], [], [], []
for d in dims:
    a = (rng.standard_normal(d) * 3).astype(np.float16)
    b = (rng.standard_normal(d) * 3).astype(np.float16)
    H_A.append(a)
    H_B.append(b)
    H_L2.append(o.dist16(O.METRIC_L2, a, b, use_ref=True))
    H_MIP.append(o.dist16(O.METRIC_IP, a, b, use_ref=True))
out["h_a"] = np.concatenate(H_A).view(np.uint16)
out["h_b"] = np.concatenate(H_B).view(np.uint16)
out["h_l2"] = np.array(H_L2, np.float32)
out["h_minus_ip"] = np.array(H_MIP, np.float32)
cv = np.concatenate([(rng.standard_normal(4096) * s_).astype(np.float32) for s_ in (1e-6, 1e-3, 1.0, 300.0, 7e4)])
out["tofp16_in"] = cv
out["tofp16_out"] = o.to_fp16_ref(cv).view(np.uint16)
out["heap_scores"] = np.concatenate(hs)
out["heap_meta"] = np.array(hl, np.int32)
out["heap_index"] = np.concatenate(hi)
out["heap_kept_scores"] = np.concatenate(hv)
# M x N block kernels (drawn after everything above so that the older arrays keep their values)
BM, BQ, BL2, BIP, BSHAPE = [], [], [], [], []
for M, N in ((8, 1), (8, 8), (16, 2), (16, 16), (32, 1), (32, 4), (32, 32)):
    for d in (1, 3, 17, 128, 769):
        mb = rng.standard_normal((d, M)).astype(np.float32)
        qb = (rng.standard_normal((d, N)) * 2 + 0.25).astype(np.float32)
        BM.append(mb.ravel())
        BQ.append(qb.ravel())
        BL2.append(o.block_dist(O.METRIC_L2, mb, qb, use_ref=True).ravel())
        BIP.append(o.block_dist(O.METRIC_IP, mb, qb, use_ref=True).ravel())
        BSHAPE.append((M, N, d))
out["block_shapes"] = np.array(BSHAPE, np.int32)
out["block_m"] = np.concatenate(BM)
out["block_q"] = np.concatenate(BQ)
out["block_l2"] = np.concatenate(BL2)
out["block_minus_ip"] = np.concatenate(BIP)
# fp16 blocks (again drawn last)
HM, HQ, HL2, HIP, HSHAPE = [], [], [], [], []
for M, N in ((8, 2), (16, 16), (32, 1), (32, 32)):
    for d in (1, 17, 128, 769):
        mb = (rng.standard_normal((d, M)) * 2).astype(np.float16)
        qb = (rng.standard_normal((d, N)) * 2).astype(np.float16)
        HM.append(mb.ravel().view(np.uint16))
        HQ.append(qb.ravel().view(np.uint16))
        HL2.append(o.block_dist(O.METRIC_L2, mb, qb, use_ref=True).ravel())
        HIP.append(o.block_dist(O.METRIC_IP, mb, qb, use_ref=True).ravel())
        HSHAPE.append((M, N, d))
out["hblock_shapes"] = np.array(HSHAPE, np.int32)
out["hblock_m"] = np.concatenate(HM)
out["hblock_q"] = np.concatenate(HQ)
out["hblock_l2"] = np.concatenate(HL2)
out["hblock_minus_ip"] = np.concatenate(HIP)
# small-M blocks (M = 2, 4: several k steps share one SIMD register, so a pair's sum is NOT one sequential chain;
# the column-major scan uses them for the left-over rows) — fp32 and fp16, drawn last again
for tag, dt, scale in (("sblock", np.float32, 2.0), ("hsblock", np.float16, 2.0)):
    SM, SQ, SL2, SIP, SSHAPE = [], [], [], [], []
    for M, N in ((2, 1), (2, 2), (4, 1), (4, 2), (4, 4)):
        for d in (1, 2, 3, 5, 6, 7, 17, 128, 769, 770, 771):
            mb = (rng.standard_normal((d, M)) * scale).astype(dt)
            qb = (rng.standard_normal((d, N)) * scale).astype(dt)
            SM.append(mb.ravel().view(np.uint16) if dt == np.float16 else mb.ravel())
            SQ.append(qb.ravel().view(np.uint16) if dt == np.float16 else qb.ravel())
            SL2.append(o.block_dist(O.METRIC_L2, mb, qb, use_ref=True).ravel())
            SIP.append(o.block_dist(O.METRIC_IP, mb, qb, use_ref=True).ravel())
            SSHAPE.append((M, N, d))
    out[tag + "_shapes"] = np.array(SSHAPE, np.int32)
    out[tag + "_m"] = np.concatenate(SM)
    out[tag + "_q"] = np.concatenate(SQ)
    out[tag + "_l2"] = np.concatenate(SL2)
    out[tag + "_minus_ip"] = np.concatenate(SIP)
# one-to-many cosine (CosineMetric::batch_distance -> BaseDistance<CosineDistanceMatrix, T, 12, 2>::ComputeBatch): 29 rows per
# dimension (two full batches of 12 + a remainder of 5), rows = converted rows (vector + norm slot); drawn last again
CB_DIMS = [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 24, 25, 31, 32, 33, 34, 40, 41, 47, 48, 49, 56, 57, 63, 64, 65, 96, 97,
           128, 129, 768, 769, 770]
for tag, dt, extra in (("cosb", np.float32, 1), ("hcosb", np.float16, 2)):
    R_, Q_, O_ = [], [], []
    for d in CB_DIMS:
        rows = (rng.standard_normal((29, d + extra)) * 0.5).astype(dt)
        q = (rng.standard_normal(d + extra) * 0.5).astype(dt)
        R_.append(rows.ravel().view(np.uint16) if dt == np.float16 else rows.ravel())
        Q_.append(q.view(np.uint16) if dt == np.float16 else q)
        O_.append(o.cosine_batch(rows, q, use_ref=True))
    out[tag + "_dims"] = np.array([d + extra for d in CB_DIMS], np.int32)
    out[tag + "_rows"] = np.concatenate(R_)
    out[tag + "_q"] = np.concatenate(Q_)
    out[tag + "_out"] = np.concatenate(O_)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kernel_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
