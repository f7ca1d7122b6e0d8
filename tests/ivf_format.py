"""Test-side WRITER of the reference's dumped IVF segments, restated from its writer code; pinned byte for byte against
files the reference's own writers dumped (tests/golden/ref_index_files.npz, tests/test_golden_files_cpu.py):
  IVFDumper::dump_inverted_vector / dump_block / dump_inverted_vector_finished   src/core/algorithm/ivf/ivf_dumper.cc:19-32,118-200,388-406
  IVFDumper::Block::do_emplace / transpose                                       src/core/algorithm/ivf/ivf_dumper.h:131-160
  InvertedIndexHeader / InvertedListMeta                                         src/core/algorithm/ivf/ivf_index_format.h:26-47
  IndexMetaFormatHeader                                                          src/core/framework/index_meta.cc:23-34
"""
import struct

import numpy as np

DT_FP16, DT_FP32 = 1, 2          # IndexMeta::DataType (index_meta.h:31-41)
MO_ROW, MO_COLUMN = 1, 2         # IndexMeta::MajorOrder (index_meta.h:45-49)


def dump_ivf_segments(lists, dim, dtype=np.float32, column_major=False, block_vector_count=32):
    """lists: list of (vectors [n_l][dim], keys [n_l]) per inverted list -> dict of segment payloads"""
    unit = np.dtype(dtype).itemsize
    elem = dim * unit
    block_size = (block_vector_count * elem + 31) // 32 * 32
    body = bytearray()
    keys = []
    metas = []
    total = blocks = 0
    for vecs, ks in lists:
        vecs = np.ascontiguousarray(vecs, dtype).reshape(-1, dim)
        off, id_off, nblk = len(body), total, 0
        for b0 in range(0, vecs.shape[0], block_vector_count):
            blk = vecs[b0:b0 + block_vector_count]
            if column_major and blk.shape[0] == block_vector_count:
                raw = np.ascontiguousarray(blk.T).tobytes()          # unit u of vector i at (u*bvc + i)*unit
            else:
                raw = blk.tobytes()
            raw += b"\0" * ((len(raw) + 31) // 32 * 32 - len(raw))    # dump_block: ailego_align(bytes, 32)
            body += raw
            nblk += 1
        keys.append(np.asarray(ks, np.uint64))
        metas.append([off, nblk, vecs.shape[0], id_off])
        total += vecs.shape[0]
        blocks += nblk
    # empty lists AFTER the last dumped vector keep the zeroed meta they were created with: check_dump_inverted_list fills
    # offset / id_offset of skipped lists only up to the next list that receives a vector (ivf_dumper.cc:284-291)
    last = max([i for i, m in enumerate(metas) if m[2]], default=-1)
    for m in metas[last + 1:]:
        m[0] = m[3] = 0
    metas = [struct.pack("<QIII16x4x", *m) for m in metas]          # sizeof(InvertedListMeta) == 40 (8-byte aligned)
    index_meta = struct.pack("<9I", 4128, 1, MO_COLUMN if column_major else MO_ROW, DT_FP16 if unit == 2 else DT_FP32,
                             dim, unit, 0, 0, 0) + b"\0" * 4092          # meta_type 1 = MT_DENSE
    header = struct.pack("<IIQIIIII28x", 64 + len(index_meta), total, len(body), len(lists), block_vector_count,
                         block_size, blocks, len(index_meta)) + index_meta
    return {"ivf.inverted_header": bytes(header), "ivf.inverted_meta": b"".join(metas), "ivf.inverted_body": bytes(body),
            "hc.keys": np.concatenate(keys).astype("<u8").tobytes() if keys else b""}


def pack_container(segments, version=b"zvec-test-writer"):
    """Test-side WRITER of the container framing of a dumped index file, restated from IndexPacker::setup / pack / finish /
    pack_version (src/include/zvec/core/framework/index_packer.h:98-231) and IndexFormat (index_format.h:26-200):
    [MetaHeader 64 B][segments' data, each padded to 32 B]["IndexVersion" segment][padding to 32][SegmentMeta[count] +
    NUL-terminated ids, padded to 32][MetaFooter 128 B].  segments: list of (id, bytes).  Checked against files the
    reference's own MemoryDumper wrote (tests/test_golden_files_cpu.py): same segment table and content, byte for byte."""
    from oracle.roaring import crc32c
    content = bytearray()
    stab = []
    for sid, data in segments:
        data = bytes(data)
        pad = (len(data) + 31) // 32 * 32 - len(data)
        stab.append((sid, len(data), pad, crc32c(data, 0)))
        content += data + b"\0" * pad
    vpad = (len(version) + 31) // 32 * 32 - len(version)
    stab.append(("IndexVersion", len(version), vpad, crc32c(version, 0)))
    content += version + b"\0" * vpad
    content_crc = crc32c(bytes(content), 0)
    cpad = (len(content) + 31) // 32 * 32 - len(content)
    metas, ids, off = bytearray(), bytearray(), 0
    ids_base = 32 * len(stab)
    for sid, size, pad, crc in stab:
        metas += struct.pack("<IIQQQ", ids_base + len(ids), crc, off, size, pad)
        ids += sid.encode() + b"\0"
        off += size + pad
    meta_block = bytes(metas + ids)
    meta_block += b"\0" * ((len(meta_block) + 31) // 32 * 32 - len(meta_block))

    def with_crc(buf):                       # crc field first, zero while the crc is taken
        return struct.pack("<I", crc32c(buf, 0)) + buf[4:]
    header = with_crc(struct.pack("<IHHIIHHIQQ24x", 0, 0, 2, 0, 0x5A564543, 64, 128, (1 << 32) - 128, 64, 1700000000))
    total = 64 + len(content) + cpad + len(meta_block) + 128
    footer = with_crc(struct.pack("<IIIIIIQQQQ56xQQ", 0, crc32c(meta_block, 0), content_crc, len(stab), len(meta_block), 0,
                                  len(content), cpad, 0, 1700000001, 0, total))
    image = header + bytes(content) + b"\0" * cpad + meta_block + footer
    assert len(header) == 64 and len(footer) == 128 and len(image) == total
    return image


def index_meta_blob(dim, dtype=np.float32, column_major=False, metric="SquaredEuclidean"):
    """the "IndexMeta" segment: IndexMetaFormatHeader + JSON attachment (index_meta.cc:23-80), test-side writer"""
    import json
    unit = np.dtype(dtype).itemsize
    att = json.dumps({"metric": {"name": metric, "revision": 0, "params": {}}}).encode()
    return struct.pack("<9I", 4128, 1, MO_COLUMN if column_major else MO_ROW, DT_FP16 if unit == 2 else DT_FP32, dim, unit, 0,
                       4128, len(att)) + b"\0" * 4092 + att      # meta_type 1 = dense; attachment_offset from the blob start


def flat_features_blob(base, column_major):
    """FlatBuilder::write_row_index / write_column_index (flat_builder.cc:188-276)"""
    blob = bytearray()
    for b0 in range(0, base.shape[0], 32):
        blk = base[b0:b0 + 32]
        blob += (np.ascontiguousarray(blk.T) if (column_major and blk.shape[0] == 32) else blk).tobytes()
    return bytes(blob)


def flat_index_file(base, keys, column_major=False, metric="SquaredEuclidean"):
    return pack_container([("IndexMeta", index_meta_blob(base.shape[1], base.dtype, column_major, metric)),
                           ("flat.keys", np.asarray(keys, "<u8").tobytes()),
                           ("flat.features", flat_features_blob(base, column_major))])


def ivf_index_file(centroids, lists, column_major=False, centroid_column_major=False, metric="SquaredEuclidean", centroid_perm=None):
    """a dumped IVF index file: the inverted segments + the centroid index as a NESTED flat index file in "ivf.centroid";
    centroid_perm: storage order of the centroid rows inside that nested index (flat.keys = their centroid ids)"""
    dim, dtype = centroids.shape[1], centroids.dtype
    perm = np.arange(centroids.shape[0]) if centroid_perm is None else np.asarray(centroid_perm)
    nested = flat_index_file(np.ascontiguousarray(centroids[perm]), perm.astype(np.uint64), centroid_column_major, metric)
    segs = dump_ivf_segments(lists, dim, dtype, column_major)
    return pack_container([("IndexMeta", index_meta_blob(dim, dtype, column_major, metric)), ("ivf.centroid", nested)] +
                          [(k, segs[k]) for k in ("ivf.inverted_header", "ivf.inverted_meta", "ivf.inverted_body", "hc.keys")])
